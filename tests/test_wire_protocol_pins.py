"""The test platform's wire protocol, pinned against the REFERENCE'S OWN TEXT where it is mounted.

server.py cannot be imported here (it needs cv2), and the firmware's C needs Xilinx headers -- so the board client
(csrc/sgm_board_client.c) is exercised against a stand-in server (tests/platform_server.py).  What keeps the stand-in and the client
honest is this file: it reads the reference's sources AS TEXT (nothing is imported, compiled or copied) and checks every constant of
the protocol -- struct formats, type codes, image size, the calibration block's place and size, the byte order of the result header
-- against what the stand-in sends and the client parses.  Skipped on machines without /root/reference (the GPU box)."""
import os
import re
import struct

import pytest

from conftest import ROOT

REF = os.environ.get("SGM_REFERENCE_DIR", "/root/reference")
SERVER = os.path.join(REF, "HostScript_Server", "server.py")
FW_SRC = os.path.join(REF, "ZedBoard", "Vitis", "lwip_tcp_perf_client", "src")

pytestmark = pytest.mark.skipif(not os.path.exists(SERVER), reason="the reference is not mounted here")


def text(path):
    with open(path, encoding="utf-8", errors="replace") as f:
        return f.read()


def test_server_side_formats_match_the_stand_in_and_the_client():
    ref = text(SERVER)
    ours = text(os.path.join(ROOT, "tests", "platform_server.py"))
    client = text(os.path.join(ROOT, "soc_project_stereo_matching_amd", "csrc", "sgm_board_client.c"))
    # image header server -> board: type (1 B), seq (int32), width, height (uint16), little-endian (server.py send_image)
    assert re.search(r"struct\.pack\('<BiHH',\s*type_id,\s*seq,\s*width,\s*height\)", ref)
    assert 'struct.pack("<BiHH", type_id, seq, w, h)' in ours and struct.calcsize("<BiHH") == 9
    assert "uint8_t hdr[9];" in client and "hdr[5] | (hdr[6] << 8)" in client and "hdr[7] | (hdr[8] << 8)" in client
    # close status: one zero byte (send_close_status); the client stops on type 0
    assert re.search(r"def send_close_status.*?struct\.pack\('<B',\s*0\)", ref, re.S)
    assert 'struct.pack("<B", 0)' in ours and "if (hdr[0] == 0) break;" in client
    # the calibration block follows the header only for type 1 (server.py: `if type_id == 1 and calib is not None: conn.send(calib.pack())`)
    assert re.search(r"if type_id == 1 and calib is not None:.*?conn\.send\(calib\.pack\(\)\)", ref, re.S)
    assert "if (hdr[0] == 1) {" in client and "recv_all(fd, cal, 80)" in client
    # planes: left B, G, R then right B, G, R, each `height` rows of `width` bytes (two loops over range(3) x range(height))
    assert len(re.findall(r"for ch in range\(3\):\s*\n\s*for y in range\(height\):\s*\n\s*conn\.send\(img_(?:left|right)\[y, :, ch\]\.tobytes\(\)\)", ref)) == 2
    # result board -> server: type byte 3 already read, then <iHH (seq, width, height), then height rows of width float32
    assert re.search(r"struct\.unpack\('<iHH',\s*header_bytes\)", ref) and "recv_exact(conn, 8)" in ref
    assert re.search(r"row_size\s*=\s*width\s*\*\s*4", ref) and "np.frombuffer(row, dtype=np.float32)" in ref
    assert 'struct.unpack("<iHH", self._recv(conn, 8))' in ours
    assert "uint8_t out[9] = {3, hdr[1], hdr[2], hdr[3], hdr[4], hdr[5], hdr[6], hdr[7], hdr[8]};" in client
    # request codes the server answers: 1 = image + calibration, 2 = image, 3 = result follows, 0 = stop
    for code in (0, 1):
        assert re.search(rf"request == {code}", ref)
    assert re.search(r"request == 1 or request == 2", ref)
    # the platform's fixed frame size
    assert re.search(r"^WIDTH = 1280$", ref, re.M) and re.search(r"^HEIGHT = 720$", ref, re.M)


def test_firmware_side_constants_match_the_client():
    fb = text(os.path.join(FW_SRC, "frame_buffer.h"))
    fw = text(os.path.join(FW_SRC, "tcp_perf_client.c"))
    client = text(os.path.join(ROOT, "soc_project_stereo_matching_amd", "csrc", "sgm_board_client.c"))
    assert re.search(r"#define IMG_WIDTH 1280", fb) and re.search(r"#define IMG_HEIGHT 720", fb)
    # the firmware's receive state machine: 9 header bytes + 80 calibration bytes = 89, then six planes
    assert "89 + IMG_WIDTH * IMG_HEIGHT * 6" in fw
    assert struct.calcsize("<BiHH") + 20 * 4 == 89
    # its result header: byte 0 = 3, frame id little-endian in bytes 1..4, width in 5..6, height in 7..8; then W*H*4 bytes of depth
    assert "header[0] = 3;" in fw and "header[4] = (uint8_t)(shared_memory->last_depth_frame.frame_id >> 24);" in fw
    assert "header[6] = (uint8_t)(shared_memory->last_depth_frame.width >> 8);" in fw
    assert "header[8] = (uint8_t)(shared_memory->last_depth_frame.height >> 8);" in fw
    assert "IMG_HEIGHT * IMG_WIDTH * 4" in fw
    assert "uint8_t out[9] = {3," in client
    # message types the firmware accepts from the server
    assert "type == 1 || type == 2" in fw and "type == 3" in fw and "type == 0" in fw


def test_firmware_grey_conversion_weights():
    """stereo_matching.c:15-24 (the file needs Xilinx headers, so it is read, not built): grey = (76 R + 150 G + 29 B) >> 8 as uint8,
    stored as float -- the weights of sgm_gray_planes_k's board mode, of oracle/platform_oracle.board_gray and of --placeholder-gray."""
    import numpy as np
    from oracle.platform_oracle import board_gray
    fw = text(os.path.join(FW_SRC, "stereo_matching.c"))
    m = re.search(r"uint32_t gray =\s*(\d+)u\s*\*[^;]*?left_red\[i\]\s*\+\s*(\d+)u\s*\*[^;]*?left_green\[i\]\s*\+\s*(\d+)u\s*\*[^;]*?left_blue\[i\];", fw, re.S)
    assert m and tuple(int(v) for v in m.groups()) == (76, 150, 29)
    assert "uint8_t gray8 = (uint8_t)(gray >> 8);" in fw and "depth[i] = (float)gray8;" in fw
    r, g, b = np.array([255, 0, 13], np.uint8), np.array([255, 0, 200], np.uint8), np.array([255, 0, 77], np.uint8)
    assert board_gray(b, g, r, 76).tolist() == [(76 * 255 + 150 * 255 + 29 * 255) >> 8, 0, (76 * 13 + 150 * 200 + 29 * 77) >> 8]
    client = text(os.path.join(ROOT, "soc_project_stereo_matching_amd", "csrc", "sgm_board_client.c"))
    assert "76" in client and "150" in client and "29" in client
