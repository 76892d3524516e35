"""The board stand-in client (csrc/sgm_board_client.c) against a stand-in of the reference's test-platform server:
wire framing, calibration block, grey conversion, depth conversion.  The --placeholder-gray mode (what the
ZedBoard firmware returns today, main.c:227-233) needs no GPU; the SGM mode is a GPU test."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_npz
from platform_server import PlatformServer

EXE = os.path.join(ROOT, "soc_project_stereo_matching_amd", "sgm_board_client")


def bgr_frames(n, w, h, d, oracle, seed):
    """Colour frames whose board-grey is a synthetic stereo pair (channels perturbed so the weights matter)."""
    rng = np.random.RandomState(seed)
    out = []
    for k in range(n):
        l, r = oracle.synth_pair(w, h, d, seed + k)
        frames = []
        for g in (l, r):
            img = np.stack([g, g, g], axis=2).astype(np.int32) + rng.randint(-6, 7, (h, w, 3))
            frames.append(np.clip(img, 0, 255).astype(np.uint8))
        out.append(tuple(frames))
    return out


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        import __graft_entry__
        __graft_entry__.build()
    return EXE


def test_calibration_block_layout():
    """The 80-byte block the reference's stereo_calibration.py packs (golden, generated from the reference):
    20 little-endian float32 = cam0 row-major, cam1 row-major, doffs, baseline."""
    z = load_npz("platform_calib.npz")
    f = np.frombuffer(z["packed"].tobytes(), "<f4")
    assert f.shape == (20,)
    assert f[0] == np.float32(z["fx"]) and f[18] == np.float32(z["doffs"]) and f[19] == np.float32(z["baseline"])
    assert f[8] == 1.0 and f[17] == 1.0 and f[1] == 0.0


def test_placeholder_mode_framing_no_gpu(exe, oracle):
    from oracle.platform_oracle import board_gray
    frames = bgr_frames(3, 96, 40, 16, oracle, 4100)
    srv = PlatformServer(frames, load_npz("platform_calib.npz")["packed"].tobytes())
    out = subprocess.run([exe, "127.0.0.1", str(srv.port), "--placeholder-gray"], capture_output=True, text=True, timeout=60)
    srv.join()
    assert out.returncode == 0, out.stderr
    assert srv.requests == [1, 3, 2, 3, 2, 3, 2] + [0] * (len(srv.requests) - 7)     # calib once, then images; 4th request answered with close
    assert sorted(srv.results) == [0, 1, 2]
    for k, (l, _) in enumerate(frames):
        want = board_gray(l[:, :, 0], l[:, :, 1], l[:, :, 2]).astype(np.float32)
        assert np.array_equal(srv.results[k], want)


def test_scores_restatement():
    from oracle.platform_oracle import compare_depth, disparity_to_depth
    gt = np.array([[1000.0, 2000.0, np.nan], [500.0, np.inf, 40.0]], np.float32)
    test = np.array([[1005.0, 2100.0, 3.0], [np.nan, 7.0, 40.0]], np.float32)
    rmse, bpr, n = compare_depth(gt, test)
    assert n == 3 and abs(rmse - np.sqrt((25 + 10000 + 0) / 3)) < 1e-3 and abs(bpr - 1 / 3) < 1e-9
    assert compare_depth(np.full((2, 2), np.nan, np.float32), test[:, :2]) == (pytest.approx(float("nan"), nan_ok=True), pytest.approx(float("nan"), nan_ok=True), 0)
    d = disparity_to_depth(np.array([10.0, np.inf, -5.0], np.float32), fx=1000.0, baseline=500.0, doffs=5.0)
    assert d[0] == np.float32(np.float32(500000.0) / np.float32(15.0)) and np.isnan(d[1]) and np.isnan(d[2])


@pytest.mark.gpu
def test_sgm_mode_depth_bit_exact(exe, oracle):
    """Full loop: frames over TCP -> board grey -> SGM on the MI355X -> depth in mm -> back over TCP; every returned
    float equals the oracle's disparity pushed through the platform's depth formula."""
    from oracle.pyoracle import default_option
    from oracle.platform_oracle import board_gray, disparity_to_depth
    z = load_npz("platform_calib.npz")
    w, h, d = 320, 96, 64
    frames = bgr_frames(3, w, h, d, oracle, 5200)
    srv = PlatformServer(frames, z["packed"].tobytes())
    out = subprocess.run([exe, "127.0.0.1", str(srv.port), "--max-disparity", str(d)], capture_output=True, text=True, timeout=120)
    srv.join()
    assert out.returncode == 0, out.stderr
    f = np.frombuffer(z["packed"].tobytes(), "<f4")
    opt = default_option(d)
    for k, (l, r) in enumerate(frames):
        gl = board_gray(l[:, :, 0], l[:, :, 1], l[:, :, 2])
        gr = board_gray(r[:, :, 0], r[:, :, 1], r[:, :, 2])
        disp = oracle.run(gl, gr, opt)["final"]
        want = disparity_to_depth(disp, fx=float(f[0]), baseline=float(f[19]), doffs=float(f[18]))
        got = srv.results[k]
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(got[~np.isnan(got)].view(np.uint32), want[~np.isnan(want)].view(np.uint32)), k


def test_client_is_asan_clean_with_the_stub_device(tmp_path, oracle):
    """sgm_board_client.c + the C host under AddressSanitizer / UBSan, linked with the stub device layer (no GPU; the maps it
    returns are meaningless): socket framing, pinned-buffer lifetime and the platform-frame entry for frames of two sizes in one
    session (buffers re-sized), then the server closing the connection."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    csrc = os.path.join(ROOT, "soc_project_stereo_matching_amd", "csrc")
    exe_asan = str(tmp_path / "board_client_asan")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", csrc,
                           "-o", exe_asan, os.path.join(csrc, "sgm_board_client.c"), os.path.join(csrc, "sgm_host.c"),
                           os.path.join(ROOT, "tests", "stub_device.c"), "-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99")
    env.pop("LD_PRELOAD", None)
    frames = bgr_frames(2, 96, 40, 16, oracle, 4300) + bgr_frames(2, 64, 24, 16, oracle, 4400)
    for extra in ([], ["--placeholder-gray"], ["--max-frames", "3"]):
        srv = PlatformServer(frames, load_npz("platform_calib.npz")["packed"].tobytes())
        out = subprocess.run([exe_asan, "127.0.0.1", str(srv.port), "--max-disparity", "16"] + extra, capture_output=True, text=True,
                             env=env, timeout=120)
        srv.join()
        assert out.returncode == 0, (extra, out.returncode, out.stderr[-2000:])
        assert "Sanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-2000:]
        assert len(srv.results) == (3 if extra and extra[0] == "--max-frames" else 4)
        for k, res in srv.results.items():
            assert res.shape == frames[k][0].shape[:2]
