"""A minimal stand-in for the SERVER side of the reference's test platform (HostScript_Server/server.py:105-131,
148-177, 183-280) -- test infrastructure.  It serves a list of BGR stereo frames over the reference's wire protocol
and collects the depth images the client returns.  One client, one thread."""
import socket
import struct
import threading

import numpy as np


class PlatformServer:
    def __init__(self, frames, calib_bytes):
        """frames: list of (left_bgr, right_bgr) uint8 [H][W][3]; calib_bytes: the 80-byte calibration block."""
        self.frames, self.calib = frames, bytes(calib_bytes)
        self.results, self.requests, self.error = {}, [], None
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.sock.bind(("127.0.0.1", 0))
        self.sock.listen(1)
        self.port = self.sock.getsockname()[1]
        self.thread = threading.Thread(target=self._serve, daemon=True)
        self.thread.start()

    @staticmethod
    def _recv(conn, n):
        buf = b""
        while len(buf) < n:
            chunk = conn.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("closed")
            buf += chunk
        return buf

    def _send_frame(self, conn, type_id, seq):
        left, right = self.frames[seq]
        h, w = left.shape[:2]
        conn.sendall(struct.pack("<BiHH", type_id, seq, w, h))          # server.py:113
        if type_id == 1:
            conn.sendall(self.calib)                                     # server.py:116-118
        for img in (left, right):                                        # server.py:124-131: B, G, R planes, row by row
            for ch in range(3):
                conn.sendall(np.ascontiguousarray(img[:, :, ch]).tobytes())

    def _serve(self):
        try:
            conn, _ = self.sock.accept()
            seq = 0
            while True:
                b = conn.recv(1)
                if not b:
                    break
                req = b[0]
                self.requests.append(req)
                if req in (1, 2) and seq >= len(self.frames):            # server.py:212-215
                    conn.sendall(struct.pack("<B", 0))
                    break
                if req == 0:
                    break
                if req in (1, 2):
                    self._send_frame(conn, req, seq)
                    seq += 1
                elif req == 3:                                           # server.py:148-177
                    s, w, h = struct.unpack("<iHH", self._recv(conn, 8))
                    data = self._recv(conn, w * h * 4)
                    self.results[s] = np.frombuffer(data, "<f4").reshape(h, w).copy()
                else:
                    raise ValueError(f"unknown request {req}")
            conn.close()
        except Exception as e:                                           # surfaced by the test
            self.error = e
        finally:
            self.sock.close()

    def join(self, timeout=60):
        self.thread.join(timeout)
        if self.error:
            raise self.error
