"""The CPU restatement is AddressSanitizer / UBSan clean (the reference itself is not: SURVEY.md Q6)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_is_asan_ubsan_clean(tmp_path):
    exe = str(tmp_path / "asan_check")
    src = [os.path.join(ROOT, "oracle", f) for f in ("sgm_oracle.c", "asan_check.c")]
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-o", exe] + src + ["-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1")
    env.pop("LD_PRELOAD", None)
    out = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("asan_check ok")
