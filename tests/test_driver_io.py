"""Image I/O of the C command-line driver (no GPU): PNG / PPM / PGM decode to the grey stb_image produces for the
reference's main.c ((77 r + 150 g + 29 b) >> 8), and the PNG writer round-trips."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

EXE = os.path.join(ROOT, "soc_project_stereo_matching_amd", "sgm_main")


def luma(rgb):
    a = rgb.astype(np.uint32)
    return ((a[..., 0] * 77 + a[..., 1] * 150 + a[..., 2] * 29) >> 8).astype(np.uint8)


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        import __graft_entry__
        __graft_entry__.build()
    return EXE


def test_decoders_and_png_writer(exe, tmp_path):
    from PIL import Image
    rng = np.random.RandomState(1)
    rgb = rng.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    rgb[:, :20] = (rgb[:, :20] // 32) * 32               # flat areas so every PNG filter type gets used
    want = luma(rgb)
    alpha = rng.randint(0, 256, (37, 53, 1)).astype(np.uint8)
    pal = Image.fromarray(rgb).quantize(200)
    files = {"rgb.png": (Image.fromarray(rgb), want), "rgba.png": (Image.fromarray(np.dstack([rgb, alpha])), want),
             "gray.png": (Image.fromarray(want), want), "ga.png": (Image.fromarray(np.dstack([want, alpha[..., 0]]), "LA"), want),
             "pal.png": (pal, luma(np.asarray(pal.convert("RGB")))), "rgb.ppm": (Image.fromarray(rgb), want),
             "gray.pgm": (Image.fromarray(want), want)}
    for name, (im, expect) in files.items():
        src = str(tmp_path / name)
        im.save(src)
        for ext in ("png", "pgm"):
            dst = str(tmp_path / f"out_{name}.{ext}")
            subprocess.check_call([exe, "--convert", src, dst])
            got = np.asarray(Image.open(dst))
            assert np.array_equal(got, expect), (name, ext)


def test_rejects_what_it_cannot_read(exe, tmp_path):
    from PIL import Image
    bad = str(tmp_path / "bad.png")
    open(bad, "wb").write(b"not a png at all")
    assert subprocess.call([exe, "--convert", bad, str(tmp_path / "o.png")], stderr=subprocess.DEVNULL) != 0
    im16 = str(tmp_path / "deep.png")
    Image.fromarray((np.arange(64, dtype=np.uint16) * 900).reshape(8, 8)).save(im16)     # 16-bit grey: unsupported
    assert subprocess.call([exe, "--convert", im16, str(tmp_path / "o.png")], stderr=subprocess.DEVNULL) != 0
    assert subprocess.call([exe, "--convert", str(tmp_path / "missing.png"), str(tmp_path / "o.png")],
                           stderr=subprocess.DEVNULL) != 0


def test_extension_flags_are_parsed_before_anything_else(exe, tmp_path):
    """--census wants WxH; unknown options are refused (no GPU is touched for either)."""
    for bad in (["--census", "7"], ["--census"], ["--right-ref"]):
        out = subprocess.run([exe, "a.png", "b.png", str(tmp_path / "o.png")] + bad, capture_output=True, text=True)
        assert out.returncode == 2, (bad, out.stderr)


def test_image_io_is_asan_clean_on_damaged_files(tmp_path):
    """The driver's own PNG / PNM decoders (csrc/sgm_image_io.c; zlib inflate + filters written here) under AddressSanitizer /
    UBSan: every valid variant decodes, and truncated or bit-flipped copies are either decoded or refused -- never a memory
    error (exit codes other than 0 / 1 or sanitizer output fail the test).  CPU only: the driver is linked with the stub device."""
    import shutil
    from PIL import Image
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    csrc = os.path.join(ROOT, "soc_project_stereo_matching_amd", "csrc")
    exe = str(tmp_path / "sgm_main_asan")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", csrc,
                           "-o", exe, os.path.join(csrc, "sgm_main.c"), os.path.join(csrc, "sgm_image_io.c"),
                           os.path.join(csrc, "sgm_host.c"), os.path.join(ROOT, "tests", "stub_device.c"), "-lz", "-lm"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99")
    env.pop("LD_PRELOAD", None)
    rng = np.random.RandomState(7)
    rgb = rng.randint(0, 256, (23, 31, 3)).astype(np.uint8)
    rgb[:, :12] = (rgb[:, :12] // 64) * 64
    good = {"a.png": Image.fromarray(rgb), "b.png": Image.fromarray(luma(rgb)), "c.png": Image.fromarray(rgb).quantize(60),
            "d.ppm": Image.fromarray(rgb), "e.pgm": Image.fromarray(luma(rgb))}
    n_cases = 0
    for name, im in good.items():
        src = str(tmp_path / name)
        im.save(src)
        data = open(src, "rb").read()
        variants = [data] + [data[:k] for k in (0, 1, 8, 20, 33, len(data) // 2, len(data) - 5, len(data) - 1)]
        for _ in range(40):
            b = bytearray(data)
            for _ in range(rng.randint(1, 4)):
                b[rng.randint(0, len(b))] ^= 1 << rng.randint(0, 8)
            variants.append(bytes(b))
        for k, v in enumerate(variants):
            p = str(tmp_path / f"v{k}_{name}")
            open(p, "wb").write(v)
            out = subprocess.run([exe, "--convert", p, str(tmp_path / "out.pgm")], capture_output=True, text=True, env=env, timeout=60)
            assert out.returncode in (0, 1), (name, k, out.returncode, out.stderr[-1500:])
            assert "Sanitizer" not in out.stderr and "runtime error" not in out.stderr, (name, k, out.stderr[-1500:])
            n_cases += 1
        assert subprocess.run([exe, "--convert", src, str(tmp_path / "ok.pgm")], env=env).returncode == 0
    assert n_cases == len(good) * 49
