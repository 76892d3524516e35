"""The C-ABI library loads without a GPU and exports every symbol include/sgm_mi355x.h and include/sgm_tiles.h declare.
No compute call is made here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT

HEADERS = [os.path.join(ROOT, "include", h) for h in ("sgm_mi355x.h", "sgm_tiles.h")]


def _declared_functions():
    names = set()
    for header in HEADERS:
        text = open(header).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"typedef[^;{]*\(\s*\*\s*\w+\s*\)[^;]*;", "", text)          # function-pointer typedefs are not symbols
        names.update(re.findall(r"\b((?:SGM_|sgm_)\w+)\s*\(", text))
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    import soc_project_stereo_matching_amd as S
    if not os.path.exists(S.library_path()):
        import __graft_entry__
        __graft_entry__.build()
    return S.load_library()


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    # the reference boundary (SemiGlobalMatching.h:78-80) must be there verbatim
    for must in ("SGM_Initialize", "SGM_Reset", "SGM_Match"):
        assert must in names
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert len(names) >= 25
    for must in ("sgm_tile_step", "sgm_tiles_create", "sgm_tiles_submit", "sgm_tiles_finish", "sgm_tiles_rccl_transport",
                 "sgm_tiles_local_transport"):
        assert must in names


def test_option_struct_abi():
    """x86-64 SysV layout of the reference's SGMOption: 28 bytes, these offsets."""
    from soc_project_stereo_matching_amd import SGMOption
    assert C.sizeof(SGMOption) == 28 and C.alignment(SGMOption) == 4
    want = {"num_paths": 0, "min_disparity": 2, "max_disparity": 4, "is_check_unique": 6, "uniqueness_ratio": 8,
            "is_check_lr": 12, "lrcheck_thres": 16, "is_remove_speckles": 20, "min_speckle_area": 22, "p1": 24,
            "p2_init": 26}
    for k, off in want.items():
        assert getattr(SGMOption, k).offset == off, k


def test_argument_errors_need_no_gpu(lib):
    """false for w==0, h==0, max<=min, Match before Initialize, NULL images (SemiGlobalMatching.c:43-48,70,73)."""
    import soc_project_stereo_matching_amd as S
    g = S.SGM()
    assert not g.initialize(0, 10, S.default_option(16))
    assert not g.initialize(10, 0, S.default_option(16))
    assert not g.initialize(10, 10, S.default_option(8, 8))
    assert not g.initialize(10, 10, S.default_option(4, 9))
    img = np.zeros((10, 10), np.uint8)
    assert g.match(img, img) is None            # never initialised
    assert g.match(None, None) is None
    assert not g.synchronize()
    assert b"gfx950" in lib.SGM_Version()


def test_fails_loudly_without_gpu(lib, capfd):
    """No CPU fallback: on a machine without a gfx950 device a valid Initialize returns false and says why."""
    import soc_project_stereo_matching_amd as S
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert not S.SGM().initialize(64, 32, S.default_option(16))
    assert "no HIP device" in capfd.readouterr().err
    with pytest.raises(RuntimeError):
        S.SGMInstance(0)


def test_synth_generator_matches_oracle(lib, oracle):
    import soc_project_stereo_matching_amd as S
    for (w, h, d, seed) in [(24, 16, 8, 1), (130, 37, 64, 0x5EED0002), (1242, 375, 128, 0x5EED0002)]:
        a = S.synth_pair(w, h, d, seed)
        b = oracle.synth_pair(w, h, d, seed)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
