"""Parity of the HIP path (through the C-ABI of libsgm_mi355x.so) with the CPU oracle -- needs an MI355X.

Bit-exact on every stage: integer stages byte-identical, float stages identical bit patterns (the
north_star's +-1 LSB allowance after the median is not used).  Tolerance: 0."""
import numpy as np
import pytest

from conftest import case_inputs, load_npz, option_from_dict
from oracle.pyoracle import STAGE_NAMES, sha

pytestmark = pytest.mark.gpu

GOLDEN_CASES = ["cone", "t24x16_d8", "t70x33_d16", "t20x31_d8_tall", "t40x24_d16_dmin3", "t33x33_d12_square",
                "t64x20_d40", "v_default", "v_no_unique", "v_no_lr", "v_no_speckle", "v_plain", "v_p1_0_p2_0",
                "v_p2_small", "v_p_big", "v_ratio_095", "v_lr_thres_0", "v_speckle_area_400", "v_num_paths_4_ignored",
                "c1_synth_450x375_d64", "c2_kitti_1242x375_d128", "d256_400x48", "d192_300x60",
                # real scenes: the pairs the reference ships beside cone (Data/Cloth3, Reindeer, Wood2; make_golden_scenes.py)
                "scene_cloth3", "scene_reindeer", "scene_wood2"]


def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_same(got, want, what):
    if not np.array_equal(bits(got), bits(want)):
        bad = np.argwhere(bits(got) != bits(want))
        first = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {got.size} elements differ; first at {first}: "
                             f"gpu={got[first]} oracle={want[first]}")


@pytest.fixture(scope="module", params=["fused", "separate"])
def inst(request):
    """Every test on this instance runs twice: with the library's default cost-sum/WTA path (the fused row kernel
    where it exists, Dp <= 256) and with the separate sum / right-view kernels forced (what D > 256 always uses;
    SGM_FUSED_WTA is read at SGM_Initialize / SGM_Reset)."""
    import os
    import soc_project_stereo_matching_amd as S
    if request.param == "separate":
        os.environ["SGM_FUSED_WTA"] = "0"
    i = S.SGMInstance(0)
    i.keep_stages(True)
    yield i
    i.close()
    os.environ.pop("SGM_FUSED_WTA", None)


@pytest.fixture
def gsgm():
    """The reference-shaped global entry points.  Their default instance keeps its census buffers like the reference's statics
    (a frame's border words depend on the frames of other shapes before it, tests/test_census_history.py), so
    every test starts from SGM_Shutdown = a new process."""
    import soc_project_stereo_matching_amd as S
    g = S.SGM()
    g.shutdown()
    yield g
    g.shutdown()


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_golden_case_all_stages(inst, oracle, golden_cases, name):
    """Every stage the device materialises against the golden digests produced by the reference itself."""
    case = golden_cases[name]
    left, right = case_inputs(case, oracle)
    opt = option_from_dict(case["option"])
    assert inst.reset(case["w"], case["h"], opt)
    out = inst.match(left, right)
    assert out is not None
    st = inst.read_stages()
    if not case["option"]["is_check_lr"]:
        st["disp_r"][:] = 0
    for n in STAGE_NAMES:
        assert sha(st[n]) == case["sha256"][n], f"{name}: stage {n} differs from the reference"
    assert sha(out) == case["sha256"]["final"]
    if "file" in case:
        z = load_npz(case["file"])
        for n in STAGE_NAMES:
            assert_same(st[n], z[n], f"{name}:{n}")


def test_cone_through_reference_entry_points(gsgm, oracle, golden_cases):
    """The main.c flow (Initialize -> Match -> normalise) through SGM_Initialize / SGM_Match proper,
    compared with the reference's committed Data/cone/im2.d.png: 168749/168750 px."""
    import soc_project_stereo_matching_amd as S
    z = load_npz("cone_inputs.npz")
    assert gsgm.initialize(450, 375, S.default_option(64))
    disp = gsgm.match(z["left"], z["right"])
    assert disp is not None
    assert_same(disp, load_npz("cone_final.npz")["final"], "cone final")
    u8 = oracle.normalize_u8(disp)
    assert np.argwhere(u8 != z["im2_d_png"]).tolist() == [[374, 153]]


@pytest.mark.parametrize("shape", [(37, 21, 0, 8), (15, 29, 0, 8), (130, 47, 2, 50), (257, 64, 0, 100), (64, 7, 0, 16),
                                   (9, 9, 0, 4), (40, 1100, 0, 8), (600, 300, 0, 256), (520, 40, 0, 512),
                                   (1762, 120, 0, 192), (2880, 90, 0, 256)])    # full widths of configs 4 and 2
def test_random_shapes_against_oracle(inst, oracle, shape):
    """Seeded shapes incl. W<H, H > one median band (1024 rows), odd D, the largest D."""
    w, h, dmin, dmax = shape
    from oracle.pyoracle import default_option
    left, right = oracle.synth_pair(w, h, dmax - dmin, 0xABC000 + w * 7 + h)
    opt = default_option(dmax, dmin, min_speckle_area=9)
    want = oracle.run(left, right, opt)
    assert inst.reset(w, h, opt)
    out = inst.match(left, right)
    assert out is not None
    got = inst.read_stages()
    for n in STAGE_NAMES:
        assert_same(got[n], want[n], f"{shape}:{n}")
    assert_same(out, want["final"], f"{shape}:result")


@pytest.mark.parametrize("fused", ["0", "1"])
def test_q14_match_without_reset_accumulates(gsgm, fused, monkeypatch):
    """a second SGM_Match without SGM_Reset adds onto the previous frame's S.  With the fused
    kernel S is not written per frame: the library has to materialise it when the second Match arrives."""
    import soc_project_stereo_matching_amd as S
    monkeypatch.setenv("SGM_FUSED_WTA", fused)
    z = load_npz("q14_no_reset_48x20_d16.npz")
    opt = S.default_option(16, min_speckle_area=8)
    assert gsgm.reset(48, 20, opt)
    assert_same(gsgm.match(z["left"], z["right"]), z["first"], "first")
    assert_same(gsgm.match(z["left2"], z["right2"]), z["second"], "second (no reset)")
    # a third Match on the same S: the oracle's ctx API is the checker (the fixture holds two frames)
    from oracle.pyoracle import Oracle, default_option as oracle_option
    orc = Oracle()
    assert orc.reset(48, 20, oracle_option(16, min_speckle_area=8))
    orc.match(z["left"], z["right"])
    assert_same(orc.match(z["left2"], z["right2"]), z["second"], "oracle second (no reset)")
    third = orc.match(z["left"], z["right"])
    assert_same(gsgm.match(z["left"], z["right"]), third, "third (no reset)")
    assert_same(gsgm.read_stage("aggr"), orc.stage("aggr"), "S after three frames")
    assert gsgm.reset(48, 20, opt)
    assert_same(gsgm.match(z["left2"], z["right2"]), z["second_fresh"], "second after reset")


@pytest.mark.parametrize("name", ["p4_cone", "p4_scene_reindeer", "p4_t70x33_d16", "p4_t20x31_d8_tall", "p4_t40x24_d16_dmin3",
                                  "p4_kitti_1242x375_d128"])
def test_four_path_mode_equals_the_references_first_four_calls(inst, oracle, golden_cases, name):
    """BASELINE config 0 ("4 paths"): num_paths == 4 with SGM_SetHonorNumPaths(1) against digests of the reference's OWN stage functions
    run with the first four of its eight CostAggregate calls (SemiGlobalMatching.c:213-216; tests/golden/cases_paths4.json) -- every
    stage the device materialises, on the cone pair, a real scene, tiny / tall / dmin shapes and a KITTI-size frame."""
    case = golden_cases[name]
    left, right = case_inputs(case, oracle)
    inst.set_honor_num_paths(True)
    try:
        assert inst.reset(case["w"], case["h"], option_from_dict(case["option"]))
        out = inst.match(left, right)
        assert out is not None
        st = inst.read_stages()
        for n in STAGE_NAMES:
            assert sha(st[n]) == case["sha256"][n], f"{name}: stage {n} differs from the reference's four-call run"
        assert sha(out) == case["sha256"]["final"]
    finally:
        inst.set_honor_num_paths(False)


def test_four_path_mode_extension(inst, oracle):
    """num_paths == 4 with SGM_SetHonorNumPaths(1): the first four directions only, here against the oracle on a further shape (the
    reference-made digests of this mode: test_four_path_mode_equals_the_references_first_four_calls)."""
    from oracle.pyoracle import default_option
    left, right = oracle.synth_pair(450, 375, 64, 0x5EED0001)
    opt = default_option(64, num_paths=4)
    oracle.set_honor_num_paths(True)
    inst.set_honor_num_paths(True)
    try:
        want = oracle.run(left, right, opt)
        assert inst.reset(450, 375, opt)
        out = inst.match(left, right)
        assert_same(inst.read_stage("aggr"), want["aggr"], "4-path aggr")
        assert_same(out, want["final"], "4-path final")
    finally:
        oracle.set_honor_num_paths(False)
        inst.set_honor_num_paths(False)


def test_device_resident_frames_and_instances_in_flight(oracle):
    """sgm_match_device on HBM-resident frames, three instances in flight on their own streams."""
    import torch
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    w, h, d = 320, 96, 64
    opt = default_option(d)
    insts = [S.SGMInstance(0) for _ in range(3)]
    frames, outs, wants = [], [], []
    for k in range(6):
        l, r = oracle.synth_pair(w, h, d, 0x77000 + k)
        wants.append(oracle.run(l, r, opt)["final"])
        frames.append((torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()))
        outs.append(torch.empty((h, w), dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    for k in range(6):
        i = insts[k % 3]
        assert i.reset(w, h, opt)
        assert i.match_device(frames[k][0].data_ptr(), frames[k][1].data_ptr(), outs[k].data_ptr())
    for i in insts:
        assert i.synchronize()
    for k in range(6):
        assert_same(outs[k].cpu().numpy(), wants[k], f"frame {k}")
    for i in insts:
        i.close()


def test_full_size_properties_kitti_batch(oracle):
    """Size-independent checks at the headline shape on frames nobody ran on the CPU: (a) the result
    is deterministic across repeats and instances, (b) S of a frame equals the sum of its
    per-direction planes wherever a pixel is visited once (sum kernel linearity), (c) a checksum of
    checksums over a batch of 8 frames is reproducible."""
    import hashlib
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    opt = default_option(128)
    a, b = S.SGMInstance(0), S.SGMInstance(0)
    digests = []
    for k in range(8):
        l, r = S.synth_pair(1242, 375, 128, 0x5EED0100 + k)
        assert a.reset(1242, 375, opt) and b.reset(1242, 375, opt)
        ra, rb = a.match(l, r), b.match(l, r)
        assert_same(ra, rb, f"frame {k} instance a vs b")
        assert a.reset(1242, 375, opt)
        assert_same(a.match(l, r), ra, f"frame {k} repeat")
        digests.append(hashlib.sha256(ra.tobytes()).digest())
        if k == 0:
            S_total = a.read_stage("aggr").astype(np.uint32)
            planes = sum(a.read_stage(10 + i).astype(np.uint32) for i in range(8))
            diff = (S_total != planes).any(axis=2)
            # only pixels on the four anomalous lines may differ (their extra visits), < 4*H pixels
            assert diff.sum() <= 4 * 375
            assert (S_total >= planes).all()
    assert len(set(digests)) == 8
    a.close(); b.close()


@pytest.mark.parametrize("fast", ["1", "0"])
@pytest.mark.parametrize("batch", [1, 3])
def test_aggregation_step_families_at_their_penalty_limits(oracle, fast, batch, monkeypatch):
    """The aggregation step comes in three families chosen from the penalties: FAST (max(P1, P2_init) <= 223: three-way packed
    minimum through v_pk_minimum3_f16, no uint8 mask away from the left border), plain non-negative P1, generic (negative P1).
    Penalties at and either side of the FAST limit, where C + bracket reaches exactly 255 / passes it, with the left-border
    wraps (C = 127) in every case; SGM_AGG_FAST=0 runs the same cases on the plain step.  S and the final map against the oracle."""
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    monkeypatch.setenv("SGM_AGG_FAST", fast)
    i = S.SGMInstance(0, batch=batch)
    i.keep_stages(True)
    try:
        for (p1, p2) in [(10, 150), (0, 223), (223, 0), (223, 223), (10, 224), (224, 10), (200, 231), (0, 0), (31488, 5), (32767, 32767)]:
            for (w, h, dmin, dmax) in [(300, 40, 0, 128), (150, 25, 3, 93), (90, 33, 0, 64)]:
                d = dmax - dmin
                pairs = [oracle.synth_pair(w, h, d, 0xFA57 + 7 * p1 + p2 + w + j) for j in range(batch)]
                opt = default_option(dmax, dmin, min_speckle_area=10, p1=p1, p2_init=p2)
                assert i.reset(w, h, opt)
                L = np.stack([p[0] for p in pairs]) if batch > 1 else pairs[0][0]
                R = np.stack([p[1] for p in pairs]) if batch > 1 else pairs[0][1]
                out = i.match(L, R)
                for j, (l, r) in enumerate(pairs):
                    want = oracle.run(l, r, opt)
                    i.select_frame(j)
                    assert_same(i.read_stage("aggr"), want["aggr"], f"S P1={p1} P2={p2} {w}x{h} d{dmin}-{dmax} frame {j}")
                    assert_same(out[j] if batch > 1 else out, want["final"], f"final P1={p1} P2={p2} {w}x{h} frame {j}")
    finally:
        i.close()


@pytest.mark.parametrize("hl", ["0", "32", "64"])
@pytest.mark.parametrize("lanes", ["16", "8"])
def test_lane_layouts_of_the_aggregation_kernel(oracle, hl, lanes, monkeypatch):
    """The aggregation kernel's lane layouts (lanes per pixel on the horizontal lines: as the others / 32 / 64;
    8 or 16 on the vertical and diagonal ones) are tuning knobs only: every combination must give the oracle's S
    and final map (combinations a disparity range does not admit fall back inside the library)."""
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    monkeypatch.setenv("SGM_HL", hl)
    monkeypatch.setenv("SGM_LANES_PER_PIXEL", lanes)
    i = S.SGMInstance(0)
    try:
        for (w, h, dmin, dmax) in [(300, 40, 0, 128), (200, 30, 0, 256), (150, 25, 3, 93), (90, 33, 0, 64), (70, 20, 0, 512),
                                   (260, 22, 0, 192), (420, 18, 4, 180)]:
            left, right = oracle.synth_pair(w, h, dmax - dmin, 0x1A7E5 + w)
            opt = default_option(dmax, dmin, min_speckle_area=10)
            want = oracle.run(left, right, opt)
            assert i.reset(w, h, opt)
            out = i.match(left, right)
            assert_same(i.read_stage("aggr"), want["aggr"], f"S {w}x{h} d{dmin}-{dmax} HL={hl} lanes={lanes}")
            assert_same(out, want["final"], f"final {w}x{h} d{dmin}-{dmax} HL={hl} lanes={lanes}")
    finally:
        i.close()


@pytest.mark.parametrize("segments", ["1", "2", "3", "4"])
def test_row_segments_of_the_fused_sum_kernel(oracle, segments, monkeypatch):
    """The fused cost-sum/WTA kernel cuts rows into segments when a launch has few rows; segments overlap by
    dmin + D - 1 columns (re-summed) and must not change either disparity view."""
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    monkeypatch.setenv("SGM_SUM_SEGMENTS", segments)
    monkeypatch.setenv("SGM_FUSED_WTA", "1")
    i = S.SGMInstance(0)
    i.keep_stages(False)                       # S not stored: segments are only used then
    try:
        for (w, h, dmin, dmax) in [(1242, 24, 0, 128), (530, 20, 5, 98), (300, 17, 0, 64), (257, 9, 2, 30), (2000, 6, 0, 128),
                                   (1100, 8, 0, 192), (1300, 7, 3, 250)]:
            left, right = oracle.synth_pair(w, h, dmax - dmin, 0x5E6 + w)
            opt = default_option(dmax, dmin, min_speckle_area=10)
            want = oracle.run(left, right, opt)
            assert i.reset(w, h, opt)
            out = i.match(left, right)
            assert_same(i.read_stage("disp_r"), want["disp_r"], f"right view {w}x{h} d{dmin}-{dmax} segments={segments}")
            assert_same(out, want["final"], f"final {w}x{h} d{dmin}-{dmax} segments={segments}")
            assert_same(i.read_stage("aggr"), want["aggr"], f"S (materialised afterwards) {w}x{h}")
    finally:
        i.close()


def _speckle_maps(rng, h, w):
    """Disparity maps that stress the connected-component labelling: long straight tile edges, thin stripes,
    holes, one giant component, noise around the |delta| <= 1 link threshold."""
    inf = np.float32(np.inf)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    maps = {}
    maps["flat"] = np.full((h, w), 7.25, np.float32)
    maps["ramp"] = (xx * 0.4 + yy * 0.3).astype(np.float32)                    # every neighbour linked
    maps["steps64"] = (np.floor(xx / 64) * 3 + np.floor(yy / 16) * 5).astype(np.float32)   # edges ON the tile grid
    maps["steps63"] = (np.floor((xx + 1) / 63) * 3 + np.floor((yy + 3) / 17) * 5).astype(np.float32)
    maps["stripes_v"] = ((xx.astype(np.int32) % 3) * 2.5).astype(np.float32)
    maps["stripes_h"] = ((yy.astype(np.int32) % 2) * 4.0).astype(np.float32)
    maps["checker"] = (((xx.astype(np.int32) // 5 + yy.astype(np.int32) // 3) % 2) * 9.0).astype(np.float32)
    noise = rng.random((h, w), dtype=np.float32)
    maps["noise_thr"] = (10 + noise * 2.2).astype(np.float32)                  # links flip around the threshold
    holes = (xx * 0.2).astype(np.float32)
    holes[rng.random((h, w)) < 0.3] = inf
    maps["holes"] = holes
    diag = np.where((xx.astype(np.int32) + yy.astype(np.int32)) % 7 == 0, np.float32(50), inf).astype(np.float32)
    maps["diagonals"] = diag                                                   # 8-connectivity only
    blobs = np.floor(rng.random((h // 8 + 1, w // 8 + 1)) * 6).astype(np.float32).repeat(8, 0).repeat(8, 1)[:h, :w] * 3
    blobs[rng.random((h, w)) < 0.05] = inf
    maps["blobs"] = np.ascontiguousarray(blobs)
    return maps


@pytest.mark.parametrize("tile_rows", ["16", "32", "64"])
@pytest.mark.parametrize("shape", [(64, 16), (65, 33), (200, 150), (257, 130), (130, 70), (640, 48)])
def test_speckle_and_median_on_crafted_maps(oracle, shape, tile_rows, monkeypatch):
    """Speckle removal (two-level union-find with implied-union skipping) + in-place median on crafted disparity
    maps, through sgm_tile_post (the library's whole-frame post-filter entry) against the oracle's BFS."""
    import torch
    import soc_project_stereo_matching_amd as S
    w, h = shape
    monkeypatch.setenv("SGM_SPECKLE_TILE_ROWS", tile_rows)
    rng = np.random.default_rng(w * 1000 + h)
    i = S.SGMInstance(0)
    try:
        for min_area in (1, 9, 50, 700, 65535):
            opt = S.default_option(16, min_speckle_area=min_area)
            assert i.reset(w, h, opt)
            for name, m in _speckle_maps(rng, h, w).items():
                want = oracle.median(oracle.remove_speckles(m.copy(), min_area))
                t = torch.from_numpy(m.copy()).cuda()
                torch.cuda.synchronize()
                assert i.tile_post(t.data_ptr()) and i.synchronize()
                assert_same(t.cpu().numpy(), want, f"{name} {w}x{h} min_area={min_area}")
    finally:
        i.close()


def _big_cases():
    import json
    import os
    from conftest import ROOT
    path = os.path.join(ROOT, "tests", "golden", "cases_big.json")
    if not os.path.exists(path):
        return []
    with open(path) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _big_cases(), ids=lambda c: c["name"])
def test_full_size_digests(case):
    """BASELINE.json's two large shapes (2880x1988 D=256 = the maximum volume the library accepts short of 2^32
    cells; 1762x800 D=192) against digests the REFERENCE produced for every stage (tests/golden/cases_big.json,
    minutes of CPU time, generated in the build container)."""
    import gc
    import soc_project_stereo_matching_amd as S
    w, h, d = case["w"], case["h"], case["d"]
    left, right = S.synth_pair(w, h, d, case["seed"])
    assert sha(left) == case["sha256_inputs"]["left"] and sha(right) == case["sha256_inputs"]["right"]
    i = S.SGMInstance(0)
    try:
        i.keep_stages(True)
        assert i.reset(w, h, option_from_dict(case["option"]))
        out = i.match(left, right)
        assert out is not None
        assert sha(out) == case["sha256"]["final"], "final"
        assert int(np.isinf(out).sum()) == case["invalid_final"]
        for name in STAGE_NAMES:
            if name == "final":
                continue
            got = i.read_stage(name)
            assert sha(got) == case["sha256"][name], name
            del got
            gc.collect()
    finally:
        i.close()


def test_batched_frames_one_launch_per_stage(oracle):
    """sgm_set_batch: every kernel processes all frames of a batch in one launch; each frame must still equal
    the oracle on every stage (different content per frame; incl. a W < H shape that clears the planes)."""
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    for (w, h, d, B) in [(200, 64, 64, 3), (30, 70, 8, 4), (1242, 375, 128, 2)]:
        opt = default_option(d, min_speckle_area=12)
        inst = S.SGMInstance(0, batch=B)
        inst.keep_stages(True)
        frames = [oracle.synth_pair(w, h, d, 0xBA7C00 + 31 * k + w) for k in range(B)]
        left = np.stack([f[0] for f in frames])
        right = np.stack([f[1] for f in frames])
        assert inst.reset(w, h, opt)
        out = inst.match(left, right)
        assert out is not None and out.shape == (B, h, w)
        for k in range(B):
            want = oracle.run(frames[k][0], frames[k][1], opt)
            inst.select_frame(k)
            got = inst.read_stages()
            for n in STAGE_NAMES:
                assert_same(got[n], want[n], f"batch {w}x{h} frame {k}: {n}")
            assert_same(out[k], want["final"], f"batch {w}x{h} frame {k}: result")
        inst.close()


def _corner_cases():
    from test_oracle_vs_reference import _cases
    return _cases()


@pytest.mark.parametrize("case", _corner_cases(), ids=lambda c: f"{c[0]}x{c[1]}_d{c[2]}-{c[3]}_s{c[4]}")
def test_random_options_and_degenerate_shapes(inst, oracle, case):
    """The option / shape sweep the oracle is pinned on against the reference (test_oracle_vs_reference.py), incl.
    one-row / one-column images, 2x2, census no-op sizes, D = 1, negative and huge penalties."""
    w, h, dmin, dmax, seed, kw = case
    from oracle.pyoracle import default_option
    left, right = oracle.synth_pair(w, h, dmax - dmin, seed)
    opt = default_option(dmax, dmin, **kw)
    want = oracle.run(left, right, opt)
    assert inst.reset(w, h, opt)
    out = inst.match(left, right)
    assert out is not None
    got = inst.read_stages()
    for n in STAGE_NAMES:
        if n == "disp_r" and not opt.is_check_lr:
            continue
        assert_same(got[n], want[n], f"{case}:{n}")
    assert_same(out, want["final"], f"{case}:result")


def test_c_driver_reproduces_the_reference_fixture(tmp_path):
    """soc_project_stereo_matching_amd/sgm_main = the reference's main.c flow in C (load -> SGM_Initialize ->
    SGM_Match -> normalise -> PNG).  On the cone pair its PNG equals the reference's committed
    Data/cone/im2.d.png except the one pixel of SURVEY.md Q6, and its raw disparities equal the golden floats."""
    import os
    import subprocess
    from PIL import Image
    from conftest import ROOT
    exe = os.path.join(ROOT, "soc_project_stereo_matching_amd", "sgm_main")
    z = load_npz("cone_inputs.npz")
    Image.fromarray(z["left"]).save(str(tmp_path / "im2.pgm"))
    Image.fromarray(z["right"]).save(str(tmp_path / "im6.png"))          # one PGM, one PNG: both readers
    out_png, out_raw = str(tmp_path / "im2.d.png"), str(tmp_path / "im2.d.f32")
    subprocess.check_call([exe, str(tmp_path / "im2.pgm"), str(tmp_path / "im6.png"), out_png, "--raw", out_raw])
    got = np.asarray(Image.open(out_png))
    assert np.argwhere(got != z["im2_d_png"]).tolist() == [[374, 153]]
    raw = np.fromfile(out_raw, np.float32).reshape(375, 450)
    assert_same(raw, load_npz("cone_final.npz")["final"], "driver raw disparities")


def test_c_driver_on_a_colour_scene(tmp_path):
    """sgm_main on the RGB files of a second scene (Data/Reindeer view1 / view5, 671x555, stored as arrays by
    make_golden_scenes.py and written back as an RGB PNG and an RGB PPM here): the driver's own colour -> grey conversion
    (stb's formula, main.c's loader) must give the grey images the reference's loader gave, and the raw disparities the
    final map of the compiled reference for that pair (max_disparity 128 from drange.txt)."""
    import json
    import os
    import subprocess
    from PIL import Image
    from conftest import GOLDEN, ROOT
    exe = os.path.join(ROOT, "soc_project_stereo_matching_amd", "sgm_main")
    z = load_npz("scene_reindeer.npz")
    with open(os.path.join(GOLDEN, "cases_scenes.json")) as f:
        case = {c["name"]: c for c in json.load(f)["cases"]}["scene_reindeer"]
    h, w = z["left"].shape
    Image.fromarray(z["rgb_left"], "RGB").save(str(tmp_path / "view1.png"))
    Image.fromarray(z["rgb_right"], "RGB").save(str(tmp_path / "view5.ppm"))
    for src, want in (("view1.png", z["left"]), ("view5.ppm", z["right"])):       # the readers alone (--convert)
        subprocess.check_call([exe, "--convert", str(tmp_path / src), str(tmp_path / "g.pgm")])
        assert np.array_equal(np.asarray(Image.open(str(tmp_path / "g.pgm"))), want), src
    out_png, out_raw = str(tmp_path / "d.png"), str(tmp_path / "d.f32")
    subprocess.check_call([exe, str(tmp_path / "view1.png"), str(tmp_path / "view5.ppm"), out_png, "--max-disparity", "128", "--raw", out_raw])
    raw = np.fromfile(out_raw, np.float32).reshape(h, w)
    assert sha(raw) == case["sha256"]["final"]
    assert_same(raw, z["final"], "driver raw disparities, Reindeer")
    assert int(np.isinf(raw).sum()) == case["invalid_final"]


def _bench_frames(workload):
    import json
    import os
    from conftest import ROOT
    path = os.path.join(ROOT, "tests", "golden", "bench_frames.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)["workloads"].get(workload)


def test_batched_fast_path_kitti_against_reference_digests():
    """The configuration bench.py times: batches of 8 KITTI frames, keep_stages OFF (the non-SLOW instantiation of the
    fused cost-sum/WTA kernel, S never stored), device-resident frames, TWO instances interleaved on their own
    streams.  Every frame's final map and right-view map against the digests the reference's own C produced for the
    16 seeds bench.py uses (tests/golden/bench_frames.json)."""
    import torch
    import soc_project_stereo_matching_amd as S
    wl = _bench_frames("kitti_1242x375_d128_p8")
    assert wl is not None and len(wl["frames"]) >= 16
    w, h, d, seed, B = wl["w"], wl["h"], wl["d"], wl["first_seed"], 8
    opt = S.default_option(d)
    insts = [S.SGMInstance(0, batch=B) for _ in range(2)]
    ins, outs = [], []
    for k in range(2):
        ps = [S.synth_pair(w, h, d, seed + k * B + j) for j in range(B)]
        for j, p in enumerate(ps):
            e = wl["frames"][str(seed + k * B + j)]["sha256_inputs"]
            assert sha(p[0]) == e["left"] and sha(p[1]) == e["right"]
        ins.append((torch.from_numpy(np.stack([p[0] for p in ps])).cuda(), torch.from_numpy(np.stack([p[1] for p in ps])).cuda()))
        outs.append(torch.empty((B, h, w), dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    for i in insts:
        i.keep_stages(False)
        assert i.reset(w, h, opt)
    for rep in range(3):                                   # interleaved, several rounds in flight
        for k in range(2):
            assert insts[k].reset(w, h, opt)
            assert insts[k].match_device(ins[k][0].data_ptr(), ins[k][1].data_ptr(), outs[k].data_ptr())
    for i in insts:
        assert i.synchronize()
    for k in range(2):
        got = outs[k].cpu().numpy()
        for j in range(B):
            e = wl["frames"][str(seed + k * B + j)]["sha256"]
            assert sha(got[j]) == e["final"], f"batch {k} frame {j}: final"
            insts[k].select_frame(j)
            assert sha(insts[k].read_stage("disp_r")) == e["disp_r"], f"batch {k} frame {j}: right view"
    for i in insts:
        i.close()


def test_batched_fast_path_kitti_without_speckle_removal():
    """SURVEY.md 8(d) asks for throughput with speckle removal on and off: the speckle-off rate in bench.py's line is checked against
    these digests -- the reference's own C on the same 16 KITTI frames with is_remove_speckles = false (main.c:60 flipped)."""
    import torch
    import soc_project_stereo_matching_amd as S
    wl = _bench_frames("kitti_1242x375_d128_p8_nospeckle")
    assert wl is not None and len(wl["frames"]) >= 16
    w, h, d, seed, B = wl["w"], wl["h"], wl["d"], wl["first_seed"], 8
    opt = S.default_option(d, is_remove_speckles=False)
    inst = S.SGMInstance(0, batch=B)
    inst.keep_stages(False)
    out = torch.empty((B, h, w), dtype=torch.float32, device="cuda")
    for k in range(2):
        ps = [S.synth_pair(w, h, d, seed + k * B + j) for j in range(B)]
        l, r = torch.from_numpy(np.stack([p[0] for p in ps])).cuda(), torch.from_numpy(np.stack([p[1] for p in ps])).cuda()
        assert inst.reset(w, h, opt) and inst.match_device(l.data_ptr(), r.data_ptr(), out.data_ptr()) and inst.synchronize()
        got = out.cpu().numpy()
        for j in range(B):
            e = wl["frames"][str(seed + k * B + j)]["sha256"]
            assert e["after_speckle"] == e["after_lr"]                    # the reference skipped the stage
            assert sha(got[j]) == e["final"], f"batch {k} frame {j}: final"
    inst.close()


@pytest.mark.parametrize("shape", [(600, 140, 0, 100, 8), (333, 150, 2, 66, 8), (500, 130, 0, 192, 4), (420, 260, 0, 256, 4)])
def test_batched_fast_path_padded_disparity_ranges(oracle, shape):
    """The same fast path (batch, keep_stages off, two instances in flight) where D is not the padded stride (D = 100
    in a 128-wide cell, 64 in 64 with dmin 2), and for the 192 / 256 ranges of the large BASELINE shapes; H*B >= 1024 so
    the fused kernel runs whole rows like the bench's.  Against the oracle."""
    import torch
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    w, h, dmin, dmax, B = shape
    d = dmax - dmin
    opt = default_option(dmax, dmin, min_speckle_area=20)
    insts = [S.SGMInstance(0, batch=B) for _ in range(2)]
    frames = [[oracle.synth_pair(w, h, d, 0xFA57000 + 97 * k + j + w) for j in range(B)] for k in range(2)]
    want = [[oracle.run(l, r, opt) for l, r in fr] for fr in frames]
    ins = [(torch.from_numpy(np.stack([p[0] for p in fr])).cuda(), torch.from_numpy(np.stack([p[1] for p in fr])).cuda()) for fr in frames]
    outs = [torch.empty((B, h, w), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(2):
        for k in range(2):
            insts[k].keep_stages(False)
            assert insts[k].reset(w, h, opt)
            assert insts[k].match_device(ins[k][0].data_ptr(), ins[k][1].data_ptr(), outs[k].data_ptr())
    for k in range(2):
        assert insts[k].synchronize()
        got = outs[k].cpu().numpy()
        for j in range(B):
            assert_same(got[j], want[k][j]["final"], f"{shape} batch {k} frame {j}: final")
            insts[k].select_frame(j)
            assert_same(insts[k].read_stage("disp_r"), want[k][j]["disp_r"], f"{shape} batch {k} frame {j}: right view")
        insts[k].close()


@pytest.mark.parametrize("pinned", [False, True])
def test_async_host_pointer_pipeline(oracle, pinned):
    """sgm_match_async / sgm_match_wait (the pipelined host-pointer boundary): three instances round-robined, batches
    of 2 frames, caller buffers pageable (staged) or from sgm_host_alloc (used in place); every result against the
    oracle; a Reset to another shape with a match still pending must hand that match over first."""
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    w, h, d, B = 320, 96, 64, 2
    opt = default_option(d)
    insts = [S.SGMInstance(0, batch=B) for _ in range(3)]
    for i in insts:
        assert i.reset(w, h, opt)
    mk = (lambda i, shp, dt: i.host_array(shp, dt)) if pinned else (lambda i, shp, dt: np.empty(shp, dt))
    bufs = [(mk(i, (B, h, w), np.uint8), mk(i, (B, h, w), np.uint8), mk(i, (B, h, w), np.float32)) for i in insts]
    n_rounds = 7
    wants, gots = {}, {}
    for k in range(n_rounds):
        i, (L, R, O) = insts[k % 3], bufs[k % 3]
        if k >= 3:
            assert i.match_wait()
            gots[k - 3] = O.copy()
        for j in range(B):
            l, r = oracle.synth_pair(w, h, d, 0xA5C000 + k * B + j)
            L[j], R[j] = l, r
            wants[(k, j)] = oracle.run(l, r, opt)["final"]
        assert i.reset(w, h, opt)
        assert i.match_async(L, R, O)
    for k in range(n_rounds - 3, n_rounds):
        if k == n_rounds - 1:
            assert insts[k % 3].reset(w + 8, h, opt)          # another shape: waits for and delivers the pending match
        else:
            assert insts[k % 3].match_wait()
        gots[k] = bufs[k % 3][2].copy()
    for k in range(n_rounds):
        for j in range(B):
            assert_same(gots[k][j], wants[(k, j)], f"round {k} frame {j}")
    for i in insts:
        i.close()


def test_sgm_compute_is_reset_plus_match(gsgm, oracle):
    """north_star's one-call entry: sgm_compute(left, right, w, h, &opt, disp) == SGM_Reset + SGM_Match; a second call
    must not accumulate onto the first (Q14) and argument errors return false like the two calls."""
    import ctypes as C
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    l, r = oracle.synth_pair(200, 60, 32, 0xC0FFEE)
    l2, r2 = oracle.synth_pair(200, 60, 32, 0xC0FFEF)
    opt = default_option(32, min_speckle_area=10)
    assert_same(gsgm.compute(l, r, opt), oracle.run(l, r, opt)["final"], "first")
    assert_same(gsgm.compute(l2, r2, opt), oracle.run(l2, r2, opt)["final"], "second (reset in between)")
    lib = S.load_library()
    out = np.empty((60, 200), np.float32)
    assert not lib.sgm_compute(l.ctypes.data, r.ctypes.data, 0, 60, C.byref(opt), out.ctypes.data)
    assert not lib.sgm_compute(None, r.ctypes.data, 200, 60, C.byref(opt), out.ctypes.data)
    assert not lib.sgm_compute(l.ctypes.data, r.ctypes.data, 200, 60, C.byref(default_option(5, 9)), out.ctypes.data)


@pytest.mark.parametrize("window", [(7, 7), (9, 7), (3, 3), (13, 3), (5, 5)])
@pytest.mark.parametrize("shape", [(130, 47, 2, 50), (257, 64, 0, 100), (37, 61, 0, 8), (320, 96, 0, 64)])
def test_census_window_extension(oracle, window, shape):
    """census windows other than the reference's 5x5 (u64 words, materialised cost volume, volume-fed
    aggregation).  The reference has no such option: the oracle's sgmo_census_window defines it ("parity unpinned by the
    reference"); every stage against the oracle, and 5x5 through this API must be the reference path again."""
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    w, h, dmin, dmax = shape
    cw, ch = window
    left, right = oracle.synth_pair(w, h, dmax - dmin, 0xCE115 + w + cw)
    opt = default_option(dmax, dmin, min_speckle_area=9)
    i = S.SGMInstance(0)
    try:
        assert oracle.set_census_window(cw, ch) and i.set_census_window(cw, ch)
        want = oracle.run(left, right, opt)
        i.keep_stages(True)
        assert i.reset(w, h, opt)
        out = i.match(left, right)
        assert out is not None
        got = i.read_stages()
        for n in STAGE_NAMES:
            assert_same(got[n], want[n], f"{window} {shape}:{n}")
        assert_same(out, want["final"], f"{window} {shape}:result")
    finally:
        oracle.set_census_window(5, 5)
        i.close()


def test_census_window_in_batches_and_tiles(oracle):
    import torch
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    from soc_project_stereo_matching_amd.tiling import DeviceTileEngine, match_tiled_in_process, tile_rows
    w, h, d, B = 200, 64, 48, 3
    opt = default_option(d, min_speckle_area=12)
    frames = [oracle.synth_pair(w, h, d, 0x7A7 + k) for k in range(B)]
    assert oracle.set_census_window(9, 7)
    try:
        wants = [oracle.run(l, r, opt)["final"] for l, r in frames]
        inst = S.SGMInstance(0, batch=B)
        assert inst.set_census_window(9, 7) and inst.reset(w, h, opt)
        out = inst.match(np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]))
        for k in range(B):
            assert_same(out[k], wants[k], f"batch frame {k}")
        inst.close()
        engines = []
        for rows in tile_rows(h, 3):
            e = DeviceTileEngine.__new__(DeviceTileEngine)
            e.torch, e.dev = torch, torch.device("cuda", 0)
            e.w, e.h, e.rows, e.option = w, h, rows, opt
            e.inst = S.SGMInstance(0)
            assert e.inst.set_census_window(9, 7) and e.inst.set_rows(*rows) and e.inst.reset(w, h, opt)
            e.disp = torch.empty((h, w), dtype=torch.float32, device=e.dev)
            e.nbytes = e.inst.tile_boundary_bytes()
            engines.append(e)
        got = match_tiled_in_process(engines, torch.from_numpy(frames[0][0]).cuda(), torch.from_numpy(frames[0][1]).cuda())
        assert_same(got.cpu().numpy(), wants[0], "three row tiles")
        for e in engines:
            e.close()
    finally:
        oracle.set_census_window(5, 5)


@pytest.mark.parametrize("check_lr", [True, False])
@pytest.mark.parametrize("shape", [(130, 47, 2, 50), (450, 375, 0, 64), (600, 40, 0, 300), (37, 61, 0, 8)])
def test_right_reference_view_extension(oracle, gsgm, shape, check_lr):
    """the right image as the reference view -- the right-view WTA map (reference .c:395-408) validated by
    the mirror image of LRCheck, then speckle + median.  Defined by the oracle only ("parity unpinned by the reference").
    Through the global entry points (SGM_SetReferenceView) incl. D > 256 (separate sum / right-view kernels)."""
    from oracle.pyoracle import default_option
    w, h, dmin, dmax = shape
    left, right = oracle.synth_pair(w, h, dmax - dmin, 0x81647 + w)
    opt = default_option(dmax, dmin, min_speckle_area=9, is_check_lr=check_lr)
    oracle.set_reference_view(True)
    gsgm.set_reference_view(True)
    try:
        want = oracle.run(left, right, opt)
        gsgm.keep_stages(True)
        assert gsgm.reset(w, h, opt)
        gsgm.keep_stages(True)
        out = gsgm.match(left, right)
        assert_same(gsgm.read_stage("disp_r"), want["disp_r"], f"{shape}: right-view WTA")
        assert_same(gsgm.read_stage("after_lr"), want["after_lr"], f"{shape}: mirrored LR check")
        assert_same(out, want["final"], f"{shape}: result")
        assert not np.array_equal(out.view(np.uint32), oracle.run(left, right, opt)["disp_l"].view(np.uint32))
    finally:
        oracle.set_reference_view(False)
        gsgm.set_reference_view(False)
        gsgm.keep_stages(False)
    # and back: the reference behaviour is untouched
    assert gsgm.reset(w, h, opt)
    assert_same(gsgm.match(left, right), oracle.run(left, right, opt)["final"], "left view again")


@pytest.mark.parametrize("chain", ["1", "0"])
@pytest.mark.parametrize("shape", [(64, 600), (300, 1500), (257, 2160), (1242, 515), (40, 4100)])
def test_median_of_tall_frames(oracle, shape, chain, monkeypatch):
    """Frames taller than one median band (8 waves x 64 rows): the bands run as a chain of workgroups that hand their last
    row down through tagged granules in global memory (SGM_MEDIAN_CHAIN=1, default), or one after the other in one
    workgroup (=0).  Crafted maps through the post-filter entry, twice on the same instance (the granule generation)."""
    import torch
    import soc_project_stereo_matching_amd as S
    w, h = shape
    monkeypatch.setenv("SGM_MEDIAN_CHAIN", chain)
    rng = np.random.default_rng(w + h)
    i = S.SGMInstance(0)
    try:
        opt = S.default_option(16, is_remove_speckles=False)
        assert i.reset(w, h, opt)
        maps = _speckle_maps(rng, h, w)
        for name in ("ramp", "noise_thr", "holes", "blobs", "checker"):
            m = maps[name]
            want = oracle.median(m.copy())
            for rep in range(2):
                t = torch.from_numpy(m.copy()).cuda()
                torch.cuda.synchronize()
                assert i.tile_post(t.data_ptr()) and i.synchronize()
                assert_same(t.cpu().numpy(), want, f"{name} {w}x{h} chain={chain} rep {rep}")
    finally:
        i.close()


def test_tall_frames_in_a_batch(oracle):
    """Chained median bands with several frames per launch (one band ticket counter per frame)."""
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    w, h, d, B = 90, 700, 8, 3
    opt = default_option(d, min_speckle_area=12)
    frames = [oracle.synth_pair(w, h, d, 0x7A11 + k) for k in range(B)]
    inst = S.SGMInstance(0, batch=B)
    try:
        assert inst.reset(w, h, opt)
        for rep in range(2):
            out = inst.match(np.stack([f[0] for f in frames]), np.stack([f[1] for f in frames]))
            for k in range(B):
                assert_same(out[k], oracle.run(frames[k][0], frames[k][1], opt)["final"], f"frame {k} rep {rep}")
    finally:
        inst.close()


def test_depth_and_scoring_on_device(oracle):
    """SURVEY.md 8(f)-3 on device buffers: disparity -> depth in mm and RMSE / bad-pixel rate / valid count, against the host
    restatement oracle/platform_oracle.py (the reference's depth_image.py imports cv2, which is not
    installed: no reference-made vectors, "parity unpinned").  Depth: bit-exact (one float32 divide).  Scores: counts exact,
    rmse within 1e-6 relative (numpy sums float32 squares pairwise, the device sums doubles in block order)."""
    import torch
    import soc_project_stereo_matching_amd as S
    from oracle.platform_oracle import compare_depth, disparity_to_depth
    w, h, d = 640, 200, 64
    left, right = oracle.synth_pair(w, h, d, 0xDE97)
    inst = S.SGMInstance(0)
    try:
        assert inst.reset(w, h, S.default_option(d))
        disp = inst.match(left, right)                               # +inf where invalid
        fx, baseline, doffs = 3979.911, 193.001, 124.343             # a Middlebury calib.txt's magnitudes
        t_disp = torch.from_numpy(disp).cuda()
        t_depth = torch.empty_like(t_disp)
        torch.cuda.synchronize()
        assert inst.disparity_to_depth(t_disp.data_ptr(), disp.size, fx, baseline, doffs, t_depth.data_ptr()) and inst.synchronize()
        got = t_depth.cpu().numpy()
        want = disparity_to_depth(disp, fx, baseline, doffs)
        assert np.array_equal(np.isnan(got), np.isnan(want)) and int(np.isnan(want).sum()) > 0
        ok = ~np.isnan(want)
        assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))
        # a denominator of exactly zero and a NaN disparity also give NaN
        odd = torch.tensor([-doffs, float("nan"), float("inf"), 10.0], dtype=torch.float32).cuda()
        out = torch.empty_like(odd)
        assert inst.disparity_to_depth(odd.data_ptr(), 4, fx, baseline, float(np.float32(doffs)), out.data_ptr()) and inst.synchronize()
        assert torch.isnan(out[:3]).all() and torch.isfinite(out[3])
        # scoring against a perturbed "ground truth" with holes
        rng = np.random.default_rng(5)
        gt = (want + rng.normal(0, 8, want.shape).astype(np.float32)).astype(np.float32)
        gt[rng.random(want.shape) < 0.1] = np.nan
        t_gt = torch.from_numpy(gt).cuda()
        torch.cuda.synchronize()
        rmse, bpr, n = inst.compare_depth(t_gt.data_ptr(), t_depth.data_ptr(), want.size, 10.0)
        w_rmse, w_bpr, w_n = compare_depth(gt, want, 10.0)
        assert n == w_n and abs(bpr - w_bpr) < 1e-12 and abs(rmse - w_rmse) <= 1e-6 * w_rmse
        # nothing valid in common -> (nan, nan, 0)
        none = torch.full_like(t_gt, float("nan"))
        r2 = inst.compare_depth(none.data_ptr(), t_depth.data_ptr(), want.size, 10.0)
        assert r2[2] == 0 and np.isnan(r2[0]) and np.isnan(r2[1])
    finally:
        inst.close()


def test_device_depth_and_scores_equal_the_references_own_functions():
    """The same two device entries against vectors the REFERENCE'S OWN `disparity_to_depth` / `compare_img` produced
    (tests/golden/platform_depth.npz; make_golden_depth.py compiles those two numpy-only functions from depth_image.py's text, the
    module itself needs cv2): depth bit for bit wherever the denominator is finite and non-zero -- NaN by design elsewhere, where the
    reference's bare formula gives 0 or inf --, valid count and bad-pixel rate exact, RMSE within 2e-6 relative (float32 pairwise mean
    there, float64 sums here)."""
    import torch
    import soc_project_stereo_matching_amd as S
    from test_platform_oracle import _depth_cases, check_depth_against_reference
    z = load_npz("platform_depth.npz")
    inst = S.SGMInstance(0)
    try:
        n = 0
        for what, disp, fx, baseline, doffs, ref in _depth_cases():
            t_disp = torch.from_numpy(np.ascontiguousarray(disp)).cuda()
            t_depth = torch.empty_like(t_disp)
            torch.cuda.synchronize()
            assert inst.disparity_to_depth(t_disp.data_ptr(), disp.size, fx, baseline, doffs, t_depth.data_ptr()) and inst.synchronize()
            check_depth_against_reference(t_depth.cpu().numpy(), disp, doffs, ref, what)
            n += 1
        assert n == 7
        for k in range(int(z["n_scores"][0])):
            t_gt, t_te = torch.from_numpy(z[f"score_{k}_gt"]).cuda(), torch.from_numpy(z[f"score_{k}_test"]).cuda()
            torch.cuda.synchronize()
            for rmse_r, bpr_r, n_r, thr in z[f"score_{k}_results"]:
                rmse, bpr, nv = inst.compare_depth(t_gt.data_ptr(), t_te.data_ptr(), t_gt.numel(), float(thr))
                assert nv == int(n_r) and abs(bpr - bpr_r) < 1e-12 and abs(rmse - rmse_r) <= 2e-6 * rmse_r, (k, thr, rmse, rmse_r)
    finally:
        inst.close()


@pytest.mark.parametrize("split", [None, "post=0:4,main=4:28", "post=0:2,sum=2:10,main=12:20", "sum=0:0", "sum=0:8"])
@pytest.mark.parametrize("batch", [1, 4])
def test_overlapped_post_pass(oracle, batch, split):
    """sgm_set_overlap_post: LR check / speckle / median of a match on the instance's second stream beside the next match's
    aggregation -- and sgm_set_stage_cus (`split`): the cost sum and / or the post pass on streams of their own, restricted to
    some compute units of every XCD, the main stream to the others.  A stream of matches on ONE instance with one output buffer per match, results read after one
    sgm_synchronize at the end; then the same instance without Reset (Q14 accumulation through the lazy S materialisation,
    which shares scratch with the post pass), through the host-pointer async entry, and with the option switched off again."""
    import torch
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import Oracle, default_option
    w, h, d, n = 400, 120, 64, 5
    opt = default_option(d, min_speckle_area=20)
    inst = S.SGMInstance(0, batch=batch)
    assert inst.set_cu_split(split) if split else inst.set_overlap_post(True)
    try:
        frames = [[oracle.synth_pair(w, h, d, 0x0FE7 + 10 * k + j) for j in range(batch)] for k in range(n)]
        ins = [(torch.from_numpy(np.stack([p[0] for p in fr])).cuda(), torch.from_numpy(np.stack([p[1] for p in fr])).cuda()) for fr in frames]
        outs = [torch.empty((batch, h, w), dtype=torch.float32, device="cuda") for _ in range(n)]
        torch.cuda.synchronize()
        for k in range(n):
            assert inst.reset(w, h, opt)
            assert inst.match_device(ins[k][0].data_ptr(), ins[k][1].data_ptr(), outs[k].data_ptr())
        assert inst.synchronize()
        for k in range(n):
            got = outs[k].cpu().numpy()
            for j in range(batch):
                assert_same(got[j], oracle.run(frames[k][j][0], frames[k][j][1], opt)["final"], f"match {k} frame {j}")
        # Q14 with the post pass still in flight: three matches without Reset on the same output buffer
        if batch == 1:
            orc = Oracle()
            assert orc.reset(w, h, opt) and inst.reset(w, h, opt)
            for k in range(3):
                want = orc.match(frames[k][0][0], frames[k][0][1])
                assert inst.match_device(ins[k][0].data_ptr(), ins[k][1].data_ptr(), outs[0].data_ptr())
                assert inst.synchronize()
                assert_same(outs[0].cpu().numpy()[0], want, f"no-reset match {k}")
        # host pointers, pipelined
        assert inst.reset(w, h, opt)
        L = np.stack([p[0] for p in frames[1]]) if batch > 1 else frames[1][0][0]
        R = np.stack([p[1] for p in frames[1]]) if batch > 1 else frames[1][0][1]
        O = np.empty(L.shape, np.float32)
        assert inst.match_async(np.ascontiguousarray(L), np.ascontiguousarray(R), O) and inst.match_wait()
        for j in range(batch):
            assert_same(O[j] if batch > 1 else O, oracle.run(frames[1][j][0], frames[1][j][1], opt)["final"], f"async frame {j}")
        # ... and off again: stream order of sgm_stream alone
        assert inst.set_overlap_post(False)
        for which in (inst.STAGE_SUM, inst.STAGE_POST, inst.STAGE_MAIN):
            assert inst.set_stage_cus(which, 0, -1)
        assert inst.reset(w, h, opt)
        assert inst.match_device(ins[2][0].data_ptr(), ins[2][1].data_ptr(), outs[2].data_ptr()) and inst.synchronize()
        assert_same(outs[2].cpu().numpy()[0], oracle.run(frames[2][0][0], frames[2][0][1], opt)["final"], "overlap off")
    finally:
        inst.close()


def _colour_planes(oracle, w, h, d, seed, rng):
    """Six planes (left B, G, R, right B, G, R) whose board-grey is close to a synthetic stereo pair, channels perturbed
    so that the three weights matter."""
    l, r = oracle.synth_pair(w, h, d, seed)
    planes = np.empty((6, h, w), np.uint8)
    for v, g in enumerate((l, r)):
        for c in range(3):
            planes[3 * v + c] = np.clip(g.astype(np.int32) + rng.integers(-6, 7, (h, w)), 0, 255)
    return planes


@pytest.mark.parametrize("n", [4096, 1280 * 720, 1001, 3])
@pytest.mark.parametrize("weight_r", [76, 77])
def test_gray_from_planes(n, weight_r):
    """The firmware's grey conversion (stereo_matching.c:18-25; weight 77: stb_image.h:1746-1749) of three byte planes on the
    device, all 256 values per channel present; dword path (n % 4 == 0) and byte path."""
    import torch
    import soc_project_stereo_matching_amd as S
    from oracle.platform_oracle import board_gray
    rng = np.random.default_rng(n + weight_r)
    bgr = rng.integers(0, 256, (3, n), dtype=np.uint8)
    bgr[:, :3] = [[255, 0, 255], [255, 0, 0], [255, 255, 0]]                  # the extremes
    t = torch.from_numpy(bgr).cuda()
    out = torch.zeros(n + 8, dtype=torch.uint8, device="cuda")               # canary behind the result
    torch.cuda.synchronize()
    inst = S.SGMInstance(0)
    try:
        assert inst.gray_from_planes(t.data_ptr(), n, out.data_ptr(), weight_r) and inst.synchronize()
        got = out.cpu().numpy()
        assert np.array_equal(got[:n], board_gray(bgr[0], bgr[1], bgr[2], weight_r)) and not got[n:].any()
        assert not inst.gray_from_planes(t.data_ptr(), n, out.data_ptr(), 75)
    finally:
        inst.close()


@pytest.mark.parametrize("batch,overlap,pinned", [(1, False, False), (1, True, True), (2, True, False), (3, False, True)])
def test_match_planes_depth(oracle, batch, overlap, pinned):
    """A test-platform frame end to end on the device: six colour planes in host memory -> grey -> SGM ->
    depth in mm in host memory.  Expected: the oracle's disparity for the board-grey images pushed through the platform's
    depth formula (oracle/platform_oracle.py), bit for bit; the disparity map stays readable as stage 8; a stream of frames on one
    instance, pageable and pinned buffers, with and without the post pass on the second stream."""
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    from oracle.platform_oracle import board_gray, disparity_to_depth
    w, h, d = 322, 97, 48                                                    # odd row pitch: frames of a batch are not dword-aligned
    fx, baseline, doffs = 1733.74, 536.62, 0.0
    opt = S.default_option(d)
    rng = np.random.default_rng(97)
    inst = S.SGMInstance(0, batch=batch)
    try:
        assert inst.set_overlap_post(overlap)
        shape_in, shape_out = ((batch, 6, h, w), (batch, h, w)) if batch > 1 else ((6, h, w), (h, w))
        buf_in = inst.host_array(shape_in, np.uint8) if pinned else np.empty(shape_in, np.uint8)
        buf_out = inst.host_array(shape_out, np.float32) if pinned else np.empty(shape_out, np.float32)
        for rep in range(3):
            planes = np.stack([_colour_planes(oracle, w, h, d, 0x9A00 + 10 * rep + j, rng) for j in range(batch)])
            buf_in[...] = planes if batch > 1 else planes[0]
            assert inst.reset(w, h, opt)
            assert inst.match_planes(buf_in, fx, baseline, doffs, buf_out, wait=(rep != 1))
            if rep == 1:
                assert inst.match_wait()
            got = buf_out.reshape(batch, h, w)
            for j in range(batch):
                gl = board_gray(planes[j, 0], planes[j, 1], planes[j, 2])
                gr = board_gray(planes[j, 3], planes[j, 4], planes[j, 5])
                disp = oracle.run(gl, gr, default_option(d))["final"]
                want = disparity_to_depth(disp, fx, baseline, doffs)
                assert np.array_equal(np.isnan(got[j]), np.isnan(want)), (rep, j)
                ok = ~np.isnan(want)
                assert np.array_equal(got[j][ok].view(np.uint32), want[ok].view(np.uint32)), (rep, j)
                if batch == 1:
                    assert_same(inst.read_stage("final"), disp, "disparity behind the depth map")
    finally:
        inst.close()


@pytest.mark.parametrize("first", ["match", "match_async"])
def test_match_planes_after_a_chunked_match_on_the_same_instance(oracle, first):
    """A staged disparity map of >= 256 KB comes back in pieces, an event behind each (sgm_match_async); the platform-frame entry
    queues ONE copy.  After such a match on the same instance sgm_match_planes must not walk the earlier match's chunk events
    (they completed long ago: the caller would get most of the map before this match's copy has finished).  Pageable buffers,
    a map of 410 KB, several rounds so a stale hand-over would show the previous frame's depth."""
    import soc_project_stereo_matching_amd as S
    from oracle.pyoracle import default_option
    from oracle.platform_oracle import board_gray, disparity_to_depth
    w, h, d = 640, 160, 32
    assert w * h * 4 >= 256 << 10
    fx, baseline, doffs = 1733.74, 536.62, 0.0
    opt = S.default_option(d)
    rng = np.random.default_rng(1262)
    inst = S.SGMInstance(0)
    try:
        for rep in range(3):
            l, r = oracle.synth_pair(w, h, d, 0xC400 + rep)
            assert inst.reset(w, h, opt)
            if first == "match":
                out = inst.match(l, r)
                assert out is not None
            else:
                out = np.empty((h, w), np.float32)
                assert inst.match_async(l, r, out) and inst.match_wait()
            assert_same(out, oracle.run(l, r, default_option(d))["final"], f"chunked match {rep}")
            planes = _colour_planes(oracle, w, h, d, 0xC500 + rep, rng)
            depth = np.full((h, w), -1.0, np.float32)
            assert inst.reset(w, h, opt)
            assert inst.match_planes(np.ascontiguousarray(planes), fx, baseline, doffs, depth)
            gl, gr = board_gray(planes[0], planes[1], planes[2]), board_gray(planes[3], planes[4], planes[5])
            want = disparity_to_depth(oracle.run(gl, gr, default_option(d))["final"], fx, baseline, doffs)
            assert np.array_equal(np.isnan(depth), np.isnan(want)), rep
            ok = ~np.isnan(want)
            assert np.array_equal(depth[ok].view(np.uint32), want[ok].view(np.uint32)), rep
    finally:
        inst.close()


@pytest.mark.parametrize("flags,setup", [
    (["--paths", "4"], dict(honor=True, num_paths=4)),
    (["--census", "7x7"], dict(window=(7, 7))),
    (["--right-reference"], dict(right=True)),
    (["--census", "9x7", "--right-reference", "--min-disparity", "3", "--max-disparity", "40"], dict(window=(9, 7), right=True, dmin=3, dmax=40)),
])
def test_c_driver_extension_flags(tmp_path, oracle, flags, setup):
    """sgm_main's --paths / --census / --right-reference (SURVEY.md 8f-4 as command-line options): the raw disparities equal
    the oracle's with the same extension switched on (the reference has none of them: parity unpinned by the reference)."""
    import os
    import subprocess
    from PIL import Image
    from conftest import ROOT
    from oracle.pyoracle import default_option
    exe = os.path.join(ROOT, "soc_project_stereo_matching_amd", "sgm_main")
    w, h = 203, 77
    dmin, dmax = setup.get("dmin", 0), setup.get("dmax", 64)
    left, right = oracle.synth_pair(w, h, dmax - dmin, 0xC11)
    Image.fromarray(left).save(str(tmp_path / "l.png"))
    Image.fromarray(right).save(str(tmp_path / "r.pgm"))
    out_raw = str(tmp_path / "d.f32")
    base = ["--max-disparity", "64"] if "--max-disparity" not in flags else []
    subprocess.check_call([exe, str(tmp_path / "l.png"), str(tmp_path / "r.pgm"), str(tmp_path / "d.png"), "--raw", out_raw] + base + flags)
    opt = default_option(dmax, dmin, num_paths=setup.get("num_paths", 8))
    try:
        oracle.set_honor_num_paths(setup.get("honor", False))
        oracle.set_census_window(*setup.get("window", (5, 5)))
        oracle.set_reference_view(setup.get("right", False))
        want = oracle.run(left, right, opt)["final"]
    finally:
        oracle.set_honor_num_paths(False)
        oracle.set_census_window(5, 5)
        oracle.set_reference_view(False)
    assert_same(np.fromfile(out_raw, np.float32).reshape(h, w), want, f"driver {flags}")
