"""Stage order and error paths of the product's C host (csrc/sgm_host.c) -- no GPU.

sgm_host.c is linked with tests/stub_device.c (launchers that only log and can be told to refuse a call) into a
test-only library.  Checked: the stage order of SGM_Match (reference SemiGlobalMatching.c:80-122), that a refused
launch ends the match at once, that the Q14 bookkeeping (S zero / S pending in the per-direction planes)
stays consistent across a failed match, and that a failed host-pointer match waits for the stream before returning
(queued copies read the caller's buffers)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "soc_project_stereo_matching_amd", "csrc")


@pytest.fixture(scope="module")
def host(tmp_path_factory):
    out = tmp_path_factory.mktemp("hoststub") / "libsgm_hoststub.so"
    subprocess.check_call(["gcc", "-O1", "-std=c11", "-fPIC", "-shared", "-I", CSRC, "-o", str(out),
                           os.path.join(CSRC, "sgm_host.c"), os.path.join(ROOT, "tests", "stub_device.c"), "-lm"])
    L = C.CDLL(str(out))
    L.sgm_create.restype = C.c_void_p
    L.sgm_create.argtypes = [C.c_int]
    L.sgm_destroy.argtypes = [C.c_void_p]
    for f in (L.sgm_initialize, L.sgm_reset):
        f.argtypes = [C.c_void_p, C.c_uint16, C.c_uint16, C.c_void_p]
        f.restype = C.c_bool
    for f in (L.sgm_match, L.sgm_match_async, L.sgm_match_device):
        f.argtypes = [C.c_void_p] * 4
        f.restype = C.c_bool
    L.sgm_match_wait.argtypes = [C.c_void_p]
    L.sgm_match_wait.restype = C.c_bool
    L.sgm_keep_stages.argtypes = [C.c_void_p, C.c_int]
    L.stub_log_name.restype = C.c_char_p
    L.stub_log_name.argtypes = [C.c_int]
    L.stub_log_arg.argtypes = [C.c_int]
    L.stub_fail_at.argtypes = [C.c_char_p, C.c_int]
    return L


def log(L, drop=("sync", "h2d", "d2h", "alloc", "memset")):
    return [(L.stub_log_name(i).decode(), L.stub_log_arg(i)) for i in range(L.stub_log_size())
            if L.stub_log_name(i).decode() not in drop]


class Frame:
    def __init__(self, w=48, h=20, b=1):
        self.left = np.zeros((b, h, w), np.uint8)
        self.right = np.zeros((b, h, w), np.uint8)
        self.out = np.zeros((b, h, w), np.float32)

    def args(self):
        return self.left.ctypes.data, self.right.ctypes.data, self.out.ctypes.data


def fresh(L, d=16, **kw):
    import soc_project_stereo_matching_amd as S
    s = L.sgm_create(0)
    assert s
    opt = S.default_option(d, **kw)
    assert L.sgm_reset(s, 48, 20, C.byref(opt))
    L.stub_clear()
    return s, opt


def test_stage_order_of_a_match(host):
    L = host
    s, _ = fresh(L)
    f = Frame()
    assert L.sgm_match(s, *f.args())
    assert [n for n, _ in log(L)] == ["census", "aggregate", "sum_wta_lr", "lrcheck", "speckle", "median"]
    L.sgm_destroy(s)
    # D > 256: separate sum / right-view kernels
    s, _ = fresh(L, d=300)
    assert L.sgm_match(s, *f.args())
    assert [n for n, _ in log(L)] == ["census", "aggregate", "sum_wta", "wta_right", "lrcheck", "speckle", "median"]
    L.sgm_destroy(s)
    # options off: the stages are not launched at all
    s, _ = fresh(L, is_check_lr=False, is_remove_speckles=False)
    assert L.sgm_match(s, *f.args())
    assert [n for n, _ in log(L)] == ["census", "aggregate", "sum_wta_lr", "median"]
    L.sgm_destroy(s)


@pytest.mark.parametrize("stage", ["census", "aggregate", "sum_wta_lr", "lrcheck", "speckle", "median"])
def test_a_refused_launch_ends_the_match(host, stage):
    L = host
    s, _ = fresh(L)
    f = Frame()
    L.stub_fail_at(stage.encode(), 0)
    assert not L.sgm_match(s, *f.args())
    names = [n for n, _ in log(L)]
    assert names[-1] == stage and names.count(stage) == 1           # nothing is launched after the refused call
    full = [L.stub_log_name(i).decode() for i in range(L.stub_log_size())]
    assert full[-1] == "sync" and "d2h" not in full                # waited for the queued uploads, queued no download
    L.sgm_destroy(s)


def test_failed_first_match_leaves_S_zero(host):
    """Reset, a match that dies before the cost sum, then Match without Reset: nothing was accumulated, so the sum
    must not add to S (accumulate = 0) and no lazy S materialisation may run."""
    L = host
    s, _ = fresh(L)
    f = Frame()
    L.stub_fail_at(b"aggregate", 0)
    assert not L.sgm_match(s, *f.args())
    L.stub_clear()
    assert L.sgm_match(s, *f.args())
    assert log(L)[:3] == [("census", 1), ("aggregate", 0xFF), ("sum_wta_lr", 0)]
    L.sgm_destroy(s)


def test_failed_sum_does_not_mark_planes_pending(host):
    L = host
    s, _ = fresh(L)
    f = Frame()
    L.stub_fail_at(b"sum_wta_lr", 0)
    assert not L.sgm_match(s, *f.args())
    L.stub_clear()
    assert L.sgm_match(s, *f.args())                                 # still the first frame of S
    assert ("sum_wta", 0) not in log(L) and ("sum_wta", 1) not in log(L)
    assert ("sum_wta_lr", 0) in log(L)
    L.sgm_destroy(s)


def test_q14_state_after_a_failure_in_the_second_match(host):
    """Frame 1 completes (its sum stays pending in the planes: the fused kernel does not write S).  Frame 2 without
    Reset materialises S (sum_wta, accumulate 0) and then dies at the census.  Frame 3: S already holds frame 1, so
    it must NOT be materialised again (double add) and the fused sum accumulates (accumulate = 1)."""
    L = host
    s, _ = fresh(L)
    f = Frame()
    assert L.sgm_match(s, *f.args())
    L.stub_clear()
    L.stub_fail_at(b"census", 0)
    assert not L.sgm_match(s, *f.args())
    assert log(L) == [("sum_wta", 0), ("census", 1)]
    L.stub_clear()
    assert L.sgm_match(s, *f.args())
    assert log(L)[:3] == [("census", 1), ("aggregate", 0xFF), ("sum_wta_lr", 1)]
    L.sgm_destroy(s)


def test_refused_materialisation_stays_pending(host):
    L = host
    s, _ = fresh(L)
    f = Frame()
    assert L.sgm_match(s, *f.args())
    L.stub_clear()
    L.stub_fail_at(b"sum_wta", 0)
    assert not L.sgm_match(s, *f.args())
    assert log(L) == [("sum_wta", 0)]
    L.stub_clear()
    assert L.sgm_match(s, *f.args())
    assert log(L)[:4] == [("sum_wta", 0), ("census", 1), ("aggregate", 0xFF), ("sum_wta_lr", 1)]
    L.sgm_destroy(s)


def test_separate_kernels_failure_after_the_sum(host):
    """D > 256: the sum kernel writes S itself.  If the right-view WTA behind it is refused, S already contains the
    frame: the next Match without Reset accumulates."""
    L = host
    s, _ = fresh(L, d=300)
    f = Frame()
    L.stub_fail_at(b"wta_right", 0)
    assert not L.sgm_match(s, *f.args())
    L.stub_clear()
    assert L.sgm_match(s, *f.args())
    assert ("sum_wta", 1) in log(L)
    L.sgm_destroy(s)


def test_async_match_hands_over_at_wait(host):
    L = host
    s, _ = fresh(L)
    f, g = Frame(), Frame()
    assert L.sgm_match_async(s, *f.args())
    full = [L.stub_log_name(i).decode() for i in range(L.stub_log_size())]
    assert full.count("h2d") == 2 and full.count("d2h") == 1 and "sync" not in full     # queued, not waited for
    assert L.sgm_match_async(s, *g.args())                          # implicit wait for the first one
    full = [L.stub_log_name(i).decode() for i in range(L.stub_log_size())]
    assert full.index("sync") < len(full) - 1 and full.count("d2h") == 2
    assert L.sgm_match_wait(s) and L.sgm_match_wait(s)              # idempotent
    L.sgm_destroy(s)


def test_staged_result_is_handed_over_in_pieces(host):
    """A result of 256 KiB or more that has to be staged (pageable caller buffer) comes back in pieces -- two below 4 MiB, four above
    (that size does not fit the stub's capped allocations: the GPU tests of batches through pageable buffers run it) -- with an
    event behind each but the last; sgm_match_wait copies piece i to the caller while piece i + 1 is still on the bus, and the
    caller ends up with exactly the staged bytes."""
    L = host
    L.sgm_match_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.sgm_match_wait.argtypes = [C.c_void_p]
    s, opt = fresh(L)
    w, h = 320, 240                                                  # 300 KiB of disparities
    assert L.sgm_reset(s, w, h, C.byref(opt))
    img = np.zeros((h, w), np.uint8)
    out = np.full((h, w), -1.0, np.float32)
    L.stub_clear()
    assert L.sgm_match_async(s, img.ctypes.data, img.ctypes.data, out.ctypes.data)
    names = [L.stub_log_name(i).decode() for i in range(L.stub_log_size())]
    tail = names[names.index("median") + 1:]
    assert tail == ["d2h", "event_record", "d2h"]
    sizes = [L.stub_log_arg(i) for i in range(L.stub_log_size()) if L.stub_log_name(i) == b"d2h"]
    assert sum(sizes) == w * h * 4 and all(n % 4 == 0 for n in sizes)
    assert names.index("h2d") < names.index("census")
    L.stub_clear()
    assert L.sgm_match_wait(s)
    names = [L.stub_log_name(i).decode() for i in range(L.stub_log_size())]
    assert names.count("event_sync") == 1 and names[-1] == "sync"
    assert not np.any(out == -1.0)                                   # every piece reached the caller (the stub's device memory is zeroed)
    L.sgm_destroy(s)


def test_row_tile_instance_allocates_a_tile_not_a_frame(host):
    """Row-tile mode: the per-direction planes have storage for the tile's rows + one hand-over row either side, the
    frame-sized S and cost volumes are not allocated at all (nothing on the tile path reads or writes them)."""
    import soc_project_stereo_matching_amd as S
    L = host
    L.sgm_set_rows.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.sgm_set_rows.restype = C.c_bool
    w, h, d = 640, 480, 128
    opt = S.default_option(d)

    def allocated_kib(rows):
        s = L.sgm_create(0)
        if rows:
            assert L.sgm_set_rows(s, *rows)
        L.stub_clear()
        assert L.sgm_reset(s, w, h, C.byref(opt))
        kib = sum(L.stub_log_arg(i) for i in range(L.stub_log_size()) if L.stub_log_name(i) == b"alloc")
        L.sgm_destroy(s)
        return kib

    cells_kib = w * h * d // 1024
    whole = allocated_kib(None)
    assert 8 * cells_kib <= whole < 8.2 * cells_kib + 40 * w * h * 4 // 1024          # 8 planes, no S, no cost volume
    tile = allocated_kib((120, 180))                                                  # 60 of 480 rows
    assert tile < 8 * cells_kib * 62 // 480 + 40 * w * h * 4 // 1024
    edge = allocated_kib((0, 60))
    assert edge <= tile


def test_extension_options_select_their_stages(host):
    """SURVEY.md 8(f)-4 extensions: a wide census window takes the materialised-cost path (u64 census, cost volume,
    volume-fed aggregation with 16 lanes per pixel -- also for a batch, which otherwise uses 8); the right reference view
    replaces LRCheck by its mirror image and needs the right-view WTA even with the LR check off."""
    import soc_project_stereo_matching_amd as S
    L = host
    L.sgm_set_census_window.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.sgm_set_census_window.restype = C.c_bool
    L.sgm_set_reference_view.argtypes = [C.c_void_p, C.c_int]
    L.sgm_set_batch.argtypes = [C.c_void_p, C.c_int]
    L.sgm_set_batch.restype = C.c_bool
    f = Frame()
    s = L.sgm_create(0)
    assert not L.sgm_set_census_window(s, 9, 9) and not L.sgm_set_census_window(s, 6, 5) and not L.sgm_set_census_window(s, 0, 1)
    assert L.sgm_set_census_window(s, 9, 7)
    opt = S.default_option(16)
    assert L.sgm_reset(s, 48, 20, C.byref(opt))
    L.stub_clear()
    assert L.sgm_match(s, *f.args())
    assert log(L) == [("census_window", 907), ("cost64", 0), ("aggregate_volume", 0xFF | (16 << 8)), ("sum_wta_lr", 0), ("lrcheck", 0),
                      ("speckle", 50), ("median", 0)]
    assert L.sgm_set_census_window(s, 5, 5) and L.sgm_reset(s, 48, 20, C.byref(opt))     # back to the reference window
    L.stub_clear()
    assert L.sgm_match(s, *f.args())
    assert [n for n, _ in log(L)][:2] == ["census", "aggregate"]
    L.sgm_destroy(s)

    s = L.sgm_create(0)
    assert L.sgm_set_batch(s, 2) and L.sgm_set_census_window(s, 7, 7) and L.sgm_reset(s, 48, 20, C.byref(opt))
    g = Frame(48, 40)                                                 # two frames back to back
    L.stub_clear()
    assert L.sgm_match(s, *g.args())
    assert ("aggregate_volume", 0xFF | (16 << 8)) in log(L)
    L.sgm_destroy(s)

    for check_lr in (True, False):
        s = L.sgm_create(0)
        L.sgm_set_reference_view(s, 1)
        o = S.default_option(16, is_check_lr=check_lr)
        assert L.sgm_reset(s, 48, 20, C.byref(o))
        L.stub_clear()
        assert L.sgm_match(s, *f.args())
        names = [n for n, _ in log(L, drop=("sync", "h2d", "d2h", "alloc", "memset", "d2d"))]
        assert names == ["census", "aggregate", "sum_wta_lr", "lrcheck_right", "speckle", "median"]
        assert ("lrcheck_right", int(check_lr)) in log(L)
        L.sgm_destroy(s)


def test_overlapped_post_pass_ordering(host):
    """sgm_set_overlap_post: the post pass goes to a second stream behind an event recorded after the cost sum; the NEXT
    cost sum (and a lazy S materialisation, which uses the speckle label map as scratch) waits for the event recorded after
    the post pass; sgm_synchronize waits for both streams."""
    L = host
    L.sgm_set_overlap_post.argtypes = [C.c_void_p, C.c_int]
    L.sgm_set_overlap_post.restype = C.c_bool
    L.sgm_synchronize.argtypes = [C.c_void_p]
    L.sgm_synchronize.restype = C.c_bool
    s, opt = fresh(L)
    assert L.sgm_set_overlap_post(s, 1)
    f = Frame()
    keep = ("sync", "h2d", "d2h", "alloc", "memset")
    L.stub_clear()
    assert L.sgm_match_device(s, *f.args())
    assert [n for n, _ in log(L, drop=keep)] == ["census", "aggregate", "sum_wta_lr", "event_record", "wait_event", "lrcheck", "speckle",
                                                   "median", "event_record"]
    L.stub_clear()
    assert L.sgm_reset(s, 48, 20, C.byref(opt)) and L.sgm_match_device(s, *f.args())
    names = [n for n, _ in log(L, drop=keep)]
    assert names[:4] == ["census", "aggregate", "wait_event", "sum_wta_lr"]      # the post pass of the match before still reads the maps
    L.stub_clear()
    assert L.sgm_match_device(s, *f.args())                                    # no Reset: S of the previous match is materialised first
    names = [n for n, _ in log(L, drop=keep)]
    assert names[:2] == ["sum_wta", "census"]                                  # (allocating S waited for both streams: nothing pending)
    L.stub_clear()
    assert L.sgm_match_device(s, *f.args())                                    # again without Reset, S exists now
    names = [n for n, _ in log(L, drop=keep)]
    assert names[:3] == ["wait_event", "sum_wta", "census"]                    # the materialisation scribbles on the speckle label map
    L.stub_clear()
    assert L.sgm_synchronize(s)
    assert [L.stub_log_name(i).decode() for i in range(L.stub_log_size())].count("sync") == 2    # both streams
    L.stub_clear()
    assert L.sgm_reset(s, 48, 20, C.byref(opt)) and L.sgm_match_device(s, *f.args())
    assert "wait_event" not in [n for n, _ in log(L, drop=keep)][:3]           # nothing pending after a synchronize
    L.sgm_destroy(s)


def test_platform_frame_entry_stage_order_and_errors(host):
    """sgm_match_planes: six colour planes -> two grey conversions -> the match -> depth -> one D2H; with the
    post pass on the second stream the depth conversion follows it there; a refused launch waits for the streams and
    fails; depth / scoring of an arbitrary device map wait for a pending post pass first."""
    L = host
    for f in (L.sgm_match_planes, L.sgm_match_planes_async):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]
        f.restype = C.c_bool
    L.sgm_gray_from_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.sgm_gray_from_planes.restype = C.c_bool
    L.sgm_disparity_to_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.sgm_disparity_to_depth.restype = C.c_bool
    L.sgm_set_overlap_post.argtypes = [C.c_void_p, C.c_int]
    L.sgm_set_overlap_post.restype = C.c_bool
    L.sgm_set_rows.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.sgm_set_rows.restype = C.c_bool
    s, opt = fresh(L)
    planes = np.zeros((6, 20, 48), np.uint8)
    depth = np.zeros((20, 48), np.float32)
    args = (planes.ctypes.data, 1000.0, 100.0, 0.0, depth.ctypes.data)
    keep = ("sync", "alloc", "memset")
    assert L.sgm_match_planes_async(s, *args)
    assert [n for n, _ in log(L, drop=keep)] == ["h2d", "gray", "gray", "census", "aggregate", "sum_wta_lr", "lrcheck", "speckle", "median",
                                              "depth", "d2h"]
    assert [a for n, a in log(L, drop=keep) if n in ("gray", "depth")] == [960, 960, 960]
    assert L.sgm_match_wait(s)
    # arguments
    assert not L.sgm_match_planes(s, None, 1000.0, 100.0, 0.0, depth.ctypes.data)
    assert not L.sgm_match_planes(s, planes.ctypes.data, 1000.0, 100.0, 0.0, None)
    assert not L.sgm_gray_from_planes(s, planes.ctypes.data, 960, 80, planes.ctypes.data)
    assert L.sgm_gray_from_planes(s, planes.ctypes.data, 960, 77, planes.ctypes.data)
    # a refused grey conversion / depth conversion ends the call after waiting for what was queued
    for stage, nth in (("gray", 1), ("depth", 0)):
        L.stub_clear()
        L.stub_fail_at(stage.encode(), nth)
        assert not L.sgm_match_planes(s, *args)
        full = [L.stub_log_name(i).decode() for i in range(L.stub_log_size())]
        assert full[-1] == "sync" and "d2h" not in full
    # second stream: depth + D2H follow the post pass; a later depth conversion on the main stream waits for its event
    assert L.sgm_set_overlap_post(s, 1) and L.sgm_reset(s, 48, 20, C.byref(opt))
    L.stub_clear()
    assert L.sgm_match_planes_async(s, *args)
    names = [n for n, _ in log(L, drop=keep)]
    assert names == ["h2d", "gray", "gray", "census", "aggregate", "sum_wta_lr", "event_record", "wait_event", "lrcheck", "speckle",
                     "median", "event_record", "depth", "event_record", "d2h"]          # "done" is recorded behind the depth kernel, not behind the copy
    L.stub_clear()
    assert L.sgm_disparity_to_depth(s, depth.ctypes.data, 960, 1000.0, 100.0, 0.0, depth.ctypes.data)
    assert [n for n, _ in log(L, drop=keep)] == ["wait_event", "depth"]
    assert L.sgm_match_wait(s)
    # row-tile instances have their own call sequence
    assert L.sgm_set_rows(s, 0, 10) and L.sgm_reset(s, 48, 20, C.byref(opt))
    assert not L.sgm_match_planes(s, *args)
    L.sgm_destroy(s)


def test_row_tiles_of_a_batch(host):
    """sgm_set_batch on a row-tile instance: every tile call covers the same rows of B frames -- the hand-over holds B rows per
    direction of the sweep and moves in one launch, like every stage."""
    import soc_project_stereo_matching_amd as S
    L = host
    L.sgm_set_rows.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.sgm_set_rows.restype = C.c_bool
    L.sgm_set_batch.argtypes = [C.c_void_p, C.c_int]
    L.sgm_set_batch.restype = C.c_bool
    L.sgm_tile_boundary_bytes.argtypes = [C.c_void_p]
    L.sgm_tile_boundary_bytes.restype = C.c_size_t
    for f in (L.sgm_tile_begin,):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        f.restype = C.c_bool
    for f in (L.sgm_tile_export_boundary, L.sgm_tile_import_boundary):
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        f.restype = C.c_bool
    L.sgm_tile_sweep.argtypes = [C.c_void_p, C.c_int]
    L.sgm_tile_sweep.restype = C.c_bool
    L.sgm_tile_finish.argtypes = [C.c_void_p, C.c_void_p]
    L.sgm_tile_finish.restype = C.c_bool
    w, h, d, B = 48, 20, 16, 3
    opt = S.default_option(d)
    s = L.sgm_create(0)
    assert L.sgm_set_batch(s, B) and L.sgm_set_rows(s, 5, 12) and L.sgm_reset(s, w, h, C.byref(opt))
    assert L.sgm_tile_boundary_bytes(s) == B * 3 * w * 32                  # Dp = 32 for D = 16
    img = np.zeros((B, h, w), np.uint8)
    buf = np.zeros(B * 3 * w * 32, np.uint8)
    out = np.zeros((B, h, w), np.float32)
    L.stub_clear()
    assert L.sgm_tile_begin(s, img.ctypes.data, img.ctypes.data)
    assert L.sgm_tile_import_boundary(s, 1, buf.ctypes.data) and L.sgm_tile_sweep(s, 1) and L.sgm_tile_export_boundary(s, 1, buf.ctypes.data)
    assert L.sgm_tile_finish(s, out.ctypes.data)
    names = [n for n, _ in log(L, drop=("sync", "alloc", "memset", "h2d", "d2h"))]
    assert names == ["census", "aggregate", "rows_in", "aggregate", "rows_out", "sum_wta_lr", "lrcheck"]
    assert [a for n, a in log(L) if n.startswith("rows_")] == [3 * B, 3 * B]      # one launch moves 3 directions x B frames
    L.sgm_destroy(s)


def test_stage_streams_ordering(host):
    """sgm_set_stage_cus: cost sum and post pass on streams of their own (restricted to some CUs of every XCD).  The sum
    stream waits for the aggregation, the post stream for the sum; the NEXT aggregation waits for the previous cost sum (it
    rewrites the planes that sum reads), the next cost sum for the previous post pass (it rewrites the maps that pass reads);
    a failed launch drains every stream; count < 0 returns a group to the main stream."""
    L = host
    L.sgm_set_stage_cus.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.sgm_set_stage_cus.restype = C.c_bool
    L.sgm_synchronize.argtypes = [C.c_void_p]
    L.sgm_synchronize.restype = C.c_bool
    s, opt = fresh(L)
    L.stub_clear()
    assert L.sgm_set_stage_cus(s, 2, 0, 2) and L.sgm_set_stage_cus(s, 1, 2, 10) and L.sgm_set_stage_cus(s, 0, 12, 20)
    assert [a for n, a in log(L) if n == "stream_cus"] == [2, 210, 1220]          # first * 100 + count CUs per XCD
    assert not L.sgm_set_stage_cus(s, 3, 0, 1)
    f = Frame()
    keep = ("sync", "h2d", "d2h", "alloc", "memset", "stream_cus")
    assert L.sgm_reset(s, 48, 20, C.byref(opt))
    L.stub_clear()
    assert L.sgm_match_device(s, *f.args())
    assert [n for n, _ in log(L, drop=keep)] == ["census", "aggregate", "event_record", "wait_event", "sum_wta_lr", "event_record",
                                                   "wait_event", "lrcheck", "speckle", "median", "event_record"]
    L.stub_clear()
    assert L.sgm_reset(s, 48, 20, C.byref(opt)) and L.sgm_match_device(s, *f.args())
    names = [n for n, _ in log(L, drop=keep)]
    # previous sum still reads the planes -> wait; ...; previous post pass still reads the maps -> wait before the sum
    assert names[:7] == ["wait_event", "census", "aggregate", "event_record", "wait_event", "wait_event", "sum_wta_lr"]
    # a refused launch on the post stream: everything is drained (three streams), nothing stays pending
    L.stub_clear()
    L.stub_fail_at(b"speckle", 0)
    assert L.sgm_reset(s, 48, 20, C.byref(opt)) and not L.sgm_match_device(s, *f.args())
    assert [L.stub_log_name(i).decode() for i in range(L.stub_log_size())][-3:] == ["sync"] * 3
    L.stub_clear()
    assert L.sgm_reset(s, 48, 20, C.byref(opt)) and L.sgm_match_device(s, *f.args())
    assert [n for n, _ in log(L, drop=keep)][:2] == ["census", "aggregate"]
    # sum on its own stream, post pass not: the post pass follows the sum there and the next aggregation waits for all of it
    assert L.sgm_set_stage_cus(s, 2, 0, -1) and L.sgm_reset(s, 48, 20, C.byref(opt))
    L.stub_clear()
    assert L.sgm_match_device(s, *f.args())
    assert [n for n, _ in log(L, drop=keep)] == ["census", "aggregate", "event_record", "wait_event", "sum_wta_lr", "event_record",
                                                   "lrcheck", "speckle", "median", "event_record"]
    # back to one stream
    assert L.sgm_set_stage_cus(s, 1, 0, -1) and L.sgm_set_stage_cus(s, 0, 0, -1) and L.sgm_reset(s, 48, 20, C.byref(opt))
    L.stub_clear()
    assert L.sgm_match_device(s, *f.args())
    assert [n for n, _ in log(L, drop=keep)] == ["census", "aggregate", "sum_wta_lr", "lrcheck", "speckle", "median"]
    L.sgm_destroy(s)


def test_a_priority_stream_is_replaced_by_a_later_all_cu_request(host):
    """sgm_set_stage_priority stores no CU range; a later sgm_set_stage_cus(which, 0, 0) -- 'own stream, all CUs, normal
    priority' -- and, for the main group, count < 0 ('back to the default') must make a new stream instead of finding the
    request 'already configured' and keeping the priority stream."""
    L = host
    L.sgm_set_stage_cus.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.sgm_set_stage_cus.restype = C.c_bool
    L.sgm_set_stage_priority.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.sgm_set_stage_priority.restype = C.c_bool
    s, opt = fresh(L)
    for which in (1, 0):
        L.stub_clear()
        assert L.sgm_set_stage_priority(s, which, -1)
        assert L.sgm_set_stage_cus(s, which, 0, 0)                        # replaces the priority stream ...
        assert L.sgm_set_stage_cus(s, which, 0, 0)                        # ... and only then is it "already configured"
        assert [(n, a) for n, a in log(L) if n.startswith("stream_")] == [("stream_prio", -1), ("stream_cus", 0)]
    L.stub_clear()
    assert L.sgm_set_stage_priority(s, 0, 1) and L.sgm_set_stage_cus(s, 0, 0, -1)          # main group: default = a plain all-CU stream
    assert [(n, a) for n, a in log(L) if n.startswith("stream_")] == [("stream_prio", 1), ("stream_cus", 0)]
    f = Frame()
    assert L.sgm_reset(s, 48, 20, C.byref(opt)) and L.sgm_match_device(s, *f.args())
    L.sgm_destroy(s)


def test_fused_last_sweep_stage_order_and_q14(host, monkeypatch):
    """SGM_UPSUM=1 on a batch instance with W > H and a padded range of 128: the aggregation launch is told to leave the upward
    directions to the fused kernel (bit 8 of the stub's log), the fused kernel replaces the cost sum, the left image is kept; a
    Match WITHOUT Reset then first walks the three missing directions (mask 0x68 = (0,-1), (-1,-1), (1,-1)) on the kept image,
    puts S together and runs the separate kernels; keep_stages and a refused launch keep / restore the ordinary path."""
    L = host
    monkeypatch.setenv("SGM_UPSUM", "1")
    L.sgm_fused_sweep_rows.argtypes = [C.c_void_p]
    L.sgm_set_batch.argtypes = [C.c_void_p, C.c_int]
    L.sgm_set_batch.restype = C.c_bool
    s = L.sgm_create(0)
    assert s and L.sgm_set_batch(s, 2)
    import soc_project_stereo_matching_amd as S
    opt = S.default_option(128)
    f = Frame(w=200, h=30, b=2)
    assert L.sgm_reset(s, 200, 30, C.byref(opt))
    L.stub_clear()
    assert L.sgm_match_device(s, *f.args())
    assert L.sgm_fused_sweep_rows(s) == 3
    got = log(L, drop=("sync", "h2d", "d2h", "alloc", "memset"))
    assert [n for n, _ in got] == ["census", "d2d", "aggregate", "upsum", "lrcheck", "speckle", "median"]
    assert dict(got)["d2d"] == 2 * 200 * 30 and dict(got)["aggregate"] == 0x1FF and dict(got)["upsum"] == 3
    # Match without Reset: the three planes are re-created, S materialised, then the accumulating separate path
    L.stub_clear()
    assert L.sgm_match_device(s, *f.args())
    assert L.sgm_fused_sweep_rows(s) == 0
    names = [(n, a) for n, a in log(L, drop=("sync", "h2d", "d2h", "alloc", "memset"))]
    assert names[:2] == [("aggregate", 0x68), ("sum_wta", 0)] and ("aggregate", 0xFF) in names and ("upsum", 3) not in names
    # a refused fused launch fails the match and leaves nothing pending; the next Reset + Match runs fused again
    assert L.sgm_reset(s, 200, 30, C.byref(opt))
    L.stub_fail_at(b"upsum", 0)
    assert not L.sgm_match_device(s, *f.args())
    assert L.sgm_reset(s, 200, 30, C.byref(opt))
    L.stub_clear()
    assert L.sgm_match_device(s, *f.args()) and L.sgm_fused_sweep_rows(s) == 3
    # stage read-back wants S: the ordinary kernels
    L.sgm_keep_stages.argtypes = [C.c_void_p, C.c_int]
    L.sgm_keep_stages(s, 1)
    assert L.sgm_reset(s, 200, 30, C.byref(opt)) and L.sgm_match_device(s, *f.args()) and L.sgm_fused_sweep_rows(s) == 0
    L.sgm_destroy(s)
    # shapes the fused kernel does not cover keep the separate kernels whatever SGM_UPSUM says: W <= H, a padded range of 64
    for (w, h, d) in ((20, 22, 128), (200, 30, 64)):          # (the stub allocator holds at most 1 MB of planes)
        s = L.sgm_create(0)
        assert L.sgm_set_batch(s, 2)
        o2 = S.default_option(d)
        f2 = Frame(w=w, h=h, b=2)
        assert L.sgm_reset(s, w, h, C.byref(o2)) and L.sgm_match_device(s, *f2.args()) and L.sgm_fused_sweep_rows(s) == 0
        L.sgm_destroy(s)


def test_median_band_that_gave_up_fails_the_match(host):
    """The chained median kernel of tall frames polls the band above a bounded number of times; a band that gives up sets a word
    of page-locked host memory (sgmd_median's status argument).  The host reads it after every wait for the streams: the
    match that produced the map fails (sgm_synchronize / sgm_match_wait / sgm_match return false), the next one is clean."""
    L = host
    L.sgm_synchronize.argtypes = [C.c_void_p]
    L.sgm_synchronize.restype = C.c_bool
    L.stub_median_stall.argtypes = [C.c_int]
    s, opt = fresh(L)
    f = Frame()
    L.stub_median_stall(1)
    assert L.sgm_match_device(s, *f.args())            # queued; the failure shows when the host waits
    assert not L.sgm_synchronize(s)
    assert L.sgm_reset(s, 48, 20, C.byref(opt))
    assert not L.sgm_match(s, *f.args())               # blocking host-pointer form: false, the output is not handed over
    L.stub_median_stall(0)
    assert L.sgm_reset(s, 48, 20, C.byref(opt)) and L.sgm_match(s, *f.args()) and L.sgm_synchronize(s)
    L.sgm_destroy(s)
