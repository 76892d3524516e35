"""The C host of the multi-GPU row-tile path (include/sgm_tiles.h; csrc/sgm_tile_sched.c, csrc/sgm_tiles.c) on the CPU.

* the step schedule (sgm_tile_step) against an independent Python restatement of DESIGN.md section 6's step order, call by
  call, for 1..8 ranks, leads 0..3 and streams shorter and longer than the pipeline;
* the device pipeline (sgm_tiles_*) END TO END with ranks as threads: csrc/sgm_host.c + sgm_tiles.c + sgm_tile_sched.c linked
  with tests/stub_device.c in its toy-compute mode (a recurrence with the data dependencies of SGM's vertical / diagonal paths,
  in host memory), through the library's local transport and through its RCCL transport bound to tests/stub_rccl.c
  (SGM_RCCL_LIBRARY) -- results of 2..4 ranks, batches and leads must equal the same recurrence computed on the whole frame.
No GPU anywhere here; the GPU tests of the same pipeline are in tests/test_gpu_tiling.py."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "soc_project_stereo_matching_amd", "csrc")
INC = os.path.join(ROOT, "include")


# ------------------------------------------------------------------------------------------ the schedule, restated
def restated_trace(rank, world, height, slots, lead, n_frames, rows_of):
    """DESIGN.md section 6's step order written down again in Python (test infrastructure): the calls one rank makes."""
    r, N, F, K, R = rank, world, n_frames, lead, slots
    slot = lambda f: f % R                                           # noqa: E731
    valid = lambda f: 0 <= f < F                                     # noqa: E731
    lag = max(r, N - 1 - r)
    out = []
    for step in range(F + N + 2 + K):
        if valid(step):
            out.append(("begin", slot(step), step))
        s = step - K
        f, g = s - r, s - (N - 1 - r)
        for forward, fr in ((True, f), (False, g)):
            if not valid(fr):
                continue
            first = (r == 0) if forward else (r == N - 1)
            last = (r == N - 1) if forward else (r == 0)
            if not first:
                out.append(("import", slot(fr), forward))
            out.append(("sweep", slot(fr), forward))
            if not last:
                out.append(("export", slot(fr), forward))
        if N > 1:
            ops, touched = [], set()
            if valid(f) and r < N - 1:
                ops.append(("send", "bnd", slot(f), True, False, r + 1)); touched.add(slot(f))
            if valid(s + 1 - r) and r > 0:
                ops.append(("recv", "bnd", slot(s + 1 - r), True, True, r - 1)); touched.add(slot(s + 1 - r))
            if valid(g) and r > 0:
                ops.append(("send", "bnd", slot(g), False, False, r - 1)); touched.add(slot(g))
            if valid(s + 1 - (N - 1 - r)) and r < N - 1:
                ops.append(("recv", "bnd", slot(s + 1 - (N - 1 - r)), False, True, r + 1)); touched.add(slot(s + 1 - (N - 1 - r)))
            h = s - N
            if valid(h):
                owner = h % N
                touched.add(slot(h))
                if r != owner:
                    ops.append(("send", "rows", slot(h)) + rows_of(r) + (owner,))
                else:
                    ops += [("recv", "rows", slot(h)) + rows_of(k) + (k,) for k in range(N) if k != r]
            if ops:
                out.append(("exchange", tuple(ops), tuple(sorted(touched))))
        if valid(s - lag):
            out.append(("finish", slot(s - lag)))
        p = s - N - 1 if N > 1 else s
        if valid(p) and p % N == r:
            out.append(("post", slot(p), p))
    return out


def c_trace(rank, world, height, slots, lead, n_frames, incremental=False):
    from soc_project_stereo_matching_amd import tiles
    out = []

    def exchange(ops, sl):
        t = []
        for o in ops:
            kind = "send" if o.kind == tiles.XOP_SEND else "recv"
            if o.buf == tiles.XBUF_BOUNDARY:
                t.append((kind, "bnd", o.slot, bool(o.forward), bool(o.incoming), o.peer))
            else:
                t.append((kind, "rows", o.slot, o.row_begin, o.row_end, o.peer))
        out.append(("exchange", tuple(t), tuple(sl)))

    eng = tiles.PyEngine(lambda s, f: out.append(("begin", s, f)), lambda s, fw: out.append(("import", s, fw)),
                         lambda s, fw: out.append(("sweep", s, fw)), lambda s, fw: out.append(("export", s, fw)), exchange,
                         lambda s: out.append(("finish", s)), lambda s, f: out.append(("post", s, f)))
    for step in range(tiles.steps_total(n_frames, world, lead)):
        # incremental: the way sgm_tiles_submit calls it -- while frames arrive only those submitted so far are known
        known = min(step + 1, n_frames) if incremental else n_frames
        eng.step(rank, world, height, slots, lead, step, known)
    return out


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 8])
def test_c_schedule_equals_the_restated_one(world):
    from soc_project_stereo_matching_amd import tiles
    height = 37
    rows_of = lambda k: tiles.tile_rows(height, world, k)            # noqa: E731
    for lead in (0, 1, 2, 3):
        need = tiles.slots_needed(world, lead)
        assert need == (world + 3 if world > 1 else 2) + lead
        for n_frames in sorted({1, 2, max(1, world - 1), world, world + 3, 2 * world + 5}):
            assert tiles.steps_total(n_frames, world, lead) == n_frames + world + 2 + lead
            for slots in (need, need + 2):
                for rank in range(world):
                    want = restated_trace(rank, world, height, slots, lead, n_frames, rows_of)
                    assert c_trace(rank, world, height, slots, lead, n_frames) == want
                    assert c_trace(rank, world, height, slots, lead, n_frames, incremental=True) == want


def test_every_send_meets_its_receive_in_the_same_exchange():
    """All ranks run the same step sequence: in every step the operations rank a lists towards rank b are, in order, the mirror
    images of those rank b lists towards rank a (same buffer kind, same size class), which is what a grouped RCCL exchange needs."""
    from soc_project_stereo_matching_amd import tiles
    for world, lead, n_frames in ((2, 0, 5), (3, 2, 7), (4, 1, 9), (8, 2, 19)):
        height = 50
        slots = tiles.slots_needed(world, lead)
        per_rank = []
        for r in range(world):
            steps = [[] for _ in range(tiles.steps_total(n_frames, world, lead))]
            tr = c_trace(r, world, height, slots, lead, n_frames)
            # re-run step by step to know which step an exchange belongs to
            out = []
            eng = tiles.PyEngine(lambda *a: None, lambda *a: None, lambda *a: None, lambda *a: None,
                                 lambda ops, sl: out.append([(o.kind, o.buf, o.forward, o.row_end - o.row_begin, o.peer) for o in ops]),
                                 lambda *a: None, lambda *a: None)
            for step in range(len(steps)):
                out.clear()
                eng.step(r, world, height, slots, lead, step, n_frames)
                steps[step] = list(out[0]) if out else []
            per_rank.append(steps)
            assert sum(1 for e in tr if e[0] == "exchange") == sum(1 for s in steps if s)
        for step in range(len(per_rank[0])):
            for a in range(world):
                for b in range(world):
                    if a == b:
                        continue
                    a_to_b = [(buf, fwd, n) for kind, buf, fwd, n, peer in per_rank[a][step] if kind == tiles.XOP_SEND and peer == b]
                    b_from_a = [(buf, fwd, n) for kind, buf, fwd, n, peer in per_rank[b][step] if kind == tiles.XOP_RECV and peer == a]
                    assert a_to_b == b_from_a, (world, step, a, b)


def test_schedule_argument_checks_and_error_propagation():
    from soc_project_stereo_matching_amd import tiles
    L = tiles.lib()
    with pytest.raises(ValueError):
        tiles.tile_rows(3, 4, 0)
    assert tiles.tile_rows(10, 3, 0) == (0, 4) and tiles.tile_rows(10, 3, 2) == (7, 10)
    ok = tiles.PyEngine(*([lambda *a: None] * 7))
    for bad in (dict(rank=2, world=2), dict(slots=4, world=2), dict(lead=-1), dict(step=-1)):
        kw = dict(rank=0, world=1, height=8, slots=8, lead=0, step=0, frames_known=1)
        kw.update(bad)
        assert L.sgm_tile_step(C.byref(ok.struct), kw["rank"], kw["world"], kw["height"], kw["slots"], kw["lead"], kw["step"], kw["frames_known"]) == -1
    assert L.sgm_tile_step(None, 0, 1, 8, 2, 0, 0, 1) == -1
    # an engine function that fails ends the step at once with its value; a Python exception comes back as itself
    calls = []

    def sweep(slot, fwd):
        calls.append(("sweep", fwd))
        raise KeyError("no such plane")
    eng = tiles.PyEngine(lambda *a: calls.append("begin"), lambda *a: None, sweep, lambda *a: None, lambda *a: None,
                         lambda *a: calls.append("finish"), lambda *a: calls.append("post"))
    with pytest.raises(KeyError):
        eng.step(0, 1, 8, 2, 0, 0, 1)
    assert calls == ["begin", ("sweep", True)]


# ------------------------------------------------------------------------------------------ the pipeline on the stub device
@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    d = tmp_path_factory.mktemp("tilestub")
    host = str(d / "libsgm_tilestub.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-fPIC", "-shared", "-I", CSRC, "-o", host,
                           os.path.join(CSRC, "sgm_host.c"), os.path.join(CSRC, "sgm_tile_sched.c"), os.path.join(CSRC, "sgm_tiles.c"),
                           os.path.join(ROOT, "tests", "stub_device.c"), "-lm", "-ldl", "-lpthread"])
    rccl = str(d / "libstub_rccl.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-fPIC", "-shared", "-o", rccl,
                           os.path.join(ROOT, "tests", "stub_rccl.c"), "-lpthread"])
    os.environ["SGM_RCCL_LIBRARY"] = rccl                            # read by the stub build's rccl_bind at its first use
    from soc_project_stereo_matching_amd import tiles
    L = C.CDLL(host)
    i, p = C.c_int, C.c_void_p
    L.sgm_tiles_create.argtypes = [i, i, i, C.c_uint16, C.c_uint16, p, i, i, i, i, C.POINTER(tiles.Transport)]
    L.sgm_tiles_create.restype = p
    L.sgm_tiles_destroy.argtypes = [p]
    L.sgm_tiles_result_ring.argtypes = [p, p, i]
    L.sgm_tiles_submit.argtypes = [p, p, p, p]
    L.sgm_tiles_submit.restype = C.c_bool
    L.sgm_tiles_finish.argtypes = [p]
    L.sgm_tiles_finish.restype = C.c_bool
    L.sgm_tiles_info.argtypes = [p, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.sgm_tiles_on_result.argtypes = [p, tiles._RESULT_FN, p]
    L.sgm_tiles_local_group.argtypes = [i, i]
    L.sgm_tiles_local_group.restype = p
    L.sgm_tiles_local_transport.argtypes = [p, i, C.POINTER(tiles.Transport)]
    L.sgm_tiles_local_transport.restype = C.c_bool
    L.sgm_tiles_local_destroy.argtypes = [p]
    L.sgm_tiles_rccl_unique_id.argtypes = [p]
    L.sgm_tiles_rccl_unique_id.restype = C.c_bool
    L.sgm_tiles_rccl_transport.argtypes = [p, i, i, i, C.POINTER(tiles.Transport)]
    L.sgm_tiles_rccl_transport.restype = C.c_bool
    L.stub_toy_compute.argtypes = [i]
    L.stub_fail_at.argtypes = [C.c_char_p, i]
    L.rccl_lib = C.CDLL(rccl)
    L.rccl_lib.stub_rccl_stats.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    return L


DIRS = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (1, -1), (-1, 1)]      # (dx, dy), SemiGlobalMatching.c:213-220


def toy_expected(left):
    """The recurrence of tests/stub_device.c's toy mode on the WHOLE frame, in numpy: [H][W] float32 after the post pass."""
    h, w = left.shape
    img = left.astype(np.int64)
    total = np.zeros((h, w), np.int64)
    for d, (dx, dy) in enumerate(DIRS):
        P = np.zeros((h, w), np.int64)
        if dy == 0:
            P = (3 * d + img) & 0xFF
        else:
            ys = range(h) if dy > 0 else range(h - 1, -1, -1)
            for y in ys:
                py = y - dy
                prev = np.roll(P[py], dx) if 0 <= py < h else np.zeros(w, np.int64)      # pixel (py, x - dx), wrapping
                P[y] = (prev * 5 + img[y] + d) & 0xFF
        total += P
    return (total + 1000).astype(np.float32)


def run_ranks(L, world, w, h, d, batch, lead, n_steps, transport_of, spare=1, throttle=0):
    """`world` pipelines as threads of this process; returns {frame index: [batch][H][W] map} collected from the owners' rings."""
    import soc_project_stereo_matching_amd as S
    opt = S.default_option(d)
    rng = np.random.default_rng(world * 100 + batch * 10 + lead)
    lefts = [rng.integers(0, 256, (batch, h, w), dtype=np.uint8) for _ in range(n_steps)]
    rights = [rng.integers(0, 256, (batch, h, w), dtype=np.uint8) for _ in range(n_steps)]
    ring_frames = (n_steps + world - 1) // world
    rings = [np.full((ring_frames, batch, h, w), -1.0, np.float32) for _ in range(world)]
    errors, infos = [], [None] * world
    L.stub_toy_compute(1)

    def rank_main(r):
        tr = transport_of(r)
        t = L.sgm_tiles_create(0, r, world, w, h, C.byref(opt), batch, lead, spare, throttle, C.byref(tr) if tr is not None else None)
        if not t:
            errors.append((r, "create"))
            return
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        L.sgm_tiles_info(t, C.byref(a), C.byref(b), C.byref(c))
        infos[r] = (a.value, b.value, c.value)
        L.sgm_tiles_result_ring(t, rings[r].ctypes.data, ring_frames)
        for k in range(n_steps):
            if not L.sgm_tiles_submit(t, lefts[k].ctypes.data, rights[k].ctypes.data, None):
                errors.append((r, "submit", k))
                break
        else:
            if not L.sgm_tiles_finish(t):
                errors.append((r, "finish"))
        L.sgm_tiles_destroy(t)
        if tr is not None:
            tr.close()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(180)
    L.stub_toy_compute(0)
    assert not any(t.is_alive() for t in th), "a rank is stuck"
    assert not errors, errors
    got = {k: rings[k % world][(k // world) % ring_frames] for k in range(n_steps)}
    return got, lefts, infos


def check(got, lefts, batch):
    for k, m in got.items():
        for b in range(batch):
            want = toy_expected(lefts[k][b])
            assert np.array_equal(m[b], want), (k, b, np.argwhere(m[b] != want)[:4])


@pytest.mark.parametrize("world,batch,lead", [(1, 1, 0), (1, 2, 2), (2, 1, 0), (2, 2, 2), (3, 1, 2), (4, 2, 1), (4, 1, 3),
                                              (8, 8, 2), (8, 1, 2)])       # the target node's shape: 8 ranks, 8 frames per step
def test_pipeline_of_threads_over_the_local_transport(stub, world, batch, lead):
    L = stub
    from soc_project_stereo_matching_amd import tiles
    w, h, d = 12, 13, 16
    group = L.sgm_tiles_local_group(world, 0) if world > 1 else None

    def transport_of(r):
        if world == 1:
            return None
        t = tiles.Transport()
        assert L.sgm_tiles_local_transport(group, r, C.byref(t))
        return t
    n_steps = 2 * world + 3                                           # more frames than slots: slots are reused
    got, lefts, infos = run_ranks(L, world, w, h, d, batch, lead, n_steps, transport_of)
    check(got, lefts, batch)
    assert [i[:2] for i in infos] == [tiles.tile_rows(h, world, r) for r in range(world)]
    assert all(i[2] == tiles.slots_needed(world, lead) for i in infos)
    if group:
        L.sgm_tiles_local_destroy(group)


@pytest.mark.parametrize("world,batch,lead,throttle", [(2, 1, 2, 0), (3, 2, 1, 3), (4, 1, 2, 0), (8, 8, 2, 0), (8, 1, 2, 3)])
def test_pipeline_of_threads_over_rccl_bound_at_run_time(stub, world, batch, lead, throttle):
    """The RCCL transport of csrc/sgm_tiles.c (dlopen + ncclCommInitRank + grouped ncclSend / ncclRecv) against tests/stub_rccl.c,
    which completes a group's operations in another order than they were listed."""
    L = stub
    from soc_project_stereo_matching_amd import tiles
    uid = C.create_string_buffer(tiles.ID_BYTES)
    assert L.sgm_tiles_rccl_unique_id(uid)
    assert uid.raw.startswith(b"stub-rccl-")                          # bound to the stub, not to a librccl of this process
    m0, b0 = C.c_ulonglong(), C.c_ulonglong()
    L.rccl_lib.stub_rccl_stats(C.byref(m0), C.byref(b0))

    def transport_of(r):
        t = tiles.Transport()
        assert L.sgm_tiles_rccl_transport(uid, r, world, 0, C.byref(t))      # returns when all ranks have joined
        return t
    w, h, d = 12, 13, 16
    n_steps = world + 4
    got, lefts, _ = run_ranks(L, world, w, h, d, batch, lead, n_steps, transport_of, spare=2, throttle=throttle)
    check(got, lefts, batch)
    m1, b1 = C.c_ulonglong(), C.c_ulonglong()
    L.rccl_lib.stub_rccl_stats(C.byref(m1), C.byref(b1))
    # per frame: 2 (N - 1) hand-overs of batch x 3 x W x Dp bytes and the rows of the other N - 1 ranks -- ONE message per peer: the
    # same rows of the B maps of a batch are packed (at 8 ranks x 8 frames the owner posts 7 row receives per step, not 56)
    bnd = batch * 3 * w * 32
    want_msgs = n_steps * (2 * (world - 1) + (world - 1))
    want_bytes = n_steps * (2 * (world - 1) * bnd + batch * sum(tiles.tile_rows(h, world, r)[1] - tiles.tile_rows(h, world, r)[0]
                                                                  for r in range(world)) * w * 4)
    own_rows = [tiles.tile_rows(h, world, r)[1] - tiles.tile_rows(h, world, r)[0] for r in range(world)]
    want_bytes -= sum(own_rows[k % world] for k in range(n_steps)) * batch * w * 4      # the owner keeps its own rows
    assert m1.value - m0.value == want_msgs
    assert b1.value - b0.value == want_bytes


def test_one_rank_sends_to_itself_over_the_rccl_transport(stub):
    """What a one-GPU box can exercise of the RCCL transport (tests/test_gpu_tiling.py does it with the real library)."""
    L = stub
    from soc_project_stereo_matching_amd import tiles
    uid = C.create_string_buffer(tiles.ID_BYTES)
    assert L.sgm_tiles_rccl_unique_id(uid)
    t = tiles.Transport()
    assert L.sgm_tiles_rccl_transport(uid, 0, 1, 0, C.byref(t))
    src = np.arange(64, dtype=np.uint8)
    dst = np.zeros(64, np.uint8)
    assert t.group_start(t.ctx) == 0
    assert t.recv(t.ctx, dst.ctypes.data, 64, 0, None) == 0 and t.send(t.ctx, src.ctypes.data, 64, 0, None) == 0
    assert t.group_end(t.ctx) == 0
    assert np.array_equal(src, dst)
    t.close()


def test_results_reach_their_owner_in_order_and_a_refused_launch_fails_the_submit(stub):
    L = stub
    import soc_project_stereo_matching_amd as S
    from soc_project_stereo_matching_amd import tiles
    w, h, d = 12, 13, 16
    opt = S.default_option(d)
    seen = []
    cb = tiles._RESULT_FN(lambda _u, frame, d_map, ev: seen.append(int(frame)))
    t = L.sgm_tiles_create(0, 0, 1, w, h, C.byref(opt), 1, 2, 1, 0, None)
    assert t
    L.sgm_tiles_on_result(t, cb, None)
    img = np.zeros((h, w), np.uint8)
    for _ in range(5):
        assert L.sgm_tiles_submit(t, img.ctypes.data, img.ctypes.data, None)
    assert seen == [0, 1, 2]                                          # a lead of 2: the post pass of frame f is queued in step f + 2
    assert L.sgm_tiles_finish(t)
    assert seen == [0, 1, 2, 3, 4]
    # a new stream on the same pipeline; a launch the device refuses ends the submit with false
    assert L.sgm_tiles_submit(t, img.ctypes.data, img.ctypes.data, None)
    L.stub_fail_at(b"aggregate", 0)
    assert not L.sgm_tiles_submit(t, img.ctypes.data, img.ctypes.data, None)
    L.sgm_tiles_destroy(t)
    # more ranks than rows, a missing transport, a bad rank: no pipeline
    assert not L.sgm_tiles_create(0, 0, 20, w, h, C.byref(opt), 1, 2, 1, 0, None)
    assert not L.sgm_tiles_create(0, 0, 2, w, h, C.byref(opt), 1, 2, 1, 0, None)
    assert not L.sgm_tiles_create(0, 3, 2, w, h, C.byref(opt), 1, 2, 1, 0, None)


def test_slot_memory_estimate_and_the_guard_of_create(stub, capfd):
    """sgm_tile_slot_bytes is what sgm_tiles_create and tiling.DeviceSlotEngine check against the free device memory.  The failed
    round-3 rehearsal (2 ranks, 3840x2160 D=128, 8 frames per step) died in hipMalloc on 34 005 319 680 bytes of planes for ONE slot:
    the estimate must contain exactly that; 7 slots of it do not fit a 288 GB GPU, and create says so with the batch that would."""
    L = stub
    import soc_project_stereo_matching_amd as S
    from soc_project_stereo_matching_amd import tiles
    opt = S.default_option(128)
    w, h = 3840, 2160
    r0, r1 = tiles.tile_rows(h, 2, 0)
    per_slot = tiles.slot_bytes(r0, r1, w, h, opt, 8)
    planes = 8 * 8 * (r1 - r0 + 1) * w * 128                          # rank 0: its rows + ONE hand-over row (an inner tile has two)
    assert planes == 34005319680 and planes < per_slot < planes * 1.15             # + ~64 B per pixel of the frame
    assert tiles.slot_bytes(0, h, w, h, opt, 1) == 8 * h * w * 128 + 64 * w * h + 12 * w * 128      # one rank: no hand-over rows
    assert tiles.slot_bytes(5, 5, w, h, opt, 1) == 0 and tiles.slot_bytes(0, h + 1, w, h, opt, 1) == 0
    # DESIGN.md section 6's table: GB per GPU for N = 1, 2, 4, 8 at a lead of 2
    gb = {n: [tiles.slot_bytes(*tiles.tile_rows(h, n, 0), w, h, opt, b) * tiles.slots_needed(n, 2) / 1e9 for b in (1, 4, 8)] for n in (1, 2, 4, 8)}
    assert [round(v) for v in gb[2]] == [34, 134, 268] and [round(v) for v in gb[8]] == [21, 84, 167]
    # the stub device reports 200 GiB free: 2 ranks x batch 8 is refused with a message, batch 4 is accepted
    group = L.sgm_tiles_local_group(2, 0)
    tr = tiles.Transport()
    assert L.sgm_tiles_local_transport(group, 0, C.byref(tr))
    assert not L.sgm_tiles_create(0, 0, 2, w, h, C.byref(opt), 8, 2, 1, 0, C.byref(tr))
    err = capfd.readouterr().err
    assert "need about 268" in err and "batch of at most 6" in err
    tr.close()
    L.sgm_tiles_local_destroy(group)


def test_first_use_from_several_threads_at_once():
    """Ranks may be threads of one process and reach the library for the first time together (tools/fuzz_parity.py did): the
    prototypes must be attached exactly once -- a racing first use once left a function object without them, and `long`
    arguments went through as 32-bit ints.  A fresh interpreter, eight threads, frame numbers beyond 2^32."""
    import sys
    code = r'''
import sys, threading
sys.path.insert(0, %r)
from soc_project_stereo_matching_amd import tiles
bad, go = [], threading.Barrier(8)
def worker(k):
    go.wait()
    seen = []
    eng = tiles.PyEngine(lambda s, f: seen.append(f), *([lambda *a: None] * 6))
    big = (1 << 33) + k
    if tiles.steps_total(big, 1, 0) != big + 3:
        bad.append(("steps_total", k))
    eng.step(0, 1, 8, 2, 0, big, big + 1)
    if seen != [big]:
        bad.append(("begin", k, seen))
th = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
[t.start() for t in th]; [t.join() for t in th]
print("bad:", bad)
sys.exit(1 if bad else 0)
''' % ROOT
    for _ in range(3):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, (out.stdout[-500:], out.stderr[-1500:])
