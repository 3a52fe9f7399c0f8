"""Host engine logic without a GPU: replay the plan (tables, descriptors, schedule, packed
weights) on the CPU and compare with the oracle's STTNInpaint.inpaint()."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

from vsr_amd.synth import make_state_dict

from _replay import PlanView


@pytest.fixture(scope="module")
def host_engine(built_lib):
    from vsr_amd.engine import SttnEngine

    sd = make_state_dict(0, "auto")
    eng = SttnEngine(sd, "auto", device=None, neighbor_stride=2, ref_length=3)
    yield sd, eng
    eng.close()


def test_plan_replay_matches_oracle(built_lib):
    """Default tuning: 4 frames, stride 2 / refs every 3 -> windows T = 4, 4; counts [2,2,2,1]."""
    import _replay_check

    res = _replay_check.run()
    assert res["counts"] == [2, 2, 2, 1]


def test_plan_replay_decoder_rows(built_lib, host_engine):
    """A plan that was told which rows of the model-resolution output its caller reads (vsr_sttn_auto_chunk_rows: the strip is blended
    back only where the mask is set) runs its decoder on those rows and on what they depend on -- and nothing else: the replay writes
    ONLY the row ranges the ops carry (what lies outside stays zero), so a range that is one row too small anywhere in the chain
    (the last block's convs and patch rows, 3x3 convs, two align_corners upsamplings, the 2-row blocks of the output conv) shows
    up as rows of garbage.  Inside the range the comps are the full plan's; the FLOPs go down.  Three ranges: the bench's subtitle
    box, the top rows, a sliver in the middle."""
    from vsr_amd import _lib
    from _replay import PlanView, replay

    sd, eng = host_engine
    L = 2                # one window of two neighbour frames: the row / column logic does not depend on the window's length
    frames = np.random.default_rng(21).integers(0, 256, size=(L, 120, 640, 3), dtype=np.uint8)
    full = PlanView(_lib, eng, L)
    want, counts, _ = replay(full, eng.packed_weights(), frames)
    full_flops = full.flops
    full.close()
    for rows in ((76, 118), (0, 7), (59, 61)):
        part = PlanView(_lib, eng, L, rows=rows)
        try:
            got, counts2, _ = replay(part, eng.packed_weights(), frames)
            lo, hi = rows[0] // 2 * 2, (rows[1] + 1) // 2 * 2            # whole 2-row blocks
            assert list(counts) == list(counts2)
            # (the replay's contractions are torch-CPU matmuls, whose blocking -- hence rounding -- depends on how many rows they are
            # given: a handful of u8 truncation flips, as against the oracle; on the GPU the frames are equal bit for bit,
            # tests/test_gpu_sttn.py::test_decoder_rows_give_the_same_frames.  A range one row short gives whole rows of garbage.)
            d = np.abs(got[:, lo:hi] - want[:, lo:hi])
            assert d.max() <= 1.0 and (d > 0).mean() < 1e-3, (rows, d.max(), (d > 0).mean())
            assert part.flops < full_flops
            dec = [(i.H, int(i.ipar[1]), int(i.ipar[2])) for i, _ in part.ops if i.kind == 3]      # OP_UPSAMPLE2X
            assert dec and all(0 <= a < b <= 2 * H for H, a, b in dec) and any(b - a < 2 * H for H, a, b in dec)
        finally:
            part.close()


def test_plan_replay_decoder_box(built_lib, host_engine):
    """... and which columns (vsr_plan_create_box; the engine's VSR_DECODE_COLS=1): the GEMMs of the decoder and of the last block
    take rectangles of pixels / of patches (the elementwise ops between them keep whole rows).  Inside the rectangle the comps are
    the full plan's, the FLOPs are below those of the same rows at full width, and a rectangle one column short anywhere in the
    chain shows up as columns of garbage.  Three rectangles: a centred subtitle line, the left edge, a narrow box touching the right edge."""
    from vsr_amd import _lib
    from _replay import PlanView, replay

    sd, eng = host_engine
    L = 2                # one window of two neighbour frames: the row / column logic does not depend on the window's length
    frames = np.random.default_rng(22).integers(0, 256, size=(L, 120, 640, 3), dtype=np.uint8)
    full = PlanView(_lib, eng, L)
    want, counts, _ = replay(full, eng.packed_weights(), frames)
    full.close()
    for rows, cols in (((76, 118), (120, 520)), ((0, 30), (0, 64)), ((40, 90), (600, 640))):
        strip = PlanView(_lib, eng, L, rows=rows)
        rows_flops = strip.flops
        strip.close()
        part = PlanView(_lib, eng, L, rows=rows, cols=cols)
        try:
            got, counts2, _ = replay(part, eng.packed_weights(), frames)
            lo, hi = rows[0] // 2 * 2, (rows[1] + 1) // 2 * 2            # whole 2x4 blocks
            c0, c1 = cols[0] // 4 * 4, (cols[1] + 3) // 4 * 4
            assert list(counts) == list(counts2)
            d = np.abs(got[:, lo:hi, c0:c1] - want[:, lo:hi, c0:c1])
            assert d.max() <= 1.0 and (d > 0).mean() < 1e-3, (rows, cols, d.max(), (d > 0).mean())
            assert part.flops < rows_flops, (rows, cols, part.flops, rows_flops)
        finally:
            part.close()


def test_plan_replay_degenerate_chunks(built_lib):
    """a 1-frame and a 2-frame chunk with the reference's default stride 5 / references every 10 (the ragged tail of a video,
    sttn_auto_inpaint.py:240-245): one window of T = 1 / T = 2, every frame visited once and returned as uint8"""
    import _replay_check

    assert _replay_check.run(L=1, ns=5, rl=10, seed=3)["counts"] == [1]
    assert _replay_check.run(L=2, ns=5, rl=10, seed=4)["counts"] == [1, 1]


def test_plan_replay_sttn_det(built_lib):
    """sttn-det geometry (432x240, patch table of network_sttn.py:69), pre-masked input, model-res blend."""
    import _replay_check

    res = _replay_check.run_det()
    assert res["counts"] == [2, 2]


def test_plan_replay_sttn_det_box(built_lib):
    """sttn-det with the decoder / last block on the box of the resized mask (rows and columns): the whole composite is the full plan's"""
    import _replay_check

    res = _replay_check.run_det_box()
    assert len(res) == 3 and all(r[2] < 1.0 for r in res), res


def test_plan_flops_det_matches_survey(built_lib):
    """733.8 GFLOP per frame for a 50-frame sttn-det batch."""
    from vsr_amd.engine import SttnEngine
    from vsr_amd.synth import make_state_dict

    eng = SttnEngine(make_state_dict(1, "det"), "det", device=None)
    try:
        f50, x50 = eng.flops(50, reference=True), eng.flops(50)
    finally:
        eng.close()
    assert abs(f50 / 50 / 1e9 - 733.8) < 1.5, f50 / 50 / 1e9
    assert 0.95 * f50 < x50 < 0.985 * f50          # the last block of a window runs on its neighbour frames only


def test_plan_replay_shared_first_block_qkv(built_lib):
    """VSR_QKV0_SHARED=1 in a fresh process (the library reads its tuning once): the first block's q/k/v run once per frame of the chunk
    (one GEMM in front of the windows) and every window's first attention reads its frames' rows there -- the oracle's composites,
    one q/k/v GEMM fewer per window, fewer FLOPs than the default plan, the reference's count unchanged.  Three windows on two
    kinds of frames: reference frames that are neighbours elsewhere."""
    from vsr_amd import _lib
    from vsr_amd.engine import SttnEngine
    from vsr_amd.synth import make_state_dict
    from _replay import PlanView

    env = dict(os.environ, VSR_QKV0_SHARED="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_replay_check.py"), "--long"], env=env, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    b = json.loads(r.stdout.strip().splitlines()[-1])
    # the plan without the sharing, from a process of its own (the switch is on by default since round 5; the library reads it once)
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import vsr_amd\nfrom vsr_amd import _lib\nfrom vsr_amd.engine import SttnEngine\nfrom vsr_amd.synth import make_state_dict\n"
            "from _replay import PlanView\n"
            "eng = SttnEngine(make_state_dict(0, 'auto'), 'auto', device=None, neighbor_stride=2, ref_length=4)\n"
            "view = PlanView(_lib, eng, 5)\n"
            "print(json.dumps({'n_qkv': sum(1 for info, _ in view.ops if info.tag.decode() == 'attn.qkv'), 'flops': view.flops,\n"
            "                  'ref_flops': eng.flops(5, reference=True), 'counts': view.counts.tolist()}))\n") % (os.path.dirname(HERE), HERE)
    r0 = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VSR_QKV0_SHARED="0"), capture_output=True, text=True, timeout=600)
    assert r0.returncode == 0, r0.stdout[-2000:] + r0.stderr[-4000:]
    a = json.loads(r0.stdout.strip().splitlines()[-1])
    nwin = 3
    assert a["n_qkv"] == 8 * nwin and b["n_qkv"] == 7 * nwin + 1
    assert b["flops"] < a["flops"] and abs(a["ref_flops"] - b["ref_flops"]) <= 1e-9 * b["ref_flops"]
    assert b["counts"] == a["counts"]


def test_plan_replay_split_pv_and_square_tiles():
    """Non-default tuning in a fresh process: PV split-K + reduce-scatter, 128x128 conv tiles,
    tap-major K order (the library reads its tuning env once per process)."""
    env = dict(os.environ, VSR_PV_SPLIT_CHUNKS="10", VSR_CONV_TILE="0", VSR_PV_TILE="0", VSR_CONV_KORDER="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_replay_check.py")], env=env, capture_output=True,
                       text=True, timeout=2400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["pv_split"] >= 2 and res["has_reduce"], res
    assert res["counts"] == [2, 2, 2, 1]


def test_plan_flops_L50_matches_survey(built_lib, host_engine):
    """32.14 TFLOP per full 50-frame chunk = 642.8 GFLOP per output frame."""
    from vsr_amd.engine import SttnEngine

    sd, _ = host_engine
    eng = SttnEngine(sd, "auto", device=None)
    try:
        f50, x50 = eng.flops(50, reference=True), eng.flops(50)
    finally:
        eng.close()
    assert abs(f50 / 50 / 1e9 - 642.8) < 1.0, f50 / 50 / 1e9
    # what is contracted: the reference's count minus the reference-frame rows of every window's last block (Plan::buildWindow) --
    # windows of 10 / 14 / 15 frames with 6 / 10-11 / 11 neighbours: (T - nn) / T of one block in eight, QKV excepted
    assert 0.96 * f50 < x50 < 0.975 * f50, x50 / f50


def test_plan_tables_stay_inside_buffers(built_lib, host_engine):
    """Every gathered address of every GEMM (including padded rows) lies inside its buffer."""
    sd, eng = host_engine
    view = PlanView(built_lib, eng, 7)
    try:
        for info, items in view.ops:
            if info.kind != 1:
                continue
            bm, bn = built_lib.TILE_DIMS[info.tile_cfg]
            for it in items:
                rowA, colA = view.tables[it.tRowA], view.tables[it.tColA]
                assert len(rowA) >= it.tilesM * bm and len(colA) >= it.K // 32
                lo = it.offA + rowA.min() + colA[: it.K // 32].min()
                hi = it.offA + rowA.max() + colA[: it.K // 32].max() + 31
                assert 0 <= lo and hi < view.buf_elems[it.bufA], info.tag
                rowB, colB = view.tables[it.tRowB], view.tables[it.tColB]
                if info.bmode == 0:
                    assert len(rowB) >= it.tilesN * bn and len(colB) >= it.K // 32
                    nb = it.K // 32
                else:
                    assert len(rowB) >= it.K and len(colB) >= it.tilesN * bn // 32
                    nb = it.tilesN * bn // 32
                lo = it.offB + rowB.min() + colB[:nb].min()
                hi = it.offB + rowB.max() + colB[:nb].max() + 31
                assert 0 <= lo and hi < view.buf_elems[it.bufB], info.tag
                rowC, colC = view.tables[it.tRowC], view.tables[it.tColC]
                assert len(rowC) >= it.tilesM * bm and len(colC) >= it.tilesN * bn // 32
                ncc = (it.N + 31) // 32
                hi = it.offC + (it.splitK - 1) * it.splitStride + rowC[: it.M].max() + colC[:ncc].max() + 31
                assert hi < view.buf_elems[it.bufC] + 32, info.tag
                assert it.splitK * it.chunksPerSplit >= it.K // 32
                assert (it.splitK - 1) * it.chunksPerSplit < it.K // 32, "no empty split"
    finally:
        view.close()


@pytest.mark.parametrize("nl", [2, 3, 4])
def test_window_lanes_share_nothing_but_the_running_average(built_lib, nl):
    """vsr_sttn_set_lanes(n): window w is issued on stream w % n.  What makes that legal is checked here on the plan itself: the
    op list is the same as with one lane (same kinds, shapes, FLOPs, visit counts), every buffer an op WRITES on lane 1 is a
    lane-1 instance (ids from BUF_L1_X0 on) and on lane 0 never one, the only shared buffers a window op writes are BUF_COMP
    -- by OP_DECODE_OUT, which the engine chains from window to window with events -- and the per-instance row maxima, and no
    lane reads a window buffer of the other.  (That the laned plan computes the right thing is the replay tests above: they run
    the default, two lanes.)"""
    from vsr_amd import _lib
    from vsr_amd.engine import SttnEngine

    BUF_WEIGHTS, BUF_IN_U8, BUF_FEATS, BUF_COMP, BUF_MASK_U8, BUF_ROWMAX, L1_FIRST, BUF_QKV0 = 0, 1, 6, 20, 22, 23, 25, 70
    window_scoped = set(range(7, 20)) | {21, 24}            # X0 .. D4, PVPART, LSUM
    # (BUF_QKV0: the first block's q/k/v of every frame of the chunk, written once in front of the windows -- VSR_QKV0_SHARED, the
    # default since round 5 -- and only read by them)
    shared_read = {BUF_WEIGHTS, BUF_IN_U8, BUF_FEATS, BUF_MASK_U8, BUF_ROWMAX, BUF_QKV0}
    eng = SttnEngine(make_state_dict(0, "auto"), "auto", device=None)
    try:
        views = {}
        for lanes in (1, nl):
            eng.set_lanes(lanes)
            views[lanes] = PlanView(_lib, eng, 20)
        one, two = views[1], views[nl]

        def own(buf, lane):
            return buf in window_scoped if lane == 0 else L1_FIRST + (lane - 1) * 15 <= buf < L1_FIRST + lane * 15

        assert len(one.ops) == len(two.ops) and one.flops == two.flops and list(one.counts) == list(two.counts)
        assert len(one.buf_elems) == len(two.buf_elems) == BUF_QKV0 + 1 and all(one.buf_elems[b] == 0 for b in range(L1_FIRST, BUF_QKV0))
        lanes_seen, n_decode = set(), 0
        first_window_op = next(i for i, (b, it) in enumerate(two.ops) if b.tag == b"attn.qkv" and it[0].bufC != BUF_QKV0)
        for i, ((a, ai), (b, bi)) in enumerate(zip(one.ops, two.ops)):
            lane = _lib.lib.vsr_plan_op_lane(two.p, i)
            assert _lib.lib.vsr_plan_op_lane(one.p, i) == 0 and 0 <= lane < nl
            assert (a.kind, a.nitems, a.tile_cfg, a.bmode, a.flops, a.tag) == (b.kind, b.nitems, b.tile_cfg, b.bmode, b.flops, b.tag)
            if i < first_window_op:                              # the encoder: the caller's stream, before the lanes fork
                assert lane == 0
                continue
            lanes_seen.add(lane)
            reads, writes = [], []
            if b.kind == 1:
                for x, y in zip(ai, bi):
                    assert (x.M, x.N, x.K, x.splitK, x.act) == (y.M, y.N, y.K, y.splitK, y.act)
                    reads += [y.bufA, y.bufB] + ([y.bufR] if y.bufR >= 0 else [])
                    writes.append(y.bufC)
            elif b.kind == 2:
                reads += [y.bufS for y in bi]
                writes += [y.bufP for y in bi]
            else:
                reads += [v for v in (b.buf_src, b.ibuf[0]) if v >= 0]
                writes += [b.buf_dst] if b.buf_dst >= 0 else []
            n_decode += b.kind == 4
            for buf in writes:                                   # a lane writes its own window buffers; shared state: the running average only
                assert own(buf, lane) or (buf == BUF_COMP and b.kind == 4), (i, b.tag, lane, buf)
            for buf in reads:                                    # ... and reads its own window buffers and read-only shared ones
                assert own(buf, lane) or buf in shared_read, (i, b.tag, lane, buf)
        assert lanes_seen == set(range(nl)) and n_decode == 4               # L = 20, stride 5: four windows, one decode_out each
        for v in views.values():
            v.close()
    finally:
        eng.close()


@pytest.mark.parametrize("nl", [2, 3])
def test_lane_schedule_orders_every_conflict(built_lib, nl):
    """Race freedom of the laned schedule, by construction.  The engine (sttn_engine.hip run_plan) orders the streams with three kinds of
    events: every lane starts behind everything issued before the first window op (fork), every OP_DECODE_OUT waits for the previous
    one (chain), and the caller's stream ends behind all lanes (join).  Here those edges plus program order per lane are turned into
    vector clocks over the op list, every op's reads and writes are taken from its descriptors (buffer, and for the row-maxima array
    the instance offset), and for every pair of accesses to the same location on DIFFERENT lanes with at least one write the earlier
    op must happen-before the later one."""
    from vsr_amd import _lib
    from vsr_amd.engine import SttnEngine

    BUF_WEIGHTS, BUF_ROWMAX = 0, 23
    ACT_ROW_MAX, ACT_A_EXP = 0x400, 0x800
    eng = SttnEngine(make_state_dict(0, "auto"), "auto", device=None)
    try:
        eng.set_lanes(nl)
        view = PlanView(_lib, eng, 30)                          # six windows
        lanes = [_lib.lib.vsr_plan_op_lane(view.p, i) for i in range(len(view.ops))]
        # (the first op of the first window: the shared first-block q/k/v GEMM in front of the windows -- it writes buffer 70, BUF_QKV0 --
        # belongs to what the lanes fork behind, Plan::firstWindowOp)
        first_window = next(i for i, (b, it) in enumerate(view.ops) if b.tag in (b"attn.qkv", b"attn.qk") and not (b.tag == b"attn.qkv" and it[0].bufC == 70))

        def accesses(info, items):
            """[(location, is_write)]: location = buffer id, or (BUF_ROWMAX, offset) -- every attention instance has its own array"""
            acc = []
            if info.kind == 1:
                for y in items:
                    acc += [(y.bufA, False), (y.bufC, True)]
                    if y.bufB != BUF_WEIGHTS:
                        acc.append((y.bufB, False))
                    if y.act & ACT_ROW_MAX:                     # the epilogue leaves row maxima in R (atomic max)
                        acc.append(((BUF_ROWMAX, y.offR), True))
                    elif y.act & ACT_A_EXP:                     # reads the row maxima through the bias pointer, writes partial row sums to R
                        acc.append(((BUF_ROWMAX, y.offBias), False))
                        if y.bufR >= 0:
                            acc.append((y.bufR, True))
                    elif y.bufR >= 0:
                        acc.append((y.bufR, False))
            elif info.kind == 2:
                for y in items:
                    acc += [(y.bufS, False), (y.bufP, True)]
            else:
                if info.buf_src >= 0:
                    acc.append((info.buf_src, False))
                if info.ibuf[0] >= 0:
                    acc.append((info.ibuf[0], False))
                if info.buf_mask >= 0:
                    acc.append((info.buf_mask, False))
                if info.buf_dst >= 0:
                    acc.append((info.buf_dst, True))
                    if info.kind == 4:                          # decode_out averages into comps: read-modify-write
                        acc.append((info.buf_dst, False))
            return acc

        # ---- happens-before: vc[i][lane] = largest op index on `lane` known to precede op i (inclusive of i on its own lane)
        vc, last_on_lane, last_decode = [], {}, None
        for i, (info, _) in enumerate(view.ops):
            lane = lanes[i]
            clock = dict(vc[last_on_lane[lane]]) if lane in last_on_lane else {}
            if lane != 0 and lane not in last_on_lane:          # fork: behind everything before the first window op
                assert first_window > 0
                for k, v in vc[first_window - 1].items():
                    clock[k] = max(clock.get(k, -1), v)
            if info.kind == 4 and last_decode is not None and lanes[last_decode] != lane:      # decode chain
                for k, v in vc[last_decode].items():
                    clock[k] = max(clock.get(k, -1), v)
            clock[lane] = i
            vc.append(clock)
            last_on_lane[lane] = i
            if info.kind == 4:
                last_decode = i

        def ordered(j, i):
            return vc[i].get(lanes[j], -1) >= j

        last_write, reads_since, checked = {}, {}, 0
        for i, (info, items) in enumerate(view.ops):
            for loc, is_write in accesses(info, items):
                w = last_write.get(loc)
                if w is not None and w != i and lanes[w] != lanes[i]:
                    checked += 1
                    assert ordered(w, i), (i, info.tag, "after write by", w, view.ops[w][0].tag, loc)
                if is_write:
                    for r in reads_since.get(loc, []):
                        if r != i and lanes[r] != lanes[i]:
                            checked += 1
                            assert ordered(r, i), (i, info.tag, "overwrites what", r, view.ops[r][0].tag, "reads", loc)
            for loc, is_write in accesses(info, items):
                if is_write:
                    last_write[loc] = i
                    reads_since[loc] = []
                else:
                    reads_since.setdefault(loc, []).append(i)
        assert checked > 20                                     # the features every lane reads, the running average every window updates
        view.close()
    finally:
        eng.close()
