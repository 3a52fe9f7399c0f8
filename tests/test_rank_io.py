"""tools/rank_io.py on CPU: 1 / 2 / 3 gloo ranks each read and write their own chunks of a record file by offset; the sink is the
single-process file byte for byte; a short source and a failing rank are handled the way the serial loop handles them."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PREFIX = b"FRAME\n"
HEAD_IN = b"YUV4MPEG2 W8 H6 F25:1 Ip A1:1 C444\n"
HEAD_OUT = b"YUV4MPEG2 W8 H6 F25:1 Ip A1:1 C420mpeg2 XCOLORRANGE=LIMITED\n"
FB_IN, FB_OUT = 8 * 6 * 3, 8 * 6 * 3 // 2


def _make_source(path, total, truncate=0):
    rng = np.random.default_rng(total)
    with open(path, "wb") as f:
        f.write(HEAD_IN)
        for k in range(total):
            f.write(PREFIX)
            f.write(rng.integers(0, 256, FB_IN, dtype=np.uint8).tobytes())
    if truncate:
        os.truncate(path, os.path.getsize(path) - truncate)


def _transform(i, s, inp):
    """stand-in for upload -> convert -> inpaint -> convert -> download: depends on the chunk (its temporal context) and on every input byte"""
    n = inp.shape[0]
    ctx = inp.astype(np.uint16).sum(axis=0, dtype=np.uint64) % 251                     # "temporal context": all frames of the chunk
    out = (inp[:, :FB_OUT].astype(np.uint16) * 3 + ctx[:FB_OUT][None] + i + np.arange(n)[:, None]) % 256
    return out.astype(np.uint8)


def _serial(src, dst, total, gap):
    from vsr_amd.backend.tools import chunk_parallel as cp

    data = np.fromfile(src, dtype=np.uint8)[len(HEAD_IN):]
    nrec = len(data) // (len(PREFIX) + FB_IN)
    recs = data[: nrec * (len(PREFIX) + FB_IN)].reshape(nrec, len(PREFIX) + FB_IN)[:, len(PREFIX):]
    with open(dst, "wb") as f:
        f.write(HEAD_OUT)
        for i, (s, e) in enumerate(cp.chunk_ranges(total, gap)):
            e = min(e, nrec)
            if e > s:
                for rec in _transform(i, s, recs[s:e]):
                    f.write(PREFIX)
                    f.write(rec.tobytes())
    return nrec


def _worker(rank, world, port, src, dst, total, gap, fail_rank, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import vsr_amd  # noqa: F401
    from vsr_amd.backend.tools import chunk_parallel as cp
    from vsr_amd.backend.tools import rank_io

    d = None
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        d = dist
    ranges = cp.chunk_ranges(total, gap)
    if rank == 0:
        with open(dst, "wb") as f:
            f.write(HEAD_OUT)
        rank_io.presize(dst, len(HEAD_OUT), len(PREFIX) + FB_OUT, total)
    if d is not None:
        d.barrier()
    seen, ticks = [], []

    def work(i, inp, out):
        if rank == fail_rank and i >= world:
            raise ValueError("boom")
        seen.append(i)
        out[:] = _transform(i, ranges[i][0], inp)

    err = None
    try:
        rank_io.run_rank_local(ranges, dict(path=src, data_offset=len(HEAD_IN), prefix=PREFIX, frame_bytes=FB_IN),
                               dict(path=dst, data_offset=len(HEAD_OUT), prefix=PREFIX, frame_bytes=FB_OUT), work, dist=d, tick=ticks.append)
    except Exception as e:      # noqa: BLE001
        err = type(e).__name__
    q.put((rank, seen, sum(ticks), err))
    if d is not None:
        dist.destroy_process_group()


def _run(tmp_path, world, total, gap, truncate=0, fail_rank=-1):
    src, dst, ref = str(tmp_path / "in.y4m"), str(tmp_path / "out.y4m"), str(tmp_path / "ref.y4m")
    _make_source(src, total, truncate)
    nrec = _serial(src, ref, total, gap)
    port = 21000 + (os.getpid() % 3000) + total + 11 * world + truncate % 97
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, src, dst, total, gap, fail_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return src, dst, ref, nrec, msgs


@pytest.mark.parametrize("world,total,gap", [(1, 23, 5), (2, 23, 5), (2, 20, 5), (3, 41, 4), (2, 3, 5), (3, 2, 5), (2, 0, 5)])
def test_ranks_write_the_single_process_file(tmp_path, world, total, gap):
    from vsr_amd.backend.tools import chunk_parallel as cp

    src, dst, ref, nrec, msgs = _run(tmp_path, world, total, gap)
    assert nrec == total
    assert open(dst, "rb").read() == open(ref, "rb").read(), "byte for byte the file of the serial loop"
    nchunks = len(cp.chunk_ranges(total, gap))
    for rank, seen, ticks, err in msgs:
        assert err is None
        assert seen == [i for i in range(nchunks) if i % world == rank], "round-robin, in this rank's chunk order"
        assert ticks == sum(e - s for i, (s, e) in enumerate(cp.chunk_ranges(total, gap)) if i % world == rank)


def test_short_source_ends_with_the_frames_read(tmp_path):
    """the header promised 23 frames, the file ends inside frame 17 (sttn_auto_inpaint.py:259-261: the chunk ends with the frames read)"""
    total, gap = 23, 5
    src, dst, ref, nrec, msgs = _run(tmp_path, 2, total, gap, truncate=6 * (len(PREFIX) + FB_IN) - 10)
    assert nrec == 17
    got, want = open(dst, "rb").read(), open(ref, "rb").read()
    assert got[: len(want)] == want                      # the frames that exist are the serial loop's
    assert len(got) == len(HEAD_OUT) + total * (len(PREFIX) + FB_OUT) and not any(got[len(want):]), "the pre-sized tail stays empty"
    assert all(err is None for _, _, _, err in msgs)


def test_a_failing_rank_fails_the_run_everywhere(tmp_path):
    src, dst, ref, nrec, msgs = _run(tmp_path, 3, 41, 4, fail_rank=1)
    errs = {rank: err for rank, _, _, err in msgs}
    assert errs[1] == "ValueError" and errs[0] == "RankIOError" and errs[2] == "RankIOError"
    # the pre-sized sink of a failed run is a full-size file with holes: rank 0 moves it out of the way
    assert not os.path.exists(dst) and os.path.exists(dst + ".failed")


def test_read_into_asks_again_after_a_short_read(tmp_path, monkeypatch):
    """a pread that returns fewer bytes than asked is not the end of the file"""
    from vsr_amd.backend.tools import rank_io

    path = str(tmp_path / "r.bin")
    recs = np.random.default_rng(1).integers(0, 256, size=(3, 1000), dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"HDR\n")
        for r in recs:
            f.write(b"FRAME\n" + r.tobytes())
    real = os.preadv
    monkeypatch.setattr(os, "preadv", lambda fd, bufs, off: real(fd, [bufs[0][:337]], off))     # at most 337 bytes per call
    rf = rank_io.RecordFile(path, 4, b"FRAME\n", 1000)
    out = np.zeros((3, 1000), dtype=np.uint8)
    assert rf.read_into(0, out) == 3 and np.array_equal(out, recs)
    rf.close()
