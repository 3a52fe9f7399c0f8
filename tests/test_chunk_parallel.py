"""N > 1 path on CPU: two gloo processes deal the chunks of one clip (pipelined scatter / compute / gather of
tools/chunk_parallel.py) and rank 0 reassembles them in order."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clip(total):
    return (np.arange(total * 4 * 6 * 3) % 251).astype(np.uint8).reshape(total, 4, 6, 3)


def _worker(rank, world, port, total, gap, q, io="host"):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import vsr_amd  # noqa: F401
    from vsr_amd.backend.tools import chunk_parallel as cp

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clip = _clip(total)
    ranges = cp.chunk_ranges(total, gap)
    written, processed, loaded = {}, [], []

    def load(i, out):
        assert rank == 0
        s, e = ranges[i]
        loaded.append(i)
        if io == "device":                 # the resident loop of STTNAutoInpaint hands over / receives device tensors (here: CPU tensors)
            import torch

            assert isinstance(out, torch.Tensor)
            out[:e - s] = torch.from_numpy(clip[s:e])
        else:
            out[:e - s] = clip[s:e]

    def process(i, frames):                # stand-in for engine.auto_chunk (in place): mark which rank / chunk touched it
        processed.append(i)
        assert frames.shape[0] == ranges[i][1] - ranges[i][0]
        frames[:, 0, 0, 0] = 100 + rank
        frames[:, 0, 0, 1] = i

    def store(i, arr):
        assert rank == 0 and i == len(written), "results are written in chunk order"
        written[i] = np.array(arr, copy=True)

    cp.run_chunk_parallel(ranges, (4, 6, 3), load, process, store, dist=dist, io=io)
    if rank == 0:
        assert loaded == list(range(len(ranges))), "the source is read once, in order"
        q.put(("written", written))
    q.put(("processed", rank, processed))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total,gap,io", [(2, 23, 5, "host"), (2, 20, 5, "host"), (2, 3, 5, "host"), (3, 41, 4, "host"), (2, 0, 5, "host"),
                                                (2, 23, 5, "device"), (3, 10, 4, "device")])
def test_chunk_parallel_over_gloo_ranks(world, total, gap, io):
    from vsr_amd.backend.tools import chunk_parallel as cp

    port = 29500 + (os.getpid() % 2000) + total + 7 * world + (500 if io == "device" else 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, gap, q, io)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = [q.get(timeout=120) for _ in range(world + 1)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    written = next(m[1] for m in msgs if m[0] == "written")
    done = {m[1]: m[2] for m in msgs if m[0] == "processed"}
    ranges = cp.chunk_ranges(total, gap)
    assert sorted(written) == list(range(len(ranges)))
    clip = _clip(total)
    for i, (s, e) in enumerate(ranges):
        got = written[i]
        assert got.shape[0] == e - s
        assert (got[:, 0, 0, 0] == 100 + i % world).all() and (got[:, 0, 0, 1] == i).all()
        ref = clip[s:e].copy()
        ref[:, 0, 0, 0] = got[:, 0, 0, 0]
        ref[:, 0, 0, 1] = got[:, 0, 0, 1]
        assert np.array_equal(got, ref), "frames must come back unpermuted"
    for r in range(world):
        assert done[r] == cp.chunks_of(r, len(ranges), world), "round-robin ownership, reference chunk boundaries"


def _failing_worker(rank, world, port, where, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import vsr_amd  # noqa: F401
    from vsr_amd.backend.tools import chunk_parallel as cp

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clip = _clip(40)
    ranges = cp.chunk_ranges(40, 4)                       # 10 chunks, 5 rounds of two
    stored = []

    def load(i, out):
        if where == "load" and i == 5:
            raise IOError("reader died at chunk 5")
        out[:4] = clip[ranges[i][0]:ranges[i][1]]

    def process(i, frames):
        if where == "peer" and rank == 1 and i == 3:
            raise RuntimeError("engine fault on rank 1")
        frames += 1

    def store(i, arr):
        if where == "store" and i == 2:
            raise IOError("disk full at chunk 2")
        stored.append(i)

    err = None
    try:
        cp.run_chunk_parallel(ranges, (4, 6, 3), load, process, store, dist=dist)
    except BaseException as e:                            # noqa: BLE001
        err = (type(e).__name__, str(e))
    q.put((rank, err, stored))
    dist.barrier()                                        # the communicator is still usable: nobody was left behind in an exchange
    dist.destroy_process_group()


@pytest.mark.parametrize("where", ["load", "store", "peer"])
def test_a_failing_callback_releases_every_rank(where):
    """VERDICT r2 #5 / ADVICE r2: an exception on one rank must not leave the others blocked in a grouped exchange.  The failing rank
    raises its own exception, the other a ChunkParallelError naming it, both after walking all rounds; nothing hangs."""
    port = 31500 + (os.getpid() % 1500) + {"load": 0, "store": 1, "peer": 2}[where]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, where, q)) for r in range(2)]
    for p in procs:
        p.start()
    msgs = {m[0]: m for m in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bad, good = (1, 0) if where == "peer" else (0, 1)
    assert msgs[bad][1] is not None and msgs[bad][1][0] in ("OSError", "RuntimeError")
    assert ("reader died" in msgs[bad][1][1]) or ("disk full" in msgs[bad][1][1]) or ("engine fault" in msgs[bad][1][1])
    assert msgs[good][1] == ("ChunkParallelError", f"rank {bad} failed in its chunk callbacks; this rank ({good}) finished its rounds")
    if where == "store":
        assert msgs[0][2] == [0, 1]                       # nothing is written after the sink failed
    if where == "peer":
        assert msgs[0][2] == list(range(10))              # rank 0 kept writing what came back (the peer's chunks unprocessed)


def test_single_process_failure_is_raised():
    from vsr_amd.backend.tools import chunk_parallel as cp

    ranges = cp.chunk_ranges(8, 4)
    with pytest.raises(ValueError, match="bad chunk"):
        cp.run_chunk_parallel(ranges, (4, 6, 3), lambda i, o: None, lambda i, t: (_ for _ in ()).throw(ValueError("bad chunk")), lambda i, a: None)


def test_ring_bytes():
    from vsr_amd.backend.tools import chunk_parallel as cp

    # 4 rounds x 8 owners x 50 frames of a 1080p strip (360 rows) on rank 0; one owner on a peer
    assert cp.ring_bytes(50, (360, 1920, 3), 8, 0) == 4 * 8 * 50 * 360 * 1920 * 3
    assert cp.ring_bytes(50, (360, 1920, 3), 8, 3) == 4 * 50 * 360 * 1920 * 3


def test_single_process_path():
    """dist=None: the same driver degenerates to load -> process -> store per chunk."""
    from vsr_amd.backend.tools import chunk_parallel as cp

    clip = _clip(11)
    ranges = cp.chunk_ranges(11, 4)
    out = {}
    cp.run_chunk_parallel(ranges, (4, 6, 3), lambda i, o: o.__setitem__(slice(0, ranges[i][1] - ranges[i][0]), clip[ranges[i][0]:ranges[i][1]]),
                          lambda i, t: t.add_(1), lambda i, a: out.__setitem__(i, np.array(a, copy=True)))
    assert np.array_equal(np.concatenate([out[i] for i in range(3)]), clip + 1)


def test_chunk_ranges_match_reference_loop():
    from vsr_amd.backend.tools import chunk_parallel as cp

    assert cp.chunk_ranges(300, 50) == [(i * 50, i * 50 + 50) for i in range(6)]
    assert cp.chunk_ranges(120, 50) == [(0, 50), (50, 100), (100, 120)]
    assert cp.chunk_ranges(0, 50) == []
