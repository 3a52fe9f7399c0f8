"""N > 1 path on CPU: two gloo processes deal the chunks of one clip and rank 0 reassembles them in order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, gap, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import vsr_amd  # noqa: F401
    from vsr_amd.backend.tools import chunk_parallel as cp

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clip = (np.arange(total * 4 * 6 * 3) % 251).astype(np.uint8).reshape(total, 4, 6, 3)
    written = {}
    processed = []

    def read_chunk(s, e):
        return clip[s:e]

    def process_chunk(i, frames):          # stand-in for engine.auto_chunk: mark which rank / chunk touched it
        processed.append(i)
        out = frames.clone()
        out[:, 0, 0, 0] = 100 + rank
        out[:, 0, 0, 1] = i
        return out

    def write_chunk(i, arr):
        written[i] = arr

    cp.run_chunk_parallel(total, gap, (4, 6, 3), read_chunk, process_chunk, write_chunk, dist=dist)
    if rank == 0:
        q.put(("written", {k: v.copy() for k, v in written.items()}))
    q.put(("processed", rank, processed))
    dist.destroy_process_group()


@pytest.mark.parametrize("total,gap", [(23, 5), (20, 5), (3, 5)])
def test_two_rank_chunk_parallel(total, gap):
    from vsr_amd.backend.tools import chunk_parallel as cp

    world = 2
    port = 29500 + (os.getpid() % 2000) + total
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, gap, q)) for r in range(world)]
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
    clip = (np.arange(total * 4 * 6 * 3) % 251).astype(np.uint8).reshape(total, 4, 6, 3)
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


def test_chunk_ranges_match_reference_loop():
    from vsr_amd.backend.tools import chunk_parallel as cp

    assert cp.chunk_ranges(300, 50) == [(i * 50, i * 50 + 50) for i in range(6)]
    assert cp.chunk_ranges(120, 50) == [(0, 50), (50, 100), (100, 120)]
    assert cp.chunk_ranges(0, 50) == []
