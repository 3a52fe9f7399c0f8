"""Does torch.distributed's "nccl" backend (= RCCL) carry the point-to-point scatter / gather of tools/chunk_parallel.py?

    python scripts/nccl_try.py [world]            # spawns `world` ranks (default 2) over the visible GPUs

On a 1-GPU box all ranks land on cuda:0; RCCL (like NCCL) normally refuses that ("duplicate GPU"), which is reported, not
hidden.  With world = 1 the communicator, barrier and all_reduce are exercised.  Output goes to stdout (kept under profiles/).
"""
import os
import sys
import traceback

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    ngpu = torch.cuda.device_count()
    dev = torch.device("cuda", rank % ngpu)
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        t = torch.ones(4, device=dev) * (rank + 1)
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"[rank {rank}] nccl init + all_reduce ok on {dev} ({ngpu} GPU visible): {t.tolist()}", flush=True)
        import vsr_amd  # noqa: F401
        from vsr_amd.backend.tools import chunk_parallel as cp

        total, gap = 23, 5
        clip = (torch.arange(total * 8 * 16 * 3) % 251).to(torch.uint8).reshape(total, 8, 16, 3)
        ranges = cp.chunk_ranges(total, gap)
        out = {}
        cp.run_chunk_parallel(ranges, (8, 16, 3), lambda i, o: o.__setitem__(slice(0, ranges[i][1] - ranges[i][0]), clip[ranges[i][0]:ranges[i][1]].numpy()),
                              lambda i, t: t.add_(1 + rank), lambda i, a: out.__setitem__(i, a.copy()), dist=dist, device=dev)
        if rank == 0:
            ok = all((out[i] == ((clip[s:e].to(torch.int32) + 1 + i % world) % 256).to(torch.uint8).numpy()).all() for i, (s, e) in enumerate(ranges))
            print(f"[rank 0] chunk_parallel over nccl send/recv of uint8 tensors, world {world}: {'OK' if ok else 'MISMATCH'}", flush=True)
        dist.destroy_process_group()
    except Exception:
        print(f"[rank {rank}] FAILED:\n{traceback.format_exc()}", flush=True)


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    mp.spawn(worker, args=(world, 29871 + world), nprocs=world, join=True)
