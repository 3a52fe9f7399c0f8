"""Chunk-parallel execution of the sttn-auto loop over the GPUs of one node.

The chunks of STTNAutoInpaint.__call__ (reference backend/inpaint/sttn_auto_inpaint.py:242-328) share no state:
every chunk re-reads, re-encodes and re-decodes its own `clip_gap` frames.  They are therefore dealt round-robin to
the ranks (one process per GPU) with the reference's own chunk boundaries -- never re-chunked, because the boundaries
define the temporal context of every frame -- and there is no data-path collective: rank 0 owns the frame source and
sink and exchanges raw uint8 frames with each peer point-to-point (a flat star: on xGMI every peer has its own link
to rank 0, a ring would only add hops).  torch.distributed: backend "nccl" (= RCCL) on GPUs, "gloo" in CPU tests.
"""
import numpy as np
import torch


def chunk_ranges(total_frames, clip_gap):
    """[(start, end)] exactly as the reference splits a video (sttn_auto_inpaint.py:240-245)."""
    n = total_frames // clip_gap if total_frames % clip_gap == 0 else total_frames // clip_gap + 1
    return [(i * clip_gap, min((i + 1) * clip_gap, total_frames)) for i in range(n)]


def owner_of(chunk_index, world_size):
    return chunk_index % world_size


def chunks_of(rank, n_chunks, world_size):
    return [i for i in range(n_chunks) if owner_of(i, world_size) == rank]


def run_chunk_parallel(total_frames, clip_gap, frame_shape, read_chunk, process_chunk, write_chunk, dist=None,
                       device="cpu"):
    """Drive all chunks of one video over the ranks of `dist` (None = single process).

    read_chunk(start, end) -> uint8 ndarray [n,H,W,3]      (called on rank 0 only, in order)
    process_chunk(index, frames_tensor) -> uint8 tensor     (called on the owner, frames on `device`)
    write_chunk(index, ndarray)                              (called on rank 0 only, in chunk order)
    """
    ranges = chunk_ranges(total_frames, clip_gap)
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    H, W, C = frame_shape
    # processed in rounds of `world` chunks so that rank 0 never holds more than one round of results
    for base in range(0, len(ranges), world):
        idxs = list(range(base, min(base + world, len(ranges))))
        mine = None
        # scatter: rank 0 reads the round in order and ships each chunk to its owner
        if rank == 0:
            for i in idxs:
                s, e = ranges[i]
                frames = torch.from_numpy(np.ascontiguousarray(read_chunk(s, e)))
                o = owner_of(i, world)
                if o == 0:
                    mine = (i, frames.to(device))
                else:
                    dist.send(frames.to(device), dst=o)
        else:
            for i in idxs:
                if owner_of(i, world) == rank:
                    s, e = ranges[i]
                    buf = torch.empty((e - s, H, W, C), dtype=torch.uint8, device=device)
                    dist.recv(buf, src=0)
                    mine = (i, buf)
        out = None
        if mine is not None:
            out = process_chunk(mine[0], mine[1])
        # gather: results return to rank 0 and are written in chunk order
        if rank == 0:
            for i in idxs:
                o = owner_of(i, world)
                if o == 0:
                    write_chunk(i, out.cpu().numpy())
                else:
                    s, e = ranges[i]
                    buf = torch.empty((e - s, H, W, C), dtype=torch.uint8, device=device)
                    dist.recv(buf, src=o)
                    write_chunk(i, buf.cpu().numpy())
        elif out is not None:
            dist.send(out.contiguous(), dst=0)
    if dist is not None:
        dist.barrier()
