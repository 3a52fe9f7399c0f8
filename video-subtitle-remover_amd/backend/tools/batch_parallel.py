"""Batch-parallel execution of the detector-driven modes (sttn-det, propainter) over the GPUs of one node.

In both modes the host loop of the reference (backend/main.py:159-245 propainter_mode, :260-333 video_inpaint) walks the
video once and hands the plugin independent batches `model(batch, mask)`: no state crosses a batch, and the batch
boundaries (batch_generator over an interval of frames with one mask) define the temporal context of every frame.  The
batches are therefore dealt round-robin to the ranks exactly as the single-GPU loop cuts them -- never re-cut -- while
frames outside every interval pass through untouched.  As in chunk_parallel.py there is no data-path collective: rank 0
owns the frame source, the detector pass and the sink and exchanges raw uint8 frames (+ the batch's mask) with each
peer point-to-point; the peers only serve `process`.  torch.distributed: "nccl" (= RCCL) on GPUs, "gloo" in CPU tests.
"""
import numpy as np
import torch

PASS, WORK = "pass", "work"


def _serve(process, dist, device):
    """Peer loop: header [n, H, W] (n = 0 ends it), frames [n,H,W,3] u8, mask [H,W] u8 -> processed frames back."""
    while True:
        hdr = torch.zeros(3, dtype=torch.int64, device=device)
        dist.recv(hdr, src=0)
        n, H, W = (int(v) for v in hdr.cpu())
        if n == 0:
            return
        frames = torch.empty((n, H, W, 3), dtype=torch.uint8, device=device)
        mask = torch.empty((H, W), dtype=torch.uint8, device=device)
        dist.recv(frames, src=0)
        dist.recv(mask, src=0)
        out = process(list(frames.cpu().numpy()), mask.cpu().numpy())
        dist.send(torch.from_numpy(np.ascontiguousarray(np.stack(out))).to(device), dst=0)


def run_batch_parallel(items, process, write, dist=None, device="cpu"):
    """items: iterable consumed on rank 0 only, in video order, of (PASS, frame) or (WORK, [frames], mask [H,W] u8);
    process([frames], mask) -> [frames] runs on the batch's owner; write(frame) runs on rank 0 in video order."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    if rank != 0:
        _serve(process, dist, device)
        dist.barrier()
        return
    pending, work = [], []                    # output slots of the current round; its batches

    def flush():
        if not work:
            return
        for k, (frames, mask) in enumerate(work):            # peers first, so that they compute while rank 0 does
            o = k % world
            if o:
                arr = np.stack(frames)
                dist.send(torch.tensor([arr.shape[0], arr.shape[1], arr.shape[2]], dtype=torch.int64, device=device), dst=o)
                dist.send(torch.from_numpy(np.ascontiguousarray(arr)).to(device), dst=o)
                dist.send(torch.from_numpy(np.ascontiguousarray(mask, dtype=np.uint8)).to(device), dst=o)
        results = {}
        for k, (frames, mask) in enumerate(work):
            if k % world == 0:
                results[k] = process(frames, mask)
        for k, (frames, mask) in enumerate(work):
            o = k % world
            if o:
                buf = torch.empty((len(frames),) + tuple(frames[0].shape), dtype=torch.uint8, device=device)
                dist.recv(buf, src=o)
                results[k] = list(buf.cpu().numpy())
        for kind, v in pending:
            if kind == PASS:
                write(v)
            else:
                for f in results[v]:
                    write(f)
        pending.clear()
        work.clear()

    for item in items:
        if item[0] == PASS:
            if work:
                pending.append((PASS, item[1]))
            else:
                write(item[1])
        else:
            _, frames, mask = item
            if mask.ndim == 3:
                mask = mask[:, :, 0]
            pending.append((WORK, len(work)))
            work.append((list(frames), mask))
            if len(work) == world:
                flush()
    flush()
    if dist is not None:
        for o in range(1, world):
            dist.send(torch.zeros(3, dtype=torch.int64, device=device), dst=o)
        dist.barrier()
