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


class _Prefetch:
    """Runs the `items` generator (reader + mask construction) in a thread, about `max_frames` frames ahead -- the role the
    reference gives FramePrefetcher (tools/video_io.py:12-47): reading the next round overlaps the current round's compute.
    The budget is a frame counter under a condition variable; an item is ALWAYS admitted when nothing is queued, whatever its
    size, so a batch larger than the budget (config.sttnMaxLoadNum / propainterMaxLoadNum go up to 300, the default budget is 256
    frames) passes alone instead of waiting for room that only its own consumption could make (round 2 took the budget frame by
    frame from a semaphore before queueing the item: such a batch hung the run -- ADVICE r2)."""

    def __init__(self, items, max_frames):
        import queue
        import threading

        self.q = queue.Queue()
        self.budget = max(1, int(max_frames))
        self.ahead = 0                           # frames queued and not yet consumed
        self.cv = threading.Condition()
        self.stop = False
        self.t = threading.Thread(target=self._run, args=(items,), daemon=True)
        self.t.start()

    @staticmethod
    def _cost(item):
        return 1 if item[0] == PASS else max(1, len(item[1]))

    def _run(self, items):
        try:
            for item in items:
                c = self._cost(item)
                with self.cv:
                    while not self.stop and self.ahead > 0 and self.ahead + c > self.budget:
                        self.cv.wait()
                    if self.stop:
                        return
                    self.ahead += c
                self.q.put(("item", item))
            self.q.put(("end", None))
        except BaseException as e:          # surfaces in the consumer
            self.q.put(("error", e))

    def __iter__(self):
        while True:
            kind, v = self.q.get()
            if kind == "end":
                return
            if kind == "error":
                raise v
            yield v
            with self.cv:
                self.ahead -= self._cost(v)
                self.cv.notify_all()

    def close(self):
        with self.cv:
            self.stop = True
            self.cv.notify_all()


def run_batch_parallel(items, process, write, dist=None, device="cpu", max_pending=64, prefetch_frames=256):
    """items: iterable consumed on rank 0 only, in video order, of (PASS, frame) or (WORK, [frames], mask [H,W] u8);
    process([frames], mask) -> [frames] runs on the batch's owner; write(frame) runs on rank 0 in video order.

    A round is flushed when every rank has a batch, and also -- so that a long subtitle-free stretch after a batch neither
    buffers the video in host memory nor leaves the queued batch unprocessed -- once `max_pending` pass-through frames wait
    behind an incomplete round.  Whatever happens on rank 0 (reader, detector or plugin raising), the peers are released."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    if rank != 0:
        _serve(process, dist, device)
        dist.barrier()
        return
    pending, work = [], []                    # output slots of the current round; its batches
    dealt = [0]                               # batches dealt so far: batch b belongs to rank b % world, partial rounds included

    def flush():
        if not work:
            return
        base = dealt[0]
        dealt[0] += len(work)
        for k, (frames, mask) in enumerate(work):            # peers first, so that they compute while rank 0 does
            o = (base + k) % world
            if o:
                arr = np.stack(frames)
                dist.send(torch.tensor([arr.shape[0], arr.shape[1], arr.shape[2]], dtype=torch.int64, device=device), dst=o)
                dist.send(torch.from_numpy(np.ascontiguousarray(arr)).to(device), dst=o)
                dist.send(torch.from_numpy(np.ascontiguousarray(mask, dtype=np.uint8)).to(device), dst=o)
        results = {}
        for k, (frames, mask) in enumerate(work):
            if (base + k) % world == 0:
                results[k] = process(frames, mask)
        for k, (frames, mask) in enumerate(work):
            o = (base + k) % world
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

    source = _Prefetch(items, prefetch_frames) if prefetch_frames else None
    try:
        n_pass = 0
        for item in (source if source is not None else items):
            if item[0] == PASS:
                if work:
                    pending.append((PASS, item[1]))
                    n_pass += 1
                    if n_pass >= max_pending:
                        flush()
                        n_pass = 0
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
                    n_pass = 0
        flush()
    finally:
        if source is not None:
            source.close()
        if dist is not None:
            for o in range(1, world):           # always release the peers, also when rank 0 is unwinding an exception
                try:
                    dist.send(torch.zeros(3, dtype=torch.int64, device=device), dst=o)
                except Exception:
                    pass
            dist.barrier()
