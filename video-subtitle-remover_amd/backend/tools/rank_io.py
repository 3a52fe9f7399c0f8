"""Every rank reads and writes ITS OWN chunks: the clip-parallel loop without a rank-0 funnel.

chunk_parallel.py keeps the reference's shape -- one frame source, one frame sink (sttn_auto_inpaint.py:242-328, video_io.py:12-103) --
and moves strip rows between rank 0 and its peers.  That scales the compute; the feed stays one reader / writer thread on rank 0
(~170 fps file to file at 1080p, profiles/r04_cli.log: less than ONE GPU inpaints).  A raw planar file (*.y4m) is a header plus
fixed-size records, so a chunk's frames are a byte range known up front: here rank r opens the source and the sink itself, `pread`s the
records of chunks r, r + world, ... into (pinned) staging, works on them, and `pwrite`s the result records at their final offsets of
the pre-sized output.  No frame crosses a rank boundary and no collective is on the data path; the ranks meet once at the end.  Chunk
boundaries are the reference's (clip_gap), dealt round-robin as in chunk_parallel.py, so every frame sees the temporal context it sees
in the reference and the output file is the single-process file byte for byte (tests/test_rank_io.py with 1 / 2 / 3 gloo ranks).

Pipes (ffmpeg) have no offsets: they keep the rank-0 scatter / gather path.
"""
import os
import queue
import threading

import numpy as np


class RankIOError(RuntimeError):
    """another rank failed in its chunks; this rank finished its own"""


class RecordFile:
    """header + fixed-size records (`prefix` + `frame_bytes` payload bytes each), addressed by record index"""

    def __init__(self, path, data_offset, prefix, frame_bytes, count=None, writable=False):
        self.path, self.data_offset, self.prefix, self.frame_bytes, self.count = path, int(data_offset), bytes(prefix), int(frame_bytes), count
        self.record_bytes = len(self.prefix) + self.frame_bytes
        self.fd = os.open(path, os.O_RDWR if writable else os.O_RDONLY)

    def offset(self, k):
        return self.data_offset + k * self.record_bytes

    def read_into(self, k0, out):
        """records k0 .. k0 + len(out) - 1 -> out [n][frame_bytes] (uint8); returns how many complete records were there"""
        n = 0
        for j in range(out.shape[0]):
            off = self.offset(k0 + j)
            if j == 0 and self.prefix and os.pread(self.fd, len(self.prefix), off) != self.prefix:
                break                                    # not a record boundary: the file is not what its header said
            mv, got = memoryview(out[j]), 0
            while got < self.frame_bytes:                # a short pread is not the end of the file (signals, network filesystems): ask again
                k = os.preadv(self.fd, [mv[got:]], off + len(self.prefix) + got)
                if k <= 0:
                    break
                got += k
            if got < self.frame_bytes:
                break
            n += 1
        return n

    def write_from(self, k0, recs):
        for j in range(recs.shape[0]):
            want = self.record_bytes
            if os.pwritev(self.fd, [self.prefix, memoryview(recs[j])], self.offset(k0 + j)) != want:
                raise OSError(f"short write at record {k0 + j} of {self.path}")

    def close(self):
        if self.fd is not None:
            os.close(self.fd)
            self.fd = None


def presize(path, data_offset, record_bytes, count):
    """the sink at its final size (rank 0, before anybody writes a record): sparse where the filesystem allows it"""
    with open(path, "r+b") as f:
        f.truncate(data_offset + record_bytes * count)


def run_rank_local(ranges, src, dst, work, dist=None, alloc=None, tick=None):
    """ranges: [(s, e)] frame ranges of the chunks, in order (chunk_parallel.chunk_ranges); src / dst: dicts for RecordFile
    (path, data_offset, prefix, frame_bytes); the sink exists at its final size (presize) before the first rank gets here.
    work(i, inp, out): inp uint8 [n][src frame_bytes] holds chunk i's records, fill out uint8 [n][dst frame_bytes]; called on the calling
    thread, in this rank's chunk order; both arrays are this rank's staging (alloc(kind, b, shape) -> uint8 array for kind "in" / "out",
    b = 0 / 1, asked for on first use -- e.g. pinned memory that a helper thread page-locks in that order) and are reused two chunks later.  A reader thread is one chunk ahead of work(), a writer thread one behind.  tick(n): n frames were written.
    Raises the first error of this rank; RankIOError when only another rank failed."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist is not None else (0, 1)
    mine = [i for i in range(len(ranges)) if i % world == rank]
    maxn = max((e - s for s, e in ranges), default=0)
    alloc = alloc or (lambda kind, b, shape: np.empty(shape, dtype=np.uint8))
    failure = []
    fin, fout = None, None
    try:
        fin = RecordFile(src["path"], src["data_offset"], src["prefix"], src["frame_bytes"])
        fout = RecordFile(dst["path"], dst["data_offset"], dst["prefix"], dst["frame_bytes"], writable=True)
        if mine and maxn:
            ibuf, obuf = [None, None], [None, None]
            filled, written = queue.Queue(), [threading.Semaphore(1), threading.Semaphore(1)]
            free_in = [threading.Semaphore(1), threading.Semaphore(1)]
            towrite = queue.Queue()

            def reader():
                try:
                    for q, i in enumerate(mine):
                        free_in[q % 2].acquire()             # work() is done with this staging buffer
                        if failure:
                            break
                        s, e = ranges[i]
                        if ibuf[q % 2] is None:
                            ibuf[q % 2] = alloc("in", q % 2, (maxn, fin.frame_bytes))
                        filled.put((i, q % 2, fin.read_into(s, ibuf[q % 2][: e - s])))
                except BaseException as ex:            # noqa: BLE001 -- re-raised on the calling thread
                    failure.append(ex)
                filled.put(None)

            def writer():
                while True:
                    item = towrite.get()
                    if item is None:
                        return
                    i, b, n = item
                    try:
                        if not failure:
                            fout.write_from(ranges[i][0], obuf[b][:n])
                            if tick is not None:
                                tick(n)
                    except BaseException as ex:        # noqa: BLE001
                        failure.append(ex)
                    written[b].release()

            tr = threading.Thread(target=reader, name="vsr-rank-io-read", daemon=True)
            tw = threading.Thread(target=writer, name="vsr-rank-io-write", daemon=True)
            tr.start()
            tw.start()
            try:
                q = 0
                while True:
                    item = filled.get()
                    if item is None:
                        break
                    i, b, n = item
                    s, e = ranges[i]
                    if n < e - s:                            # :259-261: a short read ends the video with the frames read so far
                        print(f"Warning: Failed to read frame {s + n}.")
                    written[q % 2].acquire()                 # the writer is done with this output staging buffer
                    try:
                        if obuf[q % 2] is None:
                            obuf[q % 2] = alloc("out", q % 2, (maxn, fout.frame_bytes))
                        if n and not failure:
                            work(i, ibuf[b][:n], obuf[q % 2][:n])
                    except BaseException as ex:        # noqa: BLE001
                        failure.append(ex)
                    free_in[b].release()
                    towrite.put((i, q % 2, n if not failure else 0))
                    q += 1
            finally:
                towrite.put(None)
                for sem in free_in:
                    sem.release()                            # a reader blocked on a buffer sees `failure` or its end
                tw.join()
                tr.join()
    except BaseException as ex:                    # noqa: BLE001
        failure.append(ex)
    finally:
        for f in (fin, fout):
            if f is not None:
                f.close()
    bad = 1 << 30
    if dist is not None:
        import torch

        # agree on the outcome (also the closing barrier: nobody reports success over a file another rank could not finish)
        status = torch.tensor([rank + 1 if failure else 1 << 30], dtype=torch.int64)
        if dist.get_backend() == "nccl":
            status = status.cuda()
        dist.all_reduce(status, op=dist.ReduceOp.MIN)
        bad = int(status.item())
    if (failure or bad != 1 << 30) and rank == 0:
        mark_failed(dst["path"])
    if failure:
        raise failure[0]
    if bad != 1 << 30:
        raise RankIOError(f"rank {bad - 1} failed in its chunks; this rank ({rank}) finished its own")


def mark_failed(path):
    """The sink was grown to its final size before anybody wrote (presize): after a failed run it is a full-size file with holes where
    records are missing -- nothing about it says so (the rank-0 funnel left a visibly truncated file).  Rank 0 renames it to
    `<path>.failed`; the exception still reaches the caller."""
    try:
        os.replace(path, path + ".failed")
    except OSError:
        pass
