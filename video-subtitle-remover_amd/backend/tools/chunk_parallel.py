"""Chunk-parallel execution of the sttn-auto loop over the GPUs of one node.

The chunks of STTNAutoInpaint.__call__ (reference backend/inpaint/sttn_auto_inpaint.py:242-328) share no state:
every chunk re-reads, re-encodes and re-decodes its own `clip_gap` frames.  They are therefore dealt round-robin to
the ranks (one process per GPU) with the reference's own chunk boundaries -- never re-chunked, because the boundaries
define the temporal context of every frame -- and there is no data-path collective: rank 0 owns the frame source and
sink and exchanges raw uint8 frame rows with each peer point-to-point (a flat star: on xGMI every peer has its own
link to rank 0, a ring would only add hops).  torch.distributed: backend "nccl" (= RCCL) on GPUs, "gloo" in CPU tests.

Pipeline.  Chunk i belongs to round i // world and to rank i % world.  All ranks walk the rounds in lock step and
meet in one grouped point-to-point exchange per step (dist.batch_isend_irecv -> one ncclGroup on RCCL, so the order
of sends and receives inside a step cannot deadlock):

    exchange X[e] = { rank 0 -> owner : the chunks of round e        (scatter)
                      owner -> rank 0 : the results of round e - 2   (gather)  }

    step r, every rank:  launch compute(round r)  |  post X[r+1]  |  rank 0: write round r-2, read + upload round r+2

X[r+1] is posted right after compute(r) has been launched and from another stream, so the transfer of the next chunk
and of the previous result runs under the compute of the current one -- on the peers as well as on rank 0, whose host
thread meanwhile drains round r-2 to the writer and reads round r+2 into pinned memory.  Device buffers form a ring
of four rounds (288 GB of HBM: a 50-frame 4K strip chunk is 415 MB).  Only the rows that can change (the strips, see
STTNAutoInpaint._call_chunk_parallel) travel; rank 0 keeps the decoded frames and patches the rows back in.

Failures (round 3).  The ranks meet in matched exchanges, so a rank that left the loop on an exception would leave the others
blocked in a grouped send / recv until the communicator times out.  Instead a failing callback (the reader or the sink on rank 0,
the engine on any rank) is recorded, the rank keeps walking the rounds with its callbacks switched off -- every exchange still
has its partner, the data is just not meaningful any more -- and after the last round the ranks agree on a status word (one
all-reduce); every rank then raises: the one that failed its own exception, the others a RuntimeError naming the failed rank.
Before the first round the ring's device memory (RING x owners x a chunk of rows) is checked against hipMemGetInfo.
"""
import contextlib

import numpy as np
import torch

RING = 4


def chunk_ranges(total_frames, clip_gap):
    """[(start, end)] exactly as the reference splits a video (sttn_auto_inpaint.py:240-245)."""
    n = total_frames // clip_gap if total_frames % clip_gap == 0 else total_frames // clip_gap + 1
    return [(i * clip_gap, min((i + 1) * clip_gap, total_frames)) for i in range(n)]


def owner_of(chunk_index, world_size):
    return chunk_index % world_size


def chunks_of(rank, n_chunks, world_size):
    return [i for i in range(n_chunks) if owner_of(i, world_size) == rank]


class _Streams:
    """two HIP streams + events on a GPU; plain synchronous execution on the CPU (gloo tests)"""

    def __init__(self, device):
        self.gpu = torch.device(device).type == "cuda"
        self.device = torch.device(device)
        if self.gpu:
            self.io = torch.cuda.Stream(self.device)
            self.cmp = torch.cuda.Stream(self.device)

    def on(self, which):
        return torch.cuda.stream(self.io if which == "io" else self.cmp) if self.gpu else contextlib.nullcontext()

    def event(self, which):
        """record an event at the tail of a stream (None on the CPU)"""
        if not self.gpu:
            return None
        ev = torch.cuda.Event()
        ev.record(self.io if which == "io" else self.cmp)
        return ev

    def wait(self, which, ev):
        if self.gpu and ev is not None:
            (self.io if which == "io" else self.cmp).wait_event(ev)


class _HostStaged:
    """gloo moves host memory only: a dry run of the N > 1 path on GPUs without RCCL (several ranks on one GPU, where RCCL
    refuses to build a communicator) bounces every transfer through a host copy.  Never the production path."""

    def __init__(self, dist, ops):
        self.pairs = [(kind, t, t.cpu() if kind == "send" else torch.empty(t.shape, dtype=t.dtype)) for kind, t, _ in ops]
        self.works = dist.batch_isend_irecv([dist.P2POp(dist.isend if kind == "send" else dist.irecv, c, peer)
                                             for (kind, _, c), (_, _, peer) in zip(self.pairs, ops)])

    def wait(self):
        for w in self.works:
            w.wait()
        for kind, t, c in self.pairs:
            if kind == "recv":
                t.copy_(c)


def _exchange(dist, ops):
    """post one grouped exchange; returns the work handles"""
    if not ops:
        return []
    if dist.get_backend() == "gloo" and ops[0][1].is_cuda:
        return [_HostStaged(dist, ops)]
    return dist.batch_isend_irecv([dist.P2POp(dist.isend if kind == "send" else dist.irecv, t, peer) for kind, t, peer in ops])


class ChunkParallelError(RuntimeError):
    """another rank failed while this one was fine (its own exception is raised there)"""


def ring_bytes(maxn, row_shape, world, rank):
    """device memory of the row ring on `rank`: RING rounds x (every owner on rank 0, itself elsewhere) x one chunk of rows"""
    h, W, C = row_shape
    return RING * (world if rank == 0 else 1) * maxn * h * W * C


def run_chunk_parallel(ranges, row_shape, load, process, store, dist=None, device="cpu", io="host", workspace_bytes=0):
    """Drive the chunks `ranges` = [(start, end)] of one video over the ranks of `dist` (None = single process).

    row_shape = (h, W, C): the rows of a frame that travel.
    load(i, out)      rank 0 only, in chunk order: fill out[:n] (uint8 [n,h,W,C]: a pinned numpy array for io="host", a
                      device tensor for io="device") with the rows of chunk i.
    process(i, t)     on the owner of chunk i: work in place on the device tensor t uint8 [n,h,W,C]; may return while
                      its kernels are still running on the current stream.
    store(i, arr)     rank 0 only, in chunk order: the processed rows (same kind of array as load's).
    workspace_bytes   what `process` needs besides the ring (the engine's buffers, if not allocated yet): part of the memory check.

    A callback that raises does not break the lock step (module docstring): all ranks finish the rounds, then all raise.
    """
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    h, W, C = row_shape
    n_rounds = (len(ranges) + world - 1) // world
    maxn = max((e - s for s, e in ranges), default=0)
    st = _Streams(device)
    dev = st.device
    host_io = io == "host"

    def chunk_of(r, k):
        i = r * world + k
        return i if 0 <= r and i < len(ranges) else None

    def nframes(i):
        return ranges[i][1] - ranges[i][0]

    if st.gpu:
        torch.cuda.set_device(dev)             # RCCL point-to-point needs the current device to be this rank's
    if dist is not None:
        dist.barrier()                         # the grouped p2p below must not be the first call on the communicator:
                                               # ranks without a chunk in round 0 post nothing in X[0]
    if maxn == 0:
        return
    if st.gpu:
        # the ring must fit beside whatever else lives on this device (several ranks may share one in dry runs): fail with the
        # numbers instead of an allocator error in the middle of round 0
        free_b, total_b = torch.cuda.mem_get_info(dev)
        need = ring_bytes(maxn, row_shape, world, rank) + int(workspace_bytes)
        if need > free_b:
            raise MemoryError(f"chunk ring on rank {rank}: {need / 2**30:.2f} GiB needed ({RING} rounds x {world if rank == 0 else 1} owners x "
                              f"{maxn} frames of {h}x{W}x{C} rows + {workspace_bytes / 2**30:.2f} GiB workspace), {free_b / 2**30:.2f} GiB of "
                              f"{total_b / 2**30:.0f} GiB free")
    failure = []                              # first exception of a callback on this rank

    def guarded(fn, *a):
        if failure:
            return
        try:
            fn(*a)
        except BaseException as e:            # noqa: BLE001 -- recorded, re-raised after the last round
            failure.append(e)

    owners = range(world) if rank == 0 else [rank]
    pool = None
    dbuf = {(q, k): torch.empty((maxn, h, W, C), dtype=torch.uint8, device=dev) for q in range(RING) for k in owners}
    if rank == 0 and host_io and st.gpu:
        # pinned staging, page-locked by a helper thread in the order of first use (tools/pinned.py): rounds 0 and 1 in, the results
        # out, then the rest of the ring; a transfer whose buffer is not there yet goes through pageable memory once
        from .pinned import PinnedPool

        order = [("in", q, k) for q in (0, 1) for k in range(world)] + [("out", 0, k) for k in range(world)] + \
                [("in", q, k) for q in range(2, RING) for k in range(world)]
        pool = PinnedPool([(maxn, h, W, C)] * len(order), device=dev)
        slot = {key: j for j, key in enumerate(order)}
        pageable = {}

        def host_buf(kind, q, k):
            """(tensor, pinned?) for this transfer"""
            t = pool.get(slot[(kind, q, k)])
            if t is not None:
                return t, True
            if kind not in pageable:
                pageable[kind] = torch.empty((maxn, h, W, C), dtype=torch.uint8)
            return pageable[kind], False
    staged, computed = {}, {}                 # (round) -> event after upload of the own chunk / after its compute

    def stage(r):                              # rank 0: read round r and bring it to the device (io stream)
        for k in range(world):
            i = chunk_of(r, k)
            if i is None:
                continue
            n, d = nframes(i), dbuf[(r % RING, k)]
            if not host_io:
                with st.on("io"):
                    guarded(load, i, d[:n])
            elif st.gpu:
                p, pinned = host_buf("in", r % RING, k)
                guarded(load, i, p.numpy()[:n])
                with st.on("io"):
                    d[:n].copy_(p[:n], non_blocking=pinned)     # (a pageable source is copied before the call returns)
            else:
                guarded(load, i, d.numpy()[:n])
        staged[r] = st.event("io")

    def drain(r):                              # rank 0: results of round r -> sink, in chunk order
        for k in range(world):
            i = chunk_of(r, k)
            if i is None:
                continue
            n, d = nframes(i), dbuf[(r % RING, k)]
            if k == 0:
                st.wait("io", computed.get(r))
            if not host_io:
                with st.on("io"):
                    guarded(store, i, d[:n])
            elif st.gpu:
                p, pinned = host_buf("out", 0, k)
                with st.on("io"):
                    p[:n].copy_(d[:n], non_blocking=pinned)
                st.event("io").synchronize()
                guarded(store, i, p.numpy()[:n])
            else:
                guarded(store, i, d.numpy()[:n])

    def compute(r):
        i = chunk_of(r, rank)
        if i is None:
            return
        st.wait("cmp", staged.get(r))
        with st.on("cmp"):
            guarded(process, i, dbuf[(r % RING, rank)][:nframes(i)])
        computed[r] = st.event("cmp")

    def post(e):                               # X[e]: scatter round e, gather round e - 2
        ops = []
        peers = range(1, world) if rank == 0 else [rank]
        for k in peers:
            i, j = chunk_of(e, k), chunk_of(e - 2, k)
            if rank == 0:
                if i is not None:
                    ops.append(("send", dbuf[(e % RING, k)][:nframes(i)], k))
                if j is not None:
                    ops.append(("recv", dbuf[((e - 2) % RING, k)][:nframes(j)], k))
            else:
                if i is not None:
                    ops.append(("recv", dbuf[(e % RING, k)][:nframes(i)], 0))
                if j is not None:
                    st.wait("io", computed.get(e - 2))
                    ops.append(("send", dbuf[((e - 2) % RING, k)][:nframes(j)], 0))
        with st.on("io"):
            works = _exchange(dist, ops) if dist is not None else []
        return works

    def finish(works, r_recv):                 # the io stream (GPU) or the host (CPU) waits for the exchange
        with st.on("io"):
            for w in works:
                w.wait()
        if rank != 0 and r_recv is not None and chunk_of(r_recv, rank) is not None:
            staged[r_recv] = st.event("io")    # the peer's chunk of round r_recv has arrived

    try:
        if rank == 0:
            stage(0)
            stage(1)
        finish(post(0), 0)
        for r in range(n_rounds + 2):
            compute(r)
            works = post(r + 1)
            if rank == 0:
                drain(r - 2)
                stage(r + 2)
            finish(works, r + 1)
            ev = computed.get(r)
            if ev is not None:
                ev.synchronize()                   # paces the host: at most one round of launches ahead of the GPU
            staged.pop(r - RING, None)
            computed.pop(r - RING, None)
        if st.gpu:
            try:
                torch.cuda.synchronize(dev)
            except BaseException as e:            # noqa: BLE001 -- an asynchronous kernel error surfaces here
                if not failure:
                    failure.append(e)
    finally:
        if pool is not None:
            pool.close()                           # also when an exchange raised: no page-locking thread outlives the loop
                                                   # (and a short run ends before that thread does)
    if dist is not None:
        # agree on the outcome: the lowest failed rank + 1, 0 = everybody fine (all-reduce MAX of -(rank + 1) would do too; MIN
        # over a large sentinel keeps it one collective).  Also the closing barrier.
        status = torch.tensor([rank + 1 if failure else 1 << 30], dtype=torch.int64, device=dev if st.gpu else "cpu")
        dist.all_reduce(status, op=dist.ReduceOp.MIN)
        bad = int(status.item())
        if failure:
            raise failure[0]
        if bad != 1 << 30:
            raise ChunkParallelError(f"rank {bad - 1} failed in its chunk callbacks; this rank ({rank}) finished its rounds")
    elif failure:
        raise failure[0]
