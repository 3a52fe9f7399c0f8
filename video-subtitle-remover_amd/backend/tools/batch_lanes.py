"""Batch lanes of the HBM-resident detector-driven modes (opt-in: VSR_BATCH_LANES=2; default 1 = the plain loop) and detector lanes of
their sampling pass (VSR_DET_LANES, default 2 since round 4).

The batches `batch_generator` cuts out of a detected interval are independent units (main.py:229-245, :323-332: every call of the
plugin starts from the frames of its own batch), and on the resident path they are disjoint slices of one device tensor that the
plugin rewrites in place.  A plugin call is a long sequence of dependent launches with host work between its stages (the
propainter plugin runs three engines back to back), so one stream leaves the GPU idle at every launch tail and every host step;
with L lanes, L host threads -- each with its own plugin instance (own engines and workspaces) and its own stream -- pull
batches from one queue.  Every batch is computed by the same code on the same inputs as in the plain loop, so the frames are
identical whatever the interleaving.  The same idea one level down: the window lanes inside the STTN engine.
"""
import os
import queue
import threading


DEFAULT_LANES = {"VSR_BATCH_LANES": 1,      # plugin instances over the batches of a resident run: measured, no gain (profiles/r04_lanes_e2e.log)
                 "VSR_DET_LANES": 2}        # detectors over the sampled batches: 4.55 -> 4.10 s per 600 frames (r04_lanes_e2e.log, r04_e2e_store.log)


def lanes_from_env(name="VSR_BATCH_LANES", default=None):
    """the lane count `name` asks for; `default` (when given) replaces the table's default for a caller that measured otherwise"""
    dflt = DEFAULT_LANES.get(name, 1) if default is None else default
    try:
        return max(1, min(4, int(os.environ.get(name, dflt))))
    except ValueError:
        return dflt


def lane_plugins(plugin, lanes, cache):
    """[plugin] + clones for the other lanes (kept in `cache`, a dict owned by the caller, so a run builds them once).  A plugin
    without clone() -- an injected callable -- stays on one lane."""
    if lanes <= 1 or not hasattr(plugin, "clone"):
        return [plugin]
    have = cache.setdefault(id(plugin), [])
    while len(have) < lanes - 1:
        have.append(plugin.clone())
    return [plugin] + have[:lanes - 1]


def run_map(jobs, workers, call, device=None):
    """[call(worker, job) for job in jobs], the jobs spread over the workers.  One worker: a plain loop in the caller's thread.
    More: one thread per worker, each on its own stream of `device` (a torch.device; None / cpu: plain threads), all pulling from
    one queue; the caller's stream is joined behind every lane; the first exception stops the queue and is re-raised here."""
    if len(workers) <= 1 or len(jobs) <= 1:
        return [call(workers[0], job) for job in jobs]
    todo = queue.Queue()
    for item in enumerate(jobs):
        todo.put(item)
    results, errors = [None] * len(jobs), []
    cuda = device is not None and getattr(device, "type", "cpu") == "cuda"
    if cuda:
        import torch

        caller = torch.cuda.current_stream(device)
        streams = [torch.cuda.Stream(device) for _ in workers]
        for s in streams:
            s.wait_stream(caller)                   # what the caller prepared (upload, colour conversion) is ready

    def lane(k):
        def pull():
            while not errors:
                try:
                    i, job = todo.get_nowait()
                except queue.Empty:
                    return
                results[i] = call(workers[k], job)

        try:
            if cuda:
                with torch.cuda.device(device), torch.cuda.stream(streams[k]):
                    pull()
                    streams[k].synchronize()
            else:
                pull()
        except BaseException as e:                  # noqa: BLE001 -- re-raised in the caller's thread
            errors.append(e)

    threads = [threading.Thread(target=lane, args=(k,), name=f"vsr-batch-lane-{k}") for k in range(len(workers))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if cuda:
        for s in streams:
            caller.wait_stream(s)
    if errors:
        raise errors[0]
    return results


def run_jobs(jobs, plugins, device=None):
    """jobs: [(frames, mask)] with disjoint `frames`, each rewritten in place by plugin(frames, mask); plugins: one per lane"""
    run_map(jobs, plugins, lambda plugin, job: plugin(*job), device)
