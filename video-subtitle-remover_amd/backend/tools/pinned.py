"""Pinned staging buffers without the wait.

Page-locking host memory costs ~0.7 ms per MB on the GPU boxes (profiles/r04_cli_startup.log: 183 ms for the four 69 MB plane
buffers of a 720p chunk loop, 0.4-0.5 s for the five 104 MB row buffers of the 1080p host-frame loop) -- all of it in front of the
first upload, i.e. in front of everything.  `PinnedPool` allocates the buffers on a helper thread in the order they will be needed
and hands out a buffer only once it exists; a caller that comes too early gets None and goes through ordinary (pageable) memory for
that one transfer, which the runtime stages itself at a few GB/s -- slower per byte than a pinned copy, far cheaper than waiting
for the page-locking in front of it.  By the second or third chunk every buffer is there and the loop runs as before.
"""
import threading
import time

import torch


class PinnedPool:
    test_delay = 0.0          # seconds the helper thread sleeps before every allocation (tests: forces the pageable path)

    def __init__(self, shapes, device=None, dtype=torch.uint8):
        self._shapes = [tuple(int(x) for x in s) for s in shapes]
        self._dtype = dtype
        self._bufs = [None] * len(self._shapes)
        self._ready = [threading.Event() for _ in self._shapes]
        self._error = None
        self._stop = False
        self._device = device
        self._thread = threading.Thread(target=self._fill, name="vsr-pinned-pool", daemon=True)
        self._thread.start()

    def _fill(self):
        try:
            if self._device is not None and torch.cuda.is_available():
                torch.cuda.set_device(self._device)
            for i, shp in enumerate(self._shapes):
                if self.test_delay:
                    time.sleep(self.test_delay)
                if self._stop:                       # the loop is over: nobody will ask for the rest
                    return
                self._bufs[i] = torch.empty(shp, dtype=self._dtype).pin_memory()
                self._ready[i].set()
        except BaseException as e:            # noqa: BLE001 -- surfaced by get(); the callers fall back to pageable memory
            self._error = e
            for ev in self._ready:
                ev.set()

    def get(self, i, wait=False):
        """buffer i, or None while it is not page-locked yet (wait=True blocks for it instead)"""
        if wait:
            self._ready[i].wait()
        return self._bufs[i] if self._ready[i].is_set() else None

    def shape(self, i):
        return self._shapes[i]

    def close(self):
        """the loop is over: stop page-locking (a short run ends before the thread does) and wait for the thread, so that no
        allocation is in flight when the process tears the runtime down"""
        self._stop = True
        self._thread.join()
        self._bufs = [None] * len(self._shapes)
