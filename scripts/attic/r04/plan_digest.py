"""Digest of everything a plan exposes through vsr_plan_* (ops, GEMM / softmax items, tables, buffer sizes, counts, FLOPs), for a few
batch lengths and decoder row ranges of both STTN variants.  Run against two trees to show that a change to the plan builder left
the plans that were already there byte for byte alone:

    python scripts/r04/plan_digest.py <repo root>      (round 4: the column ranges of DESIGN 4.3c against the commit in front of them)
"""
import ctypes as C
import hashlib
import os
import sys

root = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np  # noqa: E402

import vsr_amd  # noqa: E402,F401
from vsr_amd import _lib  # noqa: E402
from vsr_amd.engine import SttnEngine  # noqa: E402
from vsr_amd.synth import make_state_dict  # noqa: E402
from _replay import PlanView  # noqa: E402


def struct_bytes(s):
    return bytes(memoryview(s).cast("B")) if not isinstance(s, C.Structure) else C.string_at(C.addressof(s), C.sizeof(s))


def digest(view):
    h = hashlib.sha256()
    h.update(np.asarray(view.buf_elems, dtype=np.int64).tobytes())
    for t in view.tables:
        h.update(np.asarray(t, dtype=np.int64).tobytes())
    for info, items in view.ops:
        h.update(struct_bytes(info))
        for it in items:
            h.update(struct_bytes(it))
    h.update(view.counts.tobytes())
    h.update(np.float64(view.flops).tobytes())
    return h.hexdigest()[:16], len(view.ops), len(view.tables)


for variant, ranges in (("auto", (None, (76, 118), (0, 8), (56, 64))), ("det", (None, (180, 232), (0, 40)))):
    eng = SttnEngine(make_state_dict(1, variant), variant, device=None)
    for L in (1, 4, 13, 50):
        for rows in ranges:
            v = PlanView(_lib, eng, L, rows=rows)
            print(variant, L, rows, *digest(v), "%.6e" % v.flops)
            v.close()
    eng.close()
