"""Stage-by-stage comparison of the GPU workspace with the CPU replay of the same RAFT plan (debug aid).
usage: python scripts/raft_debug.py t H W iters"""
import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import vsr_amd
from vsr_amd import _lib
from vsr_amd.engine import RaftEngine
from vsr_amd.synth import make_raft_state_dict, make_flow_frames
import _replay_raft as rr
t, H, W, iters = [int(a) for a in sys.argv[1:5]]
sd = make_raft_state_dict(0)
e = RaftEngine(sd, device=0)
frames = make_flow_frames(t, H, W, seed=5)
fwd, bwd = e.flows(torch.from_numpy(frames).cuda(), iters=iters)
torch.cuda.synchronize()
v = rr.raft_plan_view(_lib, e, t, H, W, iters)
rf, rb, bufs = rr.replay_raft(v, e.packed_weights(), frames)
names = "WEIGHTS IN_U8 IM2COL S1A S1B S1C S2A S2B S2C S3A S3B S3C STATS FMAP CMAP PYR COORDS FLOW CORRF C1 CORFLO FLOWCOL F1 HXR ZR Q FH1 DELTA MASKH MASK OUT".split()
for b in range(2, len(names)):
    n = v.buf_elems[b]
    if n == 0: continue
    g = e.read_buffer(b, n)
    r = bufs[b][:n]
    d = np.abs(g - r)
    bad = ~np.isfinite(g)
    print(f"{names[b]:8s} n={n:10d} max|ref|={np.abs(r).max():10.4f} maxerr={np.nanmax(d):.3e} at {int(np.nanargmax(d))} nonfinite={int(bad.sum())}")
print('fwd err', np.abs(fwd.cpu().numpy() - rf).max(), 'bwd err', np.abs(bwd.cpu().numpy() - rb).max())
