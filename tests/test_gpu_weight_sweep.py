"""The HIP engines against the CPU oracle on the NON-benign weight draws of vsr_amd/synth.py in every arithmetic mode.

Rounds 1-5 took every parity number on one Gaussian, variance-preserving draw per network.  Here (the oracle is pinned to the reference
modules on exactly these draws: tests/test_weight_sweep.py, tests/golden/weight_sweep.npz):
  "peaked"    query / key gain x4 -- attention rows close to one-hot;
  "heavy"     Student-t weights with activations a few times below the fp16 limit (STTN: 19 000-32 000 through encoder and blocks;
              ProPainter: the FFN hidden tensor at 24 000; flow completion: 18 000);
  "undamped"  RAFT with a full-gain flow head: flows of 100-400 pixels.
Bars: exact fp32 modes -- two grey levels (STTN) / the network's own fp32 conditioning (generator: the reference in fp32 is 5.7e-2 of the
tanh range away from its own float64 run on the peaked draw, so the HIP path is measured against the FLOAT64 oracle and allowed twice the
fp32 oracle's own distance to it); fp16-operand and split modes -- >= 50 dB, OR a counted fallback whose result is the exact mode's
bit for bit.  Two guards can produce that fallback: the kernels' RANGE guard (a value left the fp16 range) and, since this sweep found
modes that stay in range and still land at 47 / 28 / 17 dB, the engines' ACCURACY guard (engine.py AccuracyGuard: the first unit of
work of a guarded mode is also run exactly; below 50 dB the caller gets the exact result and the engine is demoted).
profiles/r06_weight_sweep_before.log is this file's output before the accuracy guard existed.
"""
import numpy as np
import pytest
import torch

from vsr_amd import synth
from oracle.sttn_auto import STTNInpaintOracle, calculate_psnr

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------------ STTN
@pytest.fixture(scope="module")
def sttn_refs():
    """oracle outputs per profile on the smoke-sized case (6 frames of 120x640, stride 2 / refs every 3: 3 windows)"""
    frames = np.random.default_rng(11).integers(0, 256, size=(6, 120, 640, 3), dtype=np.uint8)
    out = {}
    for prof in ("peaked", "heavy"):
        sd = synth.make_state_dict(0, "auto", prof)
        out[prof] = (sd, np.stack([r.astype(np.float32) for r in STTNInpaintOracle(sd, "auto", 2, 3).inpaint(list(frames))]))
    return frames, out


@pytest.mark.parametrize("precision", ["f32", "split", "split-format", "f16"])
@pytest.mark.parametrize("profile", ["peaked", "heavy"])
def test_sttn_inpaint_on_profile(built_lib, gpu_device, sttn_refs, profile, precision):
    from vsr_amd.engine import SttnEngine

    frames, refs = sttn_refs
    sd, ref = refs[profile]
    d = torch.from_numpy(frames).to(gpu_device)
    exact = SttnEngine(sd, "auto", device=0, neighbor_stride=2, ref_length=3)
    want, _ = exact.inpaint(d)
    torch.cuda.synchronize()
    want = want.cpu().numpy()
    exact.close()
    if precision == "f32":
        got, fb = want, 0
    else:
        eng = SttnEngine(sd, "auto", device=0, neighbor_stride=2, ref_length=3, precision=precision)
        g, _ = eng.inpaint(d)
        torch.cuda.synchronize()
        got, fb = g.cpu().numpy(), eng.fallbacks()
        eng.close()
    psnr, dmax = calculate_psnr(got, ref), float(np.abs(got - ref).max())
    print(f"sttn [{profile}] {precision}: PSNR vs oracle {psnr:.2f} dB, max|d| {dmax:.2f} levels, range-guard fallbacks {fb}")
    assert np.isfinite(got).all()
    if precision == "f32":
        assert psnr >= 50.0 and dmax <= 2.0
    elif fb > 0:
        assert np.array_equal(got, want), "a guarded call that fell back must reproduce the exact mode"
    else:
        assert psnr >= 50.0


# ------------------------------------------------------------------------------------------------ ProPainter generator
def _pp_case():
    from oracle.make_golden import propainter_inputs

    return (7, 5, 64, 96) + tuple(propainter_inputs(41, 7, 5, 64, 96))


@pytest.fixture(scope="module")
def pp_refs():
    from test_weight_sweep import pp_oracle_out

    return {prof: (pp_oracle_out(prof, torch.float32), pp_oracle_out(prof, torch.float64)) for prof in ("peaked", "heavy")}


@pytest.mark.parametrize("precision", ["f32", "split", "f16"])
@pytest.mark.parametrize("profile", ["peaked", "heavy"])
def test_generator_on_profile(built_lib, gpu_device, pp_refs, profile, precision):
    from oracle.propainter import ProPainterOracle
    from vsr_amd.engine import PpEngine

    t, lt, h, w, frames, masks, ff, fb = _pp_case()
    sd = synth.make_propainter_state_dict(0, profile)
    o = ProPainterOracle(sd)
    fr, mk = torch.from_numpy(frames), torch.from_numpy(masks)
    masked = fr * (1 - mk)
    prop, upd = o.img_propagation(masked[:lt], torch.from_numpy(ff), torch.from_numpy(fb), mk[:lt].clone())
    sel = torch.cat([fr[:lt] * (1 - mk[:lt]) + prop * mk[:lt], masked[lt:]])
    sel_upd = torch.cat([upd, mk[lt:]])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
    args = (dev(sel.numpy()), dev(ff), dev(fb), dev(masks[:, 0].astype(np.uint8)), dev(sel_upd[:, 0].numpy().astype(np.uint8)), lt)

    def run(mode):
        e = PpEngine(device=0, state_dict=sd)
        if mode != "f32":
            e.set_precision(mode)
        y = e.forward(*args)
        torch.cuda.synchronize()
        r = y.cpu().numpy().astype(np.float64), e.fallbacks()
        e.close()
        return r

    want, _ = run("f32")
    got, fbk = (want, 0) if precision == "f32" else run(precision)
    o32, o64 = pp_refs[profile]
    # the fp32 oracle's own distance to float64 is the network's conditioning on this draw: max and rms
    own, own_rms = float(np.abs(o32 - o64).max()), float(np.sqrt(np.mean((o32 - o64) ** 2)))
    e64, e32 = float(np.abs(got - o64).max()), float(np.abs(got - o32).max())
    rms = float(np.sqrt(np.mean((got - o64) ** 2)))
    psnr = 20 * np.log10(2.0 / max(rms, 1e-12))                # tanh output: a range of 2
    print(f"generator [{profile}] {precision}: vs float64 oracle max {e64:.2e} rms {rms:.2e} ({psnr:.1f} dB); vs fp32 oracle max {e32:.2e}; "
          f"fp32 oracle vs float64 max {own:.2e} rms {own_rms:.2e}; fallbacks {fbk}")
    assert np.isfinite(got).all()
    # An ill-conditioned draw ("peaked": near one-hot rows flip between near-tied keys) turns ANY difference in fp32 summation order into
    # the same kind of error the fp32 oracle has against float64; a single pixel's maximum is a draw from that distribution, so the bar
    # is on the rms: no more than 6x the fp32 oracle's own (measured 3.9x; a well-conditioned draw: the benign bar of 2e-3 on the maximum).
    if precision == "f32":
        assert e64 <= 2e-3 or rms <= 6 * own_rms
    elif fbk > 0:
        assert np.array_equal(got, want), "a guarded call that fell back (range guard or accuracy guard) must hand out the exact mode's result"
    else:
        assert psnr >= 50.0 or rms <= 6 * own_rms


# ------------------------------------------------------------------------------------------------ flow completion
@pytest.mark.parametrize("precision", ["f32", "split", "f16"])
def test_flow_completion_heavy(built_lib, gpu_device, precision):
    from oracle.make_golden import rfc_inputs
    from oracle.rfc import RfcOracle
    from vsr_amd.engine import RfcEngine

    sd = synth.make_rfc_state_dict(0, "heavy")
    ff, fb, masks = rfc_inputs(21, 5, 64, 96)
    cf, cb, _, _ = RfcOracle(sd).complete_bi(torch.from_numpy(ff), torch.from_numpy(fb), torch.from_numpy(masks))
    m8 = torch.from_numpy((masks[:, 0] > 0).astype(np.uint8)).to(gpu_device)

    def run(mode):
        e = RfcEngine(sd, device=0)
        if mode != "f32":
            e.set_precision(mode)
        of, ob = e.complete(torch.from_numpy(ff).to(gpu_device), torch.from_numpy(fb).to(gpu_device), m8)
        torch.cuda.synchronize()
        r = of.cpu().numpy(), ob.cpu().numpy(), e.fallbacks()
        e.close()
        return r

    wf, wb, _ = run("f32")
    of, ob, fbk = (wf, wb, 0) if precision == "f32" else run(precision)
    err = max(float(np.abs(of - cf.numpy()).max()), float(np.abs(ob - cb.numpy()).max()))
    print(f"flow completion [heavy] {precision}: max abs err {err:.3e} px (flow range {float(np.abs(cf.numpy()).max()):.1f}), fallbacks {fbk}")
    assert np.isfinite(of).all()
    if precision == "f32":
        assert err <= 1e-3
    elif fbk > 0:
        assert np.array_equal(of, wf) and np.array_equal(ob, wb)
    else:
        assert err <= (2e-3 if precision == "split" else 5e-2)          # pixels; the f16 mode on the benign draw: 1e-2 (tests/test_gpu_flow_split.py)


# ------------------------------------------------------------------------------------------------ RAFT
@pytest.mark.parametrize("precision", ["f32", "split"])
@pytest.mark.parametrize("profile", ["undamped", "heavy"])
def test_raft_on_profile(built_lib, gpu_device, profile, precision):
    from oracle.raft import RaftOracle
    from vsr_amd.engine import RaftEngine

    sd = synth.make_raft_state_dict(0, profile)
    frames = synth.make_flow_frames(3, 128, 192, seed=1)
    x = torch.from_numpy(frames).permute(0, 3, 1, 2).float().div(255) * 2 - 1
    e = RaftEngine(sd, device=0)
    if precision != "f32":
        e.set_precision(precision)
    res = {}
    for iters in (2, 20):
        fwd, bwd = e.flows(torch.from_numpy(frames).to(gpu_device), iters=iters)
        torch.cuda.synchronize()
        of, ob = RaftOracle(sd).flows_bi(x, iters)
        rng = float(of.abs().max())
        err = max(float((fwd.cpu() - of).abs().max()), float((bwd.cpu() - ob).abs().max()))
        res[iters] = (err, rng)
        print(f"raft [{profile}] {precision} iters={iters}: max abs err {err:.3e} px, flow range {rng:.1f} px, fallbacks {e.fallbacks()}")
        assert torch.isfinite(fwd).all() and torch.isfinite(bwd).all()
    e.close()
    # the bars the ORACLE is held to against the reference on these draws (tests/test_weight_sweep.py: 2e-4 px after 2 iterations, 2e-2 after
    # 20 at flows of ~50 px), scaled with the flow range: twenty recurrent lookups at flows of hundreds of pixels amplify the last bit
    for iters, tol in ((2, 4e-4), (20, 2e-2)):
        err, rng = res[iters]
        assert err <= tol * max(1.0, rng / 50.0) * (1 if precision == "f32" else 4), (iters, err, rng)
