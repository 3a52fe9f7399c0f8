"""CPU checks of the bench plumbing added in round 6: the per-leg watchdog of scripts/bench_multi.py (a stuck collective must cost one
leg, not the run: rank 0 still prints the line), the kernel-symbol parser and the traffic attachment of scripts/pmc_configs.py /
bench_configs.attach_traffic (roofline.traffic and hbm_gbps for every BASELINE configuration)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_watchdog_prints_the_line_and_leaves():
    code = textwrap.dedent(f"""
        import sys, time, json
        sys.path.insert(0, {os.path.join(ROOT, 'scripts')!r})
        import bench_multi
        results = {{}}
        head = {{"metric": "m", "value": 1.0}}
        def on_timeout(leg, phase):
            results[leg] = {{"error": "watchdog: " + phase}}
            line = dict(head); line["configs_multi"] = results
            print(json.dumps(line), flush=True)
        wd = bench_multi.Watchdog(0.5, on_timeout)
        wd.phase("5", "setup"); wd.phase("5", "timed exchange")      # re-arming cancels the first timer
        time.sleep(30)                                               # the 'stuck collective'
        print("NOT REACHED")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "NOT REACHED" not in r.stdout
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and d["configs_multi"]["5"]["error"] == "watchdog: timed exchange"


def test_watchdog_cancel():
    import time

    import bench_multi

    fired = []
    wd = bench_multi.Watchdog(0.2, lambda leg, phase: fired.append((leg, phase)))
    wd.phase("4", "setup")
    wd.cancel()
    time.sleep(0.5)
    assert fired == []


def test_kernel_symbol_parser():
    import pmc_configs

    assert pmc_configs.symbol("void gather_gemm_f32_v3<128, 64, 2, 2, 0>(GGProblem const*, int, int, unsigned int*, int)") == \
        "gather_gemm_f32_v3<128, 64, 2, 2, 0>"
    assert pmc_configs.symbol("k_softmax_rows(SMProblem const*, int)") == "k_softmax_rows"
    assert pmc_configs.symbol("void at::native::vectorized_elementwise_kernel<4, at::native::nextafter_kernel(at::TensorIteratorBase&)::{lambda()#1}>(int)") \
        .startswith("at::native::vectorized_elementwise_kernel<4, at::native::nextafter_kernel(")
    assert pmc_configs.symbol("__amd_rocclr_fillBufferAligned") == "__amd_rocclr_fillBufferAligned"


def test_attach_traffic(tmp_path, monkeypatch):
    import bench_configs

    f = tmp_path / "t.json"
    f.write_text(json.dumps({"legs": {"4h": {"hbm_bytes_per_unit": 2_000_000_000_000, "note": "n",
                                            "kernels": {"gather_gemm_f32_v3<128, 64, 2, 2, 0>": {"launches": 235, "hbm_bytes_per_launch": 4_000_000_000}}}}}))
    monkeypatch.setattr(bench_configs, "TRAFFIC_FILE", str(f))
    out = {"roofline": {"kernel": "gather_gemm_f32_v3<128, 64, 2, 2, 0>", "avg_launch_ms": 5.0, "traffic": None},
           "stages": {"raft": {"roofline": {"kernel": "gather_gemm_f32_v3<128, 64, 2, 2, 0>", "avg_launch_ms": 5.0, "traffic": None}},
                      "generator": {"roofline": {"kernel": "something else", "traffic": None}}, "other": {"s": 0.1}}}
    bench_configs.attach_traffic("4h", out, 0.5)
    assert out["roofline"]["traffic"] == 4_000_000_000 and out["roofline"]["hbm_gbps_in_kernel"] == 800.0
    assert out["stages"]["raft"]["roofline"]["traffic"] == 4_000_000_000 and out["stages"]["generator"]["roofline"]["traffic"] is None
    assert out["hbm_gb_per_unit"] == 2000.0 and out["hbm_gbps"] == 1000.0 and out["hbm_frac_of_peak"] == 0.125
    untouched = {"roofline": {"kernel": "x", "traffic": None}}
    assert bench_configs.attach_traffic("nope", untouched, 1.0) == {"roofline": {"kernel": "x", "traffic": None}}
