#!/bin/bash
# round 6, third call: the fp16-operand fused window attention
OUT=gpurun_out/r06_third; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pp.py -x -q -m gpu -s -k "fused or generator" 2>&1 | grep -v "^$" | tail -12 > $OUT/pytest_pp.log; cat $OUT/pytest_pp.log
timeout 1500 python -m pytest tests/test_gpu_zbaseline.py -x -q -m gpu -s -k "config4 and (f16 or default)" 2>&1 | grep -v "^$" | tail -14 > $OUT/pytest_config4.log; cat $OUT/pytest_config4.log
for leg in 4h 4; do
    timeout 600 python scripts/bench_configs.py $leg 2>/dev/null | grep '^{' > $OUT/cfg_${leg}.json
    python - <<PY
import json
d = json.loads(open("$OUT/cfg_${leg}.json").read().splitlines()[-1])
g = d["stages"]["generator"]
print("leg $leg:", d["value"], "fps", d["s_per_batch"], "s/batch psnr", d.get("psnr_db_vs_exact_mode"), "fallbacks", d["range_guard_fallbacks"], "| generator", g["s"], "s", g["tflops"], "TF non-gemm", g["non_gemm_kernel_ms"], "ms |", g["roofline"]["kernel"], g["roofline"]["achieved"], "| hbm", d.get("hbm_gbps"))
PY
done
python - <<'PY'
import sys
sys.path.insert(0, "scripts"); sys.path.insert(0, ".")
import torch, numpy as np
from vsr_amd import engine as E
from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint
from vsr_amd.synth import make_clip, make_propainter_state_dict, make_raft_state_dict, make_rfc_state_dict
H, W, L = 360, 1920, 68
box = (H // 2, H - H // 6, W // 6, W - W // 6)
base = make_clip(10, H, W, box, seed=4)
d = torch.from_numpy(base).cuda()
frames = torch.cat([torch.roll(d, shifts=(2 * k, 3 * k), dims=(1, 2)) for k in range((L + 9) // 10)], 0)[:L].contiguous()
mask = np.zeros((H, W), np.uint8); mask[box[0]:box[1], box[2]:box[3]] = 255
plug = PropainterInpaint("cuda:0", {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0), "propainter": make_propainter_state_dict(0)}, precision="f16")
plug.profile = {}
plug.inpaint(frames, mask); plug.profile = {}
E.flow_timing_reset(); E.flow_timing(True)
plug.inpaint(frames, mask); torch.cuda.synchronize(); E.flow_timing(False)
rows = []
for k in E.flow_timing_keys():
    ms, n, fl = E.flow_timing_get(k)
    rows.append((ms, k, n, fl))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows if r[1].startswith("pp:"))
print("generator kernels total", round(tot, 1), "ms")
for ms, k, n, fl in rows[:45]:
    if k.startswith("raft:"): continue
    print(f"{k:48s} {ms:9.2f} ms {n:6d} launches {fl / ms / 1e9 if ms > 0 else 0:8.1f} TF")
PY
