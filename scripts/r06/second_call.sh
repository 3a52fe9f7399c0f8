#!/bin/bash
# round 6, second call: the fused window attention (VSR_PP_FLASH) -- parity tests, then config 4 / 4h with and without it
OUT=gpurun_out/r06_second; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pp.py tests/test_gpu_golden_wrappers.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -40 > $OUT/pytest_pp.log; tail -25 $OUT/pytest_pp.log
for f in 1 0; do
  for leg in 4 4h; do
    VSR_PP_FLASH=$f timeout 600 python scripts/bench_configs.py $leg 2>/dev/null | grep '^{' > $OUT/cfg_${leg}_flash$f.json
    python - <<PY
import json
d = json.loads(open("$OUT/cfg_${leg}_flash$f.json").read().splitlines()[-1])
g = d["stages"]["generator"]
print("flash=$f leg $leg:", d["value"], "fps", d["s_per_batch"], "s/batch psnr", d.get("psnr_db_vs_exact_mode"), "| generator", g["s"], "s", g["tflops"], "TF non-gemm", g["non_gemm_kernel_ms"], "ms |", g["roofline"]["kernel"], g["roofline"]["achieved"])
PY
  done
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
plug = PropainterInpaint("cuda:0", {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0), "propainter": make_propainter_state_dict(0)}, precision="f32")
plug.profile = {}
plug.inpaint(frames, mask); plug.profile = {}
E.flow_timing_reset(); E.flow_timing(True)
plug.inpaint(frames, mask); torch.cuda.synchronize(); E.flow_timing(False)
rows = []
for k in E.flow_timing_keys():
    ms, n, fl = E.flow_timing_get(k)
    rows.append((ms, k, n, fl))
rows.sort(reverse=True)
for ms, k, n, fl in rows[:40]:
    print(f"{k:48s} {ms:9.2f} ms {n:6d} launches {fl / ms / 1e9 if ms > 0 else 0:8.1f} TF")
PY
