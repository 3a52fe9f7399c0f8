#!/bin/bash
# round 6, twentieth call: the generator's token GEMMs on 128 x 128 tiles (VSR_PP_TR_TILE=128x128) vs 128 x 64, both arithmetics
OUT=gpurun_out/r06_twentieth; mkdir -p $OUT; export TMPDIR=/tmp
ops() {
python - "$2" "$3" <<'PY' > $OUT/$1.log 2>&1
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "scripts")
import bench_configs as bc
from vsr_amd import engine as E
r = bc.run_propainter(sys.argv[1], sys.argv[2])
print({k: r[k] for k in ("value", "s_per_batch", "psnr_db_vs_exact_mode")})
for k in E.flow_timing_keys():
    if ":tr." in k and ":gg:" in k: print(k, [round(x, 2) for x in E.flow_timing_get(k)[:2]], round(E.flow_timing_get(k)[2] / E.flow_timing_get(k)[0] / 1e9, 1), "TF")
PY
grep -v amdgpu $OUT/$1.log
}
echo "## f16, 128x64"; ops f16_64 4h f16
echo "## f16, 128x128"; VSR_PP_TR_TILE=128x128 ops f16_128 4h f16
echo "## f32, 128x128"; VSR_PP_TR_TILE=128x128 ops f32_128 4 f32
