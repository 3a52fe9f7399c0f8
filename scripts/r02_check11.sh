#!/bin/bash
# f16 mode: tile sweep (conv / QK / PV tiles, v6 stage count) on the 1080p bench + kernel tests of the 256x128 tile
mkdir -p gpurun_out/r02k; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "fp16_operands or every_variant" 2>&1 | tail -4) > gpurun_out/r02k/pytest.log 2>&1
tail -2 gpurun_out/r02k/pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0 --precision f16"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r02k/$name.log 2>&1; python - gpurun_out/r02k/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'], {k:(v['ms'],v['tflops']) for k,v in d['op_breakdown'].items()})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
run base
run conv128 VSR_CONV_TILE=0
run conv256 VSR_CONV_TILE=4
run conv256_st2 VSR_CONV_TILE=4 VSR_V6_STAGES=2
run qk128 VSR_QK_TILE=0
run qk256 VSR_QK_TILE=4
run pv128 VSR_PV_TILE=0
run all256 VSR_CONV_TILE=4 VSR_QK_TILE=4
run all256_pv128 VSR_CONV_TILE=4 VSR_QK_TILE=4 VSR_PV_TILE=0
