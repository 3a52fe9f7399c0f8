#!/bin/bash
mkdir -p gpurun_out/r02m; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x 2>&1 | tail -4) > gpurun_out/r02m/pytest.log 2>&1
tail -2 gpurun_out/r02m/pytest.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r02m/$name.log 2>&1; python - gpurun_out/r02m/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'], {k:(v['ms'],v['tflops']) for k,v in d['op_breakdown'].items()})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
run f32
run f16 VSR_PRECISION=3
run f16_conv128st2 VSR_PRECISION=3 VSR_CONV_TILE=0 VSR_V6_STAGES=2
run f16_all128st2 VSR_PRECISION=3 VSR_CONV_TILE=0 VSR_QK_TILE=0 VSR_V6_STAGES=2
run split VSR_PRECISION=2
run f32_b
