#!/bin/bash
# fused attention, second version (2^x on pre-scaled scores, row sums only where read, halving-butterfly row maxima): tests + A/B
mkdir -p gpurun_out/r02t; export TMPDIR=/tmp
(timeout 500 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "row_maxima or exponentiates" 2>&1 | tail -30) > gpurun_out/r02t/pytest_kernels.log 2>&1
tail -4 gpurun_out/r02t/pytest_kernels.log
(timeout 900 python -m pytest tests/test_gpu_zbaseline.py tests/test_gpu_sttn.py -m gpu -q --tb=short -x -k "auto_1080p or default_windows or auto_chunk_vs_oracle or full_chunk_properties" 2>&1 | tail -30) > gpurun_out/r02t/pytest_sttn.log 2>&1
tail -4 gpurun_out/r02t/pytest_sttn.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r02t/$name.log 2>&1; python - gpurun_out/r02t/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'], {k:(v['ms'],v['tflops']) for k,v in d['op_breakdown'].items()})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
run fused_1 A=1
run unfused_1 VSR_FUSE_SOFTMAX=0
run fused_2 A=1
run unfused_2 VSR_FUSE_SOFTMAX=0
