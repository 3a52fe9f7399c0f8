#!/bin/bash
# the 288 x 256 exact-fp32 kernel inside the engine: kernel tests, the STTN engine tests, then the default bench with the kernel on
# (default) and off (VSR_F32_V8=0), interleaved
OUT=gpurun_out/r04_v8e; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "288x256 or 128x64" 2>&1 | tail -15) > $OUT/pytest_kernels.log; tail -3 $OUT/pytest_kernels.log
(timeout 900 python -m pytest tests/test_gpu_sttn.py -q -x 2>&1 | tail -15) > $OUT/pytest_sttn.log; tail -3 $OUT/pytest_sttn.log
B="python bench.py --no-cpu-baseline --no-split-half --e2e-chunks 0 --steps 8 --warmup 2"
for i in 1 2; do
  for v in 1 0; do
    VSR_F32_V8=$v timeout 600 $B > $OUT/bench_v8_${v}_$i.log 2>&1
    grep '"metric"' $OUT/bench_v8_${v}_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('V8=$v run $i:', d['value'], 'fps; single lane', d['single_lane']['value'], '; roofline', r['kernel'].split(' (')[0], r['achieved'], 'TF', r['frac'], 'launches', r['launches'], r.get('every_gemm_launch'))
print('   ', {k:(round(v['ms'],1), round(v['tflops'] or 0,1)) for k,v in d.get('op_breakdown',{}).items()})
print('   ', {k:(round(v['ms'],1), v['launches'], v['tflops']) for k,v in d.get('kernel_breakdown',{}).items()})
"
  done
done
