#!/bin/bash
# the last transformer block of a window on its neighbour frames only (VSR_TRIM_LAST_BLOCK, default 1): STTN engine + golden wrapper
# tests, the BASELINE-size chunk against the oracle, then the default bench with the trim on / off, interleaved
OUT=gpurun_out/r04_trim; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_sttn.py tests/test_gpu_golden_wrappers.py -q -x 2>&1 | tail -3) > $OUT/pytest.log; tail -1 $OUT/pytest.log
(timeout 900 python -m pytest tests/test_gpu_zbaseline.py -q -x -k "sttn" 2>&1 | tail -3) > $OUT/pytest_baseline.log; tail -1 $OUT/pytest_baseline.log
B="python bench.py --no-cpu-baseline --no-split-half --e2e-chunks 0 --steps 8 --warmup 2"
for i in 1 2; do
  for v in 1 0; do
    VSR_TRIM_LAST_BLOCK=$v timeout 600 $B > $OUT/bench_trim${v}_$i.log 2>&1
    grep '"metric"' $OUT/bench_trim${v}_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('TRIM=$v run $i:', d['value'], 'fps', d['ms_per_step'], 'ms; single lane', d['single_lane']['value'], '; GFLOP/frame', d['gflop_per_frame'], d.get('gflop_per_frame_reference'), '; model TF', d['model_tflops'], '; roofline', r['achieved'], r['frac'], r['every_gemm_launch']['achieved'])"
  done
done
