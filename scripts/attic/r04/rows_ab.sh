#!/bin/bash
# the decoder on the rows the mask needs (vsr_sttn_auto_chunk_rows; VSR_DECODE_ROWS, default 1): STTN engine + golden wrapper + IO tests,
# then the default bench with the rows on / off, interleaved
OUT=gpurun_out/r04_rows; mkdir -p $OUT
(timeout 900 python -m pytest tests/test_gpu_sttn.py tests/test_gpu_golden_wrappers.py tests/test_gpu_io.py tests/test_gpu_kernels.py -q -x 2>&1 | tail -3) > $OUT/pytest.log; tail -1 $OUT/pytest.log
B="python bench.py --no-cpu-baseline --no-split-half --e2e-chunks 0 --steps 8 --warmup 2"
for i in 1 2; do
  for v in 1 0; do
    VSR_DECODE_ROWS=$v timeout 600 $B > $OUT/bench_rows${v}_$i.log 2>&1
    grep '"metric"' $OUT/bench_rows${v}_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('ROWS=$v run $i:', d['value'], 'fps', d['ms_per_step'], 'ms; single lane', d['single_lane']['value'], '; GFLOP/frame', d['gflop_per_frame'], '|', d.get('gflop_per_frame_reference'), '; model TF', d['model_tflops'], '; roofline', r['achieved'], r['frac'], '; every GEMM launch', r['every_gemm_launch']['achieved'])
print('   ', {k:(round(v['ms'],1), v['launches'], round(v['tflops'] or 0,1)) for k,v in d.get('op_breakdown',{}).items()})"
  done
done
python bench.py --no-cpu-parallel --e2e-chunks 2 --steps 4 --warmup 1 2>/dev/null | grep '"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('default line with the CPU oracle leg:', d['value'], 'fps; psnr', d['psnr_db_vs_oracle'], d['psnr_note'])"
