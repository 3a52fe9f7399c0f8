#!/bin/bash
# the 256 x 256 fp16-operand kernel inside the engine: kernel tests, the precision-mode tests of the STTN engine, then the bench in
# --precision f16 with the kernel on (default) and off (VSR_F16_V7=0), interleaved
OUT=gpurun_out/r04_v7e; mkdir -p $OUT
(timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "256x256 or fp16_operands or split_format or kn_to_nk" 2>&1 | tail -15) > $OUT/pytest_kernels.log; tail -3 $OUT/pytest_kernels.log
(timeout 900 python -m pytest tests/test_gpu_sttn.py -q -x -k "fp16 or split or lanes" 2>&1 | tail -15) > $OUT/pytest_sttn.log; tail -3 $OUT/pytest_sttn.log
for prec in ${PRECS:-f16 split-format}; do
B="python bench.py --precision $prec --no-cpu-baseline --no-split-half --e2e-chunks 0 --steps 8 --warmup 2"
for i in 1 2; do
  for v in 1 0; do
    VSR_F16_V7=$v timeout 600 $B > $OUT/bench_${prec}_v7_${v}_$i.log 2>&1
    grep '"metric"' $OUT/bench_${prec}_v7_${v}_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$prec V7=$v run $i:', d['value'], 'fps', d.get('psnr_db_vs_oracle'), 'dB', {k:(round(v['ms'],1), round(v['tflops'] or 0,1)) for k,v in d.get('op_breakdown',{}).items()})
"
  done
done
done
