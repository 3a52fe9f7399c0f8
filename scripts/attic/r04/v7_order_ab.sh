#!/bin/bash
# tile order of the 256 x 256 kernel of the split-format modes: XCD-aware static slots (VSR_V7_ORDER=1, default) against the plain
# order (0): the probe (conv / QK^T shapes), the kernel tests, then bench.py --precision {f16, split-format}, interleaved
OUT=gpurun_out/r04_v7o; mkdir -p $OUT
for o in 1 0; do
  timeout 300 video-subtitle-remover_amd/build/v7_probe $o > $OUT/probe_order$o.log 2>&1
  grep -E "^conv|^qk|v7 256x256|RESULT|timeline|tiles:" $OUT/probe_order$o.log | sed "s/^/order $o | /"
done
(timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "256x256" 2>&1 | tail -3) > $OUT/pytest_kernels.log; tail -1 $OUT/pytest_kernels.log
for prec in f16 split-format; do
B="python bench.py --precision $prec --no-cpu-baseline --no-split-half --e2e-chunks 0 --steps 8 --warmup 2"
for i in 1 2; do
  for v in 1 0; do
    VSR_V7_ORDER=$v timeout 600 $B > $OUT/bench_${prec}_order${v}_$i.log 2>&1
    grep '"metric"' $OUT/bench_${prec}_order${v}_$i.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$prec ORDER=$v run $i:', d['value'], 'fps', {k:(round(v['ms'],1), round(v['tflops'] or 0,1)) for k,v in d.get('op_breakdown',{}).items()})
"
  done
done
done
