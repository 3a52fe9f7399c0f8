#!/bin/bash
# exact-fp32 bench, interleaved A/B of the tile knobs after the round-3 kernel rebuild and of the 288 x 256 conv kernel:
#   base (VSR_F32_V8=0) | V8 | QKV 128x128 | QK^T 128x128 | P.V 128x128
OUT=gpurun_out/r04_knobs; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-split-half --e2e-chunks 0 --steps 8 --warmup 2"
run() {   # name, env...
  name=$1; shift
  env "$@" timeout 600 $B > $OUT/$name.log 2>&1
  grep '"metric"' $OUT/$name.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-14s %7.2f fps  single lane %7.2f ' % ('$name', d['value'], d['single_lane']['value']), {k:round(v['tflops'] or 0,1) for k,v in d.get('op_breakdown',{}).items() if v['tflops']})
"
}
for i in 1 2; do
  run base_$i VSR_F32_V8=0
  run v8_$i VSR_F32_V8=1
  run qkv128_$i VSR_F32_V8=0 VSR_QKV_TILE=0
  run qk128_$i VSR_F32_V8=0 VSR_QK_TILE=0
  run pv128_$i VSR_F32_V8=0 VSR_PV_TILE=0
done
