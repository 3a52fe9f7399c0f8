#!/bin/bash
# QKV GEMM tile A/B (K = 256: 8 chunks per tile)
mkdir -p gpurun_out/r02v; export TMPDIR=/tmp
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r02v/$name.log 2>&1; python - gpurun_out/r02v/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'], {k:(v['ms'],v['tflops']) for k,v in d['op_breakdown'].items() if k in ('attn.qkv','ffn')})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
run qkv_128x64 A=1
run qkv_128x128 VSR_QKV_TILE=0
run qkv_256x64 VSR_QKV_TILE=2
run qkv_128x64_b A=1
