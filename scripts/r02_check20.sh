#!/bin/bash
# 64 -> 3 output conv over 2x4 output blocks: engine parity (auto + det), decode kernel test, bench A/B
mkdir -p gpurun_out/r02u; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_sttn.py tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "default_windows or auto_chunk_vs_oracle or det_inpaint_vs_oracle or det_plugin_call or decode or split_half_mode" 2>&1 | tail -30) > gpurun_out/r02u/pytest.log 2>&1
tail -4 gpurun_out/r02u/pytest.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r02u/$name.log 2>&1; python - gpurun_out/r02u/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'], {k:(v['ms'],v['tflops']) for k,v in d['op_breakdown'].items() if k in ('dec','ffn')}, {k:(v['ms'],v['tflops']) for k,v in d['kernel_breakdown'].items() if '256, 32' in k})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
run blocked_1 A=1
run plain_1 VSR_OUT_CONV_BLOCKED=0
run blocked_2 A=1
run plain_2 VSR_OUT_CONV_BLOCKED=0
