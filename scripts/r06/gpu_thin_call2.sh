OUT=gpurun_out/r06c; mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-configs --no-split-half --e2e-chunks 0 --no-full-work"
L=$OUT/sttn_thin_ab.log; : > $L
run() { echo "=== $*" | tee -a $L; env "$@" $B 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().splitlines()[-1])
print(d['value'], 'fps', d['ms_per_step'], 'ms; single lane', d['single_lane']['value'], '; roofline', d['roofline']['achieved'], d['roofline']['frac'], 'launches', d['roofline'].get('launches'))
print('   ', {k: (v['ms'], v['tflops']) for k, v in d['op_breakdown'].items()})
print('   ', {k: (v['ms'], v['launches'], v['tflops']) for k, v in d['kernel_breakdown'].items()})" | tee -a $L; }
run VSR_STTN_THIN=0
run VSR_STTN_THIN=1
run VSR_STTN_THIN=1 VSR_STTN_THIN_TILES=0
run VSR_STTN_THIN=1 VSR_STTN_THIN_TILES=1536 VSR_STTN_THIN_K=2600
run VSR_STTN_THIN=0
run VSR_STTN_THIN=1
VSR_STTN_THIN=1 timeout 900 python -m pytest tests/test_gpu_sttn.py -q -m gpu -x 2>&1 | tail -3 | tee -a $L
timeout 900 python scripts/r06/flow_thin_ab.py 2>&1 | grep -v amdgpu.ids | head -3 | tee $OUT/flow_thin_default.log
VSR_FLOW_THIN=0 timeout 900 python scripts/r06/flow_thin_ab.py 2>&1 | grep -v amdgpu.ids | head -3 | tee -a $OUT/flow_thin_default.log
