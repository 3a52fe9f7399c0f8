OUT=gpurun_out/r06c; mkdir -p $OUT
L=$OUT/f16_companions_ab.log; : > $L
run() { echo "=== $*" | tee -a $L; env "$@" timeout 600 python scripts/bench_configs.py 5 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read().splitlines()[-1]); r = d['roofline']
print(d['value'], 'fps', d['ms_per_chunk'], 'ms; model TF', d['model_tflops'], d['model_frac_of_peak'], '; kernel', r['achieved'], r['frac'], 'launches', r.get('launches'), 'share', r.get('share_of_gemm_time'), 'every gemm', r.get('every_gemm_launch'), 'fallbacks', d.get('fp32_fallback_chunks'))" | tee -a $L; }
run VSR_F16_V7_COMPANIONS=0
run VSR_F16_V7_COMPANIONS=1
run VSR_F16_V7_COMPANIONS=0
run VSR_F16_V7_COMPANIONS=1
B="python bench.py --no-cpu-baseline --no-configs --no-split-half --e2e-chunks 0 --no-full-work --precision f16"
for c in 0 1; do echo "=== 1080p f16 bench, companions $c" | tee -a $L; VSR_F16_V7_COMPANIONS=$c $B 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().splitlines()[-1])
print(d['value'], 'fps', d['ms_per_step'], 'ms; psnr', d.get('psnr_db_vs_oracle'), {k: (v['ms'], v['tflops']) for k, v in d['op_breakdown'].items()})" | tee -a $L; done
timeout 1500 python -m pytest tests/test_gpu_sttn.py tests/test_gpu_zbaseline.py tests/test_gpu_kernels.py -q -m gpu -k "f16 or fp16 or 256x256 or precision" --tb=short 2>&1 | tail -5 | tee -a $L
