# A/B: fp32-MFMA (v3) vs split-half f16-MFMA (v4) through tests and the real bench
mkdir -p gpurun_out/ab; export TMPDIR=/tmp
(VSR_GG_VARIANT=4 timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sttn.py -m gpu -q --tb=short 2>&1 | tail -40) > gpurun_out/ab/pytest_v4.log 2>&1
tail -12 gpurun_out/ab/pytest_v4.log
B="python bench.py --steps 4 --warmup 1 --cpu-sample-frames 6"
run() { name=$1; shift; env "$@" $B > gpurun_out/ab/$name.log 2>&1; python - gpurun_out/ab/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'],'psnr',d.get('psnr_db_vs_oracle'),'roof',d['roofline']['achieved'], {k:v['tflops'] for k,v in d['op_breakdown_timed_region'].items() if v['tflops']})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
run A_v3 VSR_GG_VARIANT=3
run B_v4 VSR_GG_VARIANT=4
run C_v4_128 VSR_GG_VARIANT=4 VSR_CONV_TILE=0 VSR_QK_TILE=0
run D_v4_pv128 VSR_GG_VARIANT=4 VSR_CONV_TILE=0 VSR_QK_TILE=0 VSR_PV_TILE=0
