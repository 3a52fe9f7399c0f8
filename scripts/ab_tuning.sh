# A/B of kernel variants / tuning knobs through the real bench (each run: 3 timed chunks)
mkdir -p gpurun_out/ab; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value scripts/gg_ablate.hip -o /tmp/gg_ablate && /tmp/gg_ablate > gpurun_out/ablate6.log 2>&1
grep -E "^tile|full|persistent|wg   0" gpurun_out/ablate6.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sttn.py -m gpu -q --tb=short -x 2>&1 | tail -30) > gpurun_out/ab/pytest.log 2>&1
tail -3 gpurun_out/ab/pytest.log
run() { name=$1; shift; env "$@" $B > gpurun_out/ab/$name.log 2>&1; python - gpurun_out/ab/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'],'roof',d['roofline']['achieved'], {k:(v['ms'],v['tflops']) for k,v in d['op_breakdown_timed_region'].items()})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
run A_v3 VSR_GG_VARIANT=3 VSR_QK_TILE=3
run B_v1 VSR_GG_VARIANT=1 VSR_QK_TILE=3
run C_v3_qk128 VSR_GG_VARIANT=3 VSR_QK_TILE=0
run D_v3_nosplit VSR_GG_VARIANT=3 VSR_QK_TILE=3 VSR_PV_SPLIT_CHUNKS=0
run E_v3_pv128 VSR_GG_VARIANT=3 VSR_QK_TILE=3 VSR_PV_TILE=0
