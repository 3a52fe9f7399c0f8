# A/B of kernel variants / tuning knobs through the real bench, alternating so that drift shows
mkdir -p gpurun_out/ab; export TMPDIR=/tmp
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > gpurun_out/ab/$name.log 2>&1; python - gpurun_out/ab/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'],'roof',d['roofline']['achieved'], {k:v['tflops'] for k,v in d['op_breakdown_timed_region'].items() if v['tflops']})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
run A1_xcdq VSR_GG_QUEUES=8
run B1_globalq VSR_GG_QUEUES=1
run A2_xcdq VSR_GG_QUEUES=8
run B2_globalq VSR_GG_QUEUES=1
run C_v1 VSR_GG_VARIANT=1
run D_pv_v3 VSR_PV_VARIANT=3
