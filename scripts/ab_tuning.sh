# interleaved A/B inside split-half mode (tile shapes)
mkdir -p gpurun_out/ab; export TMPDIR=/tmp
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-split-half"
run() { name=$1; shift; env "$@" $B > gpurun_out/ab/$name.log 2>&1; python - gpurun_out/ab/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'], {k:v['tflops'] for k,v in d['op_breakdown'].items() if v['tflops']})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
for i in 1 2 3; do
run A${i}_all128_pv64 VSR_PRECISION=split VSR_PV_TILE=3
run B${i}_conv64_pv64 VSR_PRECISION=split VSR_PV_TILE=3 VSR_CONV_TILE=3
run C${i}_all64 VSR_PRECISION=split VSR_PV_TILE=3 VSR_CONV_TILE=3 VSR_QK_TILE=3
done
run F_f32 VSR_PRECISION=f32
