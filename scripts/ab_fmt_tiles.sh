mkdir -p gpurun_out/abt; export TMPDIR=/tmp
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0"
run() { name=$1; shift; env "$@" $B > gpurun_out/abt/$name.log 2>&1; python - gpurun_out/abt/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'], {k:(v['ms'],v['tflops']) for k,v in d['op_breakdown_timed_region'].items()})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
for i in 1 2; do
run base$i VSR_PRECISION=2
run conv128_$i VSR_PRECISION=2 VSR_CONV_TILE=0
run qk128_$i VSR_PRECISION=2 VSR_QK_TILE=0
run pv128_$i VSR_PRECISION=2 VSR_PV_TILE=0
run all128_$i VSR_PRECISION=2 VSR_CONV_TILE=0 VSR_QK_TILE=0 VSR_PV_TILE=0
done
