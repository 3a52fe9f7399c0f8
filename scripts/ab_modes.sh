# per-op timing of the three arithmetic modes on the bench workload (timed region only)
mkdir -p gpurun_out/abm; export TMPDIR=/tmp
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0"
run() { name=$1; shift; env "$@" $B > gpurun_out/abm/$name.log 2>&1; python - gpurun_out/abm/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'])
        for k,v in d['op_breakdown'].items(): print('   ',k,v)
        for k,v in d['kernel_breakdown'].items(): print('   ',k,v)
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
run fmt VSR_PRECISION=2
run split VSR_PRECISION=split
"$@"
