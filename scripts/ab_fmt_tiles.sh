mkdir -p gpurun_out/abt; export TMPDIR=/tmp
(timeout 600 python -m pytest tests -m gpu -q --tb=short -x -k "split_format or fp16 or split_half" 2>&1 | tail -4)
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0"
run() { name=$1; shift; env "$@" $B > gpurun_out/abt/$name.log 2>&1; python - gpurun_out/abt/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'], {k:(v['ms'],v['tflops']) for k,v in d['op_breakdown'].items()})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
for i in 1 2; do
run s2_$i VSR_PRECISION=2
run s3_$i VSR_PRECISION=2 VSR_V5_STAGES=3
run f16s2_$i VSR_PRECISION=3
run f16s3_$i VSR_PRECISION=3 VSR_V5_STAGES=3
done
