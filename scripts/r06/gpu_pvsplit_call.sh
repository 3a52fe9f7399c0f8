B="python bench.py --no-cpu-baseline --no-configs --no-split-half --e2e-chunks 0 --no-full-work"
run() { echo "=== $*"; env "$@" $B 2>/dev/null | grep '"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read().splitlines()[-1])
print(d['value'], 'fps', d['ms_per_step'], 'ms; single lane', d['single_lane']['value'], '; pv', d['op_breakdown']['attn.pv'], d['op_breakdown']['attn.pv.reduce'])"; }
run A=1
run VSR_PV_SPLIT_CHUNKS=38
run VSR_PV_SPLIT_CHUNKS=30
run VSR_PV_SPLIT_CHUNKS=75
run A=1
run VSR_PV_SPLIT_CHUNKS=38
