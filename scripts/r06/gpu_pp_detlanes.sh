# the parked event "a second live detector instance costs the propainter plugin 2-4 s of 63": propainter file to file, detector lanes 1 / 2
CLIP=gpurun_out/e2e_clip_pp.y4m
run() { echo "=== $*"; (env "$@" timeout 900 python scripts/bench_e2e.py --clip $CLIP --cycle 50 --frames 600 --always-on --mode propainter 2>&1 | grep '"metric"') | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], 'fps', d['wall_s'], 's', d['phases_s'], 'det lanes', d['detector'].get('lanes'))"; }
run VSR_DET_LANES=1
run VSR_DET_LANES=2
run VSR_DET_LANES=1
run VSR_DET_LANES=2
rm -f $CLIP
