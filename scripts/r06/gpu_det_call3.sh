# detector NHWC plan: HBM traffic of the 16-frame forward (PMC passes), plan-only kernel stats, file-to-file knobs (lanes / frames per forward)
OUT=gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp
cp profiles/config_traffic.json $OUT/config_traffic.json
timeout 900 python scripts/pmc_configs.py --legs 3d --out $OUT/config_traffic.json --tag r06c --workdir /tmp/pmc_work 2>&1 | grep -v amdgpu.ids | tail -5
DET_AB_ONLY=plan DET_AB_CASES=ppocr_det_graph.json:8 DET_AB_REPS=5 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/plan_trace -o r -- python scripts/r06/det_nhwc_ab.py > $OUT/plan_trace.log 2>&1
rm -f $OUT/plan_trace/r_kernel_trace.csv; grep "ms/frame" $OUT/plan_trace.log | cut -c1-200
CLIP=gpurun_out/e2e_clip_det.y4m
run() { echo "=== $*"; (env "$@" timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 1200 --always-on --mode sttn-det 2>&1 | grep '"metric"') | tee -a $OUT/e2e_det_knobs.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], 'fps', d['phases_s'], d['detector'].get('lanes'), d['detector'].get('frames_per_forward'))"; }
run A=1
run VSR_DET_LANES=1
run VSR_DET_BATCH=16
run VSR_DET_LANES=1 VSR_DET_BATCH=16
run VSR_DET_LANES=3
rm -f $CLIP
