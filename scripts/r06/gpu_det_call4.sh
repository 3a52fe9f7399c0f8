# batched DB post-process read-back: its test, file to file with and without
OUT=gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ocr_det.py -q -m gpu 2>&1 | tail -8 > $OUT/pytest_det2.log; tail -4 $OUT/pytest_det2.log
CLIP=gpurun_out/e2e_clip_det.y4m
run() { echo "=== $*"; (env "$@" timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 1200 --always-on --mode sttn-det 2>&1 | grep '"metric"') | tee -a $OUT/e2e_det_post_batch.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], 'fps', d['phases_s'], d['detector'].get('lanes'), d['detector'].get('frames_per_forward'), d['detector'].get('postprocess_host_fallbacks'))"; }
run A=1
run VSR_DET_POST_BATCH=0
run A=1
run VSR_DET_POST_BATCH=0
run VSR_DET_LANES=3
rm -f $CLIP
