#!/bin/bash
# Round 5, sixth GPU call: RAFT in runs of <= 24 consecutive pairs (workspace 125 -> 44 GB), config 4 again, and the file-to-file runs of
# configs 3 and 4 with the subtitle on EVERY frame (the batch sizes BASELINE.json's 1200-frame clips give: 68 / 70-frame propainter
# batches -- rounds 3-4 measured a clip whose 100-frame intervals never made a batch longer than 50).
OUT=gpurun_out/r05_sixth; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_zbaseline.py -q -s -k "config4 and default" 2>&1 | grep -E "PSNR|passed|failed|Error|error" | tail -6) > $OUT/pytest_config4.log; cat $OUT/pytest_config4.log
(timeout 600 python -m pytest tests/test_gpu_golden_wrappers.py tests/test_gpu_pp.py -q -k "propainter or plugin" 2>&1 | tail -3) >> $OUT/pytest_config4.log; tail -1 $OUT/pytest_config4.log
(timeout 900 python scripts/bench_configs.py 4 4h 2>&1 | grep '^{') > $OUT/configs_pp.log
python - <<'PY'
import json
for line in open("gpurun_out/r05_sixth/configs_pp.log"):
    d = json.loads(line)
    if "error" in d:
        print(d); continue
    print(d["config"], "|", d["value"], "fps", d["s_per_batch"], "s/batch; PSNR vs exact", d["psnr_db_vs_exact_mode"], {k: v.get("s") for k, v in d["stages"].items()})
PY
CLIP=/tmp/vsr_e2e_clip_1080p_600_on.y4m
for p in f32 f16; do
  (timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 600 --always-on --mode propainter --precision $p 2>&1 | grep '"metric"') > $OUT/e2e_pp_$p.json
  python -c "
import json; d=json.load(open('$OUT/e2e_pp_$p.json')); print('config 4 file to file, 600 frames always on, $p:', d['value'], 'fps', d['wall_s'], 's', d['phases_s'])"
done
(timeout 900 python scripts/bench_e2e.py --frames 1200 --always-on --mode sttn-det 2>&1 | grep '"metric"') > $OUT/e2e_det.json
python -c "
import json; d=json.load(open('$OUT/e2e_det.json')); print('config 3 file to file, 1200 frames always on:', d['value'], 'fps', d['wall_s'], 's', d['phases_s'])"
(timeout 900 python scripts/bench_e2e.py --frames 1200 --mode sttn-det 2>&1 | grep '"metric"') > $OUT/e2e_det_intervals.json
python -c "
import json; d=json.load(open('$OUT/e2e_det_intervals.json')); print('config 3 file to file, 1200 frames (100 of 120 on):', d['value'], 'fps', d['wall_s'], 's', d['phases_s'])"
