# GPU check of the split-format path: its kernel + end-to-end tests, then a short bench without the CPU leg
TAG=${1:-v5}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q --tb=short -x -s -k "split or to_split or fp16" 2>&1 | tail -40) > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 --e2e-chunks 0 --cpu-sample-frames 6 > $OUT/bench.log 2>&1
python - $OUT/bench.log <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        ok=True; d=json.loads(l); print('fps',d['value'],'ms/step',d['ms_per_step'],'\n split',d.get('split_half_mode'),'\n fmt',d.get('split_format_mode'),'\n f16',d.get('fp16_mode'))
if not ok: print(open(sys.argv[1]).read()[-3000:])
PY
