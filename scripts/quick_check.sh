# GPU check: all gpu tests, default bench (all legs), 2-rank dry run of the distributed bench path on one GPU
TAG=${1:-quick}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -30) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python bench.py > $OUT/bench.log 2>&1
python - $OUT/bench.log <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        ok=True; d=json.loads(l); print('fps',d['value'],'ms/step',d['ms_per_step'],'roof',d['roofline']['kernel'][:36],d['roofline']['achieved'],d['roofline']['traffic'],'\n cpu',d.get('cpu_baseline'),'\n psnr',d.get('psnr_db_vs_oracle'),'\n e2e',d.get('pcie_inclusive'),'\n split',d.get('split_half_mode'))
if not ok: print(open(sys.argv[1]).read()[-3000:])
PY
VSR_BENCH_DRYRUN_1GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0 > $OUT/bench_2rank_dry.log 2>&1; grep -E '"metric"|Error|error' $OUT/bench_2rank_dry.log | cut -c1-260
