# quick GPU check: kernel + e2e tests, default bench (no CPU baseline), fetch/write PMC of one chunk
TAG=${1:-quick}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sttn.py -m gpu -q --tb=short -x 2>&1 | tail -30) > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1
python - $OUT/bench.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); print('fps',d['value'],'ms/step',d['ms_per_step'],d['roofline']['kernel'][:40],d['roofline']['achieved'], {k:(v['ms'],v['tflops']) for k,v in d['op_breakdown_timed_region'].items()})
PY
B1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $B1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $B1 > $OUT/pmc_write.log 2>&1
python - $OUT <<'PY'
import pandas as pd,sys
o=sys.argv[1]
for n in ('fetch','write'):
    df=pd.read_csv(f'{o}/pmc_{n}/r_counter_collection.csv'); df['k']=df.Kernel_Name.str.replace(r'\(.*','',regex=True)
    print(n, df[df.k.str.contains('gather')].groupby('k').Counter_Value.agg(['size','mean']).to_string())
PY
