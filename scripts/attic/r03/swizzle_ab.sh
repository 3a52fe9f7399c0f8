#!/bin/bash
OUT=gpurun_out/r03_swz; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 video-subtitle-remover_amd/build/v3_probe > $OUT/v3_probe.log 2>&1; grep -A8 "== dense" $OUT/v3_probe.log | grep -v xcd
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/qk_swz_fetch -o r -- video-subtitle-remover_amd/build/v3_probe qk 264 > $OUT/qk_swz_fetch.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/qk_swz_tcc -o r -- video-subtitle-remover_amd/build/v3_probe qk 264 > $OUT/qk_swz_tcc.log 2>&1
python - <<'PY'
import pandas as pd, glob
for kind in ["fetch","tcc"]:
    fs = glob.glob(f'gpurun_out/r03_swz/qk_swz_{kind}/*counter_collection.csv')
    d = pd.read_csv(fs[0]); d = d[d.Kernel_Name.str.contains('gather_gemm')]
    print('qk grouped order, 8 queues', kind, d.groupby('Counter_Name').Counter_Value.mean().to_dict())
PY
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sttn.py -m gpu -x -q 2>&1 | tail -3
for s in 0 1; do VSR_GG_SWIZZLE=$s python bench.py --no-cpu-baseline --e2e-chunks 0 --no-split-half > $OUT/bench_swz$s.log 2>&1; python - <<PY
import json
l=[x for x in open('gpurun_out/r03_swz/bench_swz$s.log') if x.startswith('{"metric"')][0]
d=json.loads(l)
print('VSR_GG_SWIZZLE=$s', d['value'], 'fps, roofline frac', d['roofline']['frac'], {k: v['tflops'] for k, v in d['op_breakdown'].items()})
PY
done
