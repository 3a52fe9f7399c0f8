#!/bin/bash
OUT=gpurun_out/r03_pmc; mkdir -p $OUT; export TMPDIR=/tmp
for shape in "qk 1" "qk 8" "conv 8" "qkv 1"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${tag}_fetch -o r -- video-subtitle-remover_amd/build/v3_probe $shape > $OUT/${tag}_fetch.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/${tag}_tcc -o r -- video-subtitle-remover_amd/build/v3_probe $shape > $OUT/${tag}_tcc.log 2>&1
  grep TF $OUT/${tag}_fetch.log
done
python - <<'PY'
import pandas as pd, glob
for tag in ["qk_1","qk_8","conv_8","qkv_1"]:
    for kind in ["fetch","tcc"]:
        fs = glob.glob(f'gpurun_out/r03_pmc/{tag}_{kind}/*counter_collection.csv')
        if not fs: print(tag, kind, 'no csv'); continue
        d = pd.read_csv(fs[0]); d = d[d.Kernel_Name.str.contains('gather_gemm')]
        print(tag, kind, d.groupby('Counter_Name').Counter_Value.mean().to_dict())
PY
