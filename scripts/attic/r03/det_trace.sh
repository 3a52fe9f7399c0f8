#!/bin/bash
OUT=gpurun_out/r03_det; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- python scripts/r03/det_trace.py > $OUT/trace.log 2>&1; tail -2 $OUT/trace.log
python - <<'PY'
import pandas as pd, glob
f = glob.glob('gpurun_out/r03_det/trace/*kernel_stats.csv')[0]
d = pd.read_csv(f)
d['Name'] = d['Name'].str.slice(0, 70)
print(d[['Name','Calls','TotalDurationNs','AverageNs','Percentage']].head(22).to_string())
PY
rm -f $OUT/trace/*kernel_trace.csv
