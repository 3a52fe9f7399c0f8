#!/bin/bash
# round 3, end of round: default bench line, rocprofv3 kernel-trace stats of the same command, PMC passes, f16 trace, configs 2/3/5
OUT=gpurun_out/r03; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py > $OUT/bench.log 2>&1; grep '"metric"' $OUT/bench.log | cut -c1-600
B="python bench.py --no-cpu-baseline --no-split-half --e2e-chunks 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $B > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_under_rocprof.json
B1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-split-half --e2e-chunks 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $B1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $B1 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -o r -- $B1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f16_trace -o r -- $B --steps 3 --warmup 1 --precision f16 > $OUT/f16_trace.log 2>&1
grep '"metric"' $OUT/f16_trace.log > $OUT/f16_bench_under_rocprof.json
rm -f $OUT/f16_trace/r_kernel_trace.csv
for b in lama rfc; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${b}_trace -o r -- python scripts/bench_$b.py > $OUT/${b}_trace.log 2>&1
  rm -f $OUT/${b}_trace/r_kernel_trace.csv; grep '"metric"' $OUT/${b}_trace.log | cut -c1-250
done
(timeout 400 python scripts/bench_configs.py 2>&1 | grep '^{') > $OUT/configs.log; cut -c1-300 $OUT/configs.log
ls $OUT $OUT/trace; du -sh $OUT
