#!/bin/bash
# round 4, end of round: default bench line, rocprofv3 kernel-trace stats (single-lane = the pass `roofline` is measured on, and the
# default two-lane command), PMC passes, f16 trace, configs 2/3/5 resident, config 2 through the CLI, the two-rank dry run
OUT=gpurun_out/r04; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py > $OUT/bench.log 2>&1; grep '"metric"' $OUT/bench.log | cut -c1-300
B="python bench.py --no-cpu-baseline --no-split-half --e2e-chunks 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $B --lanes 1 > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lanes2 -o r -- $B > $OUT/trace_lanes2.log 2>&1
grep '"metric"' $OUT/trace_lanes2.log > $OUT/bench_lanes2_under_rocprof.json
rm -f $OUT/trace/r_kernel_trace.csv $OUT/trace_lanes2/r_kernel_trace.csv
B1="python bench.py --lanes 1 --steps 1 --warmup 0 --no-cpu-baseline --no-split-half --e2e-chunks 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $B1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $B1 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -o r -- $B1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f16_trace -o r -- $B --lanes 1 --steps 3 --warmup 1 --precision f16 > $OUT/f16_trace.log 2>&1
grep '"metric"' $OUT/f16_trace.log > $OUT/f16_bench_under_rocprof.json
rm -f $OUT/f16_trace/r_kernel_trace.csv
F1="$B1 --precision f16"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f16_pmc_fetch -o r -- $F1 > $OUT/f16_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/f16_pmc_write -o r -- $F1 > $OUT/f16_pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/f16_pmc_sq -o r -- $F1 > $OUT/f16_pmc_sq.log 2>&1
(timeout 400 python scripts/bench_configs.py 2>&1 | grep '^{') > $OUT/configs.log; cut -c1-250 $OUT/configs.log
for a in "--res 720p --frames 300" "--res 1080p --frames 600"; do (timeout 300 python scripts/bench_cli.py $a 2>/dev/null | grep '^{') >> $OUT/cli.log; done; cut -c1-200 $OUT/cli.log
for corrupt in 0 1; do
  VSR_BENCH_DRYRUN_1GPU=1 VSR_BENCH_SELFTEST_CORRUPT=$corrupt timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 2951$corrupt bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/dryrun_2ranks_corrupt$corrupt.log 2>&1
  echo "corrupt=$corrupt rc=$?"; grep -o '"selftest": {[^}]*}' $OUT/dryrun_2ranks_corrupt$corrupt.log | cut -c1-300; grep "SELFTEST" $OUT/dryrun_2ranks_corrupt$corrupt.log
  grep -o '"value": [0-9.]*, "unit": "frames/s", "n_gpus": 2' $OUT/dryrun_2ranks_corrupt$corrupt.log
done
ls $OUT; du -sh $OUT
