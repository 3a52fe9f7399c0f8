#!/bin/bash
# round 3, end of round: default bench line, rocprofv3 kernel-trace stats (single-lane = the pass `roofline` is measured on, and the
# default two-lane command), PMC passes, f16 trace, LaMa / RFC traces, configs 2/3/5 resident, config 3 file to file
OUT=gpurun_out/r03; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_sttn.py tests/test_gpu_golden_wrappers.py tests/test_gpu_multirank.py tests/test_gpu_io.py -m gpu -q 2>&1 | tail -3) > $OUT/pytest_gpu_lanes.log; tail -1 $OUT/pytest_gpu_lanes.log
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
(timeout 400 python scripts/bench_configs.py 2>&1 | grep '^{') > $OUT/configs.log; cut -c1-250 $OUT/configs.log
(timeout 500 python scripts/bench_e2e.py --mode sttn-det --frames 1200 2>&1 | grep '^{') > $OUT/e2e_config3.log; cut -c1-500 $OUT/e2e_config3.log
ls $OUT; du -sh $OUT
