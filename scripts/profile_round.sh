# Round profile: GPU tests, the default bench line, rocprofv3 kernel-trace stats of the same command, PMC passes.
# usage: bash scripts/profile_round.sh <tag>     (results under gpurun_out/<tag>/)
TAG=${1:-prof}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
# PYTEST_ARGS narrows the GPU test run (the whole suite takes 16 minutes of box time, most of it waiting for the BASELINE-size CPU oracles)
(timeout 1500 python -m pytest ${PYTEST_ARGS:-tests} -m gpu -q --tb=short 2>&1 | tail -40) > $OUT/pytest_gpu.log 2>&1
tail -2 $OUT/pytest_gpu.log
(timeout 400 python -c "import __graft_entry__ as g; g.smoke()") > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench.log 2>&1; grep '"metric"' $OUT/bench.log | cut -c1-400
B="python bench.py --no-cpu-baseline --no-configs --no-split-half --e2e-chunks 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $B > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_under_rocprof.json
B1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-split-half --e2e-chunks 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $B1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $B1 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -o r -- $B1 > $OUT/pmc_sq.log 2>&1
# the fp16-operand mode (BASELINE.json config 5's arithmetic): kernel-trace stats + counters of its own kernels
F1="$B1 --precision f16"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f16_trace -o r -- $B --steps 3 --warmup 1 --precision f16 > $OUT/f16_trace.log 2>&1
grep '"metric"' $OUT/f16_trace.log > $OUT/f16_bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f16_pmc_fetch -o r -- $F1 > $OUT/f16_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/f16_pmc_write -o r -- $F1 > $OUT/f16_pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/f16_pmc_sq -o r -- $F1 > $OUT/f16_pmc_sq.log 2>&1
rm -f $OUT/f16_trace/r_kernel_trace.csv $OUT/trace/r_kernel_trace.csv.bak; ls $OUT $OUT/trace; du -sh $OUT
