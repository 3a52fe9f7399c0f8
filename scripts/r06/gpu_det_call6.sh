OUT=gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ocr_det.py -q -m gpu --tb=short 2>&1 | tail -30 > $OUT/pytest_det4.log; tail -12 $OUT/pytest_det4.log | cut -c1-250
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dbpost_trace2 -o r -- python scripts/r06/db_post_bench.py > $OUT/dbpost2.log 2>&1
rm -f $OUT/dbpost_trace2/r_kernel_trace.csv; grep "per map" $OUT/dbpost2.log; head -12 $OUT/dbpost_trace2/r_kernel_stats.csv | cut -c1-160
(timeout 900 python scripts/bench_e2e.py --frames 1200 --always-on --mode sttn-det 2>&1 | grep '"metric"') > $OUT/e2e_det2.json; cut -c1-420 $OUT/e2e_det2.json
