OUT=gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ocr_det.py -q -m gpu -k "batch_equals or hole_count" --tb=short 2>&1 | tail -30 > $OUT/pytest_det3.log; tail -30 $OUT/pytest_det3.log | cut -c1-250
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dbpost_trace -o r -- python scripts/r06/db_post_bench.py > $OUT/dbpost.log 2>&1
rm -f $OUT/dbpost_trace/r_kernel_trace.csv; grep "per map" $OUT/dbpost.log; head -14 $OUT/dbpost_trace/r_kernel_stats.csv | cut -c1-160
