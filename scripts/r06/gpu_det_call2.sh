# detector NHWC plan as the default: its GPU tests, everything that runs a detector, the config-3 detector leg, kernel stats, file to file
OUT=gpurun_out/r06c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ocr_det.py -q -m gpu 2>&1 | tail -15 > $OUT/pytest_det.log
tail -5 $OUT/pytest_det.log
(timeout 600 python scripts/bench_configs.py 3d 2>/dev/null | grep '^{') > $OUT/cfg_3d.json; cut -c1-600 $OUT/cfg_3d.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/det_trace -o r -- python scripts/bench_det.py > $OUT/det_trace.log 2>&1
rm -f $OUT/det_trace/*/r_kernel_trace.csv $OUT/det_trace/r_kernel_trace.csv
grep -v amdgpu.ids $OUT/det_trace.log | tail -12
(timeout 900 python scripts/bench_e2e.py --frames 1200 --always-on --mode sttn-det 2>&1 | grep '"metric"') > $OUT/e2e_det.json; cut -c1-600 $OUT/e2e_det.json
find $OUT -name "*kernel_stats.csv" | head
