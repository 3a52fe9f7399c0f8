set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ocr_det.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r06b_pytest_det.log
timeout 600 python scripts/r06/det_nhwc_ab.py > gpurun_out/r06b_det_nhwc_ab.log 2>&1
cd /tmp && export TMPDIR=/tmp
DET_AB_ONLY=plan DET_AB_CASES=ppocr_det_graph.json:8 DET_AB_REPS=5 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_plan -o plan -- python /root/repo/scripts/r06/det_nhwc_ab.py > /root/repo/gpurun_out/r06b_prof_plan.log 2>&1
find /tmp/prof_plan -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/r06b_detector_plan_kernel_stats.csv \;
cd /root/repo
cat gpurun_out/r06b_pytest_det.log | tail -12
cat gpurun_out/r06b_det_nhwc_ab.log
head -25 gpurun_out/r06b_detector_plan_kernel_stats.csv | cut -c1-220
