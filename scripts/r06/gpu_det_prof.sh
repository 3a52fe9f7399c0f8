# rocprofv3 kernel stats of the detector's NHWC plan (server program, 8 frames per forward at 960 x 544)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
DET_AB_ONLY=plan DET_AB_CASES=ppocr_det_graph.json:8 DET_AB_REPS=5 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_plan -o plan -- python /root/repo/scripts/r06/det_nhwc_ab.py > /root/repo/gpurun_out/r06b_prof_plan.log 2>&1
find /tmp/prof_plan -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/r06b_detector_plan_kernel_stats.csv \;
cd /root/repo
grep "ms/frame" gpurun_out/r06b_prof_plan.log
head -22 gpurun_out/r06b_detector_plan_kernel_stats.csv | cut -c1-200
