# detector NHWC plan: step-by-step GPU vs CPU replay at the shapes of the failing comparisons, then the plan's GPU tests
mkdir -p gpurun_out
L=gpurun_out/r06c_det_triage.log
: > $L
for shp in "2 64 96" "3 96 160" "1 288 480" "3 288 480"; do
  timeout 600 python scripts/r06/det_plan_steps_vs_cpu.py ppocr_det_graph.json $shp 2>&1 | grep -v amdgpu.ids | head -60 >> $L
done
timeout 600 python scripts/r06/det_plan_steps_vs_cpu.py ppocr_det_fast_graph.json 2 160 224 2>&1 | grep -v amdgpu.ids | head -60 >> $L
timeout 900 python -m pytest tests/test_gpu_ocr_det.py -q -m gpu -k "nhwc or batched" 2>&1 | tail -40 > gpurun_out/r06c_pytest_det.log
cat $L | cut -c1-400
tail -30 gpurun_out/r06c_pytest_det.log | cut -c1-300
