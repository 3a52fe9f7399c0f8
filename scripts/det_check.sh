TAG=${1:-det}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_ocr_det.py -m gpu -q --tb=short -x -s 2>&1 | tail -30) > $OUT/pytest.log 2>&1; tail -20 $OUT/pytest.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/bench.log
import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import vsr_amd
from vsr_amd.backend.tools import ocr_det
from vsr_amd.backend.tools.paddle_graph import load_graph
from oracle.ppocr_det import synthetic_weights
for fx in ("ppocr_det_fast_graph.json", "ppocr_det_graph.json"):
    g = load_graph(os.path.join('/root/repo/tests/golden', fx))
    det = ocr_det.TextDetection(g, synthetic_weights(g), device=0)
    img = np.random.default_rng(3).integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    for gemm, graph in ((True, False), (False, False)) + (((True, True),) if os.environ.get('VSR_DET_GRAPH') == '1' else ()):
        det.runner.use_gemm, det.use_graph = gemm, graph
        det.probability_map(img); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): det.probability_map(img)
        torch.cuda.synchronize()
        print(fx, "forward at 1080p (960x544 net input), dense convs on the %s, %s: %.1f ms/frame" % ("gather-GEMM" if gemm else "direct kernel", "HIP graph replay" if graph else "launch by launch", (time.perf_counter() - t0) / 5 * 1e3))
PY
