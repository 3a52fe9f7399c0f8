#!/bin/bash
mkdir -p gpurun_out/r02j
timeout 900 python -m pytest tests/test_gpu_ocr_det.py tests/test_gpu_lama.py -m gpu -q -x > gpurun_out/r02j/pytest.log 2>&1
tail -3 gpurun_out/r02j/pytest.log
timeout 300 python scripts/bench_lama.py > gpurun_out/r02j/bench_lama.log 2>&1; grep metric gpurun_out/r02j/bench_lama.log | cut -c1-200
timeout 600 python - > gpurun_out/r02j/post_bench.log 2>&1 <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import vsr_amd
from vsr_amd.backend.tools import ocr_det
from test_gpu_ocr_det import _blob_map
for nb in (6, 0):
    prob = _blob_map(1, 544, 960, nb)
    if nb == 0:
        prob[:] = 0.1; prob[470:500, 200:760] = 0.9; prob[430:455, 300:650] = 0.85      # two subtitle lines, nothing else
    d = torch.from_numpy(prob).cuda()
    post = ocr_det.DeviceDBPostProcess(torch.device("cuda", 0))
    for _ in range(3): post(d, 1080, 1920)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): b, s = post(d, 1080, 1920)
    t1 = time.perf_counter()
    for _ in range(20): b2, s2 = ocr_det.db_postprocess(d.cpu().numpy(), 1080, 1920)
    t2 = time.perf_counter()
    print(f"DBPostProcess on a 960x544 map with {len(s)} boxes: device labelling + row download {1e3*(t1-t0)/20:.2f} ms, all-host (D2H + scipy) {1e3*(t2-t1)/20:.2f} ms")
PY
tail -2 gpurun_out/r02j/post_bench.log
