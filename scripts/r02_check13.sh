#!/bin/bash
# A/B of the peek-before-steal tile queue on the fp32 headline; device DB post-process tests + timing
mkdir -p gpurun_out/r02n; export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_ocr_det.py -m gpu -q --tb=short -x -k "db_postprocess" 2>&1 | tail -15) > gpurun_out/r02n/pytest_det.log 2>&1
tail -3 gpurun_out/r02n/pytest_det.log
timeout 300 python - > gpurun_out/r02n/post_bench.log 2>&1 <<'PY'
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
    for _ in range(50): b, s = post(d, 1080, 1920)
    t1 = time.perf_counter()
    for _ in range(20): b2, s2 = ocr_det.db_postprocess(d.cpu().numpy(), 1080, 1920)
    t2 = time.perf_counter()
    print(f"DBPostProcess on a 960x544 map with {len(s)} boxes: all on the device {1e3*(t1-t0)/50:.3f} ms, all-host (D2H + scipy + numpy) {1e3*(t2-t1)/20:.2f} ms; boxes equal within 1 px: {bool(len(s)==len(s2) and (len(s)==0 or np.abs(b.astype(int)-b2).max()<=1))}")
PY
tail -2 gpurun_out/r02n/post_bench.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0"
run() { name=$1; shift; env "$@" timeout 300 $B > gpurun_out/r02n/$name.log 2>&1; python - gpurun_out/r02n/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); ok=True
        print(sys.argv[2],'fps',d['value'],'ms/step',d['ms_per_step'], {k:(v['ms'],v['tflops']) for k,v in d['op_breakdown'].items()})
if not ok: print(sys.argv[2],'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
}
L=video-subtitle-remover_amd/lib/libvsr_hip.so
cp $L /tmp/peek.so
run peek_1
cp video-subtitle-remover_amd/build/libvsr_hip_nopeek.so $L
run nopeek_1
cp /tmp/peek.so $L
run peek_2
cp video-subtitle-remover_amd/build/libvsr_hip_nopeek.so $L
run nopeek_2
cp /tmp/peek.so $L
