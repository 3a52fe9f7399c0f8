#!/bin/bash
# the engines that share the changed kernels (v3 epilogue, reduce-scatter, decode): their GPU tests once more
mkdir -p gpurun_out/r02x; export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_gpu_sttn.py tests/test_gpu_lama.py tests/test_gpu_raft.py tests/test_gpu_rfc.py tests/test_gpu_ocr_det.py tests/test_gpu_flow_split.py tests/test_gpu_pp.py -m gpu -q --tb=short -x --durations=8 -k "not propainter_plugin_matches_oracle" 2>&1 | tail -30) > gpurun_out/r02x/pytest.log 2>&1
tail -14 gpurun_out/r02x/pytest.log
