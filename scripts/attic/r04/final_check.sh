#!/bin/bash
# last check of the round: the STTN / wrapper / IO / kernel / BASELINE-size tests and the default bench line on the final tree
OUT=gpurun_out/r04; mkdir -p $OUT
(timeout 1200 python -m pytest tests/test_gpu_sttn.py tests/test_gpu_golden_wrappers.py tests/test_gpu_io.py tests/test_gpu_kernels.py tests/test_gpu_zbaseline.py tests/test_gpu_multirank.py -q 2>&1 | tail -3) > $OUT/pytest_gpu_final_subset.log; tail -1 $OUT/pytest_gpu_final_subset.log
python bench.py > $OUT/bench.log 2>&1; grep '"metric"' $OUT/bench.log | cut -c1-200
