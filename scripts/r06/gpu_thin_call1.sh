OUT=gpurun_out/r06c; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "one_workgroup_per_tile" --tb=short 2>&1 | tail -5
L=$OUT/flow_thin_ab.log; : > $L
for v in 0 1 2; do VSR_FLOW_THIN=$v timeout 900 python scripts/r06/flow_thin_ab.py 2>&1 | grep -v amdgpu.ids >> $L; done
VSR_FLOW_THIN=1 VSR_FLOW_THIN_TILES=1024 VSR_FLOW_THIN_K=4800 AB_ROWS=0 timeout 900 python scripts/r06/flow_thin_ab.py 2>&1 | grep -v amdgpu.ids >> $L
grep "VSR_FLOW_THIN\|per engine" $L
