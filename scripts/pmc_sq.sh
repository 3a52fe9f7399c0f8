# SQ stall counters of the bench (one chunk) -- PMC only, no tracing domains besides kernel dispatch
mkdir -p gpurun_out/pmc_sq; export TMPDIR=/tmp
B1="python bench.py --steps 1 --warmup 0 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_sq/a -o r1 -- $B1 > gpurun_out/pmc_sq/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL --output-format csv -d gpurun_out/pmc_sq/b -o r1 -- $B1 > gpurun_out/pmc_sq/b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_sq/fetch -o r1 -- $B1 > gpurun_out/pmc_sq/fetch.log 2>&1
VSR_CONV_KORDER=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_sq/fetch_tapmajor -o r1 -- $B1 > gpurun_out/pmc_sq/fetch_tapmajor.log 2>&1
ls -R gpurun_out/pmc_sq | head; timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "reduce or decode or blend or im2col or upsample" 2>&1 | tail -3
