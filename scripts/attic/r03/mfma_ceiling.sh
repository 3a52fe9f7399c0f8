#!/bin/bash
# round 3, call 1: the fp32-MFMA ceiling microbenchmark + the shader clock the bench's dominant kernel runs at
OUT=gpurun_out/r03_ceiling; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 video-subtitle-remover_amd/build/mfma_ceiling > $OUT/mfma_ceiling.log 2>&1; tail -30 $OUT/mfma_ceiling.log
B1="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-split-half --e2e-chunks 0"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_clk -o r -- $B1 > $OUT/pmc_clk.log 2>&1
ls -la $OUT/pmc_clk/*; tail -3 $OUT/pmc_clk.log | cut -c1-300
