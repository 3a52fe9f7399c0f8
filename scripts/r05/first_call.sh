#!/bin/bash
# Round 5, ONE gpurun call (about 15 GPU-minutes) for everything round 4 built on the CPU and could not run:
#   gpurun --timeout 1500 -- 'bash scripts/r05/first_call.sh'
# 1. scripts/r05/cols_first_call.sh    STTN column ranges (VSR_DECODE_COLS): bit-equality tests, bench A/B
# 2. scripts/r05/pp_box_first_call.sh  ProPainter decoder box + per-frame encoder cache (VSR_PP_DECODE_BOX, VSR_PP_ENC_CACHE): tests, config 4 A/B
# 3. scripts/r05/det_lanes.sh          config 3 with 2 / 3 / 4 detector lanes and with the column ranges
# Then flip the defaults that are green and faster (video-subtitle-remover_amd/switches.py DEFAULTS), re-run pytest -m gpu.
mkdir -p gpurun_out      # (bench_e2e.py --clip creates the 600-frame clip on its first use and reuses it)
bash scripts/r05/cols_first_call.sh 2>&1 | tee gpurun_out/r05_first_call_cols.log
bash scripts/r05/pp_box_first_call.sh 2>&1 | tee gpurun_out/r05_first_call_pp.log
bash scripts/r05/det_lanes.sh 2>&1 | tee gpurun_out/r05_first_call_detlanes.log
