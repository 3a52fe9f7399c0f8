#!/bin/bash
# round 6, ninth call: probe of gather_gemm_f16_v9 (ring of four 32-deep stages + complementary wave roles) against v7
OUT=gpurun_out/r06_ninth; mkdir -p $OUT
timeout 600 ./video-subtitle-remover_amd/build/v9_probe 1 > $OUT/v9_probe.log 2>&1; echo "rc=$?"; cat $OUT/v9_probe.log
