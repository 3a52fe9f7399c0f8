#!/bin/bash
OUT=gpurun_out/r03_probe; mkdir -p $OUT
for i in 1 2 3; do timeout 300 video-subtitle-remover_amd/build/v3_probe stress > $OUT/stress_$i.log 2>&1; cat $OUT/stress_$i.log; done
