#!/bin/bash
OUT=gpurun_out/r03_probe; mkdir -p $OUT
timeout 300 video-subtitle-remover_amd/build/v3_probe > $OUT/v3_probe.log 2>&1; cat $OUT/v3_probe.log
