#!/bin/bash
# probe of the 288 x 256 exact-fp32 kernel (scripts/r04/v8_probe.hip, built in the container)
OUT=gpurun_out/r04_v8; mkdir -p $OUT
timeout 300 video-subtitle-remover_amd/build/v8_probe > $OUT/probe_${1:-a}.log 2>&1; echo "rc=$?" >> $OUT/probe_${1:-a}.log
tail -120 $OUT/probe_${1:-a}.log
