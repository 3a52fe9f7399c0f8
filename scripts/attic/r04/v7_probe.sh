#!/bin/bash
# probe of the 256 x 256 fp16-operand kernel (scripts/r04/v7_probe.hip, built in the container)
OUT=gpurun_out/r04_v7; mkdir -p $OUT
timeout 300 video-subtitle-remover_amd/build/v7_probe > $OUT/probe_${1:-a}.log 2>&1; echo "rc=$?" >> $OUT/probe_${1:-a}.log
tail -120 $OUT/probe_${1:-a}.log
