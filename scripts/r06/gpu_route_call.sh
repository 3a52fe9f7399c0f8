OUT=gpurun_out/r06c; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ocr_det.py -q -m gpu -k "ccl_labels or db_postprocess" --tb=short 2>&1 | tail -5
VSR_STTN_DUMP_ROUTING=1 timeout 600 python scripts/bench_configs.py 5 > $OUT/cfg5.json 2> $OUT/routing_f16_4k.log
grep '^{' $OUT/cfg5.json | cut -c1-300
grep routing $OUT/routing_f16_4k.log | sort | uniq -c | sort -rn | head -60
