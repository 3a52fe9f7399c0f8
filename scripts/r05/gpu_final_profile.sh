#!/bin/bash
# round 5, end of round: default bench line, rocprofv3 kernel-trace stats (single-lane = the pass `roofline` is measured on, and the
# default two-lane command), PMC passes, f16 trace, configs 2/3/5 resident, config 2 through the CLI, the two-rank dry run
OUT=gpurun_out/r05; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py > $OUT/bench.log 2>&1; grep '"metric"' $OUT/bench.log | cut -c1-300
B="python bench.py --no-cpu-baseline --no-configs --no-split-half --e2e-chunks 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $B --lanes 1 > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lanes2 -o r -- $B > $OUT/trace_lanes2.log 2>&1
grep '"metric"' $OUT/trace_lanes2.log > $OUT/bench_lanes2_under_rocprof.json
rm -f $OUT/trace/r_kernel_trace.csv $OUT/trace_lanes2/r_kernel_trace.csv
B1="python bench.py --lanes 1 --steps 1 --warmup 0 --no-cpu-baseline --no-configs --no-split-half --e2e-chunks 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o r -- $B1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o r -- $B1 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -o r -- $B1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f16_trace -o r -- $B --lanes 1 --steps 3 --warmup 1 --precision f16 > $OUT/f16_trace.log 2>&1
grep '"metric"' $OUT/f16_trace.log > $OUT/f16_bench_under_rocprof.json
rm -f $OUT/f16_trace/r_kernel_trace.csv
F1="$B1 --precision f16"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f16_pmc_fetch -o r -- $F1 > $OUT/f16_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/f16_pmc_write -o r -- $F1 > $OUT/f16_pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/f16_pmc_sq -o r -- $F1 > $OUT/f16_pmc_sq.log 2>&1
# (BASELINE configs 2-5 are inside bench.log's line now: `configs`)
python - <<'PY'
import json
d = [json.loads(l) for l in open("gpurun_out/r05/bench.log") if l.startswith("{")][-1]
print("headline", d["value"], "fps", d["ms_per_step"], "ms; GFLOP/frame", d["gflop_per_frame"], "|", d["gflop_per_frame_reference"], "roofline", d["roofline"]["achieved"], d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
for k, v in d.get("configs", {}).items():
    r = v.get("roofline") or {}
    print("  config", k, v.get("value"), v.get("unit"), "|", v.get("model_tflops"), "TF |", r.get("kernel"), r.get("achieved"), r.get("frac"), v.get("error"), v.get("leg_seconds"), "s")
PY
# HBM traffic of config 4's kernels (separate PMC passes, as for the headline)
P4="python scripts/bench_configs.py 4"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pp_pmc_fetch -o r -- $P4 > $OUT/pp_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pp_pmc_write -o r -- $P4 > $OUT/pp_pmc_write.log 2>&1
python - <<'PY'
import csv, collections, json, os
agg = collections.defaultdict(lambda: [0, 0.0])
for d in ("pp_pmc_fetch", "pp_pmc_write"):
    p = os.path.join("gpurun_out/r05", d, "r_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for row in csv.DictReader(open(p, newline="")):
        k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
out = {}
for (k, c), (n, tot) in agg.items():
    out.setdefault(k, {})[c] = {"dispatches": n, "mean_KB": tot / n}
rows = []
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rows.append((v["FETCH_SIZE"]["dispatches"] * (2 * v["FETCH_SIZE"]["mean_KB"] + v["WRITE_SIZE"]["mean_KB"]), k, v))
rows.sort(reverse=True)
res = [{"kernel": k, "launches": v["FETCH_SIZE"]["dispatches"], "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"]["mean_KB"] + v["WRITE_SIZE"]["mean_KB"]) * 1024)} for _, k, v in rows[:12]]
json.dump({"command": "python scripts/bench_configs.py 4 (three 68-frame propainter calls)", "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950, MI355X_MICROARCH.md)", "kernels": res},
          open("gpurun_out/r05/pp_pmc_summary.json", "w"), indent=1)
for r in res[:6]:
    print(r)
PY
rm -rf $OUT/pp_pmc_fetch $OUT/pp_pmc_write
for a in "--res 720p --frames 300" "--res 1080p --frames 600"; do (timeout 300 python scripts/bench_cli.py $a 2>/dev/null | grep '^{') >> $OUT/cli.log; done; cut -c1-200 $OUT/cli.log
for corrupt in 0 1; do      # (no launcher: bench.py starts its own ranks)
  VSR_BENCH_DRYRUN_1GPU=1 VSR_BENCH_SELFTEST_CORRUPT=$corrupt timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/dryrun_2ranks_corrupt$corrupt.log 2>&1
  echo "corrupt=$corrupt rc=$?"; grep -o '"selftest": {[^}]*}' $OUT/dryrun_2ranks_corrupt$corrupt.log | cut -c1-300; grep "SELFTEST" $OUT/dryrun_2ranks_corrupt$corrupt.log
  grep -o '"value": [0-9.]*, "unit": "frames/s", "n_gpus": 2' $OUT/dryrun_2ranks_corrupt$corrupt.log
done
ls $OUT; du -sh $OUT
