#!/bin/bash
# round 6, end of round, part B (part A = final_profile_a.sh: HBM traffic, read by bench.py): default bench line; rocprofv3 kernel-trace stats of the headline (single lane = the pass `roofline` is measured
# on, and the default two-lane command) and of the f16 mode; per-ENGINE kernel stats of one propainter batch (scripts/stage_stats.py);
OUT=gpurun_out/r06_final; mkdir -p $OUT; export TMPDIR=/tmp
python bench.py > $OUT/bench.log 2>&1; grep '"metric"' $OUT/bench.log | cut -c1-300
python - <<'PY'
import json
d = [json.loads(l) for l in open("gpurun_out/r06_final/bench.log") if l.startswith("{")][-1]
print("headline", d["value"], "fps", d["ms_per_step"], "ms; GFLOP/frame", d["gflop_per_frame"], "|", d["gflop_per_frame_reference"], "roofline", d["roofline"]["achieved"], d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "hbm_gbps", d.get("hbm_gbps"), "psnr", d.get("psnr_db_vs_oracle"), "cpu", d.get("cpu_baseline", {}).get("value"))
print("  modes:", {k: d[k]["value"] for k in ("split_half_mode", "split_format_mode", "fp16_mode") if k in d}, "full_work", d.get("full_work", {}).get("value"), "pcie", d.get("pcie_inclusive", {}).get("value"))
for k, v in d.get("configs", {}).items():
    r = v.get("roofline") or {}
    print("  config", k, v.get("value"), v.get("unit"), "|", v.get("model_tflops"), "TF |", r.get("kernel"), r.get("achieved"), r.get("frac"), "traffic", r.get("traffic"), "hbm_gbps", v.get("hbm_gbps"), v.get("error"), v.get("leg_seconds"), "s")
PY
B="python bench.py --no-cpu-baseline --no-configs --no-split-half --e2e-chunks 0 --no-full-work"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o r -- $B --lanes 1 > $OUT/trace.log 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lanes2 -o r -- $B > $OUT/trace_lanes2.log 2>&1
grep '"metric"' $OUT/trace_lanes2.log > $OUT/bench_lanes2_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f16_trace -o r -- $B --lanes 1 --steps 3 --warmup 1 --precision f16 > $OUT/f16_trace.log 2>&1
grep '"metric"' $OUT/f16_trace.log > $OUT/f16_bench_under_rocprof.json
rm -f $OUT/trace/r_kernel_trace.csv $OUT/trace_lanes2/r_kernel_trace.csv $OUT/f16_trace/r_kernel_trace.csv
for leg in 4 4h; do timeout 1200 python scripts/stage_stats.py --leg $leg --out $OUT/stages > $OUT/stage_stats_$leg.log 2>&1; cat $OUT/stage_stats_$leg.log | tail -5; done
for s in raft rfc lama; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${s}_trace -o r -- python scripts/bench_$s.py > $OUT/${s}_trace.log 2>&1
  grep '^{' $OUT/${s}_trace.log | tail -1 | cut -c1-300 > $OUT/${s}_bench_under_rocprof.json; rm -f $OUT/${s}_trace/r_kernel_trace.csv
done
# the detector's forward = the compiled NHWC plan only (server program, 8 frames of 960 x 544 per forward, 5 timed rounds)
DET_AB_ONLY=plan DET_AB_CASES=ppocr_det_graph.json:8 DET_AB_REPS=5 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/det_trace -o r -- python scripts/r06/det_nhwc_ab.py > $OUT/det_trace.log 2>&1
grep "ms/frame" $OUT/det_trace.log | cut -c1-260 > $OUT/det_bench_under_rocprof.json; rm -f $OUT/det_trace/r_kernel_trace.csv
timeout 300 python scripts/bench_det.py ppocr_det_graph.json 2>&1 | grep -v amdgpu.ids > $OUT/det_bench.log; cat $OUT/det_bench.log
timeout 300 python scripts/bench_configs.py 4s 2>/dev/null | grep '^{' > $OUT/cfg_4s.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_final/cfg_4s.json").read().splitlines()[-1]); print("config 4s:", d["value"], "fps", d["s_per_batch"], "s/batch psnr", d.get("psnr_db_vs_exact_mode"))
PY
CLIP=gpurun_out/e2e_clip_always.y4m
for p in f32 f16; do (timeout 900 python scripts/bench_e2e.py --clip $CLIP --frames 600 --always-on --mode propainter --precision $p 2>&1 | grep '"metric"') > $OUT/e2e_pp_$p.json; cut -c1-400 $OUT/e2e_pp_$p.json; done
rm -f $CLIP
(timeout 900 python scripts/bench_e2e.py --frames 1200 --always-on --mode sttn-det 2>&1 | grep '"metric"') > $OUT/e2e_det.json; cut -c1-400 $OUT/e2e_det.json
for a in "--res 720p --frames 300" "--res 1080p --frames 600"; do (timeout 300 python scripts/bench_cli.py $a 2>/dev/null | grep '^{') >> $OUT/cli.log; done; cut -c1-200 $OUT/cli.log
VSR_BENCH_DRYRUN_1GPU=1 VSR_BENCH_MULTI_PP_FRAMES=20 VSR_PP_LANES=1 VSR_RAFT_LANES=1 VSR_BENCH_LEG_TIMEOUT=240 timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 > $OUT/dryrun_2ranks.log 2>&1
echo "dryrun rc=$?"; grep '^{' $OUT/dryrun_2ranks.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('dry run headline', d['value'], 'n_gpus', d['n_gpus'], 'selftest', (d.get('replicas') or {}).get('selftest', {}).get('ok'))
    for k, v in (d.get('configs_multi') or {}).items():
        print('  multi', k, {a: v.get(a) for a in ('n_ranks', 'value', 'efficiency', 'selftest_ok', 'frames_written', 'error', 'hbm_gbps')})
"
# the fail-safe of the N-rank headline: the scatter / gather leg raises -> the line falls back to the replicas' rate
VSR_BENCH_SCATTER_FAIL=1 VSR_BENCH_DRYRUN_1GPU=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-multi-configs > $OUT/dryrun_2ranks_scatter_fail.log 2>&1
echo "scatter-fail dry run rc=$?"; grep '^{' $OUT/dryrun_2ranks_scatter_fail.log | cut -c1-900
ls $OUT; du -sh $OUT
