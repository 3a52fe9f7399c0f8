#!/bin/bash
# copies the summaries of scripts/r06/final_profile.sh (gpurun_out/r06_final, scratch) into profiles/ (tracked); run from the repo root
S=gpurun_out/r06_final; D=profiles
cp $S/bench.log $D/r06_bench.log
cp $S/bench_under_rocprof.json $D/r06_bench_under_rocprof.json
cp $S/bench_lanes2_under_rocprof.json $D/r06_bench_two_lanes_under_rocprof.json
cp $S/f16_bench_under_rocprof.json $D/r06_f16_bench_under_rocprof.json
cp $S/trace/r_kernel_stats.csv $D/r06_kernel_stats.csv
cp $S/trace_lanes2/r_kernel_stats.csv $D/r06_kernel_stats_two_lanes.csv
cp $S/f16_trace/r_kernel_stats.csv $D/r06_f16_kernel_stats.csv
for s in det raft rfc lama; do
  n=$s; [ $s = det ] && n=detector
  cp $S/${s}_trace/r_kernel_stats.csv $D/r06_${n}_kernel_stats.csv
  cp $S/${s}_bench_under_rocprof.json $D/r06_${n}_bench_under_rocprof.json
done
cp $S/stages/r06_propainter_*_kernel_stats.csv $D/
cat $S/stage_stats_4.log $S/stage_stats_4h.log > $D/r06_propainter_stage_stats.log
cp $S/config_traffic.json $D/config_traffic.json
cp $S/config_traffic.json $D/r06_config_traffic.json
cp $S/pmc.log $D/r06_pmc_configs.log
cp $S/cfg_4s.json $D/r06_config_4s.json
cp $S/e2e_pp_f32.json $S/e2e_pp_f16.json $S/e2e_det.json $D/ 2>/dev/null
for f in e2e_pp_f32 e2e_pp_f16 e2e_det; do mv $D/$f.json $D/r06_$f.json; done
cp $S/cli.log $D/r06_cli.log
cp $S/det_bench.log $D/r06_detector_bench.log
cp $S/dryrun_2ranks.log $D/r06_final_dryrun_2ranks.log
ls -la $D | grep r06_ | wc -l
