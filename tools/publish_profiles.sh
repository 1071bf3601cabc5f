#!/bin/bash
# Copies the summaries tools/collect_profiles.sh left under gpurun_out/<round>/ into profiles/
# (tracked) under their committed names.  usage: bash tools/publish_profiles.sh r04
set -eu
R=${1:-r06}
S=gpurun_out/$R
D=profiles
stats() { find "$1" -name '*kernel_stats.csv' | head -1; }
trace() { find "$1" -name '*kernel_trace.csv' | head -1; }
cp $S/bench.json $D/${R}_bench.json
cp "$(stats $S/bench_prof)" $D/${R}_bench_kernel_stats.csv
for shp in 0_20_512 0_40_8192 1_40_8192 2_40_8192 1_100_2048_0_1; do
  k=$(echo $shp | cut -d_ -f1); n=$(echo $shp | cut -d_ -f2); b=$(echo $shp | cut -d_ -f3)
  cp "$(stats $S/shape_$shp)" $D/${R}_rollout_kind${k}_N${n}_B${b}_kernel_stats.csv
done
python3 tools/timeline.py "$(trace $S/shape_0_20_512)" encoder_stack > $D/${R}_timeline_tsp20_b512.txt
python3 tools/timeline.py "$(trace $S/shape_0_40_8192)" > $D/${R}_timeline_tsp40_b8192.txt
python3 tools/timeline.py "$(trace $S/shape_1_40_8192)" > $D/${R}_timeline_vrp40_b8192.txt
python3 tools/timeline.py "$(trace $S/shape_1_100_2048_0_1)" > $D/${R}_timeline_vrp100_b2048.txt
for f in tile_phases.txt stream_rate.txt gemm_rows_probe.txt gemm_tn_probe.txt bf16x3_probe.txt wall_vs_r05.txt stack_trace.txt prologue_trace.txt source_hash.txt; do [ -f $S/$f ] && cp $S/$f $D/${R}_$f; done
for t in vrp40_b2048 irp40_b1024 tsp20_b512; do
  cp "$(stats $S/train_$t)" $D/${R}_train_${t}_kernel_stats.csv
done
python3 tools/pmc_traffic.py $S/pmc $D/${R}_traffic.json
# per-kernel summaries of the counter passes themselves (the raw per-dispatch CSVs stay in gpurun_out/)
for d in $S/pmc/fetch_* $S/pmc/write_* $S/pmc/mfma_* $S/pmc/lds_* $S/pmc/gemm_rows; do
  [ -d "$d" ] && python3 tools/pmc_summary_csv.py "$d" $D/${R}_pmc_$(basename $d).csv
done
ls -la $D | grep ${R}_
