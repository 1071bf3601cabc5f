#!/bin/bash
# Collects the round's rocprofv3 summaries on the GPU box (run through gpurun from the repo root):
#   kernel stats of the default bench command, per-shape kernel stats (TSP / VRP 8192x40,
#   VRP 2048x100 sampling, TSP 512x20), training epochs of configs 3 and 4, and the PMC passes
#   (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only) behind roofline.traffic.
# usage: bash tools/collect_profiles.sh r03
set -u
R=${1:-r03}
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
# identity of the build every number below is measured on (bench.py reports roofline.traffic only
# from a file carrying the hash of the library it runs)
python3 -c "import sys; sys.path.insert(0, 'vrp-gym_amd'); import vrpgym_hip; print(vrpgym_hip.lib().vrp_source_hash().decode())" > $OUT/source_hash.txt
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_prof -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_prof.log 2>&1
for shp in 0,20,512 0,40,8192 1,40,8192 2,40,8192 1,100,2048,0,1; do
  tag=$(echo $shp | tr ',' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shape_$tag -o p -- python3 tools/rollout_loop.py $(echo $shp | cut -d, -f1-3 | tr ',' ' ') 6 $([ "$(echo $shp | cut -d, -f5)" = "1" ] && echo 0 || echo 1) > $OUT/shape_$tag.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_vrp40_b2048 -o p -- python3 tools/train_probe.py 1 40 2048 5 > $OUT/train_vrp40_b2048.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_irp40_b1024 -o p -- python3 tools/train_probe.py 2 40 1024 5 > $OUT/train_irp40_b1024.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_tsp20_b512 -o p -- python3 tools/train_probe.py 0 20 512 10 > $OUT/train_tsp20_b512.log 2>&1
for shp in 0,20,512 0,40,8192 1,40,8192 1,100,2048,0,1; do
  tag=kind$(echo $shp | cut -d, -f1)_N$(echo $shp | cut -d, -f2)_B$(echo $shp | cut -d, -f3)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/fetch_$tag -o p -- python3 tools/step_probe.py $shp > $OUT/pmc_fetch_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/write_$tag -o p -- python3 tools/step_probe.py $shp > $OUT/pmc_write_$tag.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/mfma_tsp40 -o p -- python3 tools/rollout_loop.py 0 40 8192 3 > $OUT/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/mfma_tsp20 -o p -- python3 tools/rollout_loop.py 0 20 512 10 > $OUT/pmc_mfma20.log 2>&1
ls $OUT
