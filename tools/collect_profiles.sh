#!/bin/bash
# Collects the round's rocprofv3 summaries on the GPU box (run through gpurun from the repo root):
#   kernel stats of the default bench command, per-shape kernel stats (TSP / VRP 8192x40,
#   VRP 2048x100 sampling, TSP 512x20), training epochs of configs 3 and 4, and the PMC passes
#   (FETCH_SIZE / WRITE_SIZE in separate runs, --kernel-trace only) behind roofline.traffic.
# usage: bash tools/collect_profiles.sh r06
set -u
R=${1:-r06}
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
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/mfma_vrp100 -o p -- python3 tools/rollout_loop.py 1 100 2048 3 0 > $OUT/pmc_mfma100.log 2>&1
# round 4: the tall GEMM of the training path (docs/rounds/DESIGN_rounds_1-5.md 3.4.1), the raw-tile kernel's phases and
# the per-CU streaming rate behind docs/rounds/DESIGN_rounds_1-5.md 3.5.1
VRP_GEMM_VARIANT=rows rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/gemm_rows -o p -- python3 tools/gemm_one.py 81920 384 128 > $OUT/pmc_gemm_rows.log 2>&1
{ python3 tools/tile_phase_probe.py 1 100 2048 3 1; python3 tools/tile_phase_probe.py 0 40 8192 3 0; VRP_TILE_V1=1 python3 tools/tile_phase_probe.py 1 100 2048 3 1; VRP_TILE_V1=1 python3 tools/tile_phase_probe.py 0 40 8192 3 0; } 2>/dev/null | grep "us through phase" > $OUT/tile_phases.txt
[ -x tools/micro/stream_rate ] && tools/micro/stream_rate > $OUT/stream_rate.txt 2>&1
# round 5: fp32 products on the bf16 matrix cores: accuracy / rate probe, and the
# LDS counters of the bf16-plane kernels (bank conflicts of the swizzled operand images)
[ -x tools/micro/bf16x3_probe ] && tools/micro/bf16x3_probe > $OUT/bf16x3_probe.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc/lds_tsp40 -o p -- python3 tools/rollout_loop.py 0 40 8192 3 > $OUT/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc/lds_tsp20 -o p -- python3 tools/rollout_loop.py 0 20 512 10 > $OUT/pmc_lds20.log 2>&1
{ echo "== bias + residual + ReLU (gemm_rows_kernel)"; VRP_GEMM_VARIANT=rows python3 tools/gemm_rows_probe.py run 2>/dev/null | grep "M="; echo "== bias only (N = 384 / 256, K = 128: gemm_rows_wide_kernel)"; GEMM_PROBE_PLAIN=1 VRP_GEMM_VARIANT=rows python3 tools/gemm_rows_probe.py run 2>/dev/null | grep "M="; echo "== bias only, VRP_GEMM_ROWS_NARROW=1 (gemm_rows_kernel everywhere)"; GEMM_PROBE_PLAIN=1 VRP_GEMM_ROWS_NARROW=1 VRP_GEMM_VARIANT=rows python3 tools/gemm_rows_probe.py run 2>/dev/null | grep "M="; } > $OUT/gemm_rows_probe.txt
{ echo "== gemm_tn_x3_kernel (bf16 planes, the default)"; python3 tools/gemm_tn_probe.py 2>/dev/null | grep "R="; echo "== VRP_GEMM_FP32=1: gemm_tn_kernel (fp32 MFMA), same box"; VRP_GEMM_FP32=1 python3 tools/gemm_tn_probe.py 2>/dev/null | grep "R="; } > $OUT/gemm_tn_probe.txt
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc/lds_gemm_tn -o p -- python3 tools/gemm_tn_probe.py > $OUT/pmc_lds_gemm_tn.log 2>&1
# round 6: whole-rollout wall clock against the round-5 library on THIS box (libvar_r05.so = the library
# of commit 8d3e66b, built by the caller), and the stack kernel's per-phase shader-clock trace
[ -f vrp-gym_amd/vrpgym_hip/libvar_r05.so ] && bash tools/gpu_r6_wall3.sh 2>&1 | grep "kind=" > $OUT/wall_vs_r05.txt
[ -f vrp-gym_amd/vrpgym_hip/libvar_trace.so ] && VRPGYM_HIP_LIB=$PWD/vrp-gym_amd/vrpgym_hip/libvar_trace.so python3 tools/rollout_loop.py 0 20 512 10 > $OUT/stack_trace.txt 2>&1
# ... and the fused prologue kernel's (make EXTRA=-DVRP_PRO_TRACE build of decoder_prologue.hip)
[ -f vrp-gym_amd/vrpgym_hip/libvar_ptrace.so ] && { VRPGYM_HIP_LIB=$PWD/vrp-gym_amd/vrpgym_hip/libvar_ptrace.so python3 tools/prologue_sizes.py 0 8192 40; VRPGYM_HIP_LIB=$PWD/vrp-gym_amd/vrpgym_hip/libvar_ptrace.so python3 tools/prologue_sizes.py 0 512 20; } > $OUT/prologue_trace.txt 2>&1
ls $OUT
