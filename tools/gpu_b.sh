mkdir -p gpurun_out/r04
export VRPGYM_BENCH_ONE_GPU=1
timeout 600 python bench.py --gpus 8 --steps 2 --warmup 1 --no-cpu-baseline --no-north-star --no-extras --workload irp40_b1024_train > gpurun_out/r04/rank8_train.out 2> gpurun_out/r04/rank8_train.err
echo "rank8 rc=$?"
unset VRPGYM_BENCH_ONE_GPU
grep -v "^\[W\|^W1\|warnings.warn" gpurun_out/r04/rank8_train.err | grep -iE "error|abort|terminate|what\(\)|hip|HSA|memory" | head -20
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "tile or teacher or against_oracle or in_kernel or edge_shapes or full_size" > gpurun_out/r04/tile2_tests.log 2>&1
tail -4 gpurun_out/r04/tile2_tests.log
for v in 0 1; do
  if [ $v = 1 ]; then export VRP_TILE_V1=1; else unset VRP_TILE_V1; fi
  echo "== V1=$v"
  python tools/step_probe.py 1,100,2048,4,1 0,40,8192,4 1,40,8192,4 1,100,2048,0,1 0,40,8192 1,40,8192 2>/dev/null | grep workload
done
