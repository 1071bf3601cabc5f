mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "tile or teacher or against_oracle or edge_shapes or train_mode" > gpurun_out/r04/tile2_tests_b.log 2>&1
tail -3 gpurun_out/r04/tile2_tests_b.log
