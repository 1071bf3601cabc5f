mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "non_default or training_step_against_reference" > gpurun_out/r04/arch_tests.log 2>&1
tail -30 gpurun_out/r04/arch_tests.log
