mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_backward.py -q -m gpu -x > gpurun_out/r04/bwd_tests.log 2>&1
tail -8 gpurun_out/r04/bwd_tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "training_step or train_loop or reaches_the_reference" 2>&1 | tail -3
python tools/train_probe.py 1 40 2048 8 2>/dev/null | tail -1; python tools/train_probe.py 2 40 1024 8 2>/dev/null | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/train_mfma_attn -o p -- python3 tools/train_probe.py 1 40 2048 5 > gpurun_out/r04/train_mfma_attn.log 2>&1
