mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -q -m gpu -x -k "gemm or encoder or backward or training_step" > gpurun_out/r04/gemm_tests.log 2>&1
tail -3 gpurun_out/r04/gemm_tests.log
export VRP_GEMM_VARIANT=rows
echo "== pipelined epilogue"
python tools/gemm_rows_probe.py run 2>&1 | grep "M=" | head -9
echo "== pipelined, bias only"
GEMM_PROBE_PLAIN=1 python tools/gemm_rows_probe.py run 2>&1 | grep "M=" | head -5
export VRP_GEMM_ROWS_V1=1
echo "== burst epilogue (round 3)"
python tools/gemm_rows_probe.py run 2>&1 | grep "M=" | head -9
echo "== burst, bias only"
GEMM_PROBE_PLAIN=1 python tools/gemm_rows_probe.py run 2>&1 | grep "M=" | head -5
unset VRP_GEMM_ROWS_V1
python tools/train_probe.py 1 40 2048 8 2>&1 | tail -1
python tools/train_probe.py 2 40 1024 8 2>&1 | tail -1
