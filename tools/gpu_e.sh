mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py tests/test_gpu_backward.py -q -m gpu -x -k "gemm or encoder or backward or training_step or clipping" > gpurun_out/r04/gemm_tests.log 2>&1
tail -3 gpurun_out/r04/gemm_tests.log
export VRP_GEMM_VARIANT=rows
echo "== four-wave, two workgroups per CU"
python tools/gemm_rows_probe.py run 2>&1 | grep "M=" | head -9
echo "== four-wave, bias only"
GEMM_PROBE_PLAIN=1 python tools/gemm_rows_probe.py run 2>&1 | grep "M=" | head -5
export VRP_GEMM_ROWS=8
echo "== eight-wave (round 3)"
python tools/gemm_rows_probe.py run 2>&1 | grep "M=" | head -9
unset VRP_GEMM_ROWS
python tools/train_probe.py 1 40 2048 8 2>&1 | tail -1
VRP_GEMM_ROWS=8 python tools/train_probe.py 1 40 2048 8 2>&1 | tail -1
