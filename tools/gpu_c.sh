mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_persistent_guard.py tests/test_gpu_backward.py -q -m gpu -x -k "persistent or fused or kats or edge_shapes or in_kernel or training or backward" > gpurun_out/r04/p4_tests.log 2>&1
tail -4 gpurun_out/r04/p4_tests.log
python tools/step_probe.py 0,20,512 1,40,1024 2>/dev/null | grep workload | cut -c1-330
python bench.py --no-cpu-baseline --no-north-star --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['dispersion'])"
