mkdir -p gpurun_out/r04
python3 bench.py > gpurun_out/r04/bench.json 2> gpurun_out/r04/bench.err
tail -c 300 gpurun_out/r04/bench.json
timeout 900 python -m pytest tests/test_bench_multirank.py -q -m gpu -x 2>&1 | tail -3
