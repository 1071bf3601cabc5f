mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "encoder or against_oracle or against_reference or full_size" > gpurun_out/r04/enc_tests.log 2>&1
tail -3 gpurun_out/r04/enc_tests.log
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/shape_cfg5 -o p -- python3 tools/rollout_loop.py 1 100 2048 6 0 > gpurun_out/r04/shape_cfg5.log 2>&1
python3 tools/kstat.py gpurun_out/r04/shape_cfg5 encoder gemm prologue decode | head -14
python tools/step_probe.py 1,100,2048,0,1 2>/dev/null | grep workload
