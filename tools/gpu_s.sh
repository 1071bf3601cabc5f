mkdir -p gpurun_out/r04
export VRPGYM_TRAIN_PARITY_LOG=gpurun_out/r04/train_parity_sweep_c.csv
timeout 1000 python tools/parity_sweep.py 404 900 > gpurun_out/r04/sweep_c.log 2>&1; tail -3 gpurun_out/r04/sweep_c.log
timeout 700 python tools/parity_sweep.py 405 600 train > gpurun_out/r04/sweep_train_c.log 2>&1; tail -3 gpurun_out/r04/sweep_train_c.log
