mkdir -p gpurun_out/r04
export VRPGYM_TRAIN_PARITY_LOG=gpurun_out/r04/train_parity_suite.csv
python -m pytest tests -m gpu -q > gpurun_out/r04/gputests_b.log 2>&1
tail -8 gpurun_out/r04/gputests_b.log
export VRPGYM_TRAIN_PARITY_LOG=gpurun_out/r04/train_parity_sweep_b.csv
timeout 700 python tools/parity_sweep.py 61 600 train > gpurun_out/r04/sweep_train_b.log 2>&1; tail -3 gpurun_out/r04/sweep_train_b.log
unset VRPGYM_TRAIN_PARITY_LOG
timeout 400 python tools/parity_sweep.py 7 300 > gpurun_out/r04/sweep_b.log 2>&1; tail -3 gpurun_out/r04/sweep_b.log
