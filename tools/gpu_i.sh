mkdir -p gpurun_out/r04
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/shape_cfg5_tp -o p -- python3 tools/rollout_loop.py 1 100 2048 6 0 > gpurun_out/r04/shape_cfg5_tp.log 2>&1
tail -1 gpurun_out/r04/shape_cfg5_tp.log
python3 tools/kstat.py gpurun_out/r04/shape_cfg5_tp decode persistent | head
timeout 300 python tools/step_probe.py 1,100,2048,0,1 2>&1 | grep -E "workload|rror" | cut -c1-500
