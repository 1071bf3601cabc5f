mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "paced or persistent_steps_equal" > gpurun_out/r04/paced_tests.log 2>&1
tail -4 gpurun_out/r04/paced_tests.log
for v in 1 0; do
  if [ $v = 1 ]; then export VRP_NO_THROTTLE=1; else unset VRP_NO_THROTTLE; fi
  echo "no_throttle=$v"
  python bench.py --no-cpu-baseline --no-north-star 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print({k:v['ms_per_step'] for k,v in d['other_configs'].items()}); print(d.get('roofline_cfg5',{}).get('frac'), d.get('roofline_cfg5',{}).get('rollout_us'))"
done
