mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -q > gpurun_out/r04/gputests_c.log 2>&1
tail -4 gpurun_out/r04/gputests_c.log
python bench.py > gpurun_out/r04/bench_c.json 2> gpurun_out/r04/bench_c.err
tail -c 600 gpurun_out/r04/bench_c.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04/bench_c.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')})
print('roofline', {k:d['roofline'][k] for k in ('kernel','frac','avg_launch_us')})
for k in ('roofline_step','roofline_north_star','roofline_north_star_vrp','roofline_cfg5'):
    r=d.get(k)
    if r: print(k, {x:r[x] for x in ('kernel','frac','avg_launch_us','loop_frac','rollout_us','traffic')})
for k,v in d.get('other_configs',{}).items(): print(k, {x:v[x] for x in v if x in ('ms_per_step','node_steps_per_s','error')}, v.get('roofline_train',{}).get('frac'))
print('cpu', d.get('cpu_baseline',{}).get('value'))
PY
