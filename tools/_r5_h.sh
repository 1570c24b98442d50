set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_a.json 2> gpurun_out/r5_bench_a.err
timeout 300 python -m pytest tests/test_gpu_ppo.py -q -s -k "full_size_update_vs_reference or kl_early_stop" 2>&1 | grep -v "^$" | tail -12 > gpurun_out/fs_box2.log
python - <<'P'
import json
j=json.loads(open('gpurun_out/r5_bench_a.json').read().strip().splitlines()[-1])
print(json.dumps({k:j[k] for k in ('value','ms_per_step','config')}))
print(json.dumps(j['roofline'])[:600])
print(json.dumps(j['configs']))
for r in j.get('end_to_end_shmem',[]): print({k:r.get(k) for k in ('workers','worker_processes','busy_us','env_steps_per_s','frac_of_env_bound','worker_mode')})
print(list(j.keys())[-6:], len(json.dumps(j)))
P
cat gpurun_out/fs_box2.log; tail -3 gpurun_out/r5_bench_a.err
