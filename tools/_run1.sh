cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -2 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
python -c "
import json;d=json.loads(open('gpurun_out/r03_bench.json').read().strip().splitlines()[-1]);print(d['value'],d['roofline']['frac'],d['roofline']['step_us'],d['roofline']['frac_of_latency_floor'],[round(e['env_steps_per_s']) for e in d['end_to_end_shmem']])"
