cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_actor.py tests/test_shmem_env.py tests/test_abi.py -q -x 2>&1 | tail -4
python tools/bench_shmem.py --seconds 3 | cut -c1-60,160-400
echo no-split; python tools/bench_shmem.py --seconds 3 --no-split | cut -c1-60,160-400
