cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_actor.py -q -x -k split 2>&1 | tail -2
python tools/bench_shmem.py --seconds 3
