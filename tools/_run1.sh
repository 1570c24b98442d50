cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_trust.py tests/test_gpu_group.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -8
python -m pytest tests/test_gpu_loop.py -q -x -k "cpo or trpo" 2>&1 | tail -3
FSRL_NO_CPU=1 FSRL_ONLY=cpo python tools/bench_trust.py | cut -c1-70
FSRL_NO_CPU=1 FSRL_ONLY=trpo python tools/bench_trust.py | cut -c1-70
