cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
for i in 1 2; do
FSRL_NO_CPU=1 FSRL_ONLY=cpo python tools/bench_trust.py | cut -c1-70
FSRL_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libfsrl_hip_fb.so FSRL_NO_CPU=1 FSRL_ONLY=cpo python tools/bench_trust.py | cut -c1-70
done
FSRL_NO_CPU=1 FSRL_ONLY=trpo python tools/bench_trust.py | cut -c1-70
FSRL_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libfsrl_hip_fb.so FSRL_NO_CPU=1 FSRL_ONLY=trpo python tools/bench_trust.py | cut -c1-70
python tools/bench_sac.py --rows 200000 --updates 300 --no-cpu | cut -c1-200
FSRL_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libfsrl_hip_fb.so python tools/bench_sac.py --rows 200000 --updates 300 --no-cpu | cut -c1-200
