# probe build: the Q-network tile launches of the SAC update returning early (FSRL_QTILE_PROBE: 1 behind the prologue, 2 the forward pass,
# 3 the loss head, 4 the activation backward); prints the three fb_tile_kernel launches of an update per setting
cd /tmp && export TMPDIR=/tmp
export FSRL_HIP_LIB=$GRAFT_REPO_ROOT/fsrl_amd/libfsrl_hip_probe.so
for ph in 0 1 2 3 4; do
  rm -rf /tmp/prof_sacq
  FSRL_QTILE_PROBE=$ph rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sacq -- python $GRAFT_REPO_ROOT/tools/bench_sac.py --no-cpu --updates 300 > /tmp/sacq.log 2>&1
  echo "== phase $ph"; python $GRAFT_REPO_ROOT/tools/trace_timeline.py /tmp/prof_sacq 9 | grep "fb_tile_kernel"
done
