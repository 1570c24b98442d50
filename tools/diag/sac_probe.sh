# probe build: the actors' forward launch of the SAC update returning early (1 at entry, 2 behind the prologue, 3 behind the forward pass)
cd /tmp && export TMPDIR=/tmp
export FSRL_HIP_LIB=$GRAFT_REPO_ROOT/fsrl_amd/libfsrl_hip_probe.so
for ph in 0 1 2 3; do
  rm -rf /tmp/prof_sacp
  FSRL_DBG_PHASE=$ph rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sacp -- python $GRAFT_REPO_ROOT/tools/bench_sac.py --no-cpu --updates 300 > /tmp/sacp.log 2>&1
  echo "== phase $ph"; python $GRAFT_REPO_ROOT/tools/trace_timeline.py /tmp/prof_sacp 9 | grep "sac_actor_tile_kernel<256, 16>"
done
