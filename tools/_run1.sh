cd $GRAFT_REPO_ROOT
bash tools/ab_ppo.sh $GRAFT_REPO_ROOT/ab_libs/libfsrl_hip_mv.so
