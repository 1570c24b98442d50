cd $GRAFT_REPO_ROOT
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -c processor /proc/cpuinfo; grep "model name" /proc/cpuinfo | head -1
for i in 1 2 3; do python tools/bench_shmem.py --seconds 2 | cut -c1-40,160-330 | tail -1; done
cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
