cd $GRAFT_REPO_ROOT
python tools/bench_group.py --ks 4 8 --updates 5 | cut -c1-110
python tools/bench_group.py --ks 4 8 --updates 5 --xcd | cut -c1-110
python tools/bench_group.py --ks 4 --updates 5 | cut -c1-110
python tools/bench_group.py --ks 4 --updates 5 --xcd | cut -c1-110
