set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tiles -- python $R/tools/_r5_tiles.py > /tmp/tiles.log 2>&1
python $R/tools/trace_by_grid.py /tmp/tiles fb_hvp fb_tile > $R/gpurun_out/tiles_by_grid.txt 2>&1
cat $R/gpurun_out/tiles_by_grid.txt; tail -3 /tmp/tiles.log
