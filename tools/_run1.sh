cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ppo.py -q -x -k "full_size" -s 2>&1 | grep -E "theta vs|yardstick|passed|failed|Error|assert" | head
