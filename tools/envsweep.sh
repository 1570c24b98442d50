# HIP runtime knob sweep over the bench (round 1 result: none helps; AMD_OPT_FLUSH=0, i.e. system-scope fences, costs 15 %;
# ROC_SYSTEM_SCOPE_SIGNAL=0 stalls the run and is left out)
for kv in "X=1" "AMD_OPT_FLUSH=0" "AMD_OPT_FLUSH=1" "AMD_DIRECT_DISPATCH=0" "ROC_USE_FGS_KERNARG=0" "ROC_SKIP_KERNEL_ARG_COPY=1" "DEBUG_HIP_KERNARG_COPY_OPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=100" "HSA_ENABLE_SDMA=0"; do
  v=$(env $kv timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['roofline']['avg_launch_us'],2))" 2>/dev/null)
  echo "$kv -> $v"
done
