for v in "" "IB_NO_BRANCH=time_bwd" "IB_SKIP_TIME_BWD=1" "IB_CHAIN_V1=1"; do
  echo "== $v"
  env $v python bench.py --steps 600 --warmup 50 --no-cpu-baseline --no-ddim --no-transformer 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print(o['ms_per_step'], o['step_ms'], o['final_loss'])
for b in o['step_breakdown']: print('   ', b['entry'], b['avg_launch_us'])
print('   sum', o['step_sum_of_kernel_us'])
"
done
