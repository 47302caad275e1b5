# same-box A/B of the MLP denoiser step under environment switches (edit the list)
for v in "" "IB_TN_TARGET=192" "IB_TN_TARGET=224" "IB_TN_TARGET=320" "IB_TN_TARGET=384" ""; do
  echo "== $v"
  env $v python bench.py --steps 600 --warmup 50 --no-cpu-baseline --no-ddim --no-transformer 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print(o['ms_per_step'], o['step_ms'], o['final_loss'])
for b in o['step_breakdown']: print('   ', b['entry'], b['avg_launch_us'])
print('   sum', o['step_sum_of_kernel_us'])
"
done
