for v in "" "IB_TN256_SPLITS=4" "IB_TN256_SPLITS=8" "IB_TN256_SPLITS=2" ""; do
  echo "== $v"
  env $v python bench.py --workload transformer_denoiser_T50 --steps 300 --warmup 30 --no-cpu-baseline --no-ddim --no-transformer 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print(o['ms_per_step'], o['final_loss'])
for b in o['step_breakdown'][:1]: print('   ', b['entry'], b['avg_launch_us'])
for b in o['step_breakdown']:
    if 'optim' in b['entry']: print('   ', b['entry'], b['avg_launch_us'])
"
done
