# the transformer step under IB_TN256_SPLITS / IB_NO_TN256, same box (edit the list)
for v in "IB_TN256_SPLITS=4" "IB_TN256_SPLITS=5" "IB_TN256_SPLITS=4" "IB_TN256_SPLITS=5" "IB_TN256_SPLITS=4" "IB_TN256_SPLITS=5"; do
  echo "== $v"
  env $v python bench.py --workload transformer_denoiser_T50 --steps 400 --warmup 40 --no-cpu-baseline --no-ddim --no-transformer 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print(o['ms_per_step'], o['final_loss'])
"
done
