# the transformer step under environment switches, same box (edit the list)
for v in "" "IB_NO_TRAIN_PAD=1" "" "IB_NO_TRAIN_PAD=1"; do
  echo "== $v"
  env $v python bench.py --workload transformer_denoiser_T50 --steps 400 --warmup 40 --no-cpu-baseline --no-ddim --no-transformer 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print(o['ms_per_step'], o['final_loss'])
"
done
