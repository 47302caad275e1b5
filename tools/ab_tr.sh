# A/B of the transformer T50 step with environment switches: tools/ab_tr.sh "" "IB_NO_LN_FAST=1" ...
# environment A/B switches live in the measurement build of the library only
export IB_HIP_LIB=${IB_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/inferbiomechanics_amd/lib/ab/libib_hip_ab.so}
for v in "$@"; do
  echo "== $v"
  env $v python bench.py --workload transformer_denoiser_T50 --steps 100 --warmup 10 --no-cpu-baseline --no-ddim --no-mlp 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print(o['ms_per_step'], o['step_ms'], o['final_loss'])
print('   sum', o['step_sum_of_kernel_us'])
"
done
