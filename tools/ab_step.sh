# same-box A/B of the MLP denoiser step under environment switches / flags (edit the list)
# environment A/B switches live in the measurement build of the library only
export IB_HIP_LIB=${IB_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/inferbiomechanics_amd/lib/ab/libib_hip_ab.so}
for v in "--batches 64" "--batches 1" "--batches 64" "--batches 1"; do
  echo "== $v"
  python bench.py --workload mlp_denoiser_T50 --steps 1000 --warmup 100 --no-cpu-baseline --no-ddim --no-cli-path $v 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print(o['ms_per_step'], o['step_ms'], o['final_loss'])
"
done
