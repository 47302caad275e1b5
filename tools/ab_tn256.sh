# the transformer step in data-parallel form (1-rank RCCL) under environment switches, same box (edit the list)
# environment A/B switches live in the measurement build of the library only
export IB_HIP_LIB=${IB_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/inferbiomechanics_amd/lib/ab/libib_hip_ab.so}
for v in "" "IB_NO_LAG_GROUP=1" "" "IB_NO_LAG_GROUP=1"; do
  echo "== $v"
  env $v IB_DDP_SELFTEST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 python bench.py --workload transformer_denoiser_T50 --steps 300 --warmup 30 --no-cpu-baseline --no-ddim --no-mlp 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print(o['ms_per_step'], o['final_loss'])
"
done
