# transformer step, data-parallel FORM on a 1-rank RCCL group: captured collectives with / without the per-bucket optimizer,
# alternating, same box (IB_NO_BUCKET_OPT is a tuning switch: measurement build of the library)
export IB_HIP_LIB=$(cd "$(dirname "$0")/.." && pwd)/inferbiomechanics_amd/lib/ab/libib_hip_ab.so
run() { label="$1"; shift
  timeout -k 10 200 env "$@" python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-ddim --no-mlp --no-roofline --no-cli-path > gpurun_out/ddpab.json 2> gpurun_out/ddpab.err; rc=$?
  echo "$label rc=$rc $(python -c "import json;d=json.loads(open('gpurun_out/ddpab.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['final_loss'], d['config'].get('grad_buckets'), d['config'].get('bucket_bytes'))" 2>/dev/null)"
  if [ $rc -ne 0 ]; then tail -c 800 gpurun_out/ddpab.err; fi
}
run "single                 " X=1
for i in 1 2; do
  run "captured, bucket opt   " IB_DDP_SELFTEST=1 IB_GRAPH_COLLECTIVES=1
  run "captured, one opt      " IB_DDP_SELFTEST=1 IB_GRAPH_COLLECTIVES=1 IB_NO_BUCKET_OPT=1
done
run "cut graphs, bucket opt " IB_DDP_SELFTEST=1 IB_GRAPH_COLLECTIVES=0
run "default (probe)        " IB_DDP_SELFTEST=1
run "single                 " X=1
