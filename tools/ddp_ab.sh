# Same-box comparison of one training step in its single-GPU form and its data-parallel forms on a 1-rank RCCL group
# (no wire time: what the FORM costs -- materialised gradients, graph cuts around the all-reduce, the plain optimizer).
# usage (GPU box, repo root): bash tools/ddp_ab.sh
run() { # label, env..., args
  label="$1"; shift
  timeout -k 10 200 env "$@" python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-ddim --no-mlp --no-roofline --no-cli-path --no-variant-child $ARGS > gpurun_out/ddpab.json 2> gpurun_out/ddpab.err; rc=$?
  echo "$label rc=$rc $(python -c "import json;d=json.loads(open('gpurun_out/ddpab.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['final_loss'], d['config'].get('grad_buckets'))" 2>/dev/null)"
  if [ $rc -ne 0 ]; then tail -c 800 gpurun_out/ddpab.err; fi
}
for ARGS in "--workload transformer_denoiser_T50" "--workload mlp_denoiser_T50"; do
  echo "## $ARGS"
  run "single            " X=1
  run "single, no opt fuse" IB_NO_OPT_FUSE=1
  run "ddp default (probe)" IB_DDP_SELFTEST=1
  run "ddp cut graphs    " IB_DDP_SELFTEST=1 IB_GRAPH_COLLECTIVES=0
  run "ddp captured coll " IB_DDP_SELFTEST=1 IB_GRAPH_COLLECTIVES=1
  run "ddp captured, one optimizer launch" IB_DDP_SELFTEST=1 IB_GRAPH_COLLECTIVES=1 IB_NO_BUCKET_OPT=1 IB_HIP_LIB=inferbiomechanics_amd/lib/ab/libib_hip_ab.so
done
