export IB_DDP_SELFTEST=1
for extra in "--no-graph" "" "--overlap-comm on --bucket-mb 0.5" "--workload transformer_denoiser_T50 --steps 20 --warmup 3" "--workload transformer_denoiser_T50 --steps 20 --warmup 3 --overlap-comm off"; do
  timeout -k 10 200 python -X faulthandler bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-ddim $extra > gpurun_out/ddp.json 2> gpurun_out/ddp.err; rc=$?
  echo "selftest [$extra] rc=$rc $(python -c "import json;d=json.loads(open('gpurun_out/ddp.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['final_loss'], d['config']['grad_buckets'])" 2>/dev/null)"
  if [ $rc -ne 0 ]; then tail -c 1500 gpurun_out/ddp.err; fi
done
unset IB_DDP_SELFTEST
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 60 --warmup 5 --no-cpu-baseline --no-ddim > gpurun_out/ddp_tr.json 2> gpurun_out/ddp_tr.err; echo "torchrun rc=$? $(tail -c 300 gpurun_out/ddp_tr.json | cut -c1-200)"
