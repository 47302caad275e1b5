# Stress of the captured-collectives form (tests/test_ddp_rccl_gpu.py's five-arm worker) under host + GPU load: the c10d
# watchdog race ("event last recorded in a capturing stream") showed only inside full test-suite runs.
# usage: bash tools/rccl_stress.sh [runs] ; env EVC=0 sets TORCH_NCCL_CUDA_EVENT_CACHE=0 for the test processes
RUNS=${1:-6}
pids=()
for i in $(seq 10); do    # host load: 10 busy loops
  timeout 900 python -c "
import time
t=time.time()
while time.time()-t<890: sum(i*i for i in range(100000))
" & pids+=($!)
done
# GPU load: one process looping a training leg
timeout 900 python bench.py --workload mlp_denoiser_T50 --steps 4000000 --warmup 10 --no-cpu-baseline --no-ddim --no-mlp --no-roofline --no-cli-path > /dev/null 2>&1 & pids+=($!)
sleep 20
fail=0
for i in $(seq $RUNS); do
  if [ -n "$EVC" ]; then export TORCH_NCCL_CUDA_EVENT_CACHE=$EVC; fi
  timeout -k 10 300 python -m pytest tests/test_ddp_rccl_gpu.py -q -x -p no:cacheprovider -k lagged > gpurun_out/stress_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc captured-event errors: $(grep -c 'capturing stream' gpurun_out/stress_$i.log) | $(tail -1 gpurun_out/stress_$i.log)"
  [ $rc -ne 0 ] && fail=$((fail+1))
done
echo "failures: $fail / $RUNS (TORCH_NCCL_CUDA_EVENT_CACHE=${EVC:-default})"
kill "${pids[@]}" 2>/dev/null
wait 2>/dev/null
true
