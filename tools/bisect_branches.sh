for off in "time_fwd" "loss" "time_inner" "time_fwd,loss" "time_fwd,time_inner" "loss,time_inner" "time_fwd,loss,time_inner"; do
  IB_NO_BRANCH=$off timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-ddim > gpurun_out/bis.json 2> gpurun_out/bis.err; rc=$?
  echo "off=[$off] rc=$rc $(python -c "import json;d=json.loads(open('gpurun_out/bis.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])" 2>/dev/null)"
done
