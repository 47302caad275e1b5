for v in "" opt1 opt2 opt3; do
  echo "== $v"
  L=""; [ -n "$v" ] && L="IB_HIP_LIB=$PWD/inferbiomechanics_amd/lib/ab/libib_hip_$v.so"
  env $L python bench.py --steps 600 --warmup 50 --no-cpu-baseline --no-ddim --no-transformer 2>/dev/null | python -c "
import json,sys
o=json.loads(sys.stdin.read())
print(o['ms_per_step'], o['step_ms'], o['final_loss'])
for b in o['step_breakdown']: print('   ', b['entry'], b['avg_launch_us'])
"
done
