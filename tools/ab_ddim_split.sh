# sampler row split (plans.DenoiserTransformerPlan.side_windows): threshold sweep, same box
export IB_HIP_LIB=${IB_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/inferbiomechanics_amd/lib/ab/libib_hip_ab.so}
BS=${BS:-"84 90 100 110 120 256"}
echo "== no split"; IB_NO_INFER_SPLIT=1 python tools/ddim_ab.py $BS
echo "== split, last round <= 96 panels (default)"; python tools/ddim_ab.py $BS
echo "== split, last round <= 128 panels"; IB_INFER_SPLIT_MAX_REM=128 python tools/ddim_ab.py $BS
