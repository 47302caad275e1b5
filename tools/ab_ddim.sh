# same-box A/B of the DDIM loop under environment switches (edit the list)
# environment A/B switches live in the measurement build of the library only
export IB_HIP_LIB=${IB_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/inferbiomechanics_amd/lib/ab/libib_hip_ab.so}
BS=${BS:-"128 256"}
echo "== default (frozen-weight fused launch beyond 8192 rows; a short last round of panels on the row-panel kernels)"; python tools/ddim_ab.py $BS
echo "== IB_NO_INFER_SPLIT=1 (every panel through the fused launch)"; IB_NO_INFER_SPLIT=1 python tools/ddim_ab.py $BS
echo "== IB_NO_INFER_CHAIN=1 (per-op plan)"; IB_NO_INFER_CHAIN=1 python tools/ddim_ab.py $BS
echo "== default"; python tools/ddim_ab.py $BS
echo "== IB_NO_INFER_SPLIT=1"; IB_NO_INFER_SPLIT=1 python tools/ddim_ab.py $BS
