# same-box A/B of the DDIM loop under environment switches (edit the list)
# environment A/B switches live in the measurement build of the library only
export IB_HIP_LIB=${IB_HIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/inferbiomechanics_amd/lib/ab/libib_hip_ab.so}
echo "== default"; python tools/ddim_ab.py 1 4 8 16
echo "== IB_NO_PAD=1"; IB_NO_PAD=1 python tools/ddim_ab.py 1 4 8 16
echo "== default"; python tools/ddim_ab.py 1 4 8 16
