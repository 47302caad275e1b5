echo "== default"; python tools/ddim_ab.py 2 4 8 16 32
echo "== IB_NO_NT_SPLITK=1"; IB_NO_NT_SPLITK=1 python tools/ddim_ab.py 2 4 8 16 32
echo "== IB_LINLN_K512_MAX_M=4096"; IB_LINLN_K512_MAX_M=4096 python tools/ddim_ab.py 4 8 16
