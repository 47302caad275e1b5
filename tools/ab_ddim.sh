# same-box A/B of the DDIM loop under environment switches (edit the list)
echo "== default"; python tools/ddim_ab.py 1 4 8 16
echo "== IB_NO_PAD=1"; IB_NO_PAD=1 python tools/ddim_ab.py 1 4 8 16
echo "== default"; python tools/ddim_ab.py 1 4 8 16
