"""How long the time-MLP backward's rider workgroups take inside the grouped weight-gradient launch: the combined launch with
ONE tiny GEMM problem (its single work item ends in a few us, the rest of the launch is the riders) against the standalone
ib_time_mlp_bwd.  Usage (GPU box): python tools/rider_prof.py"""
import sys
import torch
sys.path.insert(0, ".")
from inferbiomechanics_amd import hip
from tools.kbench import timeit

dev, bf = "cuda", torch.bfloat16
B, temb, hid, out = 256, 128, 512, 1024
g = torch.Generator().manual_seed(0)
mk = lambda *s: torch.randn(*s, generator=g).to(dev, bf)
de, w2, zu, s = mk(B, out), mk(out, hid), mk(B, hid), mk(B, temb)
n = hip.time_mlp_bwd_slab_count(B)
sw, sb = torch.zeros(n, hid, temb, device=dev), torch.zeros(n, hid, device=dev)
dz, x = mk(256, 256), mk(256, 128)
ws = torch.zeros(int(hip.lib().ib_linear_wgrad_slabs_workspace(256, 256, 128)), dtype=torch.uint8, device=dev)
dz2, x2 = mk(256, 256), mk(256, 128)
ws2 = torch.zeros_like(ws)
probs = [(dz, x, ws), (dz2, x2, ws2)]
assert hip.linear_wgrad_slabs_multi(probs, time_bwd=(de, w2, zu, s, sw, sb)) is not None
print("combined launch, tiny GEMM part: %.2f us" % timeit(lambda: hip.linear_wgrad_slabs_multi(probs, time_bwd=(de, w2, zu, s, sw, sb)), 20))
print("tiny GEMM part alone:            %.2f us" % timeit(lambda: hip.linear_wgrad_slabs_multi(probs), 20))
print("standalone ib_time_mlp_bwd:      %.2f us" % timeit(lambda: hip.time_mlp_bwd(de, w2, zu, s, sw, sb), 20))
