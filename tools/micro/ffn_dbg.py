import sys
sys.path.insert(0, ".")
import torch
from tests.test_ffn_chain_gpu import problem, restate_fwd, rb, BF
from inferbiomechanics_amd import hip
M, ffn, d = 777, 1024, 512
pr = problem(M, ffn, seed=M + ffn)
ex = restate_fwd(pr)
dev = {k: v.to("cuda") for k, v in pr.items()}
packed = torch.zeros(hip.ffn_chain_packed_elems(d, ffn), dtype=BF, device="cuda")
hip.ffn_chain_pack([(dev["w1"], dev["w2"], packed)])
f1 = torch.zeros((M, ffn), dtype=BF, device="cuda"); s2 = torch.zeros((M, d), dtype=BF, device="cuda"); y = torch.zeros((M, d), dtype=BF, device="cuda")
mean, rstd = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
mask = torch.zeros(hip.ffn_chain_mask_bytes(M, d, ffn), dtype=torch.uint8, device="cuda")
hip.ffn_chain_fwd(dev["x1"], packed, dev["b1"], dev["b2"], dev["gamma"], dev["beta"], f1, s2, y, mean, rstd, mask)
torch.cuda.synchronize()
gf1 = f1.cpu().double()
print("f1 mismatches:", int((gf1 != ex["f1"]).sum()), "of", gf1.numel(), "max", float((gf1 - ex["f1"]).abs().max()))
# s2 from the KERNEL's f1
s2k = rb(pr["x1"].double() + gf1 @ pr["w2"].double().t() + pr["b2"].double())
gs2 = s2.cpu().double()
dif = (gs2 - s2k).abs()
print("s2 vs restatement from kernel f1: mismatches", int((dif > 0).sum()), "max", float(dif.max()))
bad = (gs2 - ex["s2"]).abs() > 2 * 2.0 ** -8 * ex["s2"].abs().clamp_min(ex["s2"].abs().max() * 2.0 ** -6)
idx = bad.nonzero()
print("bad vs full restatement:", idx.shape[0])
for r, c in idx[:10].tolist():
    print(r, c, "got", float(gs2[r, c]), "want", float(ex["s2"][r, c]), "from kernel f1", float(s2k[r, c]),
          "exact", float((pr["x1"].double() + ex["f1"] @ pr["w2"].double().t() + pr["b2"].double())[r, c]))
