import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemevit_amd import ops
torch.manual_seed(0)
dev = "cuda:0"; bf = torch.bfloat16
B, L, C = 1, 16, 32
qkv = torch.randn(B, L, 3 * C, device=dev).to(bf)
q, k, v = qkv[..., :C].float(), qkv[..., C:2*C].float(), qkv[..., 2*C:].float()
s = (q @ k.transpose(-1, -2)) * (32 ** -0.5)
ref = s.softmax(-1) @ v
o, lse = ops.attn_fwd((qkv, 0), (qkv, C), (qkv, 2 * C), C, 32 ** -0.5, want_lse=True)
print("fwd max err", float((o.float() - ref).abs().max()), "ref max", float(ref.abs().max()))
print("lse err", float((lse[0, 0] - torch.logsumexp(s[0], -1)).abs().max()))
print("o[0,:4,:8]\n", o[0, :4, :8].float()); print("ref\n", ref[0, :4, :8])
# is o a permutation of ref rows/cols?
print("row-sum err", float((o.float().sum(-1) - ref.sum(-1)).abs().max()), "col-sum err", float((o.float().sum(1) - ref.sum(1)).abs().max()))
