"""Print the FlatGradSync chunk bounds of a model (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lemevit_amd
from lemevit_amd.dist import attach_flat_grad_sync
m = lemevit_amd.create_model(sys.argv[1] if len(sys.argv) > 1 else "lemevit_base").cuda()
opt = lemevit_amd.FlatAdamW(m, lr=1e-4)
s = attach_flat_grad_sync(m, opt)
tot = opt._flat_g.numel()
print("flat elements", tot, "chunks (forward order):", [(a, b, round((b - a) / tot, 3)) for a, b in s.bounds], "rest params", sum(p.numel() for p in s.rest))
