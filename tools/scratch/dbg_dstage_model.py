import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import numpy as np, torch
from detfill import det_tensor, fill_state_dict
import lemevit_amd
import lemevit_amd.model as Mm
from lemevit_amd import ops
DEV = "cuda:0"
m = lemevit_amd.create_model("lemevit_base", num_classes=1000)
spec = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict(fill_state_dict(spec, 1)); m = m.to(DEV).eval()
img = det_tensor((2, 3, 224, 224), "model_base_224.img", 4).to(DEV)
outs = {}
orig_d, orig_s = ops.dstage_fwd, ops.sstage_fwd
cap = {}
def dwrap(x, c, P, H, W, eps, **kw):
    cap["in"] = (x.clone(), c.clone())
    o = orig_d(x, c, P, H, W, eps, **kw); cap["out"] = o; return o
ops.dstage_fwd = dwrap
with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
    Mm._DSTAGE = True; a = m(img).float()
    Mm._DSTAGE = False; b = m(img).float()
    print("logits dstage vs per-block:", float((a - b).abs().max()), "of", float(b.abs().max()))
    x, c = cap["in"]; xo, co = cap["out"]
    print("stage-2 input: x absmax", float(x.abs().max()), "c absmax", float(c.abs().max()))
    xr, cr = x, c
    for blk in m.stages[2]:
        xr, cr = blk.forward_tokens(xr, cr, 28, 28, masks=None)
        print("  per-block: x absmax", float(xr.float().abs().max()), "c absmax", float(cr.float().abs().max()))
    print("stage out x: err", float((xo.float() - xr.float()).abs().max()), "of", float(xr.float().abs().max()), " c: err", float((co.float() - cr.float()).abs().max()), "of", float(cr.float().abs().max()))
    for nb in (1, 2, 3):
        Pk = Mm._sstage_packed(m.stages[2][:nb], "D") if False else None
    # block by block: dstage with 1 block each, fed with the per-block schedule's inputs
    xr, cr = x, c
    for j, blk in enumerate(m.stages[2]):
        Pj = Mm._sstage_packed(torch.nn.ModuleList([blk]), "D")
        xd, cd_ = orig_d(xr.contiguous(), cr.contiguous(), Pj, 28, 28, Mm.BLOCK_LN_EPS)
        xr2, cr2 = blk.forward_tokens(xr, cr, 28, 28, masks=None)
        print(f"  block {j}: x err {float((xd.float() - xr2.float()).abs().max()):.4f} of {float(xr2.float().abs().max()):.3f}; c err {float((cd_.float() - cr2.float()).abs().max()):.4f} of {float(cr2.float().abs().max()):.3f}")
        xr, cr = xr2, cr2
