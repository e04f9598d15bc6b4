import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops
dev="cuda:0"; bf=torch.bfloat16
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/iters*1e3
for L,C in [(196,384),(49,512)]:
    B=128
    qx=torch.randn(B,L,3*C,device=dev).to(bf); qc=torch.randn(B,16,3*C,device=dev).to(bf)
    (ox,oc),(lx,lc)=ops.attn_fwd_pair([qx,qc],C,ops.SDPA_SCALE,True)
    ox2,lx2=ops.attn_fwd((qx,0),(qx,C),(qx,2*C),C,ops.SDPA_SCALE,True); oc2,lc2=ops.attn_fwd((qc,0),(qc,C),(qc,2*C),C,ops.SDPA_SCALE,True)
    print("fwd diff", float((ox.float()-ox2.float()).abs().max()), float((oc.float()-oc2.float()).abs().max()))
    dx=torch.randn_like(ox); dc=torch.randn_like(oc); gx=torch.empty_like(qx); gc=torch.empty_like(qc); gx2=torch.empty_like(qx); gc2=torch.empty_like(qc)
    ops.attn_bwd_pair([qx,qc],[ox,oc],[lx,lc],[dx,dc],[gx,gc],C,ops.SDPA_SCALE)
    ops.attn_bwd((qx,0),(qx,C),(qx,2*C),ox,lx,dx,(gx2,0),(gx2,C),(gx2,2*C),C,ops.SDPA_SCALE); ops.attn_bwd((qc,0),(qc,C),(qc,2*C),oc,lc,dc,(gc2,0),(gc2,C),(gc2,2*C),C,ops.SDPA_SCALE)
    print("bwd diff", float((gx.float()-gx2.float()).abs().max()), float((gc.float()-gc2.float()).abs().max()))
    t1=timeit(lambda: ops.attn_fwd_pair([qx,qc],C,ops.SDPA_SCALE,True))
    t2=timeit(lambda: (ops.attn_fwd((qx,0),(qx,C),(qx,2*C),C,ops.SDPA_SCALE,True), ops.attn_fwd((qc,0),(qc,C),(qc,2*C),C,ops.SDPA_SCALE,True)))
    t3=timeit(lambda: ops.attn_bwd_pair([qx,qc],[ox,oc],[lx,lc],[dx,dc],[gx,gc],C,ops.SDPA_SCALE))
    t4=timeit(lambda: (ops.attn_bwd((qx,0),(qx,C),(qx,2*C),ox,lx,dx,(gx2,0),(gx2,C),(gx2,2*C),C,ops.SDPA_SCALE), ops.attn_bwd((qc,0),(qc,C),(qc,2*C),oc,lc,dc,(gc2,0),(gc2,C),(gc2,2*C),C,ops.SDPA_SCALE)))
    print(f"L={L}: fwd pair {t1:.1f} us vs separate {t2:.1f} | bwd pair {t3:.1f} vs separate {t4:.1f}")
