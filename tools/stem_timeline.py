"""Timeline of one tile of the fused stem kernel (csrc/stem.hip): s_memtime stamps of every wave at the phase boundaries of each workgroup's second tile.
usage: python tools/stem_timeline.py [Cm=48] [Co=96] [B=128]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lemevit_amd import ops, _lib
Cm = int(sys.argv[1]) if len(sys.argv) > 1 else 48
Co = int(sys.argv[2]) if len(sys.argv) > 2 else 96
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = "cuda:0"
x = torch.randn(B, 3, 224, 224, device=dev)
w1m = torch.zeros(Cm, 32, device=dev, dtype=torch.bfloat16); w1m[:, :27] = (torch.randn(Cm, 27, device=dev) * 0.2).bfloat16()
w2m = (torch.randn(Co, 9 * Cm, device=dev) * 0.05).bfloat16()
wpk = ops.stem_pack(w1m, w2m)
b1, b2 = torch.randn(Cm, device=dev) * 0.1, torch.randn(Co, device=dev) * 0.1
for _ in range(3): ops.stem_fwd(x, wpk, b1, b2, Cm, Co)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.stem_fwd(x, wpk, b1, b2, Cm, Co)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
mb = (x.numel() * 4 + B * 56 * 56 * Co * 2) / 1e6
print(f"stem {Cm}->{Co} B={B}: {ms * 1e3:.1f} us  ({mb / ms * 1e-3:.2f} TB/s of input + output)")
tm = torch.zeros(512 * 4 * 8, dtype=torch.int64, device=dev)
f = _lib.lib.lmv_stem_debug_timing
f(tm.data_ptr()); ops.stem_fwd(x, wpk, b1, b2, Cm, Co); torch.cuda.synchronize(); f(None)
t = tm.cpu().numpy().reshape(512, 4, 8).astype(np.float64)
t = t[t[:, 0, 0] > 0]
d = np.diff(t[:, :, :7], axis=2)
for k, n in enumerate(["patch -> LDS + barrier", "issue the next tile's loads", "conv1 + GELU -> T1", "barrier", "conv2 + stores", "barrier"]):
    print(f"  {n:30s} mean {d[:, :, k].mean():8.0f}  max {d[:, :, k].max():8.0f}  min {d[:, :, k].min():8.0f}")
print(f"  tile total {np.mean(t[:, :, 6] - t[:, :, 0]):.0f} cycles; workgroups stamped {t.shape[0]}")
