"""Timeline of one block of the persistent S-stage kernel (csrc/sstage.hip): s_memtime stamps of every wave of every workgroup at the
phase boundaries (lmv_sstage_desc.timing).  Prints, per phase, the mean / max duration over all waves (cycles of the 100 MHz-class
s_memtime counter are shader clocks on gfx950), the two workgroups of image 0 wave by wave, and the launch time.
usage: python tools/sstage_timeline.py [block=5] [B=128] [nblocks=18] [C=384]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import torch
from lemevit_amd import ops

blk = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nblocks = int(sys.argv[3]) if len(sys.argv) > 3 else 18
dev = "cuda:0"
C = int(sys.argv[4]) if len(sys.argv) > 4 else 384
HID, NWV = 4 * C, C // 48
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*shape, s=1.0): return (torch.rand(*shape, generator=g) * 2 - 1) * s
blocks = []
for j in range(nblocks):
    d = {"attn.qkv.weight": rnd(3 * C, C, s=(3.0 / C) ** 0.5).bfloat16(), "attn.proj.weight": rnd(C, C, s=(3.0 / C) ** 0.5).bfloat16(),
         "mlp.0.weight": rnd(HID, C, s=(3.0 / C) ** 0.5).bfloat16(), "mlp.3.weight": rnd(C, HID, s=(3.0 / HID) ** 0.5).bfloat16(),
         "norm1.weight": 1 + rnd(C, s=0.2), "norm1.bias": rnd(C, s=0.1), "attn.qkv.bias": rnd(3 * C, s=0.1), "attn.proj.bias": rnd(C, s=0.1),
         "norm2.weight": 1 + rnd(C, s=0.2), "norm2.bias": rnd(C, s=0.1), "mlp.0.bias": rnd(HID, s=0.1), "mlp.3.bias": rnd(C, s=0.1),
         "pos_embed.weight": rnd(C, 9, s=0.3), "pos_embed.bias": rnd(C, s=0.1)}
    blocks.append({k: v.to(dev) for k, v in d.items()})
P = ops.sstage_pack(blocks, C // 32)
x = rnd(B, 196, C).bfloat16().to(dev); c = rnd(B, 16, C).bfloat16().to(dev)
for _ in range(3):
    ops.sstage_fwd(x, c, P, 14, 14, 1e-6)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    ops.sstage_fwd(x, c, P, 14, 14, 1e-6)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flop = B * nblocks * (212 * C * (3 * C + C + 2 * HID) * 2 + (C // 32) * (196 * 196 + 256) * 32 * 4)
print(f"B={B} nblocks={nblocks}: {ms:.3f} ms per stage  ({flop / ms * 1e-9:.0f} TFLOP/s incl. attention, {ms / nblocks * 1e3:.1f} us per block)")
nwg = 2 * ((B + 7) // 8) * 8
NS = 24
tm = torch.zeros(nwg * NWV * NS, dtype=torch.int64, device=dev)
ops.sstage_fwd(x, c, P, 14, 14, 1e-6, timing=tm, timing_block=blk)
torch.cuda.synchronize()
traw = tm.cpu().numpy().reshape(nwg, NWV, NS).astype(np.float64)
t = traw[:, :, :13]
names = ["dwconv", "norm1", "kv gemm", "kv drain+barrier", "q gemm", "kv wait+barrier", "attention", "attn barrier", "proj", "norm2", "mlp", "end barrier"]
d = np.diff(t, axis=2)
valid = t[:, 0, 0] > 0
d = d[valid]
tot = (t[valid][:, :, 12] - t[valid][:, :, 0])
print(f"block {blk}: {tot.mean():.0f} cycles per block (mean over waves), s_memtime ticks; x18 = {tot.mean() * 18:.0f}")
print(f"{'phase':18s} {'mean':>8s} {'max':>8s} {'min':>8s}   waves 0-3 / waves 4-7 mean")
for k, n in enumerate(names):
    print(f"{n:18s} {d[:, :, k].mean():8.0f} {d[:, :, k].max():8.0f} {d[:, :, k].min():8.0f}   {d[:, :NWV // 2, k].mean():8.0f} {d[:, NWV // 2:, k].mean():8.0f}")
vv = traw[valid]
print(f"inside dwconv: staging own rows {np.mean(vv[:, :, 13] - vv[:, :, 0]):.0f}, flag wait + halo copy {np.mean(vv[:, :, 14] - vv[:, :, 13]):.0f}, taps {np.mean(vv[:, :, 1] - vv[:, :, 14]):.0f}")
print(f"dwconv taps: first tile of channel tile 0 {np.mean(vv[:, :, 21] - vv[:, :, 14]):.0f}, second tile {np.mean(vv[:, :, 22] - vv[:, :, 21]):.0f}, rest of channel tile 0 {np.mean(vv[:, :, 23] - vv[:, :, 22]):.0f}, channel tiles 1 + 2 {np.mean(vv[:, :, 1] - vv[:, :, 23]):.0f}")
print("MLP chunk 2: " + ", ".join(f"{n} {np.mean(vv[:, :, b] - vv[:, :, a]):.0f}" for n, a, b in (("fc1 gemm", 15, 16), ("gelu + H write", 16, 17), ("barrier", 17, 18), ("fc2 gemm", 18, 19), ("barrier", 19, 20))))
for wg in (0, 8):
    print(f"workgroup {wg} (image 0, half {wg // 8 % 2}) per wave:")
    for w in range(NWV):
        print("  wave", w, " ".join(f"{v:6.0f}" for v in np.diff(t[wg, w])))
