"""Timeline of one block of the persistent D-stage kernel (csrc/dstage.hip): s_memtime stamps of every wave of every workgroup at the phase
boundaries (lmv_dstage_desc.timing), image workgroups and meta workgroups apart, plus the launch time next to the per-launch schedule.
usage: python tools/dstage_timeline.py [block=1] [B=128] [nblocks=4] [C=192 | 96 | 128 | 64 | 384 = S blocks at 24 x 24 (kind 2)]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lemevit_amd import ops
import lemevit_amd.model as Mm
from lemevit_amd.blocks import PARAM_NAMES

blk = int(sys.argv[1]) if len(sys.argv) > 1 else 1
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
nblocks = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = "cuda:0"
C = int(sys.argv[4]) if len(sys.argv) > 4 else 192
G, NWV, KWG, MAXSLOTS = (28, 4, 7, 64) if C in (192, 128) else (24, 8, 6, 32) if C == 384 else (56, 2, 28, 32)
KIND = 2 if C == 384 else 0
HID = 4 * C
g = torch.Generator(device="cpu").manual_seed(0)
def rnd(*shape, s=1.0): return (torch.rand(*shape, generator=g) * 2 - 1) * s
blocks = []
for j in range(nblocks):
    d = {"attn.qkv1.weight": rnd(3 * C, C, s=(3.0 / C) ** 0.5).bfloat16(), "attn.qkv2.weight": rnd(3 * C, C, s=(3.0 / C) ** 0.5).bfloat16(),
         "attn.proj_x.weight": rnd(C, C, s=(3.0 / C) ** 0.5).bfloat16(), "attn.proj_c.weight": rnd(C, C, s=(3.0 / C) ** 0.5).bfloat16(),
         "mlp.0.weight": rnd(HID, C, s=(3.0 / C) ** 0.5).bfloat16(), "mlp.3.weight": rnd(C, HID, s=(3.0 / HID) ** 0.5).bfloat16(),
         "norm1.weight": 1 + rnd(C, s=0.2), "norm1.bias": rnd(C, s=0.1), "attn.qkv1.bias": rnd(3 * C, s=0.1), "attn.qkv2.bias": rnd(3 * C, s=0.1),
         "attn.proj_x.bias": rnd(C, s=0.1), "attn.proj_c.bias": rnd(C, s=0.1),
         "norm2.weight": 1 + rnd(C, s=0.2), "norm2.bias": rnd(C, s=0.1), "mlp.0.bias": rnd(HID, s=0.1), "mlp.3.bias": rnd(C, s=0.1),
         "pos_embed.weight": rnd(C, 9, s=0.3), "pos_embed.bias": rnd(C, s=0.1)}
    blocks.append({k: v.to(dev) for k, v in d.items()})
if KIND == 2:
    P = ops.s2stage_pack([{"attn.qkv.weight": b["attn.qkv1.weight"], "attn.qkv.bias": b["attn.qkv1.bias"], "attn.proj.weight": b["attn.proj_x.weight"], "attn.proj.bias": b["attn.proj_x.bias"],
                           **{k: v for k, v in b.items() if not k.startswith("attn.")}} for b in blocks], C // 32)
else:
    P = ops.dstage_pack(blocks, C // 32)
x = rnd(B, G * G, C).bfloat16().to(dev); c = rnd(B, 16, C).bfloat16().to(dev)
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = timed(lambda: ops.dstage_fwd(x, c, P, G, G, 1e-6, kind=KIND))
flop = B * nblocks * (G * G * C * (2 * C + C + 2 * HID) * 2 + 16 * C * (4 * C + 2 * HID) * 2 + (C // 32) * G * G * 16 * 32 * 8)
print(f"B={B} nblocks={nblocks}: {ms:.3f} ms per stage  ({flop / ms * 1e-9:.0f} TFLOP/s, {ms / nblocks * 1e3:.1f} us per block)")
def per_launch():
    xr, cr = x, c
    with torch.no_grad():
        for bd in blocks:
            if KIND == 2:
                bs = {"attn.qkv.weight": bd["attn.qkv1.weight"], "attn.qkv.bias": bd["attn.qkv1.bias"], "attn.proj.weight": bd["attn.proj_x.weight"], "attn.proj.bias": bd["attn.proj_x.bias"], **bd}
                params = {n: (bs[n].reshape(C, 1, 3, 3) if n == "pos_embed.weight" else bs[n]) for n in PARAM_NAMES["S"]}
                xr, cr = Mm.run_block("S", xr, cr, G, G, params, (None,) * 4)
                continue
            params = {n: (bd[n].reshape(C, 1, 3, 3) if n == "pos_embed.weight" else bd[n]) for n in PARAM_NAMES["D"]}
            xr, cr = Mm.run_block("D", xr, cr, G, G, params, (None,) * 4)
    return xr, cr
try:
    print(f"per-launch schedule (lmv_block_fwd x {nblocks}): {timed(per_launch):.3f} ms")
except Exception as e:      # (the A/B line is a convenience, the timeline below is the tool)
    print("per-launch schedule not timed:", repr(e)[:200])
nslots = min(MAXSLOTS, (B + 7) // 8 * 8)
nwg = nslots * (KWG + 1)
NS = 16
tm = torch.zeros(nwg * NWV * NS, dtype=torch.int64, device=dev)
ops.dstage_fwd(x, c, P, G, G, 1e-6, timing=tm, timing_block=blk, kind=KIND)
torch.cuda.synchronize()
traw = tm.cpu().numpy().reshape(nwg, NWV, NS).astype(np.float64)
role = (np.arange(nwg) // 8) % (KWG + 1)
img, meta = traw[role < KWG][:, :, :8], traw[role == KWG][:, :, 8:]
img, meta = img[img[:, 0, 0] > 0], meta[meta[:, 0, 0] > 0]
print(f"block {blk} (global counter): image workgroups {np.mean(img[:, :, 7] - img[:, :, 0]):.0f} cycles, meta workgroups {np.mean(meta[:, :, 7] - meta[:, :, 0]):.0f} (s_memtime ticks)")
for nm, t, names in (("image workgroup", img, (["halo wait + dwconv", "norm1", "k / v -> L2", "wait + q + attention", "proj + norm2", "mlp", "bias + halo publish"] if KIND == 2 else ["halo wait + dwconv", "norm1 + meta wait", "c-direction (scores, v1, P V)", "x-direction (q1, attention)", "proj_x + norm2", "mlp", "bias + halo publish"])),
                     ("meta workgroup", meta, ["norm1", "k2 / v2 / q2", "q~ + publish", "wait for partials", "combine", "proj_c + norm2", "mlp"])):
    d = np.diff(t, axis=2)
    print(nm)
    for k, n in enumerate(names):
        print(f"  {n:32s} mean {d[:, :, k].mean():8.0f}  max {d[:, :, k].max():8.0f}  min {d[:, :, k].min():8.0f}")
print("image workgroup 0 per wave:")
for w in range(NWV):
    print("  wave", w, " ".join(f"{v:6.0f}" for v in np.diff(traw[0, w, :8])))
# per role (row group of the image): is one of them systematically late?
tr = traw[role < KWG][:, :, :8]
rr = role[role < KWG]
ok = tr[:, 0, 0] > 0
print("per role: mean start skew vs the image's first workgroup, dwconv phase, whole block")
slot_of = (np.arange(nwg) // 8 // (KWG + 1)) * 8 + (np.arange(nwg) % 8)
so = slot_of[role < KWG]
t0 = tr[:, 0, 0]
for r in range(KWG):
    sel = ok & (rr == r)
    if not sel.any(): continue
    base = np.array([t0[ok & (so == s)].min() for s in so[sel]])
    print(f"  role {r:2d}: start +{np.mean(t0[sel] - base):7.0f}   dwconv {np.mean(tr[sel][:, 0, 1] - tr[sel][:, 0, 0]):7.0f}   block {np.mean(tr[sel][:, 0, 7] - tr[sel][:, 0, 0]):7.0f}")
# placement: HW_ID of wave 0 of every image workgroup (cu_id bits 11:8, sh_id 12, se_id 15:13 on gfx9; XCC_ID low bits)
hw = traw[:, 0, 15].astype(np.int64)
if (hw != 0).any():
    cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (((hw >> 32) & 15) << 8)
    print("placement of the first workgroups on XCD 0 (workgroup index // 8, role, cu key):", [(int(j // 8), int(role[j]), int(cu[j])) for j in range(0, 8 * 40, 8) if hw[j] != 0][:40])
    import collections
    per_cu = collections.defaultdict(list)
    for j in range(nwg):
        if hw[j] != 0: per_cu[int(cu[j])].append((int(slot_of[j]), int(role[j])))
    sizes = collections.Counter(len(v) for v in per_cu.values())
    print("workgroups per CU histogram:", dict(sizes), " CUs used:", len(per_cu))
    print("examples:", list(per_cu.items())[:6])
