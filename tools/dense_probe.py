"""Dense-prediction backbone at detection / segmentation resolutions: forward time and sanity (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lemevit_amd
cfgs = {"tiny": dict(depth=[1, 2, 2, 8, 2], embed_dim=[64, 64, 128, 192, 320]), "base": dict(depth=[2, 4, 4, 18, 4], embed_dim=[96, 96, 192, 384, 512])}
for name, res, B in [("tiny", 512, 2), ("base", 512, 2), ("base", 1024, 1)]:
    m = lemevit_amd.LeMeViTBackbone(head_dim=32, queries_len=16, attn_type=["C", "D", "D", "S", "S"], **cfgs[name]).cuda().eval()
    x = torch.randn(B, 3, res, res, device="cuda")
    with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
        for _ in range(2):
            outs = m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            outs = m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name} {res}x{res} B={B}: {dt * 1e3:8.2f} ms/forward  outs {[tuple(o.shape) for o in outs]} finite {all(bool(torch.isfinite(o).all()) for o in outs)}", flush=True)
