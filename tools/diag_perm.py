import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lemevit_amd as L
torch.manual_seed(0)
dev = "cuda:0"
m = L.create_model("lemevit_base", num_classes=1000).to(dev).eval()
B = 128
x = torch.randn(B, 3, 224, 224, device=dev)
perm = torch.randperm(B, device=dev)
with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
    s = m.downsample_layers[0]
    a = s(x.contiguous(memory_format=torch.channels_last)); b = s(x[perm].contiguous(memory_format=torch.channels_last))
    print("stem equal:", torch.equal(a[perm], b), float((a[perm].float()-b.float()).abs().max()))
    xt, H, W = m._to_tokens(a, torch.bfloat16)
    c = m.meta_token_downsample[0](m.meta_tokens.unsqueeze(0)).expand(B, -1, -1).to(torch.bfloat16).contiguous()
    xp = xt[perm].contiguous()
    x1, c1 = xt, c
    x2, c2 = xp, c.clone()
    for i in range(5):
        if i > 0 and not isinstance(m.downsample_layers[i], torch.nn.Identity):
            d1 = m.downsample_layers[i](m._to_nchw(x1, H, W)); d2 = m.downsample_layers[i](m._to_nchw(x2, H, W))
            print(f"  downsample {i} equal:", torch.equal(d1[perm], d2))
            x1, H, W = m._to_tokens(d1, torch.bfloat16); x2, _, _ = m._to_tokens(d2, torch.bfloat16)
            x2 = x1[perm].contiguous()      # re-sync so only OUR kernels are tested below
        if i > 0:
            c1 = m.meta_token_downsample[i](c1).to(torch.bfloat16).contiguous(); c2 = c1[perm].contiguous()
        for j, blk in enumerate(m.stages[i]):
            x1, c1 = blk.forward_tokens(x1, c1, H, W); x2, c2 = blk.forward_tokens(x2, c2, H, W)
            ex, ec = torch.equal(x1[perm], x2), torch.equal(c1[perm], c2)
            if not (ex and ec):
                print(f"  stage {i} block {j}: x equal {ex} c equal {ec}  dx {float((x1[perm].float()-x2.float()).abs().max()):.3e} dc {float((c1[perm].float()-c2.float()).abs().max()):.3e}")
                x2 = x1[perm].contiguous(); c2 = c1[perm].contiguous()
    print("done")
