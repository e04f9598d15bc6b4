import torch
p = torch.nn.Parameter(torch.randn(1024, device="cuda"))
for kw in (dict(fused=True), dict(fused=True, capturable=True), dict(foreach=True), dict()):
    opt = torch.optim.AdamW([p], lr=1e-3, **kw)
    p.grad = torch.randn_like(p)
    v0 = p._version; d0 = p.detach().clone()
    opt.step()
    print(kw, "version", v0, "->", p._version, "changed", bool((p.detach() != d0).any()))
