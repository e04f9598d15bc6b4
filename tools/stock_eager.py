#!/usr/bin/env python3
"""The "un-tuned GPU" column of SURVEY section 8(d): LeMeViT on stock PyTorch-ROCm library ops (hipBLASLt Linear, MIOpen
convolutions / BatchNorm, F.scaled_dot_product_attention, F.layer_norm, F.gelu) on one MI355X, timed with benchmark.py's
protocol (one synthetic batch, 10 warm-up + 40 timed steps, benchmark.py:120-130,462-467,517-518,572-596).

It consumes the state_dict of ``lemevit_amd.create_model`` (identical keys to the reference) through a small functional
module written for this tool -- torch ops only, channels-last, bf16 autocast -- so the speed-up of the hand-written HIP
kernels is separable from the speed-up of the GPU.  Not a product path; nothing under lemevit_amd/ imports it.

    python tools/stock_eager.py [--model lemevit_base] [--batch 128] [--img 224] [--out gpurun_out/stock_eager.json]
"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F


class StockBlock(nn.Module):
    """One LeMeBlock on library ops; x is NCHW (channels_last memory), c is [B, M, C]."""

    def __init__(self, kind, dim, heads, drop_path):
        super().__init__()
        self.kind, self.h, self.dp = kind, heads, drop_path
        self.pos_embed = nn.Conv2d(dim, dim, 3, padding=1, groups=dim)
        self.norm1, self.norm2 = nn.LayerNorm(dim, eps=1e-6), nn.LayerNorm(dim, eps=1e-6)
        self.mlp = nn.Sequential(nn.Linear(dim, 4 * dim), nn.GELU(), nn.Identity(), nn.Linear(4 * dim, dim))
        a = nn.Module()
        if kind == "C":
            a.q, a.kv, a.proj = nn.Linear(dim, dim), nn.Linear(dim, 2 * dim), nn.Linear(dim, dim)
        elif kind == "D":
            a.qkv1, a.qkv2, a.proj_x, a.proj_c = nn.Linear(dim, 3 * dim), nn.Linear(dim, 3 * dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        else:
            a.qkv, a.proj = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim)
        self.attn = a

    def _dpath(self, t):
        if not self.training or self.dp == 0.0:
            return t
        keep = 1.0 - self.dp
        return t * (torch.rand((t.shape[0],) + (1,) * (t.dim() - 1), device=t.device) < keep).to(t.dtype) / keep

    def _heads(self, t):
        B, L, C = t.shape
        return t.view(B, L, self.h, C // self.h).transpose(1, 2)

    def _merge(self, t):
        B, h, L, d = t.shape
        return t.transpose(1, 2).reshape(B, L, h * d)

    def _sa(self, t):
        q, k, v = self.attn.qkv(t).chunk(3, dim=-1)
        return self.attn.proj(self._merge(F.scaled_dot_product_attention(self._heads(q), self._heads(k), self._heads(v))))

    def forward(self, x, c):
        B, C, H, W = x.shape
        xp = x + self.pos_embed(x)
        t = xp.flatten(2).transpose(1, 2)
        if self.kind == "C":
            tn, cn = self.norm1(t), self.norm1(c)
            k, v = self.attn.kv(tn).chunk(2, dim=-1)
            o = F.scaled_dot_product_attention(self._heads(self.attn.q(cn)), self._heads(k), self._heads(v))
            c = c + self._dpath(self.attn.proj(self._merge(o)))
            c = c + self._dpath(self.mlp(self.norm2(c)))
            return x, c
        if self.kind == "D":
            N, M = t.shape[1], c.shape[1]
            sx, sc = math.log(M) / math.log(N) * C ** -0.5, C ** -0.5
            q1, k1, v1 = self.attn.qkv1(self.norm1(t)).chunk(3, dim=-1)
            q2, k2, v2 = self.attn.qkv2(self.norm1(c)).chunk(3, dim=-1)
            ox = F.scaled_dot_product_attention(self._heads(q1), self._heads(k2), self._heads(v2), scale=sx)
            oc = F.scaled_dot_product_attention(self._heads(q2), self._heads(k1), self._heads(v1), scale=sc)
            ax, ac = self.attn.proj_x(self._merge(ox)), self.attn.proj_c(self._merge(oc))
        else:
            ax, ac = self._sa(self.norm1(t)), self._sa(self.norm1(c))
        t = t + self._dpath(ax)
        t = t + self._dpath(self.mlp(self.norm2(t)))
        c = c + self._dpath(ac)
        c = c + self._dpath(self.mlp(self.norm2(c)))
        return t.transpose(1, 2).reshape(B, C, H, W), c


class StockLeMeViT(nn.Module):
    def __init__(self, depth, dims, heads, kinds, num_classes=1000, drop_path_rate=0.0):
        super().__init__()
        d0 = dims[0]
        self.downsample_layers = nn.ModuleList([nn.Sequential(nn.Conv2d(3, d0 // 2, 3, 2, 1), nn.BatchNorm2d(d0 // 2), nn.GELU(), nn.Conv2d(d0 // 2, d0, 3, 2, 1), nn.BatchNorm2d(d0))])
        for i in range(len(kinds) - 1):
            self.downsample_layers.append(nn.Identity() if kinds[i] == "C" else nn.Sequential(nn.Conv2d(dims[i], dims[i + 1], 3, 2, 1), nn.BatchNorm2d(dims[i + 1])))
        self.meta_tokens = nn.Parameter(torch.randn(16, d0))
        self.meta_token_downsample = nn.ModuleList()
        for i in range(len(kinds)):
            cin = d0 if i == 0 else dims[i - 1]
            self.meta_token_downsample.append(nn.Sequential(nn.Linear(cin, 4 * cin), nn.LayerNorm(4 * cin), nn.GELU(), nn.Linear(4 * cin, dims[i]), nn.LayerNorm(dims[i])))
        rates = torch.linspace(0, drop_path_rate, sum(depth)).tolist()
        self.stages, cur = nn.ModuleList(), 0
        for i, k in enumerate(kinds):
            self.stages.append(nn.ModuleList([StockBlock(k, dims[i], heads[i], rates[cur + j]) for j in range(depth[i])]))
            cur += depth[i]
        self.norm, self.norm_c, self.head = nn.BatchNorm2d(dims[-1]), nn.LayerNorm(dims[-1]), nn.Linear(dims[-1], num_classes)

    def forward(self, x):
        c = self.meta_tokens.unsqueeze(0).expand(x.shape[0], -1, -1)
        for i, st in enumerate(self.stages):
            x = self.downsample_layers[i](x)
            c = self.meta_token_downsample[i](c)
            for blk in st:
                x, c = blk(x, c)
        x = self.norm(x).flatten(2).mean(-1)
        return self.head(x + self.norm_c(c).mean(1))


VARIANTS = {"lemevit_tiny": ([1, 2, 2, 8, 2], [64, 64, 128, 192, 320]), "lemevit_small": ([1, 2, 2, 6, 2], [96, 96, 192, 320, 384]),
            "lemevit_base": ([2, 4, 4, 18, 4], [96, 96, 192, 384, 512])}


def timed(step, batch, warm=10, iters=40):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dict(images_per_sec=round(batch * iters / dt, 1), ms_per_step=round(1e3 * dt / iters, 3), warmup=warm, iters=iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="lemevit_base"); ap.add_argument("--batch", type=int, default=128); ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--out", default="gpurun_out/stock_eager.json"); ap.add_argument("--modes", default="infer,train")
    a = ap.parse_args()
    import lemevit_amd
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    depth, dims = VARIANTS[a.model]
    torch.manual_seed(0)
    ours = lemevit_amd.create_model(a.model, num_classes=1000, drop_path_rate=0.1)
    stock = StockLeMeViT(depth, dims, [d // 32 for d in dims], ["C", "D", "D", "S", "S"], 1000, 0.1)
    missing, unexpected = stock.load_state_dict(ours.state_dict(), strict=False)
    assert not missing and not unexpected, (missing, unexpected)          # identical state_dict layout to the reference
    stock = stock.to(dev).to(memory_format=torch.channels_last)
    ours = ours.to(dev)
    x = torch.randn(a.batch, 3, a.img, a.img, device=dev).contiguous(memory_format=torch.channels_last)
    res = dict(model=a.model, batch=a.batch, img=a.img, torch=torch.__version__, device=torch.cuda.get_device_name(0), protocol="benchmark.py: 10 warm-up + 40 timed steps, one synthetic batch")
    # parity of the two module trees (same weights): eval forward, bf16 autocast
    stock.eval(); ours.eval()
    with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
        d = (stock(x[:8]).float() - ours(x[:8]).float()).abs().max().item()
        ref = stock(x[:8]).float().abs().max().item()
    res["logit_diff_stock_vs_hip"] = dict(max_abs=round(d, 5), ref_max_abs=round(ref, 5))
    if "infer" in a.modes:
        def inf(m):
            def step():
                with torch.no_grad(), torch.autocast("cuda", torch.bfloat16):
                    m(x)
            return step
        res["infer_stock_eager"] = timed(inf(stock), a.batch)
        res["infer_hip_eager"] = timed(inf(ours), a.batch)
    if "train" in a.modes:
        lf = nn.CrossEntropyLoss()
        def trn(m, opt):
            def step():
                opt.zero_grad(set_to_none=True)
                with torch.autocast("cuda", torch.bfloat16):
                    loss = lf(m(x), torch.empty((a.batch,), device=dev, dtype=torch.long).random_(1000))
                loss.backward()
                opt.step()
            return step
        stock.train(); ours.train()
        decay = lambda m: [dict(params=[p for p in m.parameters() if p.ndim > 1], weight_decay=0.05), dict(params=[p for p in m.parameters() if p.ndim <= 1], weight_decay=0.0)]
        res["train_stock_eager"] = timed(trn(stock, torch.optim.AdamW(decay(stock), lr=1e-4, eps=1e-8, fused=True)), a.batch, 5, 20)
        res["train_hip_torch_adamw"] = timed(trn(ours, torch.optim.AdamW(decay(ours), lr=1e-4, eps=1e-8, fused=True)), a.batch, 5, 20)   # caller unchanged (benchmark.py:559-561)
        ours2 = lemevit_amd.create_model(a.model, num_classes=1000, drop_path_rate=0.1).to(dev).train()
        res["train_hip_flat_adamw"] = timed(trn(ours2, lemevit_amd.FlatAdamW(ours2, lr=1e-4, eps=1e-8, weight_decay=0.05)), a.batch, 5, 20)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
