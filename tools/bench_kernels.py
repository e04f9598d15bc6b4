#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels on the LeMeViT-Base B=128 shapes (HIP-event timed, bf16).
usage: python tools/bench_kernels.py [gemm] [attn] [ln] [conv]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lemevit_amd import ops
from lemevit_amd.ops import Prob

dev = "cuda:0"
bf = torch.bfloat16
B = int(os.environ.get("B", "128"))


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


STAGES = [(3136, 96), (784, 192), (196, 384), (49, 512)]   # (N tokens, C) for stages 1..4


def gemm():
    print(f"{'shape':44s} {'fwd us':>9s} {'TF':>7s} {'dx us':>9s} {'TF':>7s} {'dw us':>9s} {'TF':>7s}")
    tot = [0, 0, 0]
    nblk = [4, 4, 18, 4]
    for si, (N, C) in enumerate(STAGES):
        rx, rc = B * N, B * 16
        for name, n, k in [("qkv", 3 * C, C), ("proj", C, C), ("fc1", 4 * C, C), ("fc2", C, 4 * C)]:
            ax = torch.randn(rx, k, device=dev).to(bf); ac = torch.randn(rc, k, device=dev).to(bf)
            w = (torch.randn(n, k, device=dev) * 0.05).to(bf); bias = torch.zeros(n, device=dev)
            ox = torch.empty(rx, n, device=dev, dtype=bf); oc = torch.empty(rc, n, device=dev, dtype=bf)
            dyx = torch.randn(rx, n, device=dev).to(bf); dyc = torch.randn(rc, n, device=dev).to(bf)
            dxx = torch.empty(rx, k, device=dev, dtype=bf); dxc = torch.empty(rc, k, device=dev, dtype=bf)
            dw = torch.zeros(n, k, device=dev); db = torch.zeros(n, device=dev)
            fl = 2.0 * (rx + rc) * n * k
            t_f = timeit(lambda: ops.linear_fwd([Prob(ax, w, ox, bias=bias), Prob(ac, w, oc, bias=bias)], n, k))
            t_x = timeit(lambda: ops.linear_dx([Prob(dyx, w, dxx), Prob(dyc, w, dxc)], n, k))
            t_w = timeit(lambda: ops.linear_dw([Prob(dyx, ax, dw, bias_grad=db), Prob(dyc, ac, dw, bias_grad=db)], n, k))
            print(f"s{si+1} {name:5s} rows={rx+rc:7d} N={n:5d} K={k:5d}        {t_f:9.1f} {fl/t_f/1e6:7.1f} {t_x:9.1f} {fl/t_x/1e6:7.1f} {t_w:9.1f} {fl/t_w/1e6:7.1f}")
            for i, t in enumerate((t_f, t_x, t_w)): tot[i] += t * nblk[si]
    print(f"per-step totals (x blocks per stage): fwd {tot[0]/1e3:.2f} ms, dx {tot[1]/1e3:.2f} ms, dw {tot[2]/1e3:.2f} ms")


def attn():
    print(f"{'case':40s} {'fwd us':>9s} {'bwd us':>9s}")
    nblk = [4, 4, 18, 4]
    tf = tb = 0
    for si, (N, C) in enumerate(STAGES):
        h = C // 32
        if si < 2:   # DCA
            q1 = torch.randn(B, N, 3 * C, device=dev).to(bf); q2 = torch.randn(B, 16, 3 * C, device=dev).to(bf)
            sx, sc = ops.dca_scales(N, 16, C)
            for nm, q, k, v, s in [("dca x (N q x 16 k)", (q1, 0), (q2, C), (q2, 2 * C), sx), ("dca c (16 q x N k)", (q2, 0), (q1, C), (q1, 2 * C), sc)]:
                o, lse = ops.attn_fwd(q, k, v, C, s, want_lse=True)
                do = torch.randn_like(o); d1 = torch.empty_like(q1); d2 = torch.empty_like(q2)
                dq = (d1, 0) if q[0] is q1 else (d2, 0); dk = (d2, C) if k[0] is q2 else (d1, C); dv = (d2, 2 * C) if v[0] is q2 else (d1, 2 * C)
                t1 = timeit(lambda: ops.attn_fwd(q, k, v, C, s, want_lse=True)); t2 = timeit(lambda: ops.attn_bwd(q, k, v, o, lse, do, dq, dk, dv, C, s))
                print(f"s{si+1} {nm:24s} C={C:4d} N={N:5d}  {t1:9.1f} {t2:9.1f}"); tf += t1 * nblk[si]; tb += t2 * nblk[si]
        else:
            for L in (N, 16):
                qkv = torch.randn(B, L, 3 * C, device=dev).to(bf)
                o, lse = ops.attn_fwd((qkv, 0), (qkv, C), (qkv, 2 * C), C, ops.SDPA_SCALE, want_lse=True)
                do = torch.randn_like(o); dq = torch.empty_like(qkv)
                t1 = timeit(lambda: ops.attn_fwd((qkv, 0), (qkv, C), (qkv, 2 * C), C, ops.SDPA_SCALE, want_lse=True))
                t2 = timeit(lambda: ops.attn_bwd((qkv, 0), (qkv, C), (qkv, 2 * C), o, lse, do, (dq, 0), (dq, C), (dq, 2 * C), C, ops.SDPA_SCALE))
                fl = 4.0 * B * h * L * L * 32
                print(f"s{si+1} sa L={L:4d} C={C:4d} h={h:2d}              {t1:9.1f} {t2:9.1f}   fwd {fl/t1/1e6:6.1f} TF  bwd {2.5*fl/t2/1e6:6.1f} TF")
                tf += t1 * nblk[si]; tb += t2 * nblk[si]
    print(f"per-step totals: attn fwd {tf/1e3:.2f} ms, bwd {tb/1e3:.2f} ms")


def ln():
    for si, (N, C) in enumerate(STAGES):
        rows = B * N
        x = torch.randn(rows, C, device=dev).to(bf); g = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        y, st = ops.layernorm_fwd(x, g, b, 1e-6, True)
        dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
        t1 = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, True)); t2 = timeit(lambda: ops.layernorm_bwd(y, x, st, g, dg, db, dres=y))
        by = rows * C * 2
        print(f"s{si+1} LN rows={rows:7d} C={C:4d}: fwd {t1:8.1f} us {2*by/t1/1e6:6.2f} TB/s   bwd {t2:8.1f} us {4*by/t2/1e6:6.2f} TB/s")


def conv():
    for si, (N, C) in enumerate(STAGES):
        H = int(N ** 0.5)
        x = torch.randn(B, N, C, device=dev).to(bf); w = torch.randn(C, 1, 3, 3, device=dev); b = torch.zeros(C, device=dev)
        dw = torch.zeros_like(w); db = torch.zeros_like(b)
        t1 = timeit(lambda: ops.dwconv_residual_fwd(x, w, b, H, H)); t2 = timeit(lambda: ops.dwconv_residual_bwd_data(x, w, H, H))
        t3 = timeit(lambda: ops.dwconv_bwd_weight(x, x, dw, db, H, H))
        by = B * N * C * 2
        print(f"s{si+1} dwconv {H}x{H} C={C:4d}: fwd {t1:8.1f} us {2*by/t1/1e6:6.2f} TB/s  bwd_data {t2:8.1f} us  bwd_w {t3:8.1f} us {2*by/t3/1e6:6.2f} TB/s")


def fused():
    """MLP half of a block: LayerNorm + fc1(GELU) + fc2(+residual) launches vs lmv_mlp_fused_fwd; and LN + qkv vs lmv_ln_linear_fwd"""
    print(f"{'case':34s} {'unfused us':>11s} {'fused us':>9s} {'fused TF':>9s} {'speed-up':>9s}")
    nblk = [4, 4, 18, 4]
    tu = tfz = 0
    for si, (N, C) in enumerate(STAGES):
        Hd = 4 * C
        x = torch.randn(B, N, C, device=dev).to(bf); c = torch.randn(B, 16, C, device=dev).to(bf)
        g = torch.ones(C, device=dev); be = torch.zeros(C, device=dev)
        w1 = torch.randn(Hd, C, device=dev) * 0.05; b1 = torch.zeros(Hd, device=dev)
        w2 = (torch.randn(C, Hd, device=dev) * 0.05).to(bf); b2 = torch.zeros(C, device=dev)
        w1b = w1.to(bf)
        hx = torch.empty(B, N, Hd, device=dev, dtype=bf); hc = torch.empty(B, 16, Hd, device=dev, dtype=bf)
        ox = torch.empty_like(x); oc = torch.empty_like(c)

        def unfused():
            (xn, cn), _ = ops.layernorm_fwd_multi([x, c], g, be, 1e-6)
            ops.linear_fwd([Prob(xn, w1b, hx, bias=b1), Prob(cn, w1b, hc, bias=b1)], Hd, C, ops.ACT_GELU)
            ops.linear_fwd([Prob(hx, w2, ox, bias=b2, res=x), Prob(hc, w2, oc, bias=b2, res=c)], C, Hd)
        t_u = timeit(unfused)
        fl = 4.0 * B * (N + 16) * C * Hd
        if ops.mlp_fused_supported(C, Hd, bf):
            F = ops.ln_fold(w1, b1, g, be, bf)
            t_f = timeit(lambda: ops.mlp_fused_fwd([x, c], F, w2, b2, 1e-6))
            print(f"s{si+1} mlp  rows={B*(N+16):7d} C={C:4d}      {t_u:11.1f} {t_f:9.1f} {fl/t_f/1e6:9.1f} {t_u/t_f:9.2f}")
            tu += t_u * nblk[si]; tfz += t_f * nblk[si]
        else:
            print(f"s{si+1} mlp  rows={B*(N+16):7d} C={C:4d}      {t_u:11.1f}   (unsupported C)")
        # LayerNorm + qkv projection
        wq = torch.randn(3 * C, C, device=dev) * 0.05; bq = torch.zeros(3 * C, device=dev); wqb = wq.to(bf)
        qx = torch.empty(B, N, 3 * C, device=dev, dtype=bf); qc = torch.empty(B, 16, 3 * C, device=dev, dtype=bf)

        def unfused_qkv():
            (xn, cn), _ = ops.layernorm_fwd_multi([x, c], g, be, 1e-6)
            ops.linear_fwd([Prob(xn, wqb, qx, bias=bq), Prob(cn, wqb, qc, bias=bq)], 3 * C, C)
        Fq = ops.ln_fold(wq, bq, g, be, bf)
        t_u = timeit(unfused_qkv)
        t_f = timeit(lambda: ops.ln_linear_fwd([Prob(x, Fq.wf, qx, bias=Fq.bf, aux=Fq.colsum), Prob(c, Fq.wf, qc, bias=Fq.bf, aux=Fq.colsum)], 3 * C, C, 1e-6))
        flq = 2.0 * B * (N + 16) * C * 3 * C
        print(f"s{si+1} ln+qkv rows={B*(N+16):7d} C={C:4d}    {t_u:11.1f} {t_f:9.1f} {flq/t_f/1e6:9.1f} {t_u/t_f:9.2f}")
    print(f"MLP halves per forward pass (x blocks per stage, fused stages only): unfused {tu/1e3:.2f} ms, fused {tfz/1e3:.2f} ms")


if __name__ == "__main__":
    what = sys.argv[1:] or ["gemm", "attn", "ln", "conv"]
    for w in what:
        print(f"==== {w} (B={B}) ====")
        globals()[w]()
