"""Per-kernel parity on a real MI355X: every C-ABI op vs a float64 CPU restatement of the same math
(oracle primitives) on identical (bf16-rounded, in bf16 mode) inputs.

Tolerances (BASELINE.json north_star): fp32 1e-5, bf16 1e-3, both relative to the output's max-abs;
in bf16 mode one bf16 output rounding (2^-8 * |ref|, elementwise) is allowed on top, because the
kernel's *output* is itself stored in bf16."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from detfill import det_tensor
from oracle import lemevit_oracle as O

DTYPES = [torch.float32, torch.bfloat16]


def dev():
    return torch.device("cuda:0")


def ops():
    from lemevit_amd import ops as _ops
    return _ops


def rnd(shape, name, dtype, scale=1.0):
    """deterministic tensor, rounded to `dtype`; returns (gpu tensor in dtype, float64 cpu copy of the rounded values)"""
    t = det_tensor(shape, name, 7, scale).to(dtype)
    return t.to(dev()), t.to(torch.float64)


def assert_close(out, ref, dtype, what, tol32=1e-5, tol16=1e-3):
    out = out.detach().to("cpu", torch.float64)
    ref = ref.to(torch.float64)
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    assert torch.isfinite(out).all(), what + ": non-finite output"
    mx = max(float(ref.abs().max()), 1e-30)
    err = (out - ref).abs()
    if dtype == torch.float32:
        assert float(err.max()) <= tol32 * mx, f"{what}: max-abs err {float(err.max()):.3e} vs {tol32 * mx:.3e}"
    else:
        bound = tol16 * mx + ref.abs() * 2.0 ** -8
        bad = err > bound
        assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements off, worst {float((err - bound).max()):.3e} over (max-abs {mx:.3e})"


def gelu64(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def gelu_grad64(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,N,K", [(300, 96, 64), (4096, 288, 96), (1000, 1152, 384), (129, 384, 1536), (16, 64, 256), (2051, 1000, 512),
                                      (140001, 384, 128), (70003, 200, 64)])      # the last two: > 1024 tiles (several generations of workgroups)
def test_linear_fwd(dtype, rows, N, K):
    o = ops()
    a, a64 = rnd((rows, K), "a", dtype)
    w, w64 = rnd((N, K), "w", dtype, 1 / math.sqrt(K))
    bias = det_tensor((N,), "b", 7, 0.5).to(dev()); res, res64 = rnd((rows, N), "res", dtype)
    rps = 7
    nsamp = (rows + rps - 1) // rps
    rs = (det_tensor((nsamp,), "rs", 7).abs() + 0.5).to(dev())
    # plain
    out = torch.empty((rows, N), device=dev(), dtype=dtype)
    o.linear_fwd([o.Prob(a, w, out)], N, K)
    assert_close(out, a64 @ w64.t(), dtype, "plain")
    # bias + gelu + pre-activation copy + DropPath scale + residual
    pre = torch.empty_like(out)
    o.linear_fwd([o.Prob(a, w, out, bias=bias, res=res, row_scale=rs, out_pre=pre, rps=rps)], N, K, o.ACT_GELU)
    u = a64 @ w64.t() + bias.cpu().double()
    scale_rows = rs.cpu().double()[torch.arange(rows) // rps][:, None]
    assert_close(pre, u, dtype, "pre")
    assert_close(out, res64 + scale_rows * gelu64(u), dtype, "epilogue")


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_dual(dtype):
    o = ops()
    N, K = 192, 96
    ax, ax64 = rnd((1500, K), "ax", dtype); ac, ac64 = rnd((48, K), "ac", dtype)
    w1, w164 = rnd((N, K), "w1", dtype, 0.1); w2, w264 = rnd((N, K), "w2", dtype, 0.1)
    b1 = det_tensor((N,), "b1", 7).to(dev()); b2 = det_tensor((N,), "b2", 7).to(dev())
    ox = torch.empty((1500, N), device=dev(), dtype=dtype); oc = torch.empty((48, N), device=dev(), dtype=dtype)
    o.linear_fwd([o.Prob(ax, w1, ox, bias=b1), o.Prob(ac, w2, oc, bias=b2)], N, K)
    assert_close(ox, ax64 @ w164.t() + b1.cpu().double(), dtype, "x")
    assert_close(oc, ac64 @ w264.t() + b2.cpu().double(), dtype, "c")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,N,K", [(300, 96, 64), (4100, 288, 96), (777, 1536, 384), (130, 384, 1536), (513, 1000, 512)])
def test_linear_dx(dtype, rows, N, K):
    o = ops()
    dy, dy64 = rnd((rows, N), "dy", dtype)
    w, w64 = rnd((N, K), "w", dtype, 1 / math.sqrt(N))
    u, u64 = rnd((rows, K), "u", dtype, 2.0); res, res64 = rnd((rows, K), "res", dtype)
    out = torch.empty((rows, K), device=dev(), dtype=dtype)
    o.linear_dx([o.Prob(dy, w, out)], N, K)
    assert_close(out, dy64 @ w64, dtype, "dx")
    o.linear_dx([o.Prob(dy, w, out, aux=u, res=res)], N, K, o.ACT_GELU_GRAD)
    assert_close(out, res64 + (dy64 @ w64) * gelu_grad64(u64), dtype, "dx*gelu'+res")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,N,K", [(300, 96, 64), (40000, 288, 96), (5000, 1152, 384), (2049, 384, 1536), (128, 1000, 512)])
def test_linear_dw(dtype, rows, N, K):
    o = ops()
    dy, dy64 = rnd((rows, N), "dy", dtype)
    x, x64 = rnd((rows, K), "x", dtype)
    dyc, dyc64 = rnd((33, N), "dyc", dtype); xc, xc64 = rnd((33, K), "xc", dtype)
    dw = torch.zeros((N, K), device=dev()); db = torch.zeros((N,), device=dev())
    # two problems accumulating into the SAME weight gradient (shared weights of x- and c-path)
    o.linear_dw([o.Prob(dy, x, dw, bias_grad=db), o.Prob(dyc, xc, dw, bias_grad=db)], N, K)
    ref_w = dy64.t() @ x64 + dyc64.t() @ xc64
    ref_b = dy64.sum(0) + dyc64.sum(0)
    # fp32 atomics: order-dependent rounding, scale tolerance with sqrt(rows)
    assert_close(dw, ref_w, torch.float32, "dw", tol32=3e-6 * math.sqrt(rows) if dtype == torch.float32 else 2e-5)
    assert_close(db, ref_b, torch.float32, "db", tol32=3e-6 * math.sqrt(rows) if dtype == torch.float32 else 2e-5)


# ---- the BigK tile (128 x 256 / 384, register-pipelined k-loop): long bf16 reductions, whole 64-deep k-steps ----------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,rows_c,N,K", [(6400, 128, 1536, 384), (4096, 64, 384, 1536), (4160, 128, 512, 2048), (2048, 0, 1152, 384),
                                             (1024, 2048, 2048, 512), (12800, 0, 768, 192), (1088, 64, 384, 384)])
def test_linear_dw_bigk(dtype, rows, rows_c, N, K):
    """Weight gradients whose token counts are whole k-steps (every LeMeViT layer at B % 4 == 0): x rows and meta-token rows accumulate
    into the same dW / db as ONE concatenated reduction; also accumulation on top of an existing gradient."""
    o = ops()
    dy, dy64 = rnd((rows, N), "dy", dtype); x, x64 = rnd((rows, K), "x", dtype)
    dw0 = det_tensor((N, K), "dw0", 7).to(dev()); db0 = det_tensor((N,), "db0", 7).to(dev())
    dw, db = dw0.clone(), db0.clone()
    probs = [o.Prob(dy, x, dw, bias_grad=db)]
    ref_w = dy64.t() @ x64; ref_b = dy64.sum(0)
    if rows_c:
        dyc, dyc64 = rnd((rows_c, N), "dyc", dtype); xc, xc64 = rnd((rows_c, K), "xc", dtype)
        probs.append(o.Prob(dyc, xc, dw, bias_grad=db))
        ref_w = ref_w + dyc64.t() @ xc64; ref_b = ref_b + dyc64.sum(0)
    o.linear_dw(probs, N, K)
    tol = 3e-6 * math.sqrt(rows) if dtype == torch.float32 else 2e-5
    assert_close(dw - dw0, ref_w, torch.float32, "dw", tol32=tol)
    assert_close(db - db0, ref_b, torch.float32, "db", tol32=tol)
    # two problems with DIFFERENT outputs (the qkv1 / qkv2 launch of a D block)
    if rows_c:
        dw1 = torch.zeros((N, K), device=dev()); db1 = torch.zeros((N,), device=dev()); dw2 = torch.zeros_like(dw1); db2 = torch.zeros_like(db1)
        o.linear_dw([o.Prob(dy, x, dw1, bias_grad=db1), o.Prob(dyc, xc, dw2, bias_grad=db2)], N, K)
        assert_close(dw1, dy64.t() @ x64, torch.float32, "dw1", tol32=tol); assert_close(db1, dy64.sum(0), torch.float32, "db1", tol32=tol)
        assert_close(dw2, dyc64.t() @ xc64, torch.float32, "dw2", tol32=tol); assert_close(db2, dyc64.sum(0), torch.float32, "db2", tol32=tol)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,rows_c,N,K", [(3000, 48, 384, 1536), (1000, 16, 512, 2048), (777, 0, 256, 768), (129, 0, 1000, 1024)])
def test_linear_fwd_dx_bigk(dtype, rows, rows_c, N, K):
    """Forward and dX on long reductions (fc2 forward: K = 4C; dX of qkv / fc1: reduction over 3C / 4C), dual problems, ragged rows /
    columns, every epilogue."""
    o = ops()
    a, a64 = rnd((rows, K), "a", dtype); w, w64 = rnd((N, K), "w", dtype, 1 / math.sqrt(K))
    bias = det_tensor((N,), "b", 7, 0.5).to(dev()); res, res64 = rnd((rows, N), "res", dtype)
    rps = 7
    rs = (det_tensor(((rows + rps - 1) // rps,), "rs", 7).abs() + 0.5).to(dev())
    out = torch.empty((rows, N), device=dev(), dtype=dtype); pre = torch.empty_like(out)
    probs = [o.Prob(a, w, out, bias=bias, res=res, row_scale=rs, out_pre=pre, rps=rps)]
    if rows_c:
        ac, ac64 = rnd((rows_c, K), "ac", dtype); oc = torch.empty((rows_c, N), device=dev(), dtype=dtype); prec = torch.empty_like(oc)
        probs.append(o.Prob(ac, w, oc, bias=bias, out_pre=prec))
    o.linear_fwd(probs, N, K, o.ACT_GELU)
    u = a64 @ w64.t() + bias.cpu().double()
    assert_close(pre, u, dtype, "pre")
    assert_close(out, res64 + rs.cpu().double()[torch.arange(rows) // rps][:, None] * gelu64(u), dtype, "epilogue")
    if rows_c:
        uc = ac64 @ w64.t() + bias.cpu().double()
        assert_close(prec, uc, dtype, "pre c"); assert_close(oc, gelu64(uc), dtype, "gelu c")
    # dX with the roles of N and K swapped: dy [rows, K'] x W [K', N'] with K' = K (the long reduction), N' = N
    dy, dy64 = rnd((rows, K), "dy", dtype); wt, wt64 = rnd((K, N), "wt", dtype, 1 / math.sqrt(K))
    uu, uu64 = rnd((rows, N), "u", dtype, 2.0)
    dx = torch.empty((rows, N), device=dev(), dtype=dtype)
    o.linear_dx([o.Prob(dy, wt, dx)], K, N)
    assert_close(dx, dy64 @ wt64, dtype, "dx")
    o.linear_dx([o.Prob(dy, wt, dx, aux=uu, res=res)], K, N, o.ACT_GELU_GRAD)
    assert_close(dx, res64 + (dy64 @ wt64) * gelu_grad64(uu64), dtype, "dx*gelu'+res")


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [(1000, 64), (3137, 96), (800, 192), (212, 320), (333, 384), (65, 512), (48, 1280), (16, 2048), (5, 128)])
def test_layernorm(dtype, rows, C):
    o = ops()
    x, x64 = rnd((rows, C), "x", dtype, 2.0)
    x64 = x64 + 0.0
    g = (det_tensor((C,), "g", 7, 0.3) + 1).to(dev()); b = det_tensor((C,), "be", 7, 0.2).to(dev())
    y, stats = o.layernorm_fwd(x, g, b, 1e-6, want_stats=True)
    g64, b64 = g.cpu().double(), b.cpu().double()
    assert_close(y, O.layer_norm(x64, g64, b64, 1e-6), dtype, "ln fwd")
    mu = x64.mean(-1); rstd = 1 / torch.sqrt(x64.var(-1, unbiased=False) + 1e-6)
    assert_close(stats[:, 0], mu, torch.float32, "mean", 2e-5); assert_close(stats[:, 1], rstd, torch.float32, "rstd", 2e-5)
    # backward
    dy, dy64 = rnd((rows, C), "dy", dtype); dres, dres64 = rnd((rows, C), "dres", dtype)
    xr = x64.clone().requires_grad_(True); gr = g64.clone().requires_grad_(True); br = b64.clone().requires_grad_(True)
    (O.layer_norm(xr, gr, br, 1e-6) * dy64).sum().backward()
    dg = torch.zeros(C, device=dev()); dbt = torch.zeros(C, device=dev())
    dx = o.layernorm_bwd(dy, x, stats, g, dg, dbt, dres=dres)
    assert_close(dx, xr.grad + dres64, dtype, "ln dx", tol32=2e-5, tol16=2e-3)
    assert_close(dg, gr.grad, torch.float32, "dgamma", 2e-5); assert_close(dbt, br.grad, torch.float32, "dbeta", 2e-5)
    # fused activation: y = GELU(LN(x)) and its backward (the meta-token MLPs)
    (ya,), (sta,) = o.layernorm_fwd_multi([x], g, b, 1e-6, want_stats=True, gelu=True)
    xr2 = x64.clone().requires_grad_(True); gr2 = g64.clone().requires_grad_(True); br2 = b64.clone().requires_grad_(True)
    ya_ref = gelu64(O.layer_norm(xr2, gr2, br2, 1e-6))
    assert_close(ya, ya_ref.detach(), dtype, "ln+gelu fwd")
    (ya_ref * dy64).sum().backward()
    dg2 = torch.zeros(C, device=dev()); db2 = torch.zeros(C, device=dev())
    (dxa,) = o.layernorm_bwd_multi([dy], [x], [sta], g, dg2, db2, [None], gelu_beta=b)
    assert_close(dxa, xr2.grad, dtype, "ln+gelu dx", tol32=2e-5, tol16=2e-3)
    assert_close(dg2, gr2.grad, torch.float32, "ln+gelu dgamma", 2e-5 if dtype == torch.float32 else 2e-4)
    assert_close(db2, br2.grad, torch.float32, "ln+gelu dbeta", 2e-5 if dtype == torch.float32 else 2e-4)
    # optional second output of the same launch: dx pre-scaled by the NEXT stage's per-sample DropPath vector
    if rows % 5 == 0:
        x3 = x.view(rows // 5, 5, C); sc = (det_tensor((rows // 5,), "sc", 7).abs() + 0.25).to(dev())
        (dx2,), (dxs,) = o.layernorm_bwd_multi([dy.view_as(x3)], [x3], [stats], g, torch.zeros_like(dg), torch.zeros_like(dbt), [dres.view_as(x3)],
                                               next_scales=[sc])
        assert torch.equal(dx2.view_as(dx), dx)
        assert_close(dxs.view_as(dx), (xr.grad + dres64) * sc.cpu().double().repeat_interleave(5)[:, None], dtype, "ln dx scaled", tol32=2e-5, tol16=2e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 56, 56, 96), (3, 7, 7, 512), (1, 35, 35, 64), (2, 14, 14, 384), (1, 5, 3, 32)])
def test_dwconv(dtype, B, H, W, C):
    o = ops()
    x, x64 = rnd((B, H * W, C), "x", dtype)
    wt = det_tensor((C, 1, 3, 3), "w", 7, 0.4).to(dev()); bs = det_tensor((C,), "b", 7, 0.2).to(dev())
    sd = {"p.pos_embed.weight": wt.cpu().double().requires_grad_(True), "p.pos_embed.bias": bs.cpu().double().requires_grad_(True)}
    xr = x64.clone().requires_grad_(True)
    ref = O.pos_embed_residual(sd, "p.", xr, H, W)
    y = o.dwconv_residual_fwd(x, wt, bs, H, W)
    assert_close(y, ref.detach(), dtype, "dwconv fwd")
    dy, dy64 = rnd((B, H * W, C), "dy", dtype)
    (ref * dy64).sum().backward()
    dx = o.dwconv_residual_bwd_data(dy, wt, H, W)
    assert_close(dx, xr.grad, dtype, "dwconv dx")
    dw = torch.zeros_like(wt); db = torch.zeros_like(bs)
    o.dwconv_bwd_weight(dy, x, dw, db, H, W)
    assert_close(dw, sd["p.pos_embed.weight"].grad, torch.float32, "dwconv dw", 3e-5); assert_close(db, sd["p.pos_embed.bias"].grad, torch.float32, "dwconv db", 3e-5)


# ------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, h, scale):
    """q [B,Lq,C], k/v [B,Lk,C] float64 -> o [B,Lq,C], via the oracle's sdpa."""
    (qh,) = O.split_heads(q, 1, h); (kh,) = O.split_heads(k, 1, h); (vh,) = O.split_heads(v, 1, h)
    return O.merge_heads(O.sdpa(qh, kh, vh, scale))


ATTN_CASES = [  # name, B, Lq, Lk, C, self (packed qkv) ?
    ("sa196", 2, 196, 196, 384, True), ("sa49", 3, 49, 49, 512, True), ("sa16", 2, 16, 16, 64, True), ("sa577", 1, 577, 577, 64, True),
    ("fewq", 2, 16, 3136, 96, False), ("fewq_odd", 1, 16, 1225, 64, False), ("fewq_short", 2, 16, 70, 64, False),
    ("fewk", 2, 3136, 16, 96, False), ("fewk_odd", 1, 1225, 16, 64, False), ("gen", 2, 100, 37, 64, False),
    # many queries x 225..640 keys: the whole-row MFMA kernels with K / V in dynamic LDS (SA at 384^2 is 576 + 16 tokens)
    ("sa592", 2, 592, 592, 384, True), ("long_640", 1, 300, 640, 64, False), ("long_225", 1, 33, 225, 64, False),
    # beyond 640 keys (bf16): K / V streamed through LDS in 256-key chunks, online softmax (dense backbones: 4096 image tokens)
    ("stream_700", 1, 40, 700, 64, False), ("stream_sa1040", 1, 1040, 1040, 96, True), ("stream_ragged", 2, 333, 2049, 64, False),
    ("stream_4096", 1, 4096, 4096, 64, True),
    # BASELINE config 5 (Base at 384^2): DCA with N = 9216 (stage 1, C = 96) and N = 2304 (stage 2, C = 192) image tokens against 16 meta tokens
    ("fewq_9216", 1, 16, 9216, 96, False), ("fewk_9216", 1, 9216, 16, 96, False), ("fewq_2304", 2, 16, 2304, 192, False), ("fewk_2304", 2, 2304, 16, 192, False),
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", ATTN_CASES, ids=[c[0] for c in ATTN_CASES])
def test_attention(dtype, case):
    o = ops()
    name, B, Lq, Lk, C, packed = case
    h = C // 32
    scale = 0.2 if not packed else 32 ** -0.5
    if packed:
        qkv, qkv64 = rnd((B, Lq, 3 * C), name + "qkv", dtype, 1.5)
        q = (qkv, 0); k = (qkv, C); v = (qkv, 2 * C)
        q64, k64, v64 = qkv64[..., :C], qkv64[..., C:2 * C], qkv64[..., 2 * C:]
    else:   # q from a packed [B,Lq,3C] (its first third), k/v from a packed [B,Lk,2C]
        qp, qp64 = rnd((B, Lq, 3 * C), name + "q", dtype, 1.5); kv, kv64 = rnd((B, Lk, 2 * C), name + "kv", dtype, 1.5)
        q = (qp, 0); k = (kv, 0); v = (kv, C)
        q64, k64, v64 = qp64[..., :C], kv64[..., :C], kv64[..., C:]
    q64 = q64.clone().requires_grad_(True); k64 = k64.clone().requires_grad_(True); v64 = v64.clone().requires_grad_(True)
    ref = _attn_ref(q64, k64, v64, h, scale)
    out, lse = o.attn_fwd(q, k, v, C, scale, want_lse=True)
    assert_close(out, ref.detach(), dtype, name + " fwd")
    (qh,) = O.split_heads(q64.detach(), 1, h); (kh,) = O.split_heads(k64.detach(), 1, h)
    assert_close(lse, torch.logsumexp(qh @ kh.transpose(-1, -2) * scale, -1), torch.float32, name + " lse", 2e-5)
    # backward
    do, do64 = rnd((B, Lq, C), name + "do", dtype)
    (ref * do64).sum().backward()
    if packed:
        dqkv = torch.full_like(qkv, float("nan"))
        o.attn_bwd(q, k, v, out, lse, do, (dqkv, 0), (dqkv, C), (dqkv, 2 * C), C, scale)
        got = dqkv; want = torch.cat([q64.grad, k64.grad, v64.grad], -1)
        assert_close(got, want, dtype, name + " dqkv", tol32=2e-5, tol16=3e-3)
    else:
        dqp = torch.zeros_like(qp); dkv = torch.full_like(kv, float("nan"))
        o.attn_bwd(q, k, v, out, lse, do, (dqp, 0), (dkv, 0), (dkv, C), C, scale)
        assert_close(dqp[..., :C], q64.grad, dtype, name + " dq", tol32=2e-5, tol16=3e-3)
        assert_close(dkv, torch.cat([k64.grad, v64.grad], -1), dtype, name + " dkv", tol32=2e-5, tol16=3e-3)


@pytest.mark.parametrize("dtype", DTYPES)
def test_named_cores(dtype):
    """lmv_sa_core_fwd / lmv_ca_core_fwd / lmv_dca_core_fwd == the oracle's attention flavours (sans projections)."""
    from lemevit_amd._lib import lib, check
    o = ops()
    B, N, M, C = 2, 784, 16, 96
    h = C // 32
    qkv1, qkv164 = rnd((B, N, 3 * C), "qkv1", dtype, 1.5); qkv2, qkv264 = rnd((B, M, 3 * C), "qkv2", dtype, 1.5)
    ox = torch.empty((B, N, C), device=dev(), dtype=dtype); oc = torch.empty((B, M, C), device=dev(), dtype=dtype)
    ws = torch.empty(lib.lmv_attn_workspace_bytes(B, h, M, N, 0) + 1024, device=dev(), dtype=torch.uint8)
    st = torch.cuda.current_stream().cuda_stream
    code = o.dtype_code(qkv1)
    check(lib.lmv_dca_core_fwd(qkv1.data_ptr(), qkv2.data_ptr(), ox.data_ptr(), oc.data_ptr(), None, None, B, N, M, C, ws.data_ptr(), ws.numel(), code, st), "dca")
    sx, sc = O.dca_scales(N, M, C)
    assert_close(ox, _attn_ref(qkv164[..., :C], qkv264[..., C:2 * C], qkv264[..., 2 * C:], h, sx), dtype, "dca x")
    assert_close(oc, _attn_ref(qkv264[..., :C], qkv164[..., C:2 * C], qkv164[..., 2 * C:], h, sc), dtype, "dca c")
    check(lib.lmv_sa_core_fwd(qkv1.data_ptr(), ox.data_ptr(), None, B, N, C, ws.data_ptr(), ws.numel(), code, st), "sa")
    assert_close(ox, _attn_ref(qkv164[..., :C], qkv164[..., C:2 * C], qkv164[..., 2 * C:], h, None), dtype, "sa")
    qc, qc64 = rnd((B, M, C), "qc", dtype, 1.5); kv, kv64 = rnd((B, N, 2 * C), "kv", dtype, 1.5)
    check(lib.lmv_ca_core_fwd(qc.data_ptr(), kv.data_ptr(), oc.data_ptr(), None, B, M, N, C, ws.data_ptr(), ws.numel(), code, st), "ca")
    assert_close(oc, _attn_ref(qc64, kv64[..., :C], kv64[..., C:], h, None), dtype, "ca")


@pytest.mark.parametrize("dtype", DTYPES)
def test_utils(dtype):
    o = ops()
    x, x64 = rnd((1001, 96), "x", dtype)
    s = (det_tensor((143,), "s", 7).abs()).to(dev())
    y = o.row_scale(x, s, 7)
    assert_close(y, x64 * s.cpu().double()[torch.arange(1001) // 7][:, None], dtype, "row_scale")
    other = torch.bfloat16 if dtype == torch.float32 else torch.float32
    z = o.cast(x.reshape(-1)[:96003], other)
    assert torch.equal(z.cpu(), x.reshape(-1)[:96003].cpu().to(other))


def test_adamw_flat():
    o = ops()
    n = 4096 * 3
    p = det_tensor((n,), "p", 7).to(dev()); g = det_tensor((n,), "g", 7, 0.1).to(dev())
    pr = torch.nn.Parameter(p.clone()); opt = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
    m = torch.zeros_like(p); v = torch.zeros_like(p)
    shadow = torch.zeros(n, device=dev(), dtype=torch.bfloat16)
    for step in range(1, 4):
        pr.grad = g.clone(); opt.step()
        o.adamw_flat(p, g, m, v, None, 1e-2, 0.9, 0.999, 1e-8, 0.05, step, shadow=shadow)
    assert_close(p, pr.detach().cpu(), torch.float32, "adamw", 1e-5)
    assert torch.equal(shadow, p.to(torch.bfloat16)), "bf16 shadow must be the rounded updated parameters"
    # step count on the device (captured graphs) + per-element weight-decay mask
    p2 = det_tensor((n,), "p", 7).to(dev()); m2 = torch.zeros_like(p2); v2 = torch.zeros_like(p2)
    mask = (torch.arange(n, device=dev()) % 3 == 0).float()
    pa = torch.nn.Parameter(p2[mask.bool()].clone()); pb = torch.nn.Parameter(p2[~mask.bool()].clone())
    opt2 = torch.optim.AdamW([dict(params=[pa], weight_decay=0.05), dict(params=[pb], weight_decay=0.0)], lr=1e-2, eps=1e-8)
    sd = torch.zeros((), device=dev(), dtype=torch.int32)
    for _ in range(3):
        pa.grad = g[mask.bool()].clone(); pb.grad = g[~mask.bool()].clone(); opt2.step()
        sd += 1
        o.adamw_flat(p2, g, m2, v2, mask, 1e-2, 0.9, 0.999, 1e-8, 0.05, 0, step_dev=sd)
    assert_close(p2[mask.bool()], pa.detach().cpu(), torch.float32, "adamw masked decay", 1e-5)
    assert_close(p2[~mask.bool()], pb.detach().cpu(), torch.float32, "adamw no decay", 1e-5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 3, 224, 224), (1, 3, 97, 131), (3, 3, 8, 5)])
def test_im2col_stem(dtype, shape):
    """Patch matrix of the stem's first convolution (3x3, stride 2, padding 1; models/lemevit.py:713): bit-exact copy of the
    pixels in the weight's own (ci, ky, kx) order, zero padding, columns 27..31 zero; NCHW and channels-last sources; and the
    convolution it stands for: patches @ W^T + b == F.conv2d."""
    o = ops()
    B, _, H, W = shape
    x = det_tensor(shape, "img", 5).to(dev())
    ref = torch.nn.functional.unfold(x.double(), 3, padding=1, stride=2).transpose(1, 2).reshape(-1, 27)       # [B*Ho*Wo, 27]
    for src in (x, x.contiguous(memory_format=torch.channels_last), x.to(torch.bfloat16)):
        p = o.im2col3x3s2_c3(src, dtype)
        want = ref if src.dtype == torch.float32 else torch.nn.functional.unfold(src.double(), 3, padding=1, stride=2).transpose(1, 2).reshape(-1, 27)
        assert p.shape == (B * ((H + 1) // 2) * ((W + 1) // 2), 32) and p.dtype == dtype
        assert torch.equal(p[:, :27].double().cpu(), want.to(dtype).double().cpu()), "patch values"
        assert not bool(p[:, 27:].any()), "padding columns must be zero"
    Co = 48
    w = det_tensor((Co, 3, 3, 3), "w", 5, 0.3).to(dev()); b = det_tensor((Co,), "b", 5, 0.1).to(dev())
    wm = torch.zeros(Co, 32, device=dev(), dtype=dtype); wm[:, :27] = w.reshape(Co, 27).to(dtype)
    p = o.im2col3x3s2_c3(x, dtype)
    y = torch.empty(p.shape[0], Co, device=dev(), dtype=dtype)
    o.linear_fwd([o.Prob(p, wm, y, bias=b)], Co, 32)
    conv = torch.nn.functional.conv2d(x.to(dtype).double().cpu(), w.to(dtype).double().cpu(), b.double().cpu(), stride=2, padding=1)
    assert_close(y, conv.permute(0, 2, 3, 1).reshape(-1, Co), dtype, "stem conv1 as GEMM")


# ------------------------------------------------------------------------------------------------
# training-mode BatchNorm2d (+ GELU) over channels-last rows (models/lemevit.py:698-704, 714-717, 773)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C,gelu", [(2 * 56 * 56, 48, True), (3 * 28 * 28, 96, False), (2 * 7 * 7, 512, False), (4 * 5 * 3, 64, True), (37, 1024, False)])
def test_batchnorm_train(rows, C, gelu, dtype):
    x, x64 = rnd((rows, C), "bn.x", dtype, 1.5)
    x64 = x64 + 0.0
    # a per-channel offset so that mean^2 is not small against the variance
    off = det_tensor((C,), "bn.off", 7, 2.0).to(dtype)
    x = (x.float() + off.to(dev())).to(dtype); x64 = x.to("cpu", torch.float64)
    g = (det_tensor((C,), "bn.g", 7, 0.3) + 1.0); b = det_tensor((C,), "bn.b", 7, 0.2)
    rm = det_tensor((C,), "bn.rm", 7, 0.2); rv = det_tensor((C,), "bn.rv", 7, 0.3) + 1.0
    dy, dy64 = rnd((rows, C), "bn.dy", dtype)
    eps, mom = 1e-5, 0.1
    # float64 reference (torch.nn.functional.batch_norm semantics)
    mean = x64.mean(0); var = x64.var(0, unbiased=False); rstd = 1.0 / torch.sqrt(var + eps)
    xh = (x64 - mean) * rstd
    u = xh * g.double() + b.double()
    y_ref = gelu64(u) if gelu else u
    rm_ref = (1 - mom) * rm.double() + mom * mean
    rv_ref = (1 - mom) * rv.double() + mom * x64.var(0, unbiased=True)
    dyp = dy64 * gelu_grad64(u) if gelu else dy64
    dbeta_ref = dyp.sum(0); dgamma_ref = (dyp * xh).sum(0)
    dx_ref = g.double() * rstd * (dyp - dyp.mean(0) - xh * (dyp * xh).mean(0))

    gd, bd, rmd, rvd = g.to(dev()), b.to(dev()), rm.to(dev()).clone(), rv.to(dev()).clone()
    y, stats = ops().batchnorm_train_fwd(x, gd, bd, rmd, rvd, mom, eps, gelu)
    assert_close(y, y_ref, dtype, "bn y")
    assert_close(stats[0], mean, torch.float32, "bn mean", tol32=2e-6)
    assert_close(stats[1], rstd, torch.float32, "bn rstd", tol32=1e-5)
    assert_close(rmd, rm_ref, torch.float32, "running_mean", tol32=2e-6)
    assert_close(rvd, rv_ref, torch.float32, "running_var", tol32=1e-5)
    dx, dgamma, dbeta = ops().batchnorm_train_bwd(dy, x, gd, bd, stats, gelu)
    assert_close(dx, dx_ref, dtype, "bn dx", tol32=2e-5)
    tolw = 1e-5 if dtype == torch.float32 else 2e-3          # bf16 mode: GELU' through the branch-free erf
    assert_close(dgamma, dgamma_ref, torch.float32, "bn dgamma", tol32=tolw)
    assert_close(dbeta, dbeta_ref, torch.float32, "bn dbeta", tol32=tolw)


# ------------------------------------------------------------------------------------------------
# skinny launches at full size: one n-tile over > 1024 row tiles, ragged last tile, two problems with different weights and DropPath
# vectors (the C = 96 stage shapes of BASELINE's configuration; sampled rows against float64)
@pytest.mark.parametrize("N,K", [(96, 384), (96, 96), (64, 256), (128, 64)])
def test_linear_fwd_skinny(N, K):
    o = ops()
    dtype = torch.bfloat16
    rows_x, rows_c = 131072 + 333, 4100                       # ragged last tile, second problem with its own weights
    ax, _ = rnd((rows_x, K), "sk.ax", dtype); ac, _ = rnd((rows_c, K), "sk.ac", dtype)
    w1, _ = rnd((N, K), "sk.w1", dtype, 1 / math.sqrt(K)); w2, _ = rnd((N, K), "sk.w2", dtype, 1 / math.sqrt(K))
    b1 = det_tensor((N,), "sk.b1", 7, 0.5).to(dev()); b2 = det_tensor((N,), "sk.b2", 7, 0.5).to(dev())
    resx, _ = rnd((rows_x, N), "sk.rx", dtype); resc, _ = rnd((rows_c, N), "sk.rc", dtype)
    rps_x, rps_c = 1029, 41
    rsx = (det_tensor(((rows_x + rps_x - 1) // rps_x,), "sk.sx", 7).abs() + 0.5).to(dev())
    rsc = (det_tensor(((rows_c + rps_c - 1) // rps_c,), "sk.sc", 7).abs() + 0.5).to(dev())
    ox = torch.empty((rows_x, N), device=dev(), dtype=dtype); oc = torch.empty((rows_c, N), device=dev(), dtype=dtype)
    o.linear_fwd([o.Prob(ax, w1, ox, bias=b1, res=resx, row_scale=rsx, rps=rps_x), o.Prob(ac, w2, oc, bias=b2, res=resc, row_scale=rsc, rps=rps_c)], N, K)
    for a, w, b, res, rs, rps, out, what in [(ax, w1, b1, resx, rsx, rps_x, ox, "x"), (ac, w2, b2, resc, rsc, rps_c, oc, "c")]:
        rows = a.shape[0]
        sel = torch.cat([torch.arange(0, min(rows, 300)), torch.arange(max(rows - 300, 0), rows), torch.arange(0, rows, 997)]).unique()
        u = a[sel].double().cpu() @ w.double().cpu().t() + b.double().cpu()
        ref = res[sel].double().cpu() + rs.double().cpu()[sel // rps][:, None] * u
        assert_close(out[sel], ref, dtype, "skinny " + what)
    # plain (no epilogue operands), single problem
    o.linear_fwd([o.Prob(ax, w1, ox)], N, K)
    sel = torch.arange(0, rows_x, 1013)
    assert_close(ox[sel], ax[sel].double().cpu() @ w1.double().cpu().t(), dtype, "skinny plain")


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 48, 96, 112, 112), (2, 96, 192, 56, 56), (2, 192, 384, 28, 28), (3, 384, 512, 14, 14), (1, 64, 128, 9, 7), (2, 8, 16, 5, 6)])
def test_conv3x3s2_native(dtype, B, Ci, Co, H, W):
    """Row f1: the second stem convolution and the three stage transitions of LeMeViT-Base (models/lemevit.py:701-703, :714-717) as
    im2col + the block GEMM -- forward, data gradient (col2im gather) and weight / bias gradient vs float64 F.conv2d on the same
    (rounded) operands; odd map sizes exercise the padding and the ragged last output row / column."""
    from lemevit_amd.model import _Conv3x3s2Fn
    x, x64 = rnd((B, Ci, H, W), "cx", dtype); w32 = det_tensor((Co, Ci, 3, 3), "cw", 7, 1.0 / math.sqrt(9 * Ci)); b32 = det_tensor((Co,), "cb", 7, 0.3)
    wq = w32.to(dtype).double() if dtype == torch.bfloat16 else w32.double()
    xg = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = w32.to(dev()).requires_grad_(True); bg = b32.to(dev()).requires_grad_(True)
    y = _Conv3x3s2Fn.apply(xg, wg, bg, dtype)
    x64 = x64.requires_grad_(True); w64 = wq.clone().requires_grad_(True); b64 = b32.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(x64, w64, b64, stride=2, padding=1)
    assert_close(y, ref.detach(), dtype, "conv fwd")
    gy, gy64 = rnd(tuple(ref.shape), "cgy", dtype)
    (y.float() * gy.float()).sum().backward()
    (ref * gy64).sum().backward()
    # dX = col2im(dY @ Wm): the GEMM stores its [rows, 9 Cin] result in bf16 and the gather then sums up to FOUR of those rounded
    # partial sums per input pixel, so the bf16 bound is 4 operand roundings (4 * 2^-9 of the largest term) instead of one output
    # rounding; measured worst case on MI355X: 1.6e-3 of max-abs beyond the single-rounding bound (Base stem shape).
    assert_close(xg.grad, x64.grad, dtype, "conv dx", tol16=4e-3)
    assert_close(wg.grad, w64.grad, torch.float32, "conv dw", tol32=2e-5 if dtype == torch.float32 else 2e-5 * 4)
    assert_close(bg.grad, b64.grad, torch.float32, "conv db", tol32=2e-5)


@pytest.mark.parametrize("B,Ci,Co,H,W", [(2, 48, 96, 112, 112), (8, 96, 192, 56, 56), (16, 192, 384, 28, 28), (64, 384, 512, 14, 14), (64, 64, 128, 9, 7), (64, 8, 16, 5, 6),
                                         (128, 24, 40, 3, 3)])
def test_conv3x3s2_implicit(B, Ci, Co, H, W):
    """Round 6: the same convolutions as an implicit GEMM (lmv_conv3x3s2_fwd / _dw: the LDS-DMA loads gather the patch elements from the NHWC map; no patch matrix).  Forward and
    weight / bias gradient against float64 F.conv2d on the rounded operands, and BIT-identical to the patch-matrix form (the same panel images reach the same kernel); odd maps
    exercise the padding taps, the ragged last row / column and the zero columns behind 9 Cin (Cin = 8, 24: several taps per 16-byte-chunk row; Cin = 24: taps straddle k-tiles)."""
    import lemevit_amd.model as M
    from lemevit_amd.model import _Conv3x3s2Fn
    dtype = torch.bfloat16
    x, x64 = rnd((B, Ci, H, W), "cix", dtype); w32 = det_tensor((Co, Ci, 3, 3), "ciw", 7, 1.0 / math.sqrt(9 * Ci)); b32 = det_tensor((Co,), "cib", 7, 0.3)
    wq = w32.to(dtype).double()
    res = {}
    for implicit in (True, False):
        was, M._CONV_IMPLICIT = M._CONV_IMPLICIT, implicit
        try:
            xg = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
            wg = w32.to(dev()).requires_grad_(True); bg = b32.to(dev()).requires_grad_(True)
            y = _Conv3x3s2Fn.apply(xg, wg, bg, dtype)
            assert ops().conv3x3s2_implicit_ok(xg.detach().permute(0, 2, 3, 1), Co, (9 * Ci + 63) // 64 * 64), "the shape list is meant to take the implicit form"
            gy, gy64 = rnd(tuple(y.shape), "cigy", dtype)
            (y.float() * gy.float()).sum().backward()
            res[implicit] = (y.detach().clone(), xg.grad.clone(), wg.grad.clone(), bg.grad.clone())
        finally:
            M._CONV_IMPLICIT = was
    x64 = x64.requires_grad_(True); w64 = wq.clone().requires_grad_(True); b64 = b32.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(x64, w64, b64, stride=2, padding=1)
    (ref * gy64).sum().backward()
    y, dx, dw, db = res[True]
    assert_close(y, ref.detach(), dtype, "implicit conv fwd")
    assert_close(dw, w64.grad, torch.float32, "implicit conv dw", tol32=2e-5 * 4)
    assert_close(db, b64.grad, torch.float32, "implicit conv db", tol32=2e-5)
    assert_close(dx, x64.grad, dtype, "implicit conv dx", tol16=4e-3)
    for a, b, what in zip(res[True], res[False], ("y", "dx", "dw", "db")):
        assert torch.equal(a, b), f"implicit vs patch-matrix form: {what} differs by {float((a.float() - b.float()).abs().max()):.3e}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_bwd_two_call_form(dtype):
    """lmv_layernorm_bwd_partial + lmv_layernorm_bwd_reduce (the native block scheduler runs the reduce on its side stream) are
    bit-identical to lmv_layernorm_bwd: dx and the accumulated dgamma / dbeta."""
    o = ops()
    C_, rows_x, rows_c = 192, 5000, 48
    g = (1.0 + 0.2 * det_tensor((C_,), "g", 7)).to(dev()); b = det_tensor((C_,), "b", 7, 0.1).to(dev())
    xs = [rnd((rows_x, C_), "x", dtype)[0].view(1, rows_x, C_), rnd((rows_c, C_), "c", dtype)[0].view(1, rows_c, C_)]
    dys = [rnd((rows_x, C_), "dy", dtype)[0].view(1, rows_x, C_), rnd((rows_c, C_), "dyc", dtype)[0].view(1, rows_c, C_)]
    _, st = o.layernorm_fwd_multi(xs, g, b, 1e-6, want_stats=True)
    outs = []
    for split in (False, True):
        dg = torch.full((C_,), 0.25, device=dev()); db = torch.full((C_,), -0.5, device=dev())
        dxs = o.layernorm_bwd_multi(dys, xs, st, g, dg, db, [None, None], split_reduce=split)
        outs.append([t.clone() for t in dxs] + [dg, db])
    for a, c in zip(*outs):
        assert torch.equal(a, c)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,rows_c,N,K", [(27136 // 8, 256, 1536, 384), (1000, 48, 576, 192), (129, 0, 64, 384), (4133, 16, 768, 192), (6400, 0, 1152, 384),
                                             (300, 300, 128, 192)])
def test_linear_rs_kernel(rows, rows_c, N, K):
    """csrc/rsgemm.hip (register-stationary token panel GEMM, bf16, K = 192 / 384, N % 64 == 0) forced on for every shape it can run
    (lmv_config_set("gemm_rs", 2)): plain / bias, GELU with and without the pre-activation copy, residual + DropPath scale, and the GELU'
    epilogue of the dX-through-transposed-weight form -- ragged row counts, a second problem with its OWN weight and bias (the qkv1 / qkv2
    launch of a D block), against float64 math on the operands the kernel reads; and against the 128 x 128 tile kernel on the same inputs."""
    from lemevit_amd import _lib
    o = ops()
    dtype = torch.bfloat16
    a, a64 = rnd((rows, K), "a", dtype); w, w64 = rnd((N, K), "w", dtype, 1 / math.sqrt(K))
    bias = det_tensor((N,), "b", 7, 0.5).to(dev()); res, res64 = rnd((rows, N), "res", dtype)
    uu, uu64 = rnd((rows, N), "u", dtype, 2.0)
    rps = 7
    rs = (det_tensor(((rows + rps - 1) // rps,), "rs", 7).abs() + 0.5).to(dev())
    rs_rows = rs.cpu().double()[torch.arange(rows) // rps][:, None]
    if rows_c:
        ac, ac64 = rnd((rows_c, K), "ac", dtype); wc, wc64 = rnd((N, K), "wc", dtype, 1 / math.sqrt(K))
        bc = det_tensor((N,), "bc", 7, 0.5).to(dev()); resc, resc64 = rnd((rows_c, N), "resc", dtype); uc, uc64 = rnd((rows_c, N), "uc", dtype, 2.0)
    u = a64 @ w64.t() + bias.cpu().double()
    outs = {}
    try:
        for mode in (2, 0):
            _lib.config_set("gemm_rs", mode)
            got = {}
            out = torch.zeros((rows, N), device=dev(), dtype=dtype); pre = torch.zeros_like(out)
            oc = torch.zeros((max(rows_c, 1), N), device=dev(), dtype=dtype); prec = torch.zeros_like(oc)

            def probs(**kw):
                ps = [o.Prob(a, w, out, **{k: v[0] for k, v in kw.items()})]
                if rows_c:
                    ps.append(o.Prob(ac, wc, oc, **{k: v[1] for k, v in kw.items()}))
                return ps
            o.linear_fwd(probs(), N, K)
            got["plain"] = (out.clone(), oc.clone())
            o.linear_fwd(probs(bias=(bias, bc if rows_c else None)), N, K)
            got["bias"] = (out.clone(), oc.clone())
            o.linear_fwd(probs(bias=(bias, bc if rows_c else None)), N, K, o.ACT_GELU)
            got["gelu"] = (out.clone(), oc.clone())
            o.linear_fwd(probs(bias=(bias, bc if rows_c else None), out_pre=(pre, prec)), N, K, o.ACT_GELU)
            got["gelu+pre"] = (out.clone(), oc.clone()); got["pre"] = (pre.clone(), prec.clone())
            o.linear_fwd(probs(bias=(bias, bc if rows_c else None), res=(res, resc if rows_c else None), row_scale=(rs, None), rps=(rps, 1)), N, K)
            got["res"] = (out.clone(), oc.clone())
            o.linear_fwd(probs(aux=(uu, uc if rows_c else None), row_scale=(rs, None), rps=(rps, 1)), N, K, o.ACT_GELU_GRAD)
            got["ggrad"] = (out.clone(), oc.clone())
            outs[mode] = got
    finally:
        _lib.config_set("gemm_rs", 1)
    ref = {"plain": a64 @ w64.t(), "bias": u, "gelu": gelu64(u), "gelu+pre": gelu64(u), "pre": u, "res": res64 + rs_rows * u,
           "ggrad": (a64 @ w64.t()) * gelu_grad64(uu64) * rs_rows}
    if rows_c:
        ucc = ac64 @ wc64.t() + bc.cpu().double()
        refc = {"plain": ac64 @ wc64.t(), "bias": ucc, "gelu": gelu64(ucc), "gelu+pre": gelu64(ucc), "pre": ucc, "res": resc64 + ucc,
                "ggrad": (ac64 @ wc64.t()) * gelu_grad64(uc64)}
    for k, r in ref.items():
        for mode in (2, 0):
            assert_close(outs[mode][k][0], r, dtype, f"{k} (gemm_rs={mode})")
            if rows_c:
                assert_close(outs[mode][k][1], refc[k], dtype, f"{k} meta rows (gemm_rs={mode})")
        # both kernels round the same fp32 accumulations: they may differ by the summation order only
        d = float((outs[2][k][0].float() - outs[0][k][0].float()).abs().max()); m = float(r.abs().max())
        assert d <= 8e-3 * m, f"{k}: RS and tile kernels differ by {d:.3e} (max-abs {m:.3e})"


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("rows,rows_c,K", [(27136 // 8, 256, 1536), (1000, 48, 384), (129, 0, 64), (4133, 16, 1152), (300, 300, 128), (128, 0, 2048)])
def test_linear_wn_kernel(rows, rows_c, K):
    """csrc/wngemm.hip (one workgroup per 128-row token panel and all 384 output columns, bf16, K % 64 == 0) forced on for every shape it
    can run (lmv_config_set("gemm_wn", 2)): plain, bias, residual + DropPath scale -- ragged row counts, a second problem with its OWN
    weight and bias, against float64 math on the operands the kernel reads; and against the 128 x 128 tile kernel on the same inputs."""
    from lemevit_amd import _lib
    o = ops()
    dtype, N = torch.bfloat16, 384
    a, a64 = rnd((rows, K), "a", dtype); w, w64 = rnd((N, K), "w", dtype, 1 / math.sqrt(K))
    bias = det_tensor((N,), "b", 7, 0.5).to(dev()); res, res64 = rnd((rows, N), "res", dtype)
    rps = 7
    rs = (det_tensor(((rows + rps - 1) // rps,), "rs", 7).abs() + 0.5).to(dev())
    rs_rows = rs.cpu().double()[torch.arange(rows) // rps][:, None]
    if rows_c:
        ac, ac64 = rnd((rows_c, K), "ac", dtype); wc, wc64 = rnd((N, K), "wc", dtype, 1 / math.sqrt(K))
        bc = det_tensor((N,), "bc", 7, 0.5).to(dev()); resc, resc64 = rnd((rows_c, N), "resc", dtype)
    u = a64 @ w64.t() + bias.cpu().double()
    outs = {}
    try:
        for mode in (2, 0):
            _lib.config_set("gemm_wn", mode)
            got = {}
            out = torch.zeros((rows, N), device=dev(), dtype=dtype)
            oc = torch.zeros((max(rows_c, 1), N), device=dev(), dtype=dtype)

            def probs(**kw):
                ps = [o.Prob(a, w, out, **{k: v[0] for k, v in kw.items()})]
                if rows_c:
                    ps.append(o.Prob(ac, wc, oc, **{k: v[1] for k, v in kw.items()}))
                return ps
            o.linear_fwd(probs(), N, K)
            got["plain"] = (out.clone(), oc.clone())
            o.linear_fwd(probs(bias=(bias, bc if rows_c else None)), N, K)
            got["bias"] = (out.clone(), oc.clone())
            o.linear_fwd(probs(bias=(bias, bc if rows_c else None), res=(res, resc if rows_c else None), row_scale=(rs, None), rps=(rps, 1)), N, K)
            got["res"] = (out.clone(), oc.clone())
            outs[mode] = got
    finally:
        _lib.config_set("gemm_wn", 1)
    ref = {"plain": a64 @ w64.t(), "bias": u, "res": res64 + rs_rows * u}
    if rows_c:
        ucc = ac64 @ wc64.t() + bc.cpu().double()
        refc = {"plain": ac64 @ wc64.t(), "bias": ucc, "res": resc64 + ucc}
    for k, r in ref.items():
        for mode in (2, 0):
            assert_close(outs[mode][k][0], r, dtype, f"{k} (gemm_wn={mode})")
            if rows_c:
                assert_close(outs[mode][k][1], refc[k], dtype, f"{k} meta rows (gemm_wn={mode})")
        d = float((outs[2][k][0].float() - outs[0][k][0].float()).abs().max()); m = float(r.abs().max())
        assert d <= 8e-3 * m, f"{k}: whole-width and tile kernels differ by {d:.3e} (max-abs {m:.3e})"


@pytest.mark.parametrize("rows,rows_c,N", [(27136 // 8, 256, 1536), (1000, 48, 1152), (129, 0, 64), (4133, 16, 384)])
def test_linear_dx_ln_bwd_fused(rows, rows_c, N):
    """lmv_linear_dx_ln_bwd (csrc/wngemm.hip, bf16, C = 384): dx = dres + LN'(dY @ wt^T) with dy never written -- against float64 math on the
    operands the kernel reads (dy stays fp32 inside the kernel, so it is compared with the UNROUNDED product), with and without the
    residual-path gradient, with the DropPath-scaled second output, a second problem (meta-token rows), dgamma / dbeta accumulated on top of
    existing values; and against the two-launch form (forward-form GEMM on wt, then lmv_layernorm_bwd) on the same inputs."""
    o = ops()
    dtype, C = torch.bfloat16, 384
    wt, wt64 = rnd((C, N), "wt", dtype, 1 / math.sqrt(N))
    gamma = (det_tensor((C,), "gam", 5, 0.3) + 1.0).to(dev())
    g64 = gamma.cpu().double()
    rps = 7
    sets = []
    for tag, r in (("x", rows), ("c", rows_c)):
        if not r:
            continue
        dy, dy64 = rnd((r, N), "dy" + tag, dtype); x, x64 = rnd((r, C), "x" + tag, dtype, 1.5); dres, dres64 = rnd((r, C), "dr" + tag, dtype)
        mean = x64.mean(1, keepdim=True); var = ((x64 - mean) ** 2).mean(1, keepdim=True); rstd = 1 / torch.sqrt(var + 1e-6)
        stats = torch.cat([mean, rstd], 1).float().to(dev()).contiguous()
        m64, r64 = stats.cpu().double()[:, :1], stats.cpu().double()[:, 1:]
        sc = (det_tensor(((r + rps - 1) // rps,), "sc" + tag, 7).abs() + 0.5).to(dev())
        sets.append(dict(dy=dy, dy64=dy64, x=x.view(1, r, C), x64=x64, dres=dres.view(1, r, C), dres64=dres64, stats=stats, m=m64, r=r64, sc=sc,
                         scr=sc.cpu().double()[torch.arange(r) // rps][:, None]))
    for with_res in (True, False):
        dgam = torch.full((C,), 0.25, device=dev()); dbet = torch.full((C,), -0.5, device=dev())
        dgam2, dbet2 = dgam.clone(), dbet.clone()
        # (the scaled output indexes rows by x.shape[1] = rows per sample: give the wrapper [samples, rps, C] views where the row count allows)
        xs = [t["x"] for t in sets]
        dxs = o.linear_dx_ln_bwd([t["dy"] for t in sets], wt, xs, [t["stats"] for t in sets], gamma, dgam, dbet, [t["dres"] if with_res else None for t in sets])
        # two-launch form
        dns = []
        for t in sets:
            dn = torch.empty((t["dy"].shape[0], C), device=dev(), dtype=dtype)
            o.linear_fwd([o.Prob(t["dy"], wt, dn)], C, N)
            dns.append(dn.view(1, -1, C))
        dxs2 = o.layernorm_bwd_multi(dns, xs, [t["stats"] for t in sets], gamma, dgam2, dbet2, [t["dres"] if with_res else None for t in sets])
        rg, rb = torch.full((C,), 0.25, dtype=torch.float64), torch.full((C,), -0.5, dtype=torch.float64)
        for t, dx, dx2 in zip(sets, dxs, dxs2):
            dyv = t["dy64"] @ wt64.t()
            xh = (t["x64"] - t["m"]) * t["r"]
            gg = dyv * g64
            ref = t["r"] * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True)) + (t["dres64"] if with_res else 0)
            rg += (dyv * xh).sum(0); rb += dyv.sum(0)
            assert_close(dx.view(-1, C), ref, dtype, f"dx (res={with_res})")
            d = float((dx.float() - dx2.float()).abs().max()); m = float(ref.abs().max())
            assert d <= 1.6e-2 * m, f"fused and two-launch forms differ by {d:.3e} (max-abs {m:.3e})"      # the two-launch form rounds dy to bf16 first
        for got, ref, what in ((dgam, rg, "dgamma"), (dbet, rb, "dbeta")):
            err = float((got.cpu().double() - ref).abs().max()) / float(ref.abs().max())
            assert err <= 2e-3, f"{what}: {err:.2e}"
    # the DropPath-scaled second output (rows per sample = rps): through the raw segments
    from lemevit_amd import _lib
    import ctypes as CT
    t = sets[0]
    r = t["dy"].shape[0]
    seg = (_lib.LnSegment * 1)(); pr = (_lib.LinearProblem * 1)()
    dx = torch.empty((r, C), device=dev(), dtype=dtype); dxs_ = torch.empty_like(dx)
    seg[0].x, seg[0].stats, seg[0].dres, seg[0].dx, seg[0].rows = t["x"].data_ptr(), t["stats"].data_ptr(), t["dres"].data_ptr(), dx.data_ptr(), r
    seg[0].dx_scale, seg[0].dx_scaled, seg[0].rows_per_sample = t["sc"].data_ptr(), dxs_.data_ptr(), rps
    pr[0].a, pr[0].w, pr[0].rows = t["dy"].data_ptr(), wt.data_ptr(), r
    ws = torch.empty(_lib.lib.lmv_linear_dx_ln_bwd_workspace_bytes(r, C), device=dev(), dtype=torch.uint8)
    nrows = CT.c_int(0)
    _lib.check(_lib.lib.lmv_linear_dx_ln_bwd(pr, seg, 1, C, N, gamma.data_ptr(), ws.data_ptr(), ws.numel(), CT.byref(nrows), _lib.LMV_BF16, o._stream()), "lmv_linear_dx_ln_bwd")
    torch.cuda.synchronize()
    assert nrows.value == (r + 127) // 128
    dyv = t["dy64"] @ wt64.t(); xh = (t["x64"] - t["m"]) * t["r"]; gg = dyv * g64
    ref = t["r"] * (gg - gg.mean(1, keepdim=True) - xh * (gg * xh).mean(1, keepdim=True)) + t["dres64"]
    assert_close(dx, ref, dtype, "dx (raw segments)")
    assert_close(dxs_, ref * t["scr"], dtype, "dx_scaled")      # scaled in fp32, rounded once


@pytest.mark.parametrize("rows,rows_c,K", [(27136 // 8, 256, 384), (1000, 48, 1536), (129, 0, 64)])
def test_linear_res_ln_fwd_fused(rows, rows_c, K):
    """lmv_linear_res_ln_fwd (csrc/wngemm.hip, bf16, N = 384): out = res + s (a W^T + b) and y = LayerNorm(out) in one launch -- `out` against
    float64 math, `y` and the (mean, rstd) rows against the float64 LayerNorm of the kernel's own ROUNDED `out` (what a separate launch
    would normalise), with a second problem that has its own weight; and `y` against lmv_layernorm_fwd on the same `out`."""
    o = ops()
    dtype, N, eps = torch.bfloat16, 384, 1e-6
    gamma = (det_tensor((N,), "gam", 5, 0.3) + 1.0).to(dev()); beta = det_tensor((N,), "bet", 5, 0.2).to(dev())
    rps = 7
    probs, refs = [], []
    for tag, r in (("x", rows), ("c", rows_c)):
        if not r:
            continue
        a, a64 = rnd((r, K), "a" + tag, dtype); w, w64 = rnd((N, K), "w" + tag, dtype, 1 / math.sqrt(K)); res, res64 = rnd((r, N), "res" + tag, dtype)
        bias = det_tensor((N,), "b" + tag, 7, 0.5).to(dev())
        sc = (det_tensor(((r + rps - 1) // rps,), "sc" + tag, 7).abs() + 0.5).to(dev())
        out = torch.zeros((r, N), device=dev(), dtype=dtype)
        probs.append(o.Prob(a, w, out, bias=bias, res=res, row_scale=sc, rps=rps))
        refs.append(res64 + sc.cpu().double()[torch.arange(r) // rps][:, None] * (a64 @ w64.t() + bias.cpu().double()))
    ys, sts = o.linear_res_ln_fwd(probs, N, K, gamma, beta, eps, want_stats=True)
    for p, ref, y, st in zip(probs, refs, ys, sts):
        assert_close(p.out, ref, dtype, "out")
        o64 = p.out.float().cpu().double()
        mean = o64.mean(1, keepdim=True); var = ((o64 - mean) ** 2).mean(1, keepdim=True); rstd = 1 / torch.sqrt(var + eps)
        assert_close(y, (o64 - mean) * rstd * gamma.cpu().double() + beta.cpu().double(), dtype, "LayerNorm(out)")
        assert float((st[:, 0].cpu().double() - mean[:, 0]).abs().max()) <= 1e-5 * float(o64.abs().max())
        assert float((st[:, 1].cpu().double() / rstd[:, 0] - 1).abs().max()) <= 1e-4
        y2, _ = o.layernorm_fwd(p.out.view(1, -1, N), gamma, beta, eps)
        d = float((y.float() - y2.view(-1, N).float()).abs().max()); m = float(y2.float().abs().max())
        assert d <= 8e-3 * m, f"fused and separate LayerNorm differ by {d:.3e} (max-abs {m:.3e})"      # one bf16 ulp where the two round differently


def test_whole_width_kernels_under_load():
    """Race screen for csrc/wngemm.hip (hand-counted vmcnt, requests split by wave, LDS reuse in the fused epilogues) and csrc/rsgemm.hip: their entry
    points at the full stage-3 shape of config 3, once on an idle chip and then 8 times while a second stream streams 1 GB copies and runs GEMMs -- every
    output must be BIT-identical to the idle run (the kernels reduce in a fixed order; an early LDS read or a short wait shows up as a mismatch
    that comes and goes with timing)."""
    o = ops()
    dtype, C, rows, rows_c, Hd = torch.bfloat16, 384, 25088, 2048, 1536
    a, _ = rnd((rows, Hd), "la", dtype); ac, _ = rnd((rows_c, Hd), "lac", dtype)
    w, _ = rnd((C, Hd), "lw", dtype, 1 / math.sqrt(Hd)); bias = det_tensor((C,), "lb", 7, 0.5).to(dev())
    res, _ = rnd((rows, C), "lres", dtype); resc, _ = rnd((rows_c, C), "lresc", dtype)
    gamma = (det_tensor((C,), "lg", 5, 0.3) + 1.0).to(dev()); beta = det_tensor((C,), "lbt", 5, 0.2).to(dev())
    sc = (det_tensor((128,), "lsc", 7).abs() + 0.5).to(dev())
    x3, _ = rnd((128, 196, C), "lx", dtype, 1.5); c3, _ = rnd((128, 16, C), "lc", dtype, 1.5)
    stx = torch.stack([x3.float().mean(-1).reshape(-1), 1 / torch.sqrt(x3.float().var(-1, unbiased=False).reshape(-1) + 1e-6)], 1).contiguous()
    stc = torch.stack([c3.float().mean(-1).reshape(-1), 1 / torch.sqrt(c3.float().var(-1, unbiased=False).reshape(-1) + 1e-6)], 1).contiguous()

    def run():
        out, outc = torch.empty((rows, C), device=dev(), dtype=dtype), torch.empty((rows_c, C), device=dev(), dtype=dtype)
        probs = [o.Prob(a, w, out, bias=bias, res=res, row_scale=sc, rps=196), o.Prob(ac, w, outc, bias=bias, res=resc, row_scale=sc, rps=16)]
        o.linear_fwd(probs, C, Hd)                                                   # residual epilogue (fc2 forward)
        r1 = (out.clone(), outc.clone())
        ys, sts = o.linear_res_ln_fwd(probs, C, Hd, gamma, beta, 1e-6, want_stats=True)      # + LayerNorm
        dg, db = torch.zeros(C, device=dev()), torch.zeros(C, device=dev())
        dxs, dxsc = o.linear_dx_ln_bwd([a, ac], w, [x3, c3], [stx, stc], gamma, dg, db, [res.view(128, 196, C), resc.view(128, 16, C)], next_scales=[sc, sc])
        # the register-stationary kernel (csrc/rsgemm.hip, counted waits of its own): fc1 forward with the pre-activation copy, GELU' dX of fc2
        h = torch.empty((rows, Hd), device=dev(), dtype=dtype); u = torch.empty_like(h); du = torch.empty_like(h)
        o.linear_fwd([o.Prob(res, w1, h, bias=b1, out_pre=u)], Hd, C, o.ACT_GELU)
        o.linear_fwd([o.Prob(res, w1, du, aux=a, row_scale=sc, rps=196)], Hd, C, o.ACT_GELU_GRAD)
        return [*r1, out, outc, *ys, *sts, *dxs, *dxsc, dg, db, h, u, du]

    w1, _ = rnd((Hd, C), "lw1", dtype, 1 / math.sqrt(C)); b1 = det_tensor((Hd,), "lb1", 7, 0.5).to(dev())
    idle = [t.clone() for t in run()]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.empty(256 << 20, device=dev(), dtype=torch.float32); big2 = torch.empty_like(big)
    ga = torch.randn(4096, 4096, device=dev(), dtype=dtype); gb = torch.randn(4096, 4096, device=dev(), dtype=dtype)
    for it in range(8):
        with torch.cuda.stream(side):
            for _ in range(3):
                big2.copy_(big); ga @ gb
        got = run()
        torch.cuda.synchronize()
        for i, (g, r) in enumerate(zip(got, idle)):
            assert torch.equal(g, r), f"round {it}: output {i} differs from the idle run by {float((g.float() - r.float()).abs().max()):.3e}"


@pytest.mark.parametrize("rows,rows_c,N", [(3136 * 4, 64, 288), (1000, 0, 192), (33, 48, 96), (4133, 16, 384)])
def test_ln_linear_exact_fused(rows, rows_c, N):
    """lmv_ln_linear_exact_fwd (csrc/rswgemm.hip, bf16, K = 96: the weight resident in LDS, LayerNorm on the register-resident rows): the
    LayerNorm output and its (mean, rstd) against float64, y against float64 math on the kernel's own ROUNDED LayerNorm output (what the
    two-launch form multiplies), a second problem with its own weight (qkv1 / qkv2 of a D block), ragged row counts; and y against
    lmv_layernorm_fwd + lmv_linear_fwd on the same inputs."""
    o = ops()
    dtype, K, eps = torch.bfloat16, 96, 1e-6
    gamma = (det_tensor((K,), "gam", 5, 0.3) + 1.0).to(dev()); beta = det_tensor((K,), "bet", 5, 0.2).to(dev())
    probs, x64s = [], []
    for tag, r in (("x", rows), ("c", rows_c)):
        if not r:
            continue
        x, x64 = rnd((r, K), "x" + tag, dtype, 1.5); w, _ = rnd((N, K), "w" + tag, dtype, 1 / math.sqrt(K)); bias = det_tensor((N,), "b" + tag, 7, 0.5).to(dev())
        probs.append(o.Prob(x, w, torch.zeros((r, N), device=dev(), dtype=dtype), bias=bias)); x64s.append(x64)
    ys, sts = o.ln_linear_exact_fwd(probs, N, K, gamma, beta, eps, want_stats=True)
    for p, x64, y, st in zip(probs, x64s, ys, sts):
        mean = x64.mean(1, keepdim=True); var = ((x64 - mean) ** 2).mean(1, keepdim=True); rstd = 1 / torch.sqrt(var + eps)
        assert_close(y, (x64 - mean) * rstd * gamma.cpu().double() + beta.cpu().double(), dtype, "LayerNorm output")
        assert float((st[:, 0].cpu().double() - mean[:, 0]).abs().max()) <= 1e-5 * float(x64.abs().max())
        assert float((st[:, 1].cpu().double() / rstd[:, 0] - 1).abs().max()) <= 1e-4
        ref = y.float().cpu().double() @ p.w.float().cpu().double().t() + p.bias.cpu().double()
        assert_close(p.out, ref, dtype, "LN(x) W^T + b")
        y2, _ = o.layernorm_fwd(p.a.view(1, -1, K), gamma, beta, eps)
        out2 = torch.empty_like(p.out)
        o.linear_fwd([o.Prob(y2.view(-1, K), p.w, out2, bias=p.bias)], N, K)
        d = float((p.out.float() - out2.float()).abs().max()); m = float(ref.abs().max())
        assert d <= 1.2e-2 * m, f"fused and two-launch forms differ by {d:.3e} (max-abs {m:.3e})"
