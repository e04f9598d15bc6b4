"""Parity of the fused block entry points (include/lemevit_hip.h: lmv_ln_fold, lmv_ln_linear_fwd, lmv_mlp_fused_fwd,
lmv_attn_out_proj_residual) on a real MI355X against float64 restatements of the reference math
(models/lemevit.py:526-530 MLP, :560-564 / :632-635 block halves; oracle.layer_norm / gelu_erf / linear).

Two references per case:
  * "kernel math": float64 on the SAME rounded operands the kernel reads (the folded weight as the kernel sees it) --
    north_star's budget: fp32 1e-5, bf16 1e-3 of the output's max-abs (+ one bf16 output rounding);
  * "reference math": LayerNorm -> Linear with the fp32 MASTER weights (what the reference module computes) -- wider
    (bf16 4e-3) because the fused path rounds gamma . W to bf16 where the unfused path rounds W (and LN(x)): the same
    one-rounding-per-operand budget, spent at a different place.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from detfill import det_tensor
from oracle import lemevit_oracle as O
from test_ops_gpu import assert_close, dev, gelu64, ops, rnd

DTYPES = [torch.float32, torch.bfloat16]
EPS = 1e-6


def _tokens(rows, C, name, dtype):
    """token rows with a per-row offset and scale (LayerNorm statistics that differ row to row)"""
    base = det_tensor((rows, C), name, 7)
    off = det_tensor((rows, 1), name + "o", 7) * 0.8
    sc = det_tensor((rows, 1), name + "s", 7).abs() + 0.5
    t = (base * sc + off).to(dtype)
    return t.to(dev()), t.to(torch.float64)


def _ln_params(C):
    g = 1.0 + 0.3 * det_tensor((C,), "g", 7)
    b = 0.2 * det_tensor((C,), "be", 7)
    return g, b


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,K", [(288, 96), (1536, 384), (64, 64)])
def test_ln_fold(dtype, N, K):
    o = ops()
    w = det_tensor((N, K), "w", 7, 1 / math.sqrt(K)); bias = det_tensor((N,), "b", 7, 0.5)
    g, be = _ln_params(K)
    F = o.ln_fold(w.to(dev()), bias.to(dev()), g.to(dev()), be.to(dev()), dtype)
    wf_ref = (w * g).to(dtype)                                   # fp32 product, one rounding -- bit-exact
    assert torch.equal(F.wf.cpu(), wf_ref)
    assert_close(F.colsum, wf_ref.double().sum(1), torch.float32, "colsum")
    assert_close(F.bf, bias.double() + w.double() @ be.double(), torch.float32, "bias'")
    F0 = o.ln_fold(w.to(dev()), None, g.to(dev()), be.to(dev()), dtype)
    assert_close(F0.bf, w.double() @ be.double(), torch.float32, "bias' (no bias)")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,N,K", [(300, 96, 64), (4096, 288, 96), (1000, 1152, 384), (2051, 1536, 512), (777, 384, 128), (129, 200, 72)])
def test_ln_linear_fwd(dtype, rows, N, K):
    o = ops()
    x, x64 = _tokens(rows, K, "x", dtype)
    w = det_tensor((N, K), "w", 7, 1 / math.sqrt(K)); bias = det_tensor((N,), "b", 7, 0.5)
    g, be = _ln_params(K)
    F = o.ln_fold(w.to(dev()), bias.to(dev()), g.to(dev()), be.to(dev()), dtype)
    xhat = O.layer_norm(x64, torch.ones(K, dtype=torch.float64), torch.zeros(K, dtype=torch.float64), EPS)
    ref_kernel = xhat @ F.wf.cpu().double().t() + F.bf.cpu().double()
    ref_module = O.linear(O.layer_norm(x64, g.double(), be.double(), EPS), w.double(), bias.double())
    out = torch.empty((rows, N), device=dev(), dtype=dtype)
    o.ln_linear_fwd([o.Prob(x, F.wf, out, bias=F.bf, aux=F.colsum)], N, K, EPS)
    assert_close(out, ref_kernel, dtype, "ln_linear (kernel math)")
    assert_close(out, ref_module, dtype, "ln_linear (reference math)", tol32=2e-5, tol16=4e-3)
    # GELU + pre-activation copy + residual + DropPath scale ride the same epilogue
    res, res64 = rnd((rows, N), "res", dtype)
    rps = 7
    rs = (det_tensor(((rows + rps - 1) // rps,), "rs", 7).abs() + 0.5).to(dev())
    pre = torch.empty_like(out)
    o.ln_linear_fwd([o.Prob(x, F.wf, out, bias=F.bf, aux=F.colsum, res=res, row_scale=rs, out_pre=pre, rps=rps)], N, K, EPS, o.ACT_GELU)
    sc = rs.cpu().double()[torch.arange(rows) // rps][:, None]
    assert_close(pre, ref_kernel, dtype, "ln_linear pre-activation")
    assert_close(out, res64 + sc * gelu64(ref_kernel), dtype, "ln_linear epilogue")


@pytest.mark.parametrize("dtype", DTYPES)
def test_ln_linear_dual(dtype):
    """image-token and meta-token rows through DIFFERENT folded weights in one launch (qkv1 | qkv2 of a D block)"""
    o = ops()
    N, K = 576, 192
    xs = [_tokens(1500, K, "x", dtype), _tokens(48, K, "c", dtype)]
    g, be = _ln_params(K)
    probs, refs, outs = [], [], []
    for i, (x, x64) in enumerate(xs):
        w = det_tensor((N, K), f"w{i}", 7, 1 / math.sqrt(K)); bias = det_tensor((N,), f"b{i}", 7, 0.5)
        F = o.ln_fold(w.to(dev()), bias.to(dev()), g.to(dev()), be.to(dev()), dtype)
        out = torch.empty((x.shape[0], N), device=dev(), dtype=dtype)
        probs.append(o.Prob(x, F.wf, out, bias=F.bf, aux=F.colsum)); outs.append(out)
        xhat = O.layer_norm(x64, torch.ones(K, dtype=torch.float64), torch.zeros(K, dtype=torch.float64), EPS)
        refs.append(xhat @ F.wf.cpu().double().t() + F.bf.cpu().double())
    o.ln_linear_fwd(probs, N, K, EPS)
    for out, ref, nm in zip(outs, refs, "xc"):
        assert_close(out, ref, dtype, "dual " + nm)


def _mlp_case(C, Hd, dtype=torch.bfloat16):
    o = ops()
    w1 = det_tensor((Hd, C), "w1", 7, 1 / math.sqrt(C)); b1 = det_tensor((Hd,), "b1", 7, 0.3)
    w2 = det_tensor((C, Hd), "w2", 7, 1 / math.sqrt(Hd)); b2 = det_tensor((C,), "b2", 7, 0.3)
    g, be = _ln_params(C)
    F = o.ln_fold(w1.to(dev()), b1.to(dev()), g.to(dev()), be.to(dev()), dtype)
    w2d = w2.to(dtype)
    return o, F, (w1, b1, w2, b2, g, be), w2d.to(dev()), b2.to(dev())


def _mlp_refs(x64, F, w2r64, b2, masters, scale_rows):
    """(kernel math with the hidden rounded to bf16 as the kernel feeds it to fc2, plain float64 reference math)"""
    w1, b1, w2, _, g, be = masters
    C = x64.shape[-1]
    xhat = O.layer_norm(x64, torch.ones(C, dtype=torch.float64), torch.zeros(C, dtype=torch.float64), EPS)
    u = xhat @ F.wf.cpu().double().t() + F.bf.cpu().double()
    h = gelu64(u).to(torch.bfloat16).double()
    ref_kernel = x64 + scale_rows * (h @ w2r64.t() + b2.double())
    y = O.layer_norm(x64, g.double(), be.double(), EPS)
    ref_module = x64 + scale_rows * O.linear(O.gelu_erf(O.linear(y, w1.double(), b1.double())), w2.double(), b2.double())
    return ref_kernel, ref_module


@pytest.mark.parametrize("tm", [0, 128])
@pytest.mark.parametrize("C", [64, 96, 128, 192, 256, 320, 384])
def test_mlp_fused_fwd(C, tm, monkeypatch, request):
    """x + fc2(GELU(fc1(LN(x)))) in one kernel: image tokens (ragged row count) + meta tokens in one launch, with / without DropPath;
    tm = 0: the library's choice of rows per workgroup (64 where two workgroups fit a CU), 128: the one-workgroup-per-CU form"""
    from lemevit_amd import _lib
    _lib.config_set("mlp_tm", tm)
    request.addfinalizer(lambda: _lib.config_set("mlp_tm", 0))
    dtype = torch.bfloat16
    Hd = 4 * C
    o, F, masters, w2, b2 = _mlp_case(C, Hd)
    assert o.mlp_fused_supported(C, Hd, dtype)
    B, N, M = 3, 331, 16
    x, x64 = _tokens(B * N, C, "x", dtype); c, c64 = _tokens(B * M, C, "c", dtype)
    x, c = x.view(B, N, C), c.view(B, M, C)
    for scales in (None, [(det_tensor((B,), "sx", 7).abs() + 0.5).to(dev()), None]):
        ox, oc = o.mlp_fused_fwd([x, c], F, w2, b2, EPS, scales)
        for out, t64, L, sc, nm in ((ox, x64, N, scales[0] if scales else None, "x"), (oc, c64, M, None, "c")):
            rows = t64.shape[0]
            sr = torch.ones(rows, 1, dtype=torch.float64) if sc is None else sc.cpu().double()[torch.arange(rows) // L][:, None]
            rk, rm = _mlp_refs(t64, F, w2.cpu().double(), b2.cpu(), masters, sr)
            assert_close(out.view(rows, C), rk, dtype, f"mlp_fused C={C} {nm} (kernel math)")
            assert_close(out.view(rows, C), rm, dtype, f"mlp_fused C={C} {nm} (reference math)", tol16=4e-3)


@pytest.mark.parametrize("C,B,N", [(96, 128, 3136), (192, 128, 784), (384, 128, 196)])
def test_mlp_fused_full_shapes(C, B, N):
    """the Base 224^2 stage-1 / 2 / 3 token matrices at batch 128 (config 3's shapes); float64 reference on the GPU (torch matmul)"""
    dtype = torch.bfloat16
    Hd = 4 * C
    o, F, masters, w2, b2 = _mlp_case(C, Hd)
    w1, b1, w2m, b2m, g, be = masters
    torch.manual_seed(0)
    x = (torch.randn(B, N, C, device=dev()) * (torch.rand(B, N, 1, device=dev()) + 0.5) + 0.5 * torch.randn(B, N, 1, device=dev())).to(dtype)
    c = torch.randn(B, 16, C, device=dev()).to(dtype)
    ox, oc = o.mlp_fused_fwd([x, c], F, w2, b2, EPS)
    for out, t in ((ox, x), (oc, c)):
        t64 = t.double().view(-1, C)
        mu = t64.mean(1, keepdim=True); xh = (t64 - mu) * torch.rsqrt(t64.var(1, unbiased=False, keepdim=True) + EPS)
        u = xh @ F.wf.double().t() + F.bf.double()
        h = (0.5 * u * (1 + torch.erf(u / math.sqrt(2)))).to(dtype).double()
        ref = t64 + h @ w2.double().t() + b2.double()
        assert_close(out.view(-1, C), ref.cpu(), dtype, f"mlp_fused full C={C}")
    # the unfused schedule (LayerNorm -> fc1 + GELU -> fc2 + residual launches) on the same inputs: same answer within two bf16 budgets
    xs = [x, c]
    xn, _ = o.layernorm_fwd_multi(xs, g.to(dev()), be.to(dev()), EPS, want_stats=False)
    w1d = w1.to(dtype).to(dev()); b1d = b1.to(dev())
    hs = [torch.empty(t.shape[:-1] + (Hd,), device=dev(), dtype=dtype) for t in xs]
    o.linear_fwd([o.Prob(a, w1d, h, bias=b1d) for a, h in zip(xn, hs)], Hd, C, o.ACT_GELU)
    outs = [torch.empty_like(t) for t in xs]
    o.linear_fwd([o.Prob(h, w2, ot, bias=b2, res=t) for h, ot, t in zip(hs, outs, xs)], C, Hd)
    for fused, unfused in ((ox, outs[0]), (oc, outs[1])):
        assert_close(fused, unfused.double().cpu(), dtype, f"fused vs unfused C={C}", tol16=6e-3)


@pytest.mark.parametrize("rows,rows_c", [(3 * 331, 48), (31, 0), (8 * 32 * 40 + 5, 16)])
def test_mlp_fused_c96_resident_and_streaming_forms(rows, rows_c, request):
    """C = 96, hidden = 384: the resident-weight kernel (csrc/rwmlp.hip, default) and the tile-streaming kernel (csrc/fused.hip, mlp_rw96 = 0)
    against the kernel-math reference and against each other, one and two problems, DropPath scale on the first, row counts off the panel size."""
    from lemevit_amd import _lib
    request.addfinalizer(lambda: _lib.config_set("mlp_rw96", 1))
    dtype, C, Hd = torch.bfloat16, 96, 384
    o, F, masters, w2, b2 = _mlp_case(C, Hd)
    L = 331 if rows % 331 == 0 else rows
    x, x64 = _tokens(rows, C, "x", dtype)
    xs, refs64 = [x.view(rows // L, L, C)], [x64]
    scales = [(det_tensor((rows // L,), "sx", 7).abs() + 0.5).to(dev())]
    if rows_c:
        c, c64 = _tokens(rows_c, C, "c", dtype)
        xs.append(c.view(rows_c // 16, 16, C)); refs64.append(c64); scales.append(None)
    outs = {}
    for form in (1, 0):
        _lib.config_set("mlp_rw96", form)
        outs[form] = o.mlp_fused_fwd(xs, F, w2, b2, EPS, scales)
        for out, t64, sc, Lr in zip(outs[form], refs64, scales, (L, 16)):
            n = t64.shape[0]
            sr = torch.ones(n, 1, dtype=torch.float64) if sc is None else sc.cpu().double()[torch.arange(n) // Lr][:, None]
            rk, rm = _mlp_refs(t64, F, w2.cpu().double(), b2.cpu(), masters, sr)
            assert_close(out.view(n, C), rk, dtype, f"form {form} (kernel math)")
            assert_close(out.view(n, C), rm, dtype, f"form {form} (reference math)", tol16=4e-3)
    for a, b in zip(outs[1], outs[0]):
        assert_close(a, b.double().cpu(), dtype, "resident vs streaming form", tol16=4e-3)


@pytest.mark.parametrize("C,Hd,rows", [(64, 128, 1), (128, 256, 63), (96, 384, 65), (192, 384, 129), (384, 1536, 127), (320, 640, 257)])
def test_mlp_fused_edges(C, Hd, rows):
    """row counts around the tile heights (1, 63 / 65, 127 / 129, 257) and MLP widths below 4 C (hidden = 2 C)"""
    dtype = torch.bfloat16
    o, F, masters, w2, b2 = _mlp_case(C, Hd)
    x, x64 = _tokens(rows, C, "x", dtype)
    (out,) = o.mlp_fused_fwd([x.view(1, rows, C)], F, w2, b2, EPS)
    rk, rm = _mlp_refs(x64, F, w2.cpu().double(), b2.cpu(), masters, torch.ones(rows, 1, dtype=torch.float64))
    assert_close(out.view(rows, C), rk, dtype, f"mlp_fused edge C={C} rows={rows} (kernel math)")
    assert_close(out.view(rows, C), rm, dtype, f"mlp_fused edge C={C} rows={rows} (reference math)", tol16=4e-3)


def test_fused_abi_errors():
    """status codes of the raw C ABI (no exception crosses it): bad dtype, unsupported width, missing folded operands"""
    import ctypes as C_
    from lemevit_amd import _lib
    lib = _lib.lib
    x = torch.zeros(128, 96, device=dev(), dtype=torch.bfloat16)
    prob = (_lib.MlpProblem * 1)()
    prob[0].x, prob[0].out, prob[0].rows = x.data_ptr(), x.data_ptr(), 128
    w = _lib.MlpWeights(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr())
    assert lib.lmv_mlp_fused_fwd(prob, 1, C_.byref(w), 96, 384, 1e-6, _lib.LMV_F32, None) == -2          # LMV_ERR_DTYPE
    assert lib.lmv_mlp_fused_fwd(prob, 1, C_.byref(w), 512, 2048, 1e-6, _lib.LMV_BF16, None) == -1       # LMV_ERR_SHAPE
    assert b"unsupported" in lib.lmv_last_error()
    w0 = _lib.MlpWeights(None, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr())
    assert lib.lmv_mlp_fused_fwd(prob, 1, C_.byref(w0), 96, 384, 1e-6, _lib.LMV_BF16, None) == -1
    lp = (_lib.LinearProblem * 1)()
    lp[0].a, lp[0].w, lp[0].out, lp[0].rows = x.data_ptr(), x.data_ptr(), x.data_ptr(), 128
    assert lib.lmv_ln_linear_fwd(lp, 1, 96, 96, 1e-6, 0, _lib.LMV_BF16, None) == -1                        # colsum / folded bias missing
    assert lib.lmv_attn_out_proj_residual(lp, 1, 96, _lib.LMV_BF16, None) == -1                            # residual missing
    assert lib.lmv_ln_fold(None, None, None, None, 96, 96, None, None, None, _lib.LMV_BF16, None) == -1


def test_mlp_fused_rejects():
    o = ops()
    assert not o.mlp_fused_supported(512, 2048, torch.bfloat16)
    assert not o.mlp_fused_supported(96, 384, torch.float32)
    assert not o.mlp_fused_supported(96, 200, torch.bfloat16)
    _, F, _, w2, b2 = _mlp_case(96, 384)
    x = torch.zeros(4, 16, 96, device=dev(), dtype=torch.float32)
    with pytest.raises((RuntimeError, TypeError)):
        o.mlp_fused_fwd([x], F, w2, b2, EPS)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attn_out_proj_residual(dtype):
    o = ops()
    C, B, N, M = 192, 5, 100, 16
    ao, ao64 = rnd((B, N, C), "ao", dtype); aoc, aoc64 = rnd((B, M, C), "aoc", dtype)
    x, x64 = rnd((B, N, C), "x", dtype); c, c64 = rnd((B, M, C), "c", dtype)
    w, w64 = rnd((C, C), "w", dtype, 1 / math.sqrt(C)); bias = det_tensor((C,), "b", 7, 0.5).to(dev())
    sx = (det_tensor((B,), "sx", 7).abs() + 0.5).to(dev())
    ox, oc = torch.empty_like(x), torch.empty_like(c)
    o.attn_out_proj_residual([o.Prob(ao, w, ox, bias=bias, res=x, row_scale=sx, rps=N), o.Prob(aoc, w, oc, bias=bias, res=c)], C)
    assert_close(ox, x64 + sx.cpu().double()[:, None, None] * (ao64 @ w64.t() + bias.cpu().double()), dtype, "proj x")
    assert_close(oc, c64 + aoc64 @ w64.t() + bias.cpu().double(), dtype, "proj c")
    with pytest.raises(RuntimeError):
        o.attn_out_proj_residual([o.Prob(ao, w, ox, bias=bias)], C)


def test_gelu_poly_bound():
    """The activation of the fused MLP kernel ON ITS OWN (lmv_gelu_poly_eval = the kernel's gelu_poly2): |poly(u) - GELU_erf(u)| against the
    bound DESIGN.md / INTEGRATION.md state -- 1.9e-4 absolute everywhere, 3.2e-5 for |u| <= 2 -- and saturation (identity / zero to fp32 rounding) beyond |u| >= 4, so a
    future refit cannot hide behind the residual of the full-kernel tests."""
    import ctypes as C
    from lemevit_amd._lib import lib, check
    u = torch.cat([torch.linspace(-8.0, 8.0, 1 << 20), torch.tensor([0.0, -0.0, 4.0, -4.0, 3.999, -3.999, 1e-8, -1e-8, 30.0, -30.0])]).to(dev())
    y = torch.empty_like(u)
    check(lib.lmv_gelu_poly_eval(u.data_ptr(), y.data_ptr(), u.numel(), torch.cuda.current_stream().cuda_stream), "lmv_gelu_poly_eval")
    torch.cuda.synchronize()
    ref = (0.5 * u.double() * (1.0 + torch.erf(u.double() / 2.0 ** 0.5)))
    err = (y.double() - ref).abs()
    assert float(err.max()) <= 1.9e-4, f"max |gelu_poly - gelu_erf| = {float(err.max()):.3e}"
    inner = u.abs() <= 2.0
    assert float(err[inner].max()) <= 3.2e-5, f"|u| <= 2: {float(err[inner].max()):.3e}"
    hi, lo = u >= 4.0, u <= -4.0
    # saturation: q(1) = 1/2 up to the fp32 rounding of the Horner evaluation -> identity / zero to a few ulp
    assert float(((y[hi] - u[hi]).abs() / u[hi]).max()) <= 1e-6 and float((y[lo].abs() / u[lo].abs()).max()) <= 1e-6
    # relative to a stored bf16 activation: below half an ulp for |h| >= 0.05
    big = ref.abs() >= 0.05
    assert float((err[big] / ref[big].abs()).max()) <= 2.0 ** -9 + 1e-6
