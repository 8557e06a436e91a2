"""GPU parity tests of the individual HIP kernels, called through the C-ABI (mafed_amd.ops -> libmafed_hip.so),
against plain fp32/fp64 PyTorch-CPU restatements and the oracle (oracle/vlpythia_ref.py)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vlpythia_ref as R

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _ops():
    from mafed_amd import ops
    return ops


def maxerr(a, b):
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def assert_close(a, b, tol, what=""):
    b = b.detach().double().cpu()
    a = a.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = float((a - b).abs().max()) if a.numel() else 0.0
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    assert err <= tol * scale, f"{what}: max err {err:.3e} > {tol:.1e} * {scale:.3g}"


# ---------------------------------------------------------------------------------------------------------------
# GEMM
# ---------------------------------------------------------------------------------------------------------------
def _gemm_ref(A, B, tA, tB):
    a = A.double().t() if tA else A.double()
    b = B.double().t() if tB else B.double()
    return a @ b


@pytest.mark.parametrize("tA,tB", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(42, 128, 128), (130, 72, 40), (64, 64, 16), (257, 132, 95)])
def test_gemm_f32(tA, tB, M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    A = torch.randn((K, M) if tA else (M, K), generator=g)
    B = torch.randn((N, K) if tB else (K, N), generator=g)
    C = ops.gemm(A.to(DEV), B.to(DEV), tA, tB)
    assert_close(C, _gemm_ref(A, B, tA, tB), 2e-5, "gemm_f32")


def _int_mat(shape, g, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, generator=g).float()


GEMM_SHAPES = [(128, 128, 64), (48, 136, 72), (264, 392, 200), (16, 8, 8), (1024, 256, 512), (384, 640, 192), (512, 768, 320), (288, 512, 192),
               (576, 384, 128), (576, 768, 256), (384, 256, 128)]
# variants on the product path: 0 automatic dispatch, 1 register-staged kernel (ragged shapes), 21 / 27 the 144x128 and 192x128
# LDS-DMA tiles the dispatcher picks for M = 9216 -- the full shape x transpose matrix.  The measured-and-kept-for-reference tile
# configurations behind mafed_gemm_set_variant get ONE qualifying shape each (all four transpose modes).
# variant = 10 + tile configuration (mafed_gemm_set_variant): 11 256x256, 12 256x128, 21 144x128, 22 144x128 (6 waves), 23 144x256,
# 24 128x64, 25 64x128, 26 288x256, 27 192x128, 28 128x128 with two K groups
GEMM_CASES = [(v, s) for v in (0, 1, 21, 27) for s in GEMM_SHAPES] + [
    (11, (512, 768, 320)), (12, (512, 768, 320)), (22, (576, 384, 128)), (23, (576, 768, 256)), (24, (384, 640, 192)), (25, (384, 640, 192)),
    (26, (576, 768, 256)), (28, (512, 768, 384)), (28, (256, 128, 1152))]


@pytest.mark.parametrize("tA,tB", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("variant,shape", GEMM_CASES)
def test_gemm_bf16_exact_integers(tA, tB, shape, variant):
    """Small-integer operands are exact in bf16 and the fp32 accumulator: any wrong fragment / transposing-read /
    swizzle mapping shows up as a hard mismatch (asymmetric data)."""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    M, N, K = shape
    if tA and M % 8:
        pytest.skip("contiguous extent must be a multiple of 8")
    if not tA and K % 8:
        pytest.skip("contiguous extent must be a multiple of 8")
    A = _int_mat((K, M) if tA else (M, K), g)
    B = _int_mat((N, K) if tB else (K, N), g)
    from mafed_amd import _lib
    # 0: automatic; 1: register-staged kernel everywhere; 10+c: LDS-DMA tile configuration c where the shape allows it
    _lib.load().mafed_gemm_set_variant(variant)
    try:
        C = ops.gemm(A.to(DEV, torch.bfloat16), B.to(DEV, torch.bfloat16), tA, tB, out_dtype=torch.float32)
    finally:
        _lib.load().mafed_gemm_set_variant(0)
    ref = _gemm_ref(A, B, tA, tB)
    assert maxerr(C, ref) == 0.0, f"bf16 MFMA gemm tA={tA} tB={tB} {M}x{N}x{K}: max err {maxerr(C, ref)}"


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_gemm_epilogues(dt):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    M, N, K = 72, 136, 64
    A, W = torch.randn(M, K, generator=g) * 0.5, torch.randn(N, K, generator=g) * 0.5
    bias = torch.randn(N, generator=g)
    r1, r2 = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    c0 = torch.randn(M, N, generator=g)
    Ad, Wd = A.to(DEV, dt), W.to(DEV, dt)
    Af, Wf = Ad.float().cpu().double(), Wd.float().cpu().double()
    pre = Af @ Wf.t() + bias.double()
    tol = 2e-5 if dt == torch.float32 else 1e-2
    # bias + gelu, pre-activation saved
    aux = torch.empty(M, N, dtype=dt, device=DEV)
    y = ops.gemm(Ad, Wd, False, True, bias=bias.to(DEV), epilogue=ops.EPI_GELU, aux=aux)
    assert_close(aux.float(), pre, tol, "aux")
    assert_close(y.float(), F.gelu(pre), tol, "gelu")
    # residuals + beta into an fp32 C
    c = c0.clone().to(DEV)
    ops.gemm(Ad, Wd, False, True, out=c, bias=bias.to(DEV), res1=r1.to(DEV), res2=r2.to(DEV), beta=1.0)
    assert_close(c, pre + r1.double() + r2.double() + c0.double(), tol, "res+beta")
    # gelu backward epilogue: C = acc * gelu'(aux)
    u = torch.randn(M, N, generator=g)
    ud = u.to(DEV, dt)
    uf = ud.float().cpu().double().requires_grad_(True)
    F.gelu(uf).sum().backward()
    y = ops.gemm(Ad, Wd, False, True, epilogue=ops.EPI_GELU_BWD, aux=ud)
    assert_close(y.float(), (Af @ Wf.t()) * uf.grad, tol, "gelu_bwd")


@pytest.mark.parametrize("M,N,K", [(32, 1024, 1024), (3, 48, 256), (64, 4096, 1024), (17, 50304, 512), (32, 1024, 4096), (1, 16, 256)])
def test_gemm_skinny_decode_shapes(M, N, K):
    """One token per sample (decode): the NT product with M <= 64 rows goes through the weight-streaming kernel (16-column
    strips, K split over eight waves); all of its epilogues and a strided output (the K/V-cache row) against fp64."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N)
    X = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    W = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
    bias = torch.randn(N, generator=g)
    ref = X.float().cpu().double() @ W.float().cpu().double().t()
    scale = float(ref.abs().max())
    y = ops.gemm(X, W, False, True)
    assert_close(y.float(), ref, 1e-2, "plain")
    assert torch.equal(y, ops.gemm(X, W, False, True)), "deterministic (LDS fold in a fixed order, no atomics)"
    y = ops.gemm(X, W, False, True, bias=bias.to(DEV), epilogue=ops.EPI_GELU)
    assert_close(y.float(), F.gelu(ref + bias.double()), 1e-2, "bias + gelu")
    r1 = torch.randn(M, N, generator=g).to(torch.bfloat16)
    r2 = torch.randn(M, N, generator=g)
    y = ops.gemm(X, W, False, True, bias=bias.to(DEV), out_dtype=torch.float32, res1=r1.to(DEV), res2=r2.to(DEV))
    assert y.dtype == torch.float32
    err = (y.cpu().double() - (ref + bias.double() + r1.double() + r2.double())).abs().max().item()
    assert err <= 2e-3 * max(1.0, scale), err  # fp32 out: only the bf16 operands round
    # strided destination: row t of a [M, cap, N] cache tensor
    cache = torch.zeros(M, 3, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm(X, W, False, True, bias=bias.to(DEV), out=cache[:, 1, :])
    assert_close(cache[:, 1, :].float(), ref + bias.double(), 1e-2, "strided out")
    assert float(cache[:, 0, :].abs().max()) == 0.0 and float(cache[:, 2, :].abs().max()) == 0.0


@pytest.mark.parametrize("kgroups", [False, True])
@pytest.mark.parametrize("split", [100, 101, 102, 104, 108])
def test_gemm_split_k_accumulate(split, kgroups):
    """dW += dY^T.X with K split over blockIdx.y and fp32 atomics into the running gradient (100 = automatic choice);
    ``kgroups``: the 8-wave tile whose two wave groups each take half of the block's K range (hand-over through LDS)."""
    ops = _ops()
    from mafed_amd import _lib
    g = torch.Generator().manual_seed(9)
    M, N, K = 256, 384, 2048
    A = _int_mat((K, M), g)
    B = _int_mat((K, N), g)
    c0 = _int_mat((M, N), g, -50, 50)
    out = c0.clone().to(DEV)
    _lib.load().mafed_gemm_set_variant(split)
    if kgroups:
        _lib.load().mafed_gemm_set_variant(28)
    try:
        ops.gemm(A.to(DEV, torch.bfloat16), B.to(DEV, torch.bfloat16), True, False, out=out, beta=1.0)
    finally:
        _lib.load().mafed_gemm_set_variant(100)
        _lib.load().mafed_gemm_set_variant(0)
    ref = c0.double() + A.double().t() @ B.double()
    assert maxerr(out, ref) == 0.0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,epi", [(256, 256, 128, "none"), (288, 128, 64, "gelu_bwd"), (3072, 4096, 64, "gelu_bwd"), (576, 512, 192, "none"),
                                        (100, 72, 40, "none")])
def test_gemm_fused_colsum(dt, M, N, K, epi):
    """mafed_gemm_colsum: colsum += column sums of the stored C, for every tile configuration with a fused epilogue
    (128x128, 144x128, 192x128), the GELU' epilogue that feeds dense_h_to_4h.bias, and the second-pass route."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g).to(dt).to(DEV)
    Bm = torch.randn(K, N, generator=g).to(dt).to(DEV)
    kw = {}
    if epi == "gelu_bwd":
        kw = dict(epilogue=ops.EPI_GELU_BWD, aux=torch.randn(M, N, generator=g).to(dt).to(DEV))
    ref = ops.gemm(A, Bm, False, False, **kw)
    cs = torch.full((N,), -1.25, dtype=torch.float32, device=DEV)
    out = ops.gemm(A, Bm, False, False, colsum=cs, **kw)
    assert torch.equal(out, ref)
    want = out.double().sum(0) - 1.25
    # bf16: the fused sums see the fp32 values before the store rounds them (each element moves by <= 2^-9 relative)
    bound = out.double().abs().sum(0).max().item() * (1e-5 if dt == torch.float32 else 2.0 ** -8) + 1e-4
    err = (cs.double() - want).abs().max().item()
    assert err <= bound, (err, bound)


def test_gemm_rejects_bad_arguments():
    ops = _ops()
    from mafed_amd._lib import MafedHipError
    a = torch.zeros(8, 12, device=DEV, dtype=torch.bfloat16)  # K = 12 not a multiple of 8
    b = torch.zeros(8, 12, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(MafedHipError):
        ops.gemm(a, b, False, True)
    with pytest.raises(MafedHipError):
        ops.gemm(torch.zeros(4, 8, device=DEV), torch.zeros(6, 8, device=DEV), False, True)  # N = 6 not a multiple of 4


def test_colsum_cast_gelu():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    for dt in (torch.float32, torch.bfloat16):
        X = torch.randn(300, 72, generator=g).to(DEV, dt)
        out = torch.ones(72, device=DEV)
        ops.colsum_(X, out)
        assert_close(out, 1.0 + X.float().cpu().double().sum(0), 1e-5 if dt == torch.float32 else 1e-4, "colsum")
    x = torch.randn(1003, generator=g).to(DEV)
    xb = ops.cast(x, torch.bfloat16)
    assert torch.equal(xb.cpu(), x.cpu().to(torch.bfloat16))
    assert torch.equal(ops.cast(xb, torch.float32).cpu(), xb.cpu().float())
    assert_close(ops.gelu(x), F.gelu(x.cpu().double()), 1e-6, "gelu")


# ---------------------------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("h", [128, 192, 768, 1024, 2048])
@pytest.mark.parametrize("dual", [True, False])
def test_layernorm_fwd_bwd(h, dual):
    ops = _ops()
    g = torch.Generator().manual_seed(h)
    rows = 37
    x = torch.randn(rows, h, generator=g) * 2 + 0.5
    w1, b1 = torch.randn(h, generator=g), torch.randn(h, generator=g)
    w2, b2 = torch.randn(h, generator=g), torch.randn(h, generator=g)
    dy1, dy2, dres = torch.randn(rows, h, generator=g), torch.randn(rows, h, generator=g), torch.randn(rows, h, generator=g)
    xd = x.double().requires_grad_(True)
    w1d, b1d, w2d, b2d = (t.double().requires_grad_(True) for t in (w1, b1, w2, b2))
    y1r = F.layer_norm(xd, (h,), w1d, b1d, 1e-5)
    y2r = F.layer_norm(xd, (h,), w2d, b2d, 1e-5)
    tot = (y1r * dy1.double()).sum() + ((y2r * dy2.double()).sum() if dual else 0) + (xd * dres.double()).sum()
    tot.backward()
    D = lambda t: t.to(DEV)
    y1, y2, mean, rstd = ops.layernorm_fwd(D(x), D(w1), D(b1), D(w2) if dual else None, D(b2) if dual else None, 1e-5, torch.float32)
    assert_close(y1, y1r, 2e-5, "y1")
    if dual:
        assert_close(y2, y2r, 2e-5, "y2")
    dw1, db1, dw2, db2 = (torch.full((h,), 0.25, device=DEV) for _ in range(4))
    dx, dx_lp = ops.layernorm_bwd(D(dy1), D(dy2) if dual else None, D(x), mean, rstd, D(w1), D(w2) if dual else None, D(dres), dw1, db1,
                                  dw2 if dual else None, db2 if dual else None)
    assert_close(dx, xd.grad, 5e-5, "dx")
    assert_close(dw1 - 0.25, w1d.grad, 5e-5, "dw1 (accumulated)")
    assert_close(db1 - 0.25, b1d.grad, 5e-5, "db1")
    if dual:
        assert_close(dw2 - 0.25, w2d.grad, 5e-5, "dw2")
        assert_close(db2 - 0.25, b2d.grad, 5e-5, "db2")
    # bf16 outputs round the fp32 result
    y1b, _, _, _ = ops.layernorm_fwd(D(x), D(w1), D(b1), None, None, 1e-5, torch.bfloat16)
    assert_close(y1b.float(), y1r, 1e-2, "y1 bf16")


def test_layernorm_bwd_many_rows_and_injection():
    """More rows than one block pass (grid-stride + slab reduction) and the fused distillation-gradient injection."""
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    B, P, T, h = 9, 250, 24, 256
    S = P + T
    rows = B * S
    x = torch.randn(rows, h, generator=g)
    t = x + 0.1 * torch.randn(rows, h, generator=g)
    w1, w2 = torch.randn(h, generator=g), torch.randn(h, generator=g)
    dy1, dy2 = torch.randn(rows, h, generator=g), torch.randn(rows, h, generator=g)
    am = torch.ones(B, T, dtype=torch.int64)
    for b in range(B):
        am[b, : (b * 3) % T] = 0
    scales = torch.tensor([0.7, -0.3])
    xd = x.double().requires_grad_(True)
    w1d, w2d = w1.double().requires_grad_(True), w2.double().requires_grad_(True)
    z = torch.zeros(h, dtype=torch.float64)
    tot = (F.layer_norm(xd, (h,), w1d, z, 1e-5) * dy1.double()).sum() + (F.layer_norm(xd, (h,), w2d, z, 1e-5) * dy2.double()).sum()
    tot.backward()
    lang, img = R.modality_masks(am, P)
    inj = (scales[0] * lang.reshape(-1, 1).double() + scales[1] * img.reshape(-1, 1).double()) * (x - t).double()
    D = lambda v: v.to(DEV)
    _, _, mean, rstd = ops.layernorm_fwd(D(x), D(w1), D(z.float()), D(w2), D(z.float()), 1e-5, torch.float32)
    dw1, db1, dw2, db2 = (torch.zeros(h, device=DEV) for _ in range(4))
    dx, dx_lp = ops.layernorm_bwd(D(dy1), D(dy2), D(x), mean, rstd, D(w1), D(w2), None, dw1, db1, dw2, db2,
                                  teacher=D(t), attention_mask=D(am), S=S, P=P, inj_scale=D(scales))
    assert_close(dx, xd.grad + inj, 1e-4, "dx + injection")
    assert_close(dw1, w1d.grad, 1e-4, "dw1")
    assert_close(db2, dy2.double().sum(0), 1e-4, "db2")


def test_layernorm_bwd_cosine_injection():
    """inj_mul < 0: the fused injection adds scale[class] * d/dx (1 - cos(x, teacher)) (mafed_distill_fwd's epsilon) per row."""
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    B, P, T, h = 5, 20, 12, 192
    S = P + T
    rows = B * S
    x = torch.randn(rows, h, generator=g)
    t = x + 0.5 * torch.randn(rows, h, generator=g)
    w1 = torch.randn(h, generator=g)
    dy1 = torch.randn(rows, h, generator=g)
    am = torch.ones(B, T, dtype=torch.int64)
    for b in range(B):
        am[b, : (b * 5) % T] = 0
    scales = torch.tensor([0.9, -0.4])
    xd = x.double().requires_grad_(True)
    z = torch.zeros(h, dtype=torch.float64)
    lang, img = R.modality_masks(am, P)
    rw = scales[0] * lang.reshape(-1).double() + scales[1] * img.reshape(-1).double()
    td = t.double()
    dist = 1.0 - (xd * td).sum(-1) / torch.sqrt(((xd * xd).sum(-1) + 1e-12) * ((td * td).sum(-1) + 1e-12))
    tot = (F.layer_norm(xd, (h,), w1.double(), z, 1e-5) * dy1.double()).sum() + (rw * dist).sum()
    tot.backward()
    D = lambda v: v.to(DEV)
    _, _, mean, rstd = ops.layernorm_fwd(D(x), D(w1), D(z.float()), None, None, 1e-5, torch.float32)
    dw1, db1 = torch.zeros(h, device=DEV), torch.zeros(h, device=DEV)
    dx, _ = ops.layernorm_bwd(D(dy1), None, D(x), mean, rstd, D(w1), None, None, dw1, db1, None, None,
                              teacher=D(t), attention_mask=D(am), S=S, P=P, inj_scale=D(scales), inj_mul=-1.0)
    assert_close(dx, xd.grad, 1e-4, "dx + cosine injection")
    # the same through the stand-alone kernel (what the generic path materialises)
    ds = ops.distill_bwd(D(x).view(B, S, h), D(t).view(B, S, h), D(am), P, D(scales), True)
    dx0, _ = ops.layernorm_bwd(D(dy1), None, D(x), mean, rstd, D(w1), None, None, dw1, db1, None, None)
    assert_close(dx, dx0 + ds.view(rows, h), 1e-5, "fused = LayerNorm backward + distill_bwd")


# ---------------------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------------------
def _attn_ref(qkv, B, S, H, D, rot, am_full):
    """Oracle attention core on [B,S,H,3,D] (double precision), returns out [B,S,H*D]."""
    cfg = R.RefConfig(hidden_size=H * D, num_attention_heads=H, rotary_pct=rot / D)
    cos, sin = R.rotary_tables(cfg, S)
    x = qkv.view(B, S, H, 3 * D).transpose(1, 2)
    q, k, v = x.chunk(3, dim=-1)
    q = R.apply_partial_rotary(q, cos.to(q.dtype), sin.to(q.dtype))
    k = R.apply_partial_rotary(k, cos.to(q.dtype), sin.to(q.dtype))
    w = torch.matmul(q, k.transpose(2, 3)) * (D ** -0.5) + R.additive_mask(am_full).to(q.dtype)
    w = F.softmax(w, dim=-1)
    return torch.matmul(w, v).transpose(1, 2).reshape(B, S, H * D)


def _attn_case(B, P, T, H, D, seed, pad=True):
    g = torch.Generator().manual_seed(seed)
    S = P + T
    qkv = torch.randn(B, S, H, 3, D, generator=g)
    am = torch.ones(B, T, dtype=torch.int64)
    if pad:
        for b in range(1, B):
            am[b, : min(T - 1, (b * 5) % T)] = 0
    dout = torch.randn(B, S, H * D, generator=g)
    rot = D // 4
    cfg = R.RefConfig(hidden_size=H * D, num_attention_heads=H)
    cos, sin = R.rotary_tables(cfg, S)
    half = rot // 2
    return qkv, am, dout, rot, cos[:, :half].contiguous(), sin[:, :half].contiguous()


@pytest.mark.parametrize("B,P,T,H,D", [(2, 8, 6, 2, 64), (2, 40, 24, 2, 128), (1, 8, 6, 1, 256), (3, 70, 13, 2, 64)])
def test_attention_f32(B, P, T, H, D):
    ops = _ops()
    qkv, am, dout, rot, cos, sin = _attn_case(B, P, T, H, D, seed=D + P)
    S = P + T
    am_full = torch.cat([torch.ones(B, P, dtype=torch.int64), am], 1)
    qd = qkv.double().requires_grad_(True)
    ref = _attn_ref(qd, B, S, H, D, rot, am_full)
    (ref * dout.double()).sum().backward()
    dv = lambda t: t.to(DEV)
    out, lse = ops.attn_fwd(dv(qkv).view(B * S, -1), B, S, H, D, rot, dv(cos), dv(sin), dv(am))
    assert_close(out.view(B, S, H * D), ref, 2e-5, "attn out")
    dqkv = ops.attn_bwd(dv(qkv).view(B * S, -1), out, dv(dout).view(B * S, -1), lse, B, S, H, D, rot, dv(cos), dv(sin), dv(am))
    assert_close(dqkv.view(B, S, H, 3, D), qd.grad, 5e-5, "attn dqkv")
    csum = torch.zeros(3 * H * D, dtype=torch.float32, device=DEV)
    ops.attn_bwd(dv(qkv).view(B * S, -1), out, dv(dout).view(B * S, -1), lse, B, S, H, D, rot, dv(cos), dv(sin), dv(am), colsum=csum)
    assert_close(csum, qd.grad.reshape(B * S, -1).sum(0), 1e-4, "attn dqkv colsum (f32)")


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("B,P,T,H,D", [(2, 8, 6, 2, 64), (2, 40, 24, 2, 128), (3, 70, 13, 2, 64), (2, 256, 32, 4, 64), (1, 200, 57, 2, 128),
                                       (1, 600, 41, 1, 64), (2, 129, 1, 2, 64),
                                       # head_dim 256 (VLPythia-1B, mafed/utils/download_models.py:6-24): K / V tiled through LDS
                                       (1, 8, 6, 1, 256), (2, 256, 32, 2, 256), (2, 70, 13, 1, 256)])
def test_attention_bf16_mfma(B, P, T, H, D, variant):
    """variant 0: one-block-per-head resident kernels where K/V fit in LDS (S <= 640 at D = 64), else tiled; 1: tiled only."""
    ops = _ops()
    from mafed_amd import _lib
    _lib.load().mafed_attn_set_variant(variant)
    try:
        _attention_bf16_case(ops, B, P, T, H, D)
    finally:
        _lib.load().mafed_attn_set_variant(0)


def _attention_bf16_case(ops, B, P, T, H, D):
    qkv, am, dout, rot, cos, sin = _attn_case(B, P, T, H, D, seed=3 * D + T)
    S = P + T
    am_full = torch.cat([torch.ones(B, P, dtype=torch.int64), am], 1)
    qb = qkv.to(torch.bfloat16)
    db = dout.to(torch.bfloat16)
    qd = qb.double().requires_grad_(True)
    ref = _attn_ref(qd, B, S, H, D, rot, am_full)
    (ref * db.double()).sum().backward()
    dv = lambda t: t.to(DEV)
    qg = dv(qb).view(B * S, -1)
    out, lse = ops.attn_fwd(qg, B, S, H, D, rot, dv(cos), dv(sin), dv(am))
    assert_close(out.float().view(B, S, H * D), ref, 2e-2, "attn out bf16")
    # exact kernel on the same bf16 data agrees with the MFMA kernel on the log-sum-exp
    out_e, lse_e = ops.attn_fwd_exact_bf16(qg, B, S, H, D, rot, dv(cos), dv(sin), dv(am))
    assert_close(lse, lse_e, 2e-2, "lse")
    csum = torch.full((3 * H * D,), 0.5, dtype=torch.float32, device=DEV)  # accumulate-into semantics: starts non-zero
    dqkv = ops.attn_bwd(qg, out, dv(db).view(B * S, -1), lse, B, S, H, D, rot, dv(cos), dv(sin), dv(am), colsum=csum)
    # fused query_key_value.bias gradient = column sums of dqkv (the kernels sum before the bf16 rounding of the store)
    want = dqkv.float().sum(0) + 0.5
    cerr = (csum - want).abs().max().item()
    assert cerr <= dqkv.float().abs().sum(0).max().item() * 2.0 ** -8 + 1e-3, f"attn dqkv colsum err {cerr}"
    dqkv2 = ops.attn_bwd(qg, out, dv(db).view(B * S, -1), lse, B, S, H, D, rot, dv(cos), dv(sin), dv(am))
    assert torch.equal(dqkv, dqkv2), "the colsum entry point must not change dqkv"
    g = qd.grad
    err = (dqkv.float().view(B, S, H, 3, D).cpu().double() - g).abs().max().item()
    assert err <= 4e-2 * max(1.0, g.abs().max().item()), f"attn dqkv bf16 err {err} (scale {g.abs().max().item()})"
    # padded keys receive exactly zero dK / dV
    for b in range(B):
        for t in range(T):
            if am[b, t] == 0:
                assert float(dqkv.view(B, S, H, 3, D)[b, P + t, :, 1:].abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------
# cross-entropy, distillation, optimiser, embedding
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_cross_entropy(dt):
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    B, T, V = 5, 9, 1000
    logits = (torch.randn(B, T, V, generator=g) * 3).to(dt)
    labels = torch.randint(0, V, (B, T), generator=g)
    labels[:, :3] = -100
    labels[2, :] = -100  # a sample without any valid label: contributes 0 / clamp(0, 1e-13)
    labels[4, -1] = -100
    lf = logits.float().double().requires_grad_(True)
    ref = R.masked_mean_loss(labels, lf.float() if False else lf)
    (ref * 1.7).backward()
    lg = logits.to(DEV)
    loss, lse = ops.ce_fwd(lg, labels.to(DEV))
    assert_close(loss.reshape(()), ref, 1e-5, "ce loss")
    dl = ops.ce_bwd(lg, labels.to(DEV), lse, torch.tensor([1.7], device=DEV))
    assert_close(dl.float(), lf.grad, 1e-5 if dt == torch.float32 else 2e-3, "ce grad")


@pytest.mark.parametrize("cosine", [False, True])
def test_distill(cosine):
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    B, P, T, h = 5, 17, 11, 192
    S = P + T
    s = torch.randn(B, S, h, generator=g)
    t = s + 0.3 * torch.randn(B, S, h, generator=g)
    am = torch.ones(B, T, dtype=torch.int64)
    am[1, :4] = 0
    am[3, :10] = 0
    lang, img = R.modality_masks(am, P)
    sd = s.double().requires_grad_(True)
    fn = R.masked_cos if cosine else R.masked_mse
    ll, vl = fn(sd, t.double(), lang), fn(sd, t.double(), img)
    (0.3 * ll + 1.1 * vl).backward()
    D = lambda v: v.to(DEV)
    out = ops.distill_fwd(D(s), D(t), D(am), P, cosine)
    assert_close(out[0] / out[2], ll, 1e-5, "lang")
    assert_close(out[1] / out[3], vl, 1e-5, "vision")
    assert float(out[2]) == float(lang.sum()) and float(out[3]) == float(img.sum())
    coef = torch.tensor([0.3 / float(lang.sum()), 1.1 / float(img.sum())], device=DEV)
    ds = ops.distill_bwd(D(s), D(t), D(am), P, coef, cosine)
    assert_close(ds, sd.grad, 1e-5, "ds")
    base = torch.ones(B, S, h, device=DEV)
    ops.distill_bwd(D(s), D(t), D(am), P, coef, cosine, out=base, accumulate=True)
    assert_close(base, sd.grad + 1.0, 1e-5, "ds accumulate")
    # CLS variant
    sd2 = s.double().requires_grad_(True)
    ref = R.cls_cos(sd2.float().double() if False else sd2, t.double())
    ref.backward()
    o = ops.distill_cls_fwd(D(s), D(t))
    assert_close(o.reshape(()), ref, 1e-5, "cls")
    dsc = ops.distill_cls_bwd(D(s), D(t), torch.tensor([1.0 / B], device=DEV))
    assert_close(dsc, sd2.grad, 1e-5, "cls grad")


def test_optimizer_kernels_against_reference_fixture():
    ops = _ops()
    from tests.helpers import load_golden
    g = load_golden("optim.npz")
    p0, b0 = torch.from_numpy(g["g5/p0"].copy()), torch.from_numpy(g["g5/b0"].copy())
    n1, n2 = p0.numel(), b0.numel()
    pad = (n1 + 63) // 64 * 64
    flat = torch.zeros(pad + n2, device=DEV)
    flat[:n1] = p0.reshape(-1).to(DEV)
    flat[pad:] = b0.to(DEV)
    grads, m, v = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
    shadow = torch.zeros(pad + n2, dtype=torch.bfloat16, device=DEV)
    lr_dev = torch.zeros(1, device=DEV)
    for i in range(3):
        grads.zero_()
        grads[:n1] = torch.from_numpy(g[f"g5/step{i}/gp"]).reshape(-1).to(DEV)
        grads[pad:] = torch.from_numpy(g[f"g5/step{i}/gb"]).to(DEV)
        out2 = ops.gradnorm_clip(grads, 2.0)
        assert_close(out2[0], torch.tensor(float(g[f"g5/step{i}/grad_norm"])), 1e-6, "grad norm")
        lr_dev.fill_(float(g["g5/lrs"][i]))
        ops.adamw_step_(flat[:pad], grads[:pad], m[:pad], v[:pad], lr_dev, 0.9, 0.98, 1e-6, 0.01, i + 1, out2, 1.0, shadow[:pad])
        ops.adamw_step_(flat[pad:], grads[pad:], m[pad:], v[pad:], lr_dev, 0.9, 0.98, 1e-6, 0.0, i + 1, out2, 1.0, shadow[pad:])
        assert_close(flat[:n1].view(p0.shape), torch.from_numpy(g[f"g5/step{i}/p"]), 2e-6, f"adamw p step {i}")
        assert_close(flat[pad:], torch.from_numpy(g[f"g5/step{i}/b"]), 2e-6, f"adamw b step {i}")
    assert torch.equal(shadow[:n1].cpu(), flat[:n1].cpu().to(torch.bfloat16))


def test_embed_concat():
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    B, P, T, h, V = 3, 5, 7, 64, 50
    img = torch.randn(B * P, h, generator=g)
    emb = torch.randn(V, h, generator=g)
    ids = torch.randint(0, V, (B, T), generator=g)
    ids[:, 0] = 0  # repeated ids collide in the backward scatter
    h0 = ops.embed_concat_fwd(img.to(DEV), emb.to(DEV), ids.to(DEV), B, P, T)
    ref = torch.cat([img.view(B, P, h), emb[ids]], 1)
    assert torch.equal(h0.view(B, P + T, h).cpu(), ref)
    dh0 = torch.randn(B * (P + T), h, generator=g)
    demb = torch.zeros(V, h, device=DEV)
    dimg = ops.embed_concat_bwd(dh0.to(DEV), ids.to(DEV), B, P, T, h, V, demb, torch.float32)
    assert torch.equal(dimg.cpu().view(B, P, h), dh0.view(B, P + T, h)[:, :P])
    ref_e = torch.zeros(V, h, dtype=torch.float64).index_add_(0, ids.reshape(-1), dh0.view(B, P + T, h)[:, P:].reshape(-1, h).double())
    assert_close(demb, ref_e, 1e-6, "embed grad")


def test_fast_gelu_polynomials_against_erf():
    """The bf16 MFMA epilogues' polynomial erf-GELU and its derivative (csrc/common.h, tools/gelu_poly_fit.py) over [-8, 8], read through
    an fp32-output product C[m, :] = x_m (A = x in column 0, B = ones in column 0): |gelu err| <= 5e-5, |gelu' err| <= 5e-5 -- below
    the bf16 rounding of the stored activation / gradient."""
    ops = _ops()
    M, N, K = 2048, 256, 256
    x = torch.linspace(-8.0, 8.0, M).to(torch.bfloat16).float()          # bf16-representable arguments
    A = torch.zeros(M, K); A[:, 0] = x
    Bm = torch.zeros(N, K); Bm[:, 0] = 1.0
    D = lambda v: v.to(DEV)
    y = ops.gemm(D(A).to(torch.bfloat16), D(Bm).to(torch.bfloat16), False, True, out_dtype=torch.float32, epilogue=ops.EPI_GELU)
    ref = F.gelu(x.double())
    assert float((y[:, 0].double().cpu() - ref).abs().max()) <= 5e-5
    assert torch.equal(y[:, 0], y[:, N - 1])
    # derivative: C = dy * gelu'(u) with dy = 1 (A = ones in column 0), u = the saved pre-activation
    A1 = torch.zeros(M, K); A1[:, 0] = 1.0
    u = x.view(M, 1).repeat(1, N).contiguous()
    g = ops.gemm(D(A1).to(torch.bfloat16), D(Bm).to(torch.bfloat16), False, True, out_dtype=torch.bfloat16, epilogue=ops.EPI_GELU_BWD,
                 aux=D(u).to(torch.bfloat16))
    xd = x.double().requires_grad_(True)
    F.gelu(xd).sum().backward()
    err = (g[:, 0].double().cpu() - xd.grad).abs()
    assert float((err - 2.0 ** -8 * xd.grad.abs()).max()) <= 5e-5, "beyond the bf16 rounding of the stored gradient"
