"""GPU parity tests of the persistent ping-pong MFMA GEMM (mafed_amd/csrc/gemm_pp.hip) through the C-ABI: exact-integer products in
every instantiated (tile configuration x operand layout x output type), several tiles per block, grouped launches, and the fused
epilogues against fp64 restatements.  Every case asserts that the ping-pong kernel really ran (mafed_gemm_pp_launches)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF, F32 = torch.bfloat16, torch.float32


def _ops():
    from mafed_amd import ops
    return ops


def _lib():
    from mafed_amd import _lib
    return _lib.load()


def _int_mat(shape, g, lo=-1, hi=2):
    return torch.randint(lo, hi, shape, generator=g).float()


def _ref(A, B, tA, tB):
    a = A.double().t() if tA else A.double()
    b = B.double().t() if tB else B.double()
    return a @ b


class forced:
    """mafed_gemm_set_variant(710 + cfg) around a block; checks that the ping-pong kernel launched `expect` times."""

    def __init__(self, cfg, expect=1):
        self.cfg, self.expect = cfg, expect

    def __enter__(self):
        self.lib = _lib()
        self.lib.mafed_gemm_set_variant(710 + self.cfg if self.cfg is not None else 701)
        self.n0 = self.lib.mafed_gemm_pp_launches()

    def __exit__(self, *a):
        self.lib.mafed_gemm_set_variant(701)
        if a[0] is None:
            assert self.lib.mafed_gemm_pp_launches() - self.n0 == self.expect, "the ping-pong kernel did not take this launch"


# (cfg, tA, tB, out dtype): the instantiated combinations (gemm_pp_launch)
# configuration 0 = 144 x 256 tiles (forward / dX), 1 = 128 x 256 tiles (dW: both operands reduction-major), 2 = 256 x 256 tiles
# (gemm_z.hip: one wave per SIMD)
COMBOS = [(0, False, True, BF), (0, False, True, F32), (0, False, False, BF), (1, True, False, F32),
          (2, False, True, BF), (2, False, True, F32), (2, False, False, BF), (2, True, False, F32)]
TM = {0: 144, 1: 128, 2: 256}


@pytest.mark.parametrize("cfg,tA,tB,od", COMBOS)
@pytest.mark.parametrize("tiles_m,tiles_n,K", [(1, 1, 256), (3, 2, 384), (2, 1, 1024), (5, 3, 256), (2, 2, 640)])
def test_pp_exact_integers(cfg, tA, tB, od, tiles_m, tiles_n, K):
    """Small-integer operands are exact in bf16, in the fp32 accumulator and (|sum| <= 256 at K = 256; checked in fp32 otherwise) in
    the output: a wrong DMA permutation, swizzle, fragment or column map is a hard mismatch."""
    ops = _ops()
    g = torch.Generator().manual_seed(11 + cfg)
    M, N = TM[cfg] * tiles_m, 256 * tiles_n
    A = _int_mat((K, M) if tA else (M, K), g)
    B = _int_mat((N, K) if tB else (K, N), g)
    if od == BF and K > 256:
        A = A * (torch.rand(A.shape, generator=g) < 0.3)   # keep |sum| small enough to be exact in bf16
    with forced(cfg):
        C = ops.gemm(A.to(DEV, BF), B.to(DEV, BF), tA, tB, out_dtype=od)
    ref = _ref(A, B, tA, tB)
    if od == BF:
        ref = ref.to(BF).double()
    err = float((C.double().cpu() - ref).abs().max())
    assert err == 0.0, f"cfg {cfg} tA={tA} tB={tB} {od} {M}x{N}x{K}: max err {err}"


@pytest.mark.parametrize("cfg,tA,tB,od", COMBOS)
def test_pp_many_tiles_per_block(cfg, tA, tB, od):
    """More tiles than CUs (several tiles per persistent block, a ragged last round) and a tile count that is not a multiple of 8."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    for tiles_m, tiles_n in ((37, 8), (7, 1)):
        M, N, K = TM[cfg] * tiles_m, 256 * tiles_n, 256
        A = _int_mat((K, M) if tA else (M, K), g)
        B = _int_mat((N, K) if tB else (K, N), g)
        with forced(cfg):
            C = ops.gemm(A.to(DEV, BF), B.to(DEV, BF), tA, tB, out_dtype=od)
        ref = _ref(A, B, tA, tB)
        assert float((C.double().cpu() - ref).abs().max()) == 0.0, (cfg, tiles_m, tiles_n)


@pytest.mark.parametrize("cfg,tA,tB,od", COMBOS)
def test_pp_grouped_launch(cfg, tA, tB, od):
    """Problems of different shapes and leading dimensions in one launch == the same problems one by one."""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    shapes = [(TM[cfg] * 2, 512, 256), (TM[cfg] * 1, 256, 512), (TM[cfg] * 3, 768, 384), (TM[cfg] * 1, 1024, 256)]
    probs, refs = [], []
    for (M, N, K) in shapes:
        A = _int_mat((K, M) if tA else (M, K), g).to(DEV, BF)
        B = _int_mat((N, K) if tB else (K, N), g).to(DEV, BF)
        c0 = _int_mat((M, N), g).to(DEV, od) if od == F32 else None
        out = c0.clone() if c0 is not None else torch.empty((M, N), dtype=od, device=DEV)
        probs.append(dict(A=A, B=B, out=out, beta=1.0 if od == F32 else 0.0))
        r = _ref(A.float().cpu(), B.float().cpu(), tA, tB)
        refs.append(r + c0.double().cpu() if c0 is not None else r)
    with forced(cfg):
        ops.gemm_grouped(probs, tA, tB)
    for q, r in zip(probs, refs):
        if od == BF:
            r = r.to(BF).double()
        assert float((q["out"].double().cpu() - r).abs().max()) == 0.0


@pytest.mark.parametrize("cfg", [0, 2])
def test_pp_epilogues_forward(cfg):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    M, N, K = TM[cfg] * 2, 512, 256
    A, W = torch.randn(M, K, generator=g) * 0.5, torch.randn(N, K, generator=g) * 0.5
    bias = torch.randn(N, generator=g)
    r1, r2 = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    Ad, Wd = A.to(DEV, BF), W.to(DEV, BF)
    pre = Ad.float().cpu().double() @ Wd.float().cpu().double().t() + bias.double()
    # bias only
    with forced(cfg):
        y = ops.gemm(Ad, Wd, False, True, bias=bias.to(DEV))
    assert float((y.double().cpu() - pre).abs().max()) <= 1e-2 * float(pre.abs().max())
    # bias + gelu, pre-activation saved / not saved
    aux = torch.empty(M, N, dtype=BF, device=DEV)
    with forced(cfg, 2):
        y = ops.gemm(Ad, Wd, False, True, bias=bias.to(DEV), epilogue=ops.EPI_GELU, aux=aux)
        y2 = ops.gemm(Ad, Wd, False, True, bias=bias.to(DEV), epilogue=ops.EPI_GELU)
    assert float((aux.double().cpu() - pre).abs().max()) <= 1e-2 * float(pre.abs().max())
    assert float((y.double().cpu() - F.gelu(pre)).abs().max()) <= 1e-2 * float(pre.abs().max())
    assert torch.equal(y, y2)
    # bf16 + fp32 residuals into an fp32 C (the parallel-residual add of the MLP down-projection)
    r1d = r1.to(DEV, BF)
    with forced(cfg):
        c = ops.gemm(Ad, Wd, False, True, out_dtype=F32, bias=bias.to(DEV), res1=r1d, res2=r2.to(DEV))
    want = pre + r1d.double().cpu() + r2.double()
    assert float((c.double().cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    # the same into a bf16 C
    with forced(cfg):
        cb = ops.gemm(Ad, Wd, False, True, out_dtype=BF, bias=bias.to(DEV), res1=r1d, res2=r2.to(DEV))
    assert float((cb.double().cpu() - want).abs().max()) <= 1e-2 * float(want.abs().max())
    # generic epilogue path (fp32 res1 only): in-place operand loads
    with forced(cfg):
        cg = ops.gemm(Ad, Wd, False, True, out_dtype=F32, res1=r2.to(DEV))
    want = pre - bias.double() + r2.double()
    assert float((cg.double().cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize("cfg", [0, 2])
def test_pp_gelu_backward_with_column_sums(cfg):
    """dX = dY.W with the GELU' epilogue and the fused bias gradient (column sums of the stored C)."""
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    M, N, K = TM[cfg] * 3, 512, 256
    dY, W = torch.randn(M, K, generator=g) * 0.5, torch.randn(K, N, generator=g) * 0.5
    u = torch.randn(M, N, generator=g)
    dYd, Wd, ud = dY.to(DEV, BF), W.to(DEV, BF), u.to(DEV, BF)
    cs = torch.zeros(N, device=DEV)
    with forced(cfg):
        du = ops.gemm(dYd, Wd, False, False, epilogue=ops.EPI_GELU_BWD, aux=ud, colsum=cs)
    uu = ud.float().cpu().double().requires_grad_(True)
    F.gelu(uu).sum().backward()
    want = (dYd.float().cpu().double() @ Wd.float().cpu().double()) * uu.grad
    assert float((du.double().cpu() - want).abs().max()) <= 1e-2 * float(want.abs().max())
    assert float((cs.double().cpu() - du.double().cpu().sum(0)).abs().max()) <= 1e-3 * float(du.double().abs().sum(0).max())
    # plain dX with column sums
    cs2 = torch.zeros(N, device=DEV)
    with forced(cfg):
        dx = ops.gemm(dYd, Wd, False, False, colsum=cs2)
    assert float((cs2.double().cpu() - dx.double().cpu().sum(0)).abs().max()) <= 1e-3 * float(dx.double().abs().sum(0).max())


@pytest.mark.parametrize("cfg", [1, 2])
def test_pp_weight_gradient_accumulates(cfg):
    """dW += dY^T.X (both operands [k][row]) through the 128x256 / 256x256 configurations, twice into the same buffer."""
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    Kd, M, N = 1152, 512, 768
    dY, X = torch.randn(Kd, M, generator=g), torch.randn(Kd, N, generator=g)
    dYd, Xd = dY.to(DEV, BF), X.to(DEV, BF)
    G = torch.zeros(M, N, device=DEV)
    with forced(cfg, 2):
        ops.gemm(dYd, Xd, True, False, out=G, beta=1.0)
        ops.gemm(dYd, Xd, True, False, out=G, beta=1.0)
    want = 2 * (dYd.float().cpu().double().t() @ Xd.float().cpu().double())
    assert float((G.double().cpu() - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("N,K,tB", [(3072, 1024, True), (1024, 1024, True), (4096, 1024, True), (1024, 4096, True),
                                    (1024, 3072, False), (4096, 1024, False), (1024, 4096, False)])
def test_pp_is_the_automatic_choice_for_the_step_shapes(N, K, tB):
    """M = 9216 rows (32 x 288 tokens): the dispatcher takes the ping-pong kernel for every 410M layer product -- forward
    (QKV / dense / fc1 / fc2) and dX (r2 of this test only had fc1; the QKV forward fell through a fill threshold unnoticed)."""
    ops = _ops()
    lib = _lib()
    lib.mafed_gemm_set_variant(701)
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn((9216, K), device=DEV, generator=g).to(BF)
    w = torch.randn((N, K) if tB else (K, N), device=DEV, generator=g).to(BF)
    n0 = lib.mafed_gemm_pp_launches()
    y = ops.gemm(x, w, False, tB)
    assert lib.mafed_gemm_pp_launches() == n0 + 1
    lib.mafed_gemm_set_variant(700)
    y0 = ops.gemm(x, w, False, tB)
    lib.mafed_gemm_set_variant(701)
    assert float((y.float() - y0.float()).abs().max()) <= 2e-2 * float(y0.float().abs().max())


def test_short_reductions_stay_on_the_small_tile_kernel():
    """The LM head's weight gradient over 256 labelled rows (K = 256): four k steps per tile do not pay for a persistent tile."""
    ops = _ops()
    lib = _lib()
    lib.mafed_gemm_set_variant(701)
    g = torch.Generator(device=DEV).manual_seed(0)
    d = torch.randn((256, 8192), device=DEV, generator=g).to(BF)
    a = torch.randn((256, 1024), device=DEV, generator=g).to(BF)
    n0 = lib.mafed_gemm_pp_launches()
    ops.gemm(d, a, True, False, out_dtype=F32)
    assert lib.mafed_gemm_pp_launches() == n0


def test_grouped_call_falls_back_to_one_launch_per_problem():
    """Shapes no persistent configuration tiles (or more than 16 problems): mafed_gemm_grouped runs the products one by one through the
    ordinary dispatcher -- same results, no persistent launch."""
    ops = _ops()
    lib = _lib()
    lib.mafed_gemm_set_variant(701)
    g = torch.Generator().manual_seed(10)
    shapes = [(200, 128, 64), (128, 256, 192), (96, 64, 128)]          # ragged rows / short reductions: nothing for 144 / 128 / 256-row tiles
    probs, refs = [], []
    for (M, N, K) in shapes:
        A = _int_mat((K, M), g).to(DEV, BF)
        B = _int_mat((K, N), g).to(DEV, BF)
        c0 = _int_mat((M, N), g).to(DEV, F32)
        probs.append(dict(A=A, B=B, out=c0.clone(), beta=1.0))
        refs.append(_ref(A.float().cpu(), B.float().cpu(), True, False) + c0.double().cpu())
    n0 = lib.mafed_gemm_pp_launches()
    ops.gemm_grouped(probs, True, False)
    assert lib.mafed_gemm_pp_launches() == n0
    for q, r in zip(probs, refs):
        assert float((q["out"].double().cpu() - r).abs().max()) == 0.0
    # 17 tileable problems exceed one launch's table: also one by one (each may still take the persistent kernel on its own)
    many = []
    for i in range(17):
        A = _int_mat((256, 128), g).to(DEV, BF)
        B = _int_mat((256, 256), g).to(DEV, BF)
        many.append(dict(A=A, B=B, out=torch.zeros((128, 256), dtype=F32, device=DEV), beta=1.0))
    ops.gemm_grouped(many, True, False)
    for q in many:
        r = _ref(q["A"].float().cpu(), q["B"].float().cpu(), True, False)
        assert float((q["out"].double().cpu() - r).abs().max()) == 0.0


# ---- ticketed tile order (round 4) --------------------------------------------------------------------------------------------------
class tile_order:
    """mafed_gemm_set_variant(720 / 721) around a block; restores the mode that was set."""

    def __init__(self, ticketed):
        self.v = 721 if ticketed else 720

    def __enter__(self):
        self.lib = _lib()
        self.prev = self.lib.mafed_gemm_get_variant(72)
        self.lib.mafed_gemm_set_variant(self.v)

    def __exit__(self, *a):
        self.lib.mafed_gemm_set_variant(self.prev)


def _occupy(stream, blocks, cycles=int(8e6)):
    """`blocks` CUs held by a sleeping 512-thread workgroup each for ~4 ms (mafed_tune_occupy), launched on `stream`."""
    from mafed_amd import _lib as L
    L.check(_lib().mafed_tune_occupy(blocks, 96 * 1024, cycles, stream.cuda_stream), "occupy")


TICKET_COMBOS = [(0, False, True, BF), (0, False, True, F32), (0, False, False, BF), (1, True, False, F32)]


@pytest.mark.parametrize("cfg,tA,tB,od", TICKET_COMBOS)
@pytest.mark.parametrize("occupied", [0, 24, 250])
def test_pp_ticketed_order_is_exact_with_cus_taken(cfg, tA, tB, od, occupied):
    """Every tile exactly once, whatever part of the chip another stream holds: integer operands, several rounds of short tiles (K = 256:
    the ticket draws follow each other as fast as they ever will), launches back to back on one stream (the two ticket slots alternate)
    while `occupied` CUs are held by a sleeping kernel -- 250 leaves six blocks to draw every tile.  The fp32 cases accumulate into C
    (beta = 1): a tile computed twice, or never, shows."""
    ops = _ops()
    lib = _lib()
    g = torch.Generator().manual_seed(21 + cfg)
    tiles_m, tiles_n, K = 40, 12, 256          # 480 tiles: one static round + 224 drawn
    M, N = TM[cfg] * tiles_m, 256 * tiles_n
    A = _int_mat((K, M) if tA else (M, K), g)
    B = _int_mat((N, K) if tB else (K, N), g)
    ref = _ref(A, B, tA, tB)
    Ad, Bd = A.to(DEV, BF), B.to(DEV, BF)
    side = torch.cuda.Stream()
    with tile_order(True), forced(cfg, expect=3):
        n0 = lib.mafed_gemm_get_variant(73)
        torch.cuda.synchronize()
        if occupied:
            _occupy(side, occupied)
            torch.cuda._sleep(400000)      # the occupier's blocks take their CUs first
        outs = []
        for rep in range(3):
            if od == F32:
                C = torch.full((M, N), float(rep + 1), dtype=F32, device=DEV)
                ops.gemm(Ad, Bd, tA, tB, out=C, beta=1.0)
            else:
                C = ops.gemm(Ad, Bd, tA, tB, out_dtype=od)
            outs.append(C)
        torch.cuda.synchronize()
        assert lib.mafed_gemm_get_variant(73) - n0 == 3, "the launches did not run in ticketed order"
    for rep, C in enumerate(outs):
        want = ref + (rep + 1) if od == F32 else ref.to(BF).double()
        err = float((C.double().cpu() - want).abs().max())
        assert err == 0.0, f"launch {rep}: max err {err} (cfg {cfg}, {occupied} CUs taken)"


def test_pp_ticketed_order_two_streams_at_once():
    """Ticketed launches of two streams at the same time use different slots: both exact."""
    ops = _ops()
    g = torch.Generator().manual_seed(31)
    M, N, K = 144 * 40, 256 * 12, 256
    A, B = _int_mat((M, K), g), _int_mat((N, K), g)
    ref = _ref(A, B, False, True).to(BF).double()
    Ad, Bd = A.to(DEV, BF), B.to(DEV, BF)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    with tile_order(True):
        for _ in range(4):
            for s in (s1, s2):
                with torch.cuda.stream(s):
                    outs.append(ops.gemm(Ad, Bd, False, True, out_dtype=BF))
        torch.cuda.synchronize()
    for C in outs:
        assert float((C.double().cpu() - ref).abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["gelu", "gelu_bwd_colsum", "residual"])
def test_pp_ticketed_order_equals_static_order_with_epilogues(mode):
    """The step's fused epilogues over several rounds: ticketed order == static order, bit for bit (column sums: same tiles, another
    order of the float atomics)."""
    from mafed_amd._lib import EPI_GELU, EPI_GELU_BWD
    ops = _ops()
    g = torch.Generator().manual_seed(41)
    M, N, K = 144 * 24, 256 * 16, 512     # 384 tiles
    A = (torch.randn(M, K, generator=g) * 0.5).to(DEV, BF)
    bias = torch.randn(N, generator=g).to(DEV)
    res = {}
    for ticketed in (False, True):
        with tile_order(ticketed):
            if mode == "gelu":
                W = (torch.randn(N, K, generator=g.manual_seed(42)) * 0.5).to(DEV, BF)
                aux = torch.empty((M, N), dtype=BF, device=DEV)
                y = ops.gemm(A, W, False, True, bias=bias, epilogue=EPI_GELU, aux=aux)
                res[ticketed] = (y, aux)
            elif mode == "gelu_bwd_colsum":
                W = (torch.randn(K, N, generator=g.manual_seed(43)) * 0.5).to(DEV, BF)
                u = torch.randn(M, N, generator=g.manual_seed(44)).to(DEV, BF)
                cs = torch.zeros(N, dtype=F32, device=DEV)
                y = ops.gemm(A, W, False, False, epilogue=EPI_GELU_BWD, aux=u, colsum=cs)
                res[ticketed] = (y, cs)
            else:
                W = (torch.randn(N, K, generator=g.manual_seed(45)) * 0.5).to(DEV, BF)
                r1 = torch.randn(M, N, generator=g.manual_seed(46)).to(DEV, BF)
                r2 = torch.randn(M, N, generator=g.manual_seed(47)).to(DEV)
                lib = _lib()
                lib.mafed_gemm_set_variant(710)   # (the fp32 + two-residual epilogue takes the persistent kernel only when forced)
                try:
                    y = ops.gemm(A, W, False, True, bias=bias, out_dtype=F32, res1=r1, res2=r2)
                finally:
                    lib.mafed_gemm_set_variant(701)
                res[ticketed] = (y,)
    torch.cuda.synchronize()
    assert torch.equal(res[False][0], res[True][0])
    if mode == "gelu":
        assert torch.equal(res[False][1], res[True][1])
    if mode == "gelu_bwd_colsum":
        a, b = res[False][1], res[True][1]
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max())
