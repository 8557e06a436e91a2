"""Per-phase cycle breakdown of the ping-pong GEMM (needs a -DMAFED_PP_TRACE build: tools/build_variant.sh pptrace gemm_pp -DMAFED_PP_TRACE,
then MAFED_HIP_LIB=mafed_amd/lib_pptrace.so python tools/pp_trace.py M N K transA transB out_f32 [epi]).
Stamps (s_memtime, shader cycles) of waves 0 (group 0) and 4 (group 1) of the first blocks: L start, DMA issued, counted wait passed,
barrier 1 passed, MFMA cluster issued."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import ops, _lib

M, N, K, tA, tB, f32 = [int(v) for v in sys.argv[1:7]] if len(sys.argv) > 6 else (9216, 4096, 1024, 0, 1, 0)
epi = int(sys.argv[7]) if len(sys.argv) > 7 else 0
lib = _lib.load()
ZK = os.environ.get("PP_TRACE_Z", "0") == "1"   # trace the 256 x 256 kernel (gemm_z.hip; force it with GEMM_BENCH_PRE=712)
set_trace = lib.mafed_gemm_z_set_trace if ZK else lib.mafed_gemm_pp_set_trace
set_trace.argtypes = [C.c_void_p]
set_trace.restype = C.c_int
for pv in os.environ.get("GEMM_BENCH_PRE", "").split(","):
    if pv:
        lib.mafed_gemm_set_variant(int(pv))
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn((K, M) if tA else (M, K), device="cuda", generator=g).to(torch.bfloat16)
B = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g).to(torch.bfloat16)
out = torch.zeros((M, N), dtype=torch.float32 if f32 else torch.bfloat16, device="cuda")
kw = {}
if epi == 1:
    kw = dict(bias=torch.randn(N, device="cuda"), epilogue=ops.EPI_GELU, aux=torch.empty_like(out))
beta = 1.0 if (tA and f32) else 0.0
for _ in range(3):
    ops.gemm(A, B, bool(tA), bool(tB), out=out, beta=beta, **kw)
REC = 240
buf = torch.zeros(8 * (REC * 2 + 1), dtype=torch.int64, device="cuda")
assert set_trace(buf.data_ptr()) == 0
n0 = lib.mafed_gemm_pp_launches()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.gemm(A, B, bool(tA), bool(tB), out=out, beta=beta, **kw)
e1.record()
torch.cuda.synchronize()
assert lib.mafed_gemm_pp_launches() == n0 + 1, "not the ping-pong kernel"
set_trace(None)
print(f"traced launch: {e0.elapsed_time(e1) * 1e3:.1f} us")
t = buf.view(8, REC * 2 + 1).cpu()
# events: 0 interval start, 1 epilogue start, 2 epilogue end, 3 next tile decoded (or kernel end)
for blk in range(2):
    for grp in range(2):
        row = t[blk * 2 + grp]
        n = int(row[0])
        r = row[1:1 + n * 2].view(n, 2)
        ts, tag = r[:, 0], r[:, 1]
        d = (ts[1:] - ts[:-1])
        kinds = {}
        for i in range(n - 1):
            key = (int(tag[i]), int(tag[i + 1]))
            kinds.setdefault(key, []).append(int(d[i]))
        print(f"block {blk} group {grp}: {n} events, span {int(ts[-1] - ts[0])} cycles")
        names = {(0, 0): "phase (interval start -> next interval start)", (0, 1): "last interval of a tile", (1, 2): "epilogue", (2, 3): "tile switch (decode)",
                 (3, 0): "accumulator reset -> first interval"}
        for key, v in sorted(kinds.items()):
            v2 = sorted(v)
            print(f"    {names.get(key, str(key)):48s} n={len(v):3d}  median {v2[len(v2) // 2]:6d}  mean {sum(v) / len(v):8.0f}  min {v2[0]:6d}  max {v2[-1]:6d}")
        if blk == 0 and grp == 0:
            print("    first 24 intervals:", " ".join(str(int(x)) for x in d[:24]))
