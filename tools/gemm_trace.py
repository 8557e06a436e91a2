"""Per-block phase timeline of one bf16 GEMM (needs a -DMAFED_GEMM_TRACE build: tools/build_variant.sh trace gemm -DMAFED_GEMM_TRACE,
then MAFED_HIP_LIB=mafed_amd/lib_trace.so python tools/gemm_trace.py [M N K transA transB epi]).
Shows how long the DMA prologue, the MFMA loop and the store epilogue of a block take and how the blocks that share a CU
line up in time (in phase = both idle the matrix pipe at once)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from collections import defaultdict
import torch
from mafed_amd import ops, _lib

M, N, K, tA, tB, epi = 9216, 4096, 1024, 0, 1, 1
if len(sys.argv) > 6:
    M, N, K, tA, tB, epi = [int(v) for v in sys.argv[1:7]]
lib = _lib.load()
for pv in os.environ.get("GEMM_BENCH_PRE", "").split(","):
    if pv:
        lib.mafed_gemm_set_variant(int(pv))
lib.mafed_gemm_set_trace.argtypes = [C.c_void_p]
lib.mafed_gemm_set_trace.restype = C.c_int
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn((K, M) if tA else (M, K), device="cuda", generator=g).to(torch.bfloat16)
B = torch.randn((N, K) if tB else (K, N), device="cuda", generator=g).to(torch.bfloat16)
out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
kw = dict(bias=torch.randn(N, device="cuda"), epilogue=ops.EPI_GELU, aux=torch.empty_like(out)) if epi else {}
for _ in range(3):
    ops.gemm(A, B, bool(tA), bool(tB), out=out, **kw)
buf = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
assert lib.mafed_gemm_set_trace(buf.data_ptr()) == 0
ops.gemm(A, B, bool(tA), bool(tB), out=out, **kw)
torch.cuda.synchronize()
lib.mafed_gemm_set_trace(None)
t = buf.view(-1, 8).cpu()
nb = int((t[:, 2] != 0).sum())
t = t[:nb]
t0 = int(t[:, 2].min())
us = lambda v: (int(v) - t0) / 100.0   # 100 MHz
pro = [(int(r[3]) - int(r[2])) / 100.0 for r in t]
loop = [(int(r[4]) - int(r[3])) / 100.0 for r in t]
ep = [(int(r[5]) - int(r[4])) / 100.0 for r in t]
mean = lambda x: sum(x) / len(x)
print(f"{nb} blocks; kernel span {us(t[:, 5].max()):.1f} us; prologue {mean(pro):.2f} us, loop {mean(loop):.2f} us, epilogue {mean(ep):.2f} us (means)")
cus = defaultdict(list)
for i, r in enumerate(t):
    hw, xcc = int(r[0]), int(r[1]) & 0xf
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
    cus[(xcc, se, sh, cu)].append((us(r[2]), us(r[3]), us(r[4]), us(r[5]), i))
print(f"{len(cus)} distinct (xcc, se, sh, cu) ids; blocks per CU: min {min(len(v) for v in cus.values())} max {max(len(v) for v in cus.values())}")
# how often are both residents of a CU in their epilogue (or prologue) at the same time?
both, one = 0.0, 0.0
for v in cus.values():
    ev = []
    for a, b, c, d, _ in v:
        ev += [(a, 1), (b, -1), (c, 1), (d, -1)]   # "not in the MFMA loop" intervals: [start, landed) and [loop end, end)
    ev.sort()
    depth, last = 0, 0.0
    for tt, dd in ev:
        if depth >= 2:
            both += tt - last
        elif depth == 1:
            one += tt - last
        last, depth = tt, depth + dd
n = len(cus)
print(f"per CU: {one / n:.1f} us with exactly one resident outside its MFMA loop, {both / n:.1f} us with two or more (matrix pipe idle)")
for key in list(sorted(cus))[:3]:
    print("CU", key)
    for a, b, c, d, i in sorted(cus[key]):
        print(f"   block {i:5d}: start {a:7.2f}  landed {b:7.2f}  loop end {c:7.2f}  end {d:7.2f}")
