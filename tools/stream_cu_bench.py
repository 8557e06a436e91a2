"""What do a FEW CUs stream from HBM?  (round 4, DESIGN.md section 6: can the step's HBM-bound passes run as small-grid kernels on a handful
of CUs beside a GEMM that tolerates missing CUs, instead of time-slicing the whole chip?)

`mafed_tune_stream` sweeps a buffer far larger than the Infinity Cache with n workgroups of 1024 threads (one per CU) and 2 / 4 / 8 16-byte
loads in flight per thread; alone on the chip, and beside a persistent GEMM loop on the main stream."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mafed_amd import _lib, ops

dev = "cuda"
lib = _lib.load()
GB = 2.0
n = int(GB * (1 << 30)) // 64 * 64
src = torch.empty(n, dtype=torch.uint8, device=dev)
src.view(torch.float32).normal_()
dst = torch.empty_like(src)


def run(blocks, threads, unroll, mode, reps=3, stream=None):
    st = stream or torch.cuda.current_stream()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record()
            _lib.check(lib.mafed_tune_stream(src.data_ptr(), dst.data_ptr(), n, blocks, threads, unroll, mode, st.cuda_stream), "tune_stream")
            e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    moved = n * (1 if mode == 0 else 2)
    return moved / best / 1e6   # GB/s


print("# alone on the chip: GB/s total (GB/s per workgroup), 1024-thread workgroups, one per CU up to 256", flush=True)
for mode, name in ((0, "read"), (1, "copy"), (2, "adamw-shaped 4r+4w")):
    for unroll in ((4, 8) if mode != 2 else (4,)):
        row = f"{name:20s} unroll {unroll}:"
        for blocks in (1, 4, 8, 16, 32, 64, 128, 256, 512, 2048):
            bw = run(blocks, 1024, unroll, mode, reps=2 if blocks < 8 else 3)
            row += f"  {blocks}: {bw:7.0f} ({bw / min(blocks, 256):5.1f})"
        print(row, flush=True)

# beside a GEMM loop: the persistent kernels need every CU (static schedule) -- this is the situation the ticketed order is for
M, N, K = 9216, 4096, 1024
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
W = torch.randn(N, K, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
side = torch.cuda.Stream()


def gemm_loop(reps=40):
    for _ in range(reps):
        ops.gemm(A, W, False, True, out=out)


for variant in (701, 700):
    lib.mafed_gemm_set_variant(variant)
    gemm_loop(5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gemm_loop(); e1.record(); torch.cuda.synchronize()
    base = e0.elapsed_time(e1) / 40 * 1e3
    print(f"# fc1-shaped GEMM loop, variant {variant}: alone {base:.1f} us per product", flush=True)
    for blocks in (8, 16, 32, 64):
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            s0.record()
            _lib.check(lib.mafed_tune_stream(src.data_ptr(), dst.data_ptr(), n, blocks, 1024, 4, 2, side.cuda_stream), "tune_stream")
            s1.record()
        torch.cuda._sleep(100000)
        e0.record(); gemm_loop(); e1.record()
        torch.cuda.synchronize()
        print(f"  {blocks:3d} streaming workgroups (adamw-shaped) beside it: GEMM {e0.elapsed_time(e1) / 40 * 1e3:6.1f} us per product ({e0.elapsed_time(e1) / 40 * 1e3 / base:4.2f}x), "
              f"stream {2 * n / s0.elapsed_time(s1) / 1e6:6.0f} GB/s over {s0.elapsed_time(s1):.2f} ms", flush=True)
lib.mafed_gemm_set_variant(701)
