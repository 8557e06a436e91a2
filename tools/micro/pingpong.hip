// Hand-over latency between two workgroups through device memory on MI355X (run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pingpong.hip -o /tmp/pingpong && /tmp/pingpong).
// Workgroup A bumps a counter (agent-scope atomic), workgroup B polls it with agent-scope loads and answers on a second counter; the
// round trip / 2 is one hand-over.  Variants: partner on the same XCD (workgroup ids 0 and 8) or the next one (0 and 1); with a data word
// written (sc1 store + vmcnt(0)) before the signal and read (sc1 load) after it, as the decode kernels do.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(64) void pingpong(unsigned* flags, float* data, int partner, int iters, int with_data, long long* cycles, unsigned* xcc) {
  const int b = blockIdx.x;
  if (b != 0 && b != partner) return;
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[b == 0 ? 0 : 1] = id & 15u;
  }
  unsigned* mine = flags + (b == 0 ? 0 : 64);
  unsigned* theirs = flags + (b == 0 ? 64 : 0);
  float acc = 0.f;
  const long long t0 = wall_clock64();
  for (int it = 1; it <= iters; ++it) {
    if (b == 0) {
      if (with_data) { __hip_atomic_store(data, (float)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
      if (with_data) acc += __hip_atomic_load(data + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) __builtin_amdgcn_s_sleep(1);
      if (with_data) {
        acc += __hip_atomic_load(data, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(data + 64, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0 && b == 0) { cycles[0] = t1 - t0; data[128] = acc; }
}

int main() {
  unsigned *flags, *xcc;
  float* data;
  long long* cyc;
  hipMalloc(&flags, 1024); hipMalloc(&data, 1024); hipMalloc(&cyc, 64); hipMalloc(&xcc, 64);
  int rate = 0;
  hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
  const int iters = 2000;
  for (int with_data = 0; with_data < 2; ++with_data)
    for (int partner : {8, 1, 4}) {
      hipMemset(flags, 0, 1024); hipMemset(data, 0, 1024);
      pingpong<<<16, 64>>>(flags, data, partner, iters, with_data, cyc, xcc);
      hipDeviceSynchronize();
      long long c; unsigned x[2];
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
      const double us = (double)c / (double)rate * 1e3 / iters;
      printf("partner workgroup %d (XCC %u <-> %u)%s: round trip %.2f us, one hand-over %.2f us\n", partner, x[0], x[1],
             with_data ? ", data word written before / read after the signal" : "", us, us / 2);
    }
  return 0;
}
