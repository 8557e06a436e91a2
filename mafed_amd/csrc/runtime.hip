// Error string + version of the C-ABI library.
#include <stdarg.h>

#include "common.h"

namespace mafed {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mafed

// ---- kernel profiler -------------------------------------------------------------------------------------------------------
#include <mutex>
#include <vector>

namespace mafed {
std::atomic<bool> g_prof_on{false};
namespace {
struct ProfRec { int tag; double work; hipEvent_t e0, e1; };
std::mutex g_prof_mu;                 // launches come from the caller's thread AND from autograd's backward thread
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;  // events are created once per process and re-used by later profiles
size_t g_prof_cap = 0;
const char* const kTagNames[K_TAG_COUNT] = {
    "gemm_bf16", "gemm_f32", "gemm_skinny", "attn_fwd", "attn_bwd_dq", "attn_bwd_dkv", "attn_exact", "layernorm_fwd", "layernorm_bwd",
    "layernorm_bwd_reduce", "ce_fwd", "ce_bwd", "distill_fwd", "distill_bwd", "adamw", "gradnorm", "embed_concat_fwd", "embed_concat_bwd",
    "colsum", "cast", "ewc", "small", "gemm_pp"};
}  // namespace

bool prof_events(int tag, double work, hipEvent_t* e0, hipEvent_t* e1) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_on.load(std::memory_order_relaxed) || g_prof_recs.size() >= g_prof_cap) return false;  // a full profile degrades to plain launches
  const size_t i = g_prof_recs.size();
  *e0 = g_prof_pool[2 * i];
  *e1 = g_prof_pool[2 * i + 1];
  g_prof_recs.push_back(ProfRec{tag, work, *e0, *e1});
  return true;
}
}  // namespace mafed

extern "C" int mafed_prof_begin(int max_records) {
  using namespace mafed;
  MAFED_CHECK_ARG(max_records > 0 && max_records <= (1 << 22), "prof_begin: max_records %d out of range", max_records);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  // one profile at a time: the event pool and the record list are process-wide, a second begin would re-use events the open
  // profile has handed out
  if (g_prof_on.load(std::memory_order_relaxed)) { set_error("prof_begin: a profile is already open (close it with mafed_prof_end)"); return MAFED_EINVAL; }
  while (g_prof_pool.size() < (size_t)2 * max_records) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { set_error("prof_begin: hipEventCreate failed"); return MAFED_ELAUNCH; }
    g_prof_pool.push_back(e);
  }
  g_prof_recs.clear();
  g_prof_cap = (size_t)max_records;
  g_prof_on.store(true, std::memory_order_release);
  return MAFED_OK;
}

extern "C" int mafed_prof_end(void) {
  std::lock_guard<std::mutex> lk(mafed::g_prof_mu);
  mafed::g_prof_on.store(false, std::memory_order_release);
  return MAFED_OK;
}

// The caller has synchronised the device.  Fills up to `max` records in launch order and returns how many a profile holds.
extern "C" int mafed_prof_collect(int* tags, double* work, float* ms, float* start_ms, int max) {
  using namespace mafed;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  const int n = (int)g_prof_recs.size();
  for (int i = 0; i < n && i < max; ++i) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, g_prof_recs[i].e0, g_prof_recs[i].e1) != hipSuccess) t = -1.f;
    if (tags) tags[i] = g_prof_recs[i].tag;
    if (work) work[i] = g_prof_recs[i].work;
    if (ms) ms[i] = t;
    if (start_ms) {  // launch time line: start of record i relative to the start of the first record
      float s0 = 0.f;
      if (i > 0 && hipEventElapsedTime(&s0, g_prof_recs[0].e0, g_prof_recs[i].e0) != hipSuccess) s0 = -1.f;
      start_ms[i] = s0;
    }
  }
  (void)hipGetLastError();
  return n;
}

extern "C" const char* mafed_prof_tag_name(int tag) { return (tag >= 0 && tag < mafed::K_TAG_COUNT) ? mafed::kTagNames[tag] : "?"; }

// ---- tuning helper ----------------------------------------------------------------------------------------------------------
// Occupies `blocks` CUs (one 512-thread block with `lds_bytes` of LDS each) for about `cycles` shader clocks: a stand-in for a
// long-running collective kernel when measuring how a GEMM behaves with part of the chip taken (tools/contention_bench.py).
namespace mafed {
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(8))) void occupy_kernel(long long cycles, unsigned* sink) {
  extern __shared__ char smem[];
  const long long t0 = (long long)__builtin_readcyclecounter();
  unsigned acc = 0;
  while ((long long)__builtin_readcyclecounter() - t0 < cycles) {   // every wave reaches this exit: bounded by the clock
    acc += (unsigned)smem[threadIdx.x & 63];
    __builtin_amdgcn_s_sleep(8);
  }
  if (acc == 0xffffffffu && sink) sink[0] = acc;
}
}  // namespace mafed
extern "C" int mafed_tune_occupy(int blocks, int lds_bytes, long long cycles, void* stream) {
  using namespace mafed;
  if (blocks < 0) {   // -n: n light blocks of 256 threads (one wave per SIMD, a handful of registers, no LDS): a co-residency probe
    MAFED_CHECK_ARG(blocks >= -4096 && cycles >= 0 && cycles <= (1ll << 33), "tune_occupy: bad arguments");
    occupy_kernel<<<dim3(-blocks), dim3(256), 0, (hipStream_t)stream>>>(cycles, nullptr);
    MAFED_CHECK_LAUNCH("tune_occupy(light)");
    return MAFED_OK;
  }
  MAFED_CHECK_ARG(blocks >= 1 && blocks <= 256 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && cycles >= 0 && cycles <= (1ll << 33),
                  "tune_occupy: bad arguments");
  if (lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
    set_error("tune_occupy: cannot raise the LDS limit");
    return MAFED_ELAUNCH;
  }
  occupy_kernel<<<dim3(blocks), dim3(512), lds_bytes, (hipStream_t)stream>>>(cycles, nullptr);
  MAFED_CHECK_LAUNCH("tune_occupy");
  return MAFED_OK;
}


// Streaming probe (tools/stream_cu_bench.py): `blocks` workgroups of `threads` threads sweep `n_bytes` of `src` with 16-byte loads,
// `unroll` independent loads in flight per thread (mode 0: read + fold, mode 1: copy to dst, mode 2: AdamW-shaped -- four read streams of
// n_bytes / 4 each, four written).  Answers: what does ONE CU (or 8, 32, ...) stream from HBM when the rest of the chip does not?
namespace mafed {
template <int U>
__global__ __launch_bounds__(1024) void stream_probe_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n16, int mode,
                                                            float* sink) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  float acc = 0.f;
  if (mode == 2) {
    const long long q = n16 / 4;
    for (; i < q; i += stride) {
      float4 a = src[i], b = src[q + i], c = src[2 * q + i], d = src[3 * q + i];
      a.x += b.x * c.x + d.x; a.y += b.y * c.y + d.y; a.z += b.z * c.z + d.z; a.w += b.w * c.w + d.w;
      dst[i] = a; dst[q + i] = b; dst[2 * q + i] = c; dst[3 * q + i] = d;
    }
  } else {
    for (; i + (U - 1) * stride < n16; i += U * stride) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (mode == 1) dst[i + u * stride] = v[u];
        else acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
      }
    }
    for (; i < n16; i += stride) {
      const float4 v = src[i];
      if (mode == 1) dst[i] = v;
      else acc += (v.x + v.y) + (v.z + v.w);
    }
  }
  if (acc == 1.2345e-30f && sink) sink[0] = acc;
}
}  // namespace mafed
// The same AdamW-shaped sweep, wrapping around the buffer until `ticks` of the 100 MHz s_memrealtime clock have passed: a stand-in for one
// bucket's all-reduce kernels (RCCL: one workgroup per channel, copy / reduce loops for the collective's duration).
namespace mafed {
__global__ __launch_bounds__(1024) void stream_for_kernel(const float4* __restrict__ src, float4* __restrict__ dst, long long n16, long long ticks) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  const long long q = n16 / 4, stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // every wave reaches the exit: the loop is bounded by the wall clock (checked once per trip: ~64 bytes per lane between checks)
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    if (i >= q) i -= (i / stride) * stride;   // wrap to this thread's first element
    float4 a = src[i], b = src[q + i], c = src[2 * q + i], d = src[3 * q + i];
    a.x += b.x * c.x + d.x; a.y += b.y * c.y + d.y; a.z += b.z * c.z + d.z; a.w += b.w * c.w + d.w;
    dst[i] = a; dst[q + i] = b; dst[2 * q + i] = c; dst[3 * q + i] = d;
    i += stride;
  }
}
}  // namespace mafed
extern "C" int mafed_tune_stream_for(const void* src, void* dst, long long n_bytes, int blocks, int threads, double microseconds, void* stream) {
  using namespace mafed;
  MAFED_CHECK_ARG(src && dst && n_bytes >= (1 << 20) && n_bytes % 64 == 0 && blocks >= 1 && blocks <= 256 && threads >= 64 && threads <= 1024 &&
                  threads % 64 == 0 && microseconds >= 0.0 && microseconds <= 1e6, "tune_stream_for: bad arguments");
  MAFED_CHECK_ARG((long long)blocks * threads <= n_bytes / 64, "tune_stream_for: buffer too small for the grid");
  stream_for_kernel<<<dim3(blocks), dim3(threads), 0, (hipStream_t)stream>>>((const float4*)src, (float4*)dst, n_bytes / 16, (long long)(microseconds * 100.0));
  MAFED_CHECK_LAUNCH("tune_stream_for");
  return MAFED_OK;
}

extern "C" int mafed_tune_stream(const void* src, void* dst, long long n_bytes, int blocks, int threads, int unroll, int mode, void* stream) {
  using namespace mafed;
  MAFED_CHECK_ARG(src && n_bytes >= 0 && n_bytes % 64 == 0 && blocks >= 1 && blocks <= 65536 && threads >= 64 && threads <= 1024 && threads % 64 == 0 &&
                  mode >= 0 && mode <= 2 && (mode == 0 || dst), "tune_stream: bad arguments");
  const long long n16 = n_bytes / 16;
  hipStream_t st = (hipStream_t)stream;
  if (unroll >= 8) stream_probe_kernel<8><<<dim3(blocks), dim3(threads), 0, st>>>((const float4*)src, (float4*)dst, n16, mode, nullptr);
  else if (unroll >= 4) stream_probe_kernel<4><<<dim3(blocks), dim3(threads), 0, st>>>((const float4*)src, (float4*)dst, n16, mode, nullptr);
  else stream_probe_kernel<2><<<dim3(blocks), dim3(threads), 0, st>>>((const float4*)src, (float4*)dst, n16, mode, nullptr);
  MAFED_CHECK_LAUNCH("tune_stream");
  return MAFED_OK;
}

extern "C" int mafed_version(void) { return 110; }
extern "C" const char* mafed_last_error_string(void) { return mafed::g_err; }
