// Host side of libce_hip.so: error reporting, the pinned host table (CachedParamMgr.weight,
// SURVEY.md Appendix A.1 / pin_weight at benchmark/benchmark_fbgemm_uvm.py:98-105) and its
// multi-threaded initialisation (CachedEmbeddingBag weight init uniform_(-1/N, 1/N), A.7 --
// the reference fills 91 GB with one thread).
#include <ctype.h>
#include <sched.h>
#include <stdarg.h>
#include <stdlib.h>
#include <sys/mman.h>

#include <algorithm>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "ce_common.h"

namespace ce {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// CPUs next to the current GPU (its PCI function's local_cpulist) that this process may run on
static bool gpu_local_cpus(cpu_set_t* out) {
  int dev = 0;
  char bdf[64] = {0};
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, dev) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
  char path[160];
  snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
  FILE* f = fopen(path, "r");
  if (!f) return false;
  char line[1024] = {0};
  const bool got = fgets(line, sizeof line, f) != nullptr;
  fclose(f);
  if (!got) return false;
  cpu_set_t allowed;
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
  CPU_ZERO(out);
  int n = 0;
  char* save = nullptr;      // (several library threads come through here at once: no strtok)
  for (char* tok = strtok_r(line, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {
    int a = 0, b = 0;
    const int k = sscanf(tok, "%d-%d", &a, &b);
    if (k < 1) continue;
    if (k == 1) b = a;
    for (int c = a; c <= b && c < CPU_SETSIZE; ++c)
      if (c >= 0 && CPU_ISSET(c, &allowed)) {
        CPU_SET(c, out);
        ++n;
      }
  }
  return n > 0;
}

bool near_gpu_cpus(cpu_set_t* out) {
  static const int on = [] { const char* e = getenv("CE_NUMA_BIND"); return e ? atoi(e) : 1; }();
  return on != 0 && gpu_local_cpus(out);
}

void bind_thread_near_gpu() {
  cpu_set_t set;
  if (near_gpu_cpus(&set)) (void)sched_setaffinity(0, sizeof set, &set);
}

template <typename F>
static void parallel_for(int64_t n, int threads, F fn) {
  if (threads < 1) threads = 1;
  const int64_t min_chunk = 1 << 16;
  int t = (int)std::min<int64_t>(threads, std::max<int64_t>(1, n / min_chunk));
  if (t <= 1) {
    fn(0, n);
    return;
  }
  std::vector<std::thread> pool;
  const int64_t per = (n + t - 1) / t;
  for (int i = 0; i < t; ++i) {
    const int64_t lo = i * per, hi = std::min<int64_t>(n, lo + per);
    if (lo >= hi) break;
    pool.emplace_back([=] { fn(lo, hi); });
  }
  for (auto& th : pool) th.join();
}

}  // namespace ce

using namespace ce;

extern "C" int ce_version(void) { return CE_API_VERSION; }
extern "C" const char* ce_last_error(void) { return g_err; }

// Blocks that came from mmap + MADV_HUGEPAGE + hipHostRegister (ce_host_free has to undo exactly that)
static std::mutex g_huge_m;
static std::unordered_map<void*, size_t> g_huge;

static void touch_pages(void* p, size_t bytes, int threads) {
  // touch every page from several threads so the zero-fill is not a single-thread walk; threads of their own (never
  // the caller's), pinned next to the GPU: first touch decides which NUMA node a page lives on, and the swap reads
  // and writes single rows of this block over the GPU's PCIe link
  int dev = 0;
  const bool have_dev = hipGetDevice(&dev) == hipSuccess;
  const int64_t pages = (int64_t)(bytes / 4096) + 1;
  const int t = (int)std::max<int64_t>(1, std::min<int64_t>(threads, pages / 4096 + 1));
  const int64_t per = (pages + t - 1) / t;
  std::vector<std::thread> pool;
  for (int i = 0; i < t; ++i) {
    const int64_t lo = i * per, hi = std::min<int64_t>(pages, lo + per);
    if (lo >= hi) break;
    pool.emplace_back([=] {
      if (have_dev) (void)hipSetDevice(dev);
      bind_thread_near_gpu();
      volatile char* c = (volatile char*)p;
      for (int64_t j = lo; j < hi; ++j) {
        const size_t off = (size_t)j * 4096;
        if (off < bytes) c[off] = 0;
      }
    });
  }
  for (auto& th : pool) th.join();
}

// The swap workers' helper threads move single 512-byte rows between the table and pinned staging, every row on a
// page of its own: on 4 KB pages each costs a TLB miss -- profiles/probes/probe_scatter2.cpp on the bench box's EPYC 9575F,
// 48 k rows, 6 threads: 71 ns per row and thread (0.57 ms per write-back job), 18.5 ns (0.15 ms) on 2 MB pages.
// hipHostMalloc gives 4 KB pages where transparent huge pages are in `madvise` mode, so a large block is mapped here,
// 2 MB-aligned, advised MADV_HUGEPAGE, first-touched and then registered.  CE_HOST_THP=0 keeps hipHostMalloc.
static void* alloc_huge(size_t bytes, int threads, void** dev_ptr) {
  constexpr size_t kHuge = (size_t)2 << 20;
  const size_t len = (bytes + kHuge - 1) / kHuge * kHuge;
  char* raw = (char*)mmap(nullptr, len + kHuge, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (raw == (char*)MAP_FAILED) return nullptr;
  char* p = (char*)(((uintptr_t)raw + kHuge - 1) / kHuge * kHuge);
  if (p > raw) (void)munmap(raw, (size_t)(p - raw));
  if (p + len < raw + len + kHuge) (void)munmap(p + len, (size_t)(raw + len + kHuge - (p + len)));
  (void)madvise(p, len, MADV_HUGEPAGE);
  // not inherited by fork() (DataLoader workers): a child would have to copy the page tables of the whole block, and a
  // parent write after the fork would COW-move the page away from the physical page the GPU mapping points at
  (void)madvise(p, len, MADV_DONTFORK);
  touch_pages(p, len, threads);
  if (hipHostRegister(p, len, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    (void)munmap(p, len);
    return nullptr;
  }
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) {
    (void)hipGetLastError();
    d = p;
  }
  *dev_ptr = d;
  std::lock_guard<std::mutex> g(g_huge_m);
  g_huge[p] = len;
  return p;
}

extern "C" int ce_host_alloc(size_t bytes, int threads, void** host_ptr, void** dev_ptr) {
  CE_REQUIRE(host_ptr && dev_ptr && bytes > 0, CE_ERR_INVALID, "bad arguments");
  static const int thp_env = [] { const char* e = getenv("CE_HOST_THP"); return e ? atoi(e) : 1; }();
  if (thp_env != 0 && bytes >= ((size_t)64 << 20)) {
    void* d = nullptr;
    if (void* p = alloc_huge(bytes, threads, &d)) {
      *host_ptr = p;
      *dev_ptr = d;
      return CE_OK;
    }
    // (no address space / registration refused: the plain pinned allocation below)
  }
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable);
  if (e != hipSuccess) {
    set_error("hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    return CE_ERR_NOMEM;
  }
  touch_pages(p, bytes, threads);
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) {
    (void)hipGetLastError();
    d = p;
  }
  *host_ptr = p;
  *dev_ptr = d;
  return CE_OK;
}

extern "C" int ce_host_free(void* host_ptr) {
  if (!host_ptr) return CE_OK;
  size_t len = 0;
  {
    std::lock_guard<std::mutex> g(g_huge_m);
    auto it = g_huge.find(host_ptr);
    if (it != g_huge.end()) len = it->second;
  }
  if (len) {
    // the entry goes only once the registration is undone: a failed unregister leaves the block known as mmap'd
    // memory, so a retry does not fall through to hipHostFree
    CE_HIP_CHECK(hipHostUnregister(host_ptr));
    {
      std::lock_guard<std::mutex> g(g_huge_m);
      g_huge.erase(host_ptr);
    }
    (void)munmap(host_ptr, len);
    return CE_OK;
  }
  CE_HIP_CHECK(hipHostFree(host_ptr));
  return CE_OK;
}

extern "C" int ce_host_register(void* host_ptr, size_t bytes, void** dev_ptr) {
  CE_REQUIRE(host_ptr && dev_ptr && bytes > 0, CE_ERR_INVALID, "bad arguments");
  hipError_t e = hipHostRegister(host_ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  if (e != hipSuccess) {
    set_error("hipHostRegister(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    return CE_ERR_HIP;
  }
  void* d = nullptr;
  e = hipHostGetDevicePointer(&d, host_ptr, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    d = host_ptr;
  }
  *dev_ptr = d;
  return CE_OK;
}

extern "C" int ce_host_unregister(void* host_ptr) {
  if (!host_ptr) return CE_OK;
  CE_HIP_CHECK(hipHostUnregister(host_ptr));
  return CE_OK;
}

extern "C" int ce_host_fill_uniform(float* dst, int64_t n, float lo, float hi, uint64_t seed, int threads) {
  CE_REQUIRE(dst && n >= 0, CE_ERR_INVALID, "bad arguments");
  const float span = hi - lo;
  parallel_for(n, threads, [=](int64_t a, int64_t b) {
    for (int64_t i = a; i < b; ++i) {
      const uint64_t r = splitmix64(seed * 0xD6E8FEB86659FD93ull + (uint64_t)i);
      const float u = (float)(r >> 40) * (1.0f / 16777216.0f);   // 24 random bits -> [0, 1)
      dst[i] = lo + span * u;
    }
  });
  return CE_OK;
}

// ---- rows of the host table by number: regenerated from the seed, or read back through the device mapping ----------

namespace ce {

// lo + span * u with TWO roundings, like the host loop (x86-64 without FMA): the device compiler would contract it
__device__ __forceinline__ float mul_then_add(float lo, float span, float u) {
#pragma clang fp contract(off)
  const float t = span * u;
  return lo + t;
}

// the values ce_host_fill_uniform gave rows[i] (same counter-based generator, same two roundings: the host code is
// compiled without FMA contraction, so multiply and add stay separate here)
__global__ __launch_bounds__(256) void k_fill_uniform_rows(const int64_t* __restrict__ rows, int64_t n, int dim,
                                                           float lo, float span, uint64_t seed_mul, float* out) {
  const int64_t total = n * dim;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t i = e / dim;
    const int d = (int)(e - i * dim);
    uint64_t x = seed_mul + (uint64_t)(rows[i] * dim + d);
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    const float u = (float)(x >> 40) * (1.0f / 16777216.0f);
    out[e] = mul_then_add(lo, span, u);
  }
}

// out[i] = table[rows[i]] over the device mapping of the pinned table (PCIe reads; lane group per row, R in flight)
template <typename VT>
__global__ __launch_bounds__(256) void k_host_rows_gather(const VT* __restrict__ table, int64_t num_rows,
                                                          const int64_t* __restrict__ rows, int64_t n, int rowlen,
                                                          int g_log2, VT* __restrict__ out) {
  constexpr int R = 8;
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> g_log2;
  for (int64_t i = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2) * R; i < n; i += gstride * R) {
    for (int c0 = 0; c0 < rowlen; c0 += G) {
      VT v[R];
      const int ch = c0 + gl;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        v[t] = vzero<VT>();
        if (i + t < n && ch < rowlen) {
          const int64_t r = rows[i + t];
          if ((uint64_t)r < (uint64_t)num_rows) v[t] = table[r * rowlen + ch];
        }
      }
#pragma unroll
      for (int t = 0; t < R; ++t)
        if (i + t < n && ch < rowlen) out[(i + t) * rowlen + ch] = v[t];
    }
  }
}

// box probe: a streaming read and a fill of `n4` 16-byte words
__global__ __launch_bounds__(256) void k_probe_read(const f32x4* __restrict__ src, int64_t n4, f32x4* sink) {
  constexpr int U = 8;
  f32x4 acc = {0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 256;
      v[u] = f32x4{0, 0, 0, 0};
      if (j < n4) v[u] = __builtin_nontemporal_load(src + j);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_probe_fill(f32x4* __restrict__ dst, int64_t n4, float val) {
  constexpr int U = 8;
  const f32x4 v = {val, val, val, val};
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 256;
      if (j < n4) dst[j] = v;
    }
  }
}

}  // namespace ce

extern "C" int ce_host_fill_uniform_rows(const int64_t* rows, int64_t n, int32_t dim, float lo, float hi,
                                         uint64_t seed, float* out, ce_stream_t stream) {
  if (n == 0) return CE_OK;
  CE_REQUIRE(rows && out && n > 0 && dim > 0, CE_ERR_INVALID, "bad arguments");
  hipLaunchKernelGGL(k_fill_uniform_rows, dim3(grid_for(n * dim, 256 * 4)), dim3(256), 0, (hipStream_t)stream, rows, n,
                     (int)dim, lo, hi - lo, seed * 0xD6E8FEB86659FD93ull, out);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int ce_host_rows_gather(const float* table_dev, int64_t num_rows, int32_t dim, const int64_t* rows,
                                   int64_t n, float* out, ce_stream_t stream) {
  if (n == 0) return CE_OK;
  CE_REQUIRE(table_dev && rows && out && n > 0 && dim > 0 && num_rows > 0, CE_ERR_INVALID, "bad arguments");
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  const bool vec = (dim % 4 == 0) && al16(table_dev) && al16(out);
  const int rowlen = vec ? dim / 4 : dim;
  int g = 1, gl2 = 0;
  while (g < rowlen && g < 64) { g <<= 1; ++gl2; }
  const int grid = grid_for(n, (256 >> gl2) * 8);
  if (vec)
    hipLaunchKernelGGL((k_host_rows_gather<f32x4>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)table_dev, num_rows, rows, n, rowlen, gl2, (f32x4*)out);
  else
    hipLaunchKernelGGL((k_host_rows_gather<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, table_dev,
                       num_rows, rows, n, rowlen, gl2, out);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int ce_box_probe(void* scratch, size_t bytes, int32_t reps, double* read_GBps, double* fill_GBps,
                            ce_stream_t stream) {
  CE_REQUIRE(scratch && bytes >= (1u << 20) && reps > 0 && read_GBps && fill_GBps, CE_ERR_INVALID, "bad arguments");
  CE_REQUIRE((((uintptr_t)scratch) & 15) == 0, CE_ERR_INVALID, "scratch must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n4 = (int64_t)(bytes / 16);
  hipEvent_t e0, e1, e2;
  CE_HIP_CHECK(hipEventCreate(&e0));
  CE_HIP_CHECK(hipEventCreate(&e1));
  CE_HIP_CHECK(hipEventCreate(&e2));
  const dim3 grid(kNumCU * 8), block(256);
  // one untimed launch of each first (page-table walks, clocks), then `reps` back to back between two events
  hipLaunchKernelGGL(k_probe_fill, grid, block, 0, s, (f32x4*)scratch, n4, 0.f);
  hipLaunchKernelGGL(k_probe_read, grid, block, 0, s, (const f32x4*)scratch, n4, (f32x4*)scratch);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(k_probe_read, grid, block, 0, s, (const f32x4*)scratch, n4, (f32x4*)scratch);
  (void)hipEventRecord(e1, s);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_probe_fill, grid, block, 0, s, (f32x4*)scratch, n4, 0.f);
  (void)hipEventRecord(e2, s);
  hipError_t e = hipEventSynchronize(e2);
  float ms_r = 0, ms_f = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms_r, e0, e1);
  if (e == hipSuccess) e = hipEventElapsedTime(&ms_f, e1, e2);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipEventDestroy(e2);
  if (e != hipSuccess) {
    set_error("box probe failed: %s", hipGetErrorString(e));
    return CE_ERR_HIP;
  }
  *read_GBps = (double)n4 * 16 * reps / (ms_r * 1e-3) / 1e9;
  *fill_GBps = (double)n4 * 16 * reps / (ms_f * 1e-3) / 1e9;
  return CE_OK;
}

// rows of a [rows, dim] buffer visited the way the hook-folded forward stores them / the streaming backward reads them:
// consecutive lane groups take rows `fold` apart (i = f * (rows / fold) + b  ->  row b * fold + f)
template <bool WRITE>
__global__ __launch_bounds__(256) void k_probe_rows(f32x4* buf, int64_t rows, int rowlen, int g_log2, int64_t fold) {
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t per = rows / fold;
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> g_log2;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2; i < per * fold; i += gstride) {
    const int64_t f = i / per, b = i - f * per;
    f32x4* row = buf + (b * fold + f) * rowlen;
    for (int c = gl; c < rowlen; c += G) {
      if (WRITE) __builtin_nontemporal_store(acc, row + c);
      else acc += __builtin_nontemporal_load(row + c);
    }
  }
  if (!WRITE && acc.x == 12345.678f) buf[0] = acc;
}

extern "C" int ce_probe_rows(float* buf, int64_t rows, int32_t dim, int64_t fold, int32_t reps, double* write_us,
                             double* read_us, ce_stream_t stream) {
  CE_REQUIRE(buf && rows > 0 && dim > 0 && dim % 4 == 0 && fold > 0 && rows % fold == 0 && reps > 0 && write_us && read_us,
             CE_ERR_INVALID, "bad arguments (dim must be a multiple of 4, fold must divide rows)");
  CE_REQUIRE((((uintptr_t)buf) & 15) == 0, CE_ERR_INVALID, "buffer must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int rowlen = dim / 4;
  int g = 1, gl2 = 0;
  while (g < rowlen && g < 64) { g <<= 1; ++gl2; }
  hipEvent_t e0, e1, e2;
  CE_HIP_CHECK(hipEventCreate(&e0));
  CE_HIP_CHECK(hipEventCreate(&e1));
  CE_HIP_CHECK(hipEventCreate(&e2));
  const dim3 grid(kNumCU * 8), block(256);
  hipLaunchKernelGGL((k_probe_rows<true>), grid, block, 0, s, (f32x4*)buf, rows, rowlen, gl2, fold);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_probe_rows<true>), grid, block, 0, s, (f32x4*)buf, rows, rowlen, gl2, fold);
  (void)hipEventRecord(e1, s);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_probe_rows<false>), grid, block, 0, s, (f32x4*)buf, rows, rowlen, gl2, fold);
  (void)hipEventRecord(e2, s);
  hipError_t e = hipEventSynchronize(e2);
  float ms_w = 0, ms_r = 0;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms_w, e0, e1);
  if (e == hipSuccess) e = hipEventElapsedTime(&ms_r, e1, e2);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipEventDestroy(e2);
  if (e != hipSuccess) {
    set_error("row probe failed: %s", hipGetErrorString(e));
    return CE_ERR_HIP;
  }
  *write_us = (double)ms_w * 1e3 / reps;
  *read_us = (double)ms_r * 1e3 / reps;
  return CE_OK;
}

extern "C" int ce_stream_create_cu_mask(const uint32_t* cu_mask, int32_t words, ce_stream_t* out) {
  CE_REQUIRE(out && words >= 0 && (words == 0 || cu_mask), CE_ERR_INVALID, "bad arguments");
  hipStream_t s = nullptr;
  if (words == 0) {
    CE_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  } else {
    CE_HIP_CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask));
  }
  *out = (ce_stream_t)s;
  return CE_OK;
}

extern "C" int ce_stream_destroy(ce_stream_t stream) {
  if (!stream) return CE_OK;
  CE_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
  return CE_OK;
}
