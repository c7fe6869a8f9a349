// Host side of libce_hip.so: error reporting, the pinned host table (CachedParamMgr.weight,
// SURVEY.md Appendix A.1 / pin_weight at benchmark/benchmark_fbgemm_uvm.py:98-105) and its
// multi-threaded initialisation (CachedEmbeddingBag weight init uniform_(-1/N, 1/N), A.7 --
// the reference fills 91 GB with one thread).
#include <stdarg.h>

#include <algorithm>
#include <thread>
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

extern "C" int ce_host_alloc(size_t bytes, int threads, void** host_ptr, void** dev_ptr) {
  CE_REQUIRE(host_ptr && dev_ptr && bytes > 0, CE_ERR_INVALID, "bad arguments");
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocPortable);
  if (e != hipSuccess) {
    set_error("hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    return CE_ERR_NOMEM;
  }
  // touch every page from several threads so the zero-fill is not a single-thread walk
  parallel_for((int64_t)(bytes / 4096) + 1, threads, [=](int64_t lo, int64_t hi) {
    volatile char* c = (volatile char*)p;
    for (int64_t i = lo; i < hi; ++i) {
      const size_t off = (size_t)i * 4096;
      if (off < bytes) c[off] = 0;
    }
  });
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
