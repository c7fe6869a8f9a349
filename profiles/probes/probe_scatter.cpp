// host scatter micro-benchmark: contiguous staging -> 54 k random 512-B rows of a big table
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include <algorithm>
#include <stdint.h>
#include <immintrin.h>
typedef float v4f __attribute__((ext_vector_type(4)));
static inline void copy_nt16(float* d, const float* s, size_t n) { const v4f* a = (const v4f*)s; v4f* b = (v4f*)d; for (size_t i = 0; i < n / 4; ++i) __builtin_nontemporal_store(a[i], b + i); }
__attribute__((target("avx512f"))) static inline void copy_nt64(float* d, const float* s, size_t n) { for (size_t i = 0; i < n; i += 16) _mm512_stream_ps(d + i, _mm512_loadu_ps(s + i)); }
__attribute__((target("avx2"))) static inline void copy_nt32(float* d, const float* s, size_t n) { for (size_t i = 0; i < n; i += 8) _mm256_stream_ps(d + i, _mm256_loadu_ps(s + i)); }
int main() {
  const size_t N = 40000000, D = 128;
  float* tab = (float*)aligned_alloc(4096, N * D * 4);
  { std::vector<std::thread> th; for (int t = 0; t < 16; ++t) th.emplace_back([=] { size_t per = N / 16; memset(tab + t * per * D, 1, per * D * 4); }); for (auto& x : th) x.join(); }
  const size_t M = 54000;
  float* st = (float*)aligned_alloc(4096, M * D * 4); memset(st, 0, M * D * 4);
  std::vector<int> rows(M); unsigned long long x = 88172645463325252ull;
  auto run = [&](int T, int mode, bool gather) {
    double best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
      for (auto& r : rows) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; r = (int)(x % N); }
      if (gather) std::sort(rows.begin(), rows.end());
      std::atomic<long> next{0};
      auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back([&] {
        for (;;) {
          long pc = next.fetch_add(1); if (pc * 2048 >= (long)M) break;
          size_t lo = pc * 2048, hi = std::min(M, lo + 2048);
          for (size_t i = lo; i < hi; ++i) {
            float* tr = tab + (size_t)rows[i] * D; float* sr = st + i * D;
            if (gather && i + 8 < hi) { const char* q = (const char*)(tab + (size_t)rows[i + 8] * D); for (int l = 0; l < 512; l += 64) __builtin_prefetch(q + l); }
            float* d = gather ? sr : tr; const float* s = gather ? tr : sr;
            if (mode == 0) memcpy(d, s, D * 4); else if (mode == 1) copy_nt16(d, s, D); else if (mode == 2) copy_nt32(d, s, D); else copy_nt64(d, s, D);
          }
          _mm_sfence();
        }
      });
      for (auto& t : th) t.join();
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("%s T=%d %s : %.3f ms (%.1f GB/s)\n", gather ? "gather " : "scatter", T, mode == 0 ? "memcpy" : mode == 1 ? "nt16  " : mode == 2 ? "nt32  " : "nt64  ", best * 1e3, M * D * 4 / best / 1e9);
  };
  for (int T : {6, 4}) for (int g = 0; g < 2; ++g) for (int m = 0; m < 4; ++m) run(T, m, g);
  return 0;
}
