// host scatter micro-benchmark 2: contiguous staging -> random 512-B rows of a big table, with TLB-warming prefetches
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
#include <sys/mman.h>
__attribute__((target("avx512f"))) static inline void copy_nt64(float* d, const float* s, size_t n) { for (size_t i = 0; i < n; i += 16) _mm512_stream_ps(d + i, _mm512_loadu_ps(s + i)); }
int main(int argc, char** argv) {
  const size_t N = argc > 1 ? atoll(argv[1]) : 100000000, D = 128;
  const bool huge = argc > 2 && atoi(argv[2]);
  float* tab = (float*)mmap(nullptr, N * D * 4, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (huge) madvise(tab, N * D * 4, MADV_HUGEPAGE); else madvise(tab, N * D * 4, MADV_NOHUGEPAGE);
  { std::vector<std::thread> th; for (int t = 0; t < 16; ++t) th.emplace_back([=] { size_t per = N / 16; memset(tab + t * per * D, 1, per * D * 4); }); for (auto& x : th) x.join(); }
  const size_t M = 48000;
  float* st = (float*)aligned_alloc(4096, M * D * 4); memset(st, 0, M * D * 4);
  std::vector<int> rows(M); unsigned long long x = 88172645463325252ull;
  auto run = [&](int T, int mode, int ahead) {
    double best = 1e9, best_in = 1e9;
    for (int rep = 0; rep < 8; ++rep) {
      for (auto& r : rows) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; r = (int)(x % N); }
      std::atomic<long> next{0};
      std::vector<double> tin(T, 0.0);
      auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
        auto a = std::chrono::steady_clock::now();
        for (;;) {
          long pc = next.fetch_add(1); if (pc * 2048 >= (long)M) break;
          size_t lo = pc * 2048, hi = std::min(M, lo + 2048);
          for (size_t i = lo; i < hi; ++i) {
            float* tr = tab + (size_t)rows[i] * D; float* sr = st + i * D;
            if (ahead && i + ahead < hi) {
              char* q = (char*)(tab + (size_t)rows[i + ahead] * D);
              if (mode == 1) __builtin_prefetch((char*)((uintptr_t)q ^ 2048), 0, 0);            // TLB only: another line of the page
              else if (mode == 2) { for (int l = 0; l < 512; l += 64) __builtin_prefetch(q + l, 1, 0); }  // the row's lines, for writing
            }
            if (mode == 2) memcpy(tr, sr, D * 4); else copy_nt64(tr, sr, D);
          }
          _mm_sfence();
        }
        tin[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
      });
      for (auto& t : th) t.join();
      best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
      best_in = std::min(best_in, *std::max_element(tin.begin(), tin.end()));
    }
    printf("scatter T=%d %s ahead=%d : %.3f ms total, %.3f ms inner (%.1f ns/row/thread)\n", T,
           mode == 0 ? "nt64            " : mode == 1 ? "nt64+tlb prefetch" : "memcpy+prefetchw ", ahead, best * 1e3, best_in * 1e3, best_in * 1e9 * T / M);
  };
  for (int T : {6, 8}) { run(T, 0, 0); for (int a : {4, 8, 16}) run(T, 1, a); for (int a : {8, 16}) run(T, 2, a); }
  return 0;
}
