// Forward probe (round 3): does it pay to route every lookup to the XCD that "owns" its row, so that the duplicate
// reads of a batch's hot rows hit that XCD's own 4 MB L2 instead of crossing the fabric to the Infinity Cache?
// A Criteo-1TB-shaped batch (26 features x 16384 samples, one id per bag, ~45 k distinct rows of 512 B in a 1.78 M-row
// table), output [B, F, D] written with non-temporal stores, lookups in feature-major order as in the benchmark.
//   base : lane group g takes 16 consecutive lookups at a time (the production kernel's shape)
//   xcd  : the lookups are split into 8 lists by (row & 7); workgroup w works on list w % 8 (workgroups are dealt to
//          the XCDs round-robin), so an XCD only ever reads 1/8 of the rows
//   none : 'base' on a batch of 425,984 DISTINCT rows (no reuse at all), for scale
// hipcc --offload-arch=gfx950 -O3 probe_fwd_xcd.hip -o probe_fwd_xcd
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <random>
#include <unordered_map>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// pairs[j] = (slot << 32 | out_row); list l occupies [starts[l], starts[l + 1]); workgroup w: list w % nlists, and
// within the list the (w / nlists)-th of (gridDim.x / nlists) strided workers
template <int U, int NT>
__global__ __launch_bounds__(256) void k_fwd(const f32x4* __restrict__ W, const unsigned long long* __restrict__ pairs,
                                             const int* __restrict__ starts, int nlists, f32x4* __restrict__ out,
                                             int affinity) {
  const int grp = threadIdx.x >> 5, gl = threadIdx.x & 31;
  // affinity: list = workgroup % nlists (= its XCD); control: list = (workgroup / nlists) % nlists, so that every XCD
  // works on every list (grid a multiple of nlists^2)
  const int l = affinity ? blockIdx.x % nlists : (blockIdx.x / nlists) % nlists;
  const int wg_in_list = affinity ? blockIdx.x / nlists
                                  : (int)(blockIdx.x % nlists) + nlists * (int)(blockIdx.x / (nlists * nlists));
  const int worker = wg_in_list * 8 + grp, nworkers = (gridDim.x / nlists) * 8;
  const int s0 = starts[l], s1 = starts[l + 1];
  for (int q = s0 + worker * U; q < s1; q += nworkers * U) {
    f32x4 v[U];
    unsigned orow[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      orow[u] = 0xffffffffu;
      if (q + u < s1) {
        const unsigned long long p = pairs[q + u];
        orow[u] = (unsigned)p;
        v[u] = W[(int64_t)(p >> 32) * 32 + gl];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (orow[u] != 0xffffffffu) {
        if (NT) __builtin_nontemporal_store(v[u], out + (int64_t)orow[u] * 32 + gl);
        else out[(int64_t)orow[u] * 32 + gl] = v[u];
      }
    }
  }
}

static float time_it(const std::function<void()>& fn, int reps = 30) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, 0));
    fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ts.push_back(ms * 1e3f);
  }
  CK(hipGetLastError());
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const int F = 26, B = 16384;
  const int64_t C = 1779442;
  const double s = 0.25;
  const int64_t sizes[F] = {45833188, 36746, 17245, 7413, 20243, 3, 7114, 1441, 62, 29275261, 1572176, 345138, 10, 2209, 11267,
                            128, 4, 974, 14, 48937457, 11316796, 40094537, 452104, 12606, 104, 35};
  std::mt19937_64 rng(1024);
  std::uniform_real_distribution<double> Uu(0.0, 1.0);
  std::vector<int> pool(C);
  for (int64_t i = 0; i < C; ++i) pool[i] = (int)i;
  for (int64_t i = C - 1; i > 0; --i) std::swap(pool[i], pool[rng() % (i + 1)]);
  int64_t next_slot = 0, uniq = 0;
  std::vector<unsigned long long> base, none;
  for (int f = 0; f < F; ++f) {
    std::unordered_map<int64_t, int> slot_of;
    const double lo = pow(1.0 / (double)sizes[f], s);
    for (int b = 0; b < B; ++b) {
      const double x = Uu(rng) * (1.0 - lo) + lo;
      int64_t id = (int64_t)floor(1.0 / pow(x, 1.0 / s)) - 1;
      id = std::max<int64_t>(0, std::min<int64_t>(id, sizes[f] - 1));
      auto it = slot_of.find(id);
      int sl;
      if (it == slot_of.end()) { sl = pool[next_slot++]; slot_of[id] = sl; ++uniq; } else sl = it->second;
      base.push_back(((unsigned long long)sl << 32) | (unsigned)(b * F + f));
      none.push_back(((unsigned long long)pool[(C - 1) - (int64_t)f * B - b] << 32) | (unsigned)(b * F + f));
    }
  }
  const int64_t T = base.size();
  printf("batch: %lld lookups, %lld distinct rows (%.1f MB)\n", (long long)T, (long long)uniq, uniq * 512 / 1e6);
  auto split = [&](const std::vector<unsigned long long>& in, int nl, std::vector<unsigned long long>& out, std::vector<int>& st,
                   int shift) {
    std::vector<std::vector<unsigned long long>> ls(nl);
    for (auto p : in) ls[((p >> 32) >> shift) % nl].push_back(p);
    out.clear(); st.assign(nl + 1, 0);
    for (int l = 0; l < nl; ++l) { st[l] = (int)out.size(); out.insert(out.end(), ls[l].begin(), ls[l].end()); }
    st[nl] = (int)out.size();
  };
  f32x4 *W, *out;
  CK(hipMalloc(&W, C * 512));
  CK(hipMalloc(&out, T * 512));
  CK(hipMemset(W, 0, C * 512));
  unsigned long long* d_pairs;
  int* d_st;
  CK(hipMalloc(&d_pairs, T * 8));
  CK(hipMalloc(&d_st, 65 * 4));
  auto upload = [&](const std::vector<unsigned long long>& p, const std::vector<int>& st) {
    CK(hipMemcpy(d_pairs, p.data(), T * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_st, st.data(), st.size() * 4, hipMemcpyHostToDevice));
  };
#define RUN(NAME, U, NT, GRID, NL) RUNA(NAME, U, NT, GRID, NL, 1)
#define RUNA(NAME, U, NT, GRID, NL, AFF)                                                                       \
  {                                                                                                            \
    float t = time_it([&] { hipLaunchKernelGGL((k_fwd<U, NT>), dim3(GRID), dim3(256), 0, 0, W, d_pairs, d_st, NL, out, AFF); }); \
    printf("%-44s U=%2d nt=%d grid=%5d : %6.1f us\n", NAME, U, NT, GRID, t);                                    \
  }
  std::vector<unsigned long long> p;
  std::vector<int> st;
  split(base, 1, p, st, 0);
  upload(p, st);
  for (int g : {1024, 2048, 4096, 6656}) { RUN("base (one list)", 16, 1, g, 1); RUN("base (one list)", 8, 1, g, 1); }
  split(none, 1, p, st, 0);
  upload(p, st);
  for (int g : {2048, 6656}) RUN("no reuse (425,984 distinct rows)", 16, 1, g, 1);
  for (int shift : {0, 3}) {
    split(base, 8, p, st, shift);
    upload(p, st);
    printf("lists by (row >> %d) & 7: sizes", shift);
    for (int l = 0; l < 8; ++l) printf(" %d", st[l + 1] - st[l]);
    printf("\n");
    for (int g : {1024, 2048, 4096, 6656}) {
      RUN("xcd (8 lists, workgroup w -> list w % 8)", 16, 1, g, 8);
      RUN("xcd (8 lists)", 8, 1, g, 8);
      RUNA("control (8 lists, every XCD on every list)", 16, 1, g / 64 * 64, 8, 0);
    }
  }
  return 0;
}
