// Memory-system probe for the bag kernels (round 3): what do streaming reads, fills, row gathers and fp32 atomic
// row updates cost on this box, alone and mixed?  Standalone: hipcc --offload-arch=gfx950 -O3 probe_mem.hip -o probe_mem
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int U, int NT>
__global__ __launch_bounds__(256) void k_read(const f32x4* __restrict__ src, int64_t n4, f32x4* sink) {
  f32x4 acc = {0, 0, 0, 0};
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 256;
      v[u] = f32x4{0, 0, 0, 0};
      if (j < n4) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x == 12345.678f) sink[0] = acc;
}

// contiguous per-block split (each block streams one contiguous range)
template <int U, int NT>
__global__ __launch_bounds__(256) void k_read_contig(const f32x4* __restrict__ src, int64_t n4, f32x4* sink) {
  f32x4 acc = {0, 0, 0, 0};
  const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const int64_t lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256 * U) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 256;
      v[u] = f32x4{0, 0, 0, 0};
      if (j < hi) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x == 12345.678f) sink[0] = acc;
}

template <int U, int NT>
__global__ __launch_bounds__(256) void k_fill(f32x4* __restrict__ dst, int64_t n4, float val) {
  const f32x4 v = {val, val, val, val};
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 256;
      if (j < n4) { if (NT) __builtin_nontemporal_store(v, dst + j); else dst[j] = v; }
    }
  }
}

template <int U, int NT>
__global__ __launch_bounds__(256) void k_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t i = (int64_t)blockIdx.x * 256 * U + threadIdx.x; i < n4; i += stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 256;
      v[u] = f32x4{0, 0, 0, 0};
      if (j < n4) v[u] = src[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * 256;
      if (j < n4) { if (NT) __builtin_nontemporal_store(v[u], dst + j); else dst[j] = v[u]; }
    }
  }
}

// row gather: 32 lanes per 512-B row, R rows in flight per lane group, sum
template <int R, int NT>
__global__ __launch_bounds__(256) void k_gather(const f32x4* __restrict__ tab, const int* __restrict__ rows, int64_t n,
                                                f32x4* sink) {
  const int gl = threadIdx.x & 31;
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
  const int64_t ngrp = ((int64_t)gridDim.x * 256) >> 5;
  const int64_t per = ((n + ngrp - 1) / ngrp + R - 1) / R * R;
  const int64_t lo = grp * per, hi = lo + per < n ? lo + per : n;
  f32x4 acc = {0, 0, 0, 0};
  for (int64_t i = lo; i < hi; i += R) {
    int r[R];
#pragma unroll
    for (int t = 0; t < R; ++t) r[t] = i + t < hi ? rows[i + t] : -1;
    f32x4 v[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
      v[t] = f32x4{0, 0, 0, 0};
      if (r[t] >= 0) v[t] = NT ? __builtin_nontemporal_load(tab + (int64_t)r[t] * 32 + gl) : tab[(int64_t)r[t] * 32 + gl];
    }
#pragma unroll
    for (int t = 0; t < R; ++t) acc += v[t];
  }
  if (acc.x == 12345.678f) sink[0] = acc;
}

// MODE 0: no-return fp32 atomics, default; 1: sc1; 2: nt; 3: sc0 sc1 (asm); 4: plain load-add-store of the row
// (f32x4); 5: plain store; 6: atomics untransposed (lane holds 4 consecutive floats: 4 atomics 16 B apart)
template <int MODE>
__device__ __forceinline__ void row_update(float* row, f32x4 v, int gl) {
  if (MODE == 4) {
    f32x4* p = (f32x4*)row + gl;
    *p = *p + v;
  } else if (MODE == 5) {
    ((f32x4*)row)[gl] = v;
  } else if (MODE == 6) {
    float* d = row + gl * 4;
    __hip_atomic_fetch_add(d + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(d + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(d + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(d + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    float* d = row + gl;      // instruction c covers floats [32c, 32c + 32): one 128-B line per lane group
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float* a = d + 32 * c;
      if (MODE == 0) __hip_atomic_fetch_add(a, x[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 1) asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(a), "v"(x[c]) : "memory");
      else if (MODE == 2) asm volatile("global_atomic_add_f32 %0, %1, off nt" ::"v"(a), "v"(x[c]) : "memory");
      else if (MODE == 3) asm volatile("global_atomic_add_f32 %0, %1, off sc1 nt" ::"v"(a), "v"(x[c]) : "memory");
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_update(float* tab, const int* __restrict__ rows, int64_t n) {
  const int gl = threadIdx.x & 31;
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
  const int64_t ngrp = ((int64_t)gridDim.x * 256) >> 5;
  const f32x4 v = {1e-6f, 2e-6f, 3e-6f, 4e-6f};
  for (int64_t i = grp; i < n; i += ngrp) row_update<MODE>(tab + (int64_t)rows[i] * 128, v, gl);
}

// mixed: every lane group gathers `per_flush` rows (R in flight) then issues one row update, like the streaming backward
template <int MODE, int R>
__global__ __launch_bounds__(256) void k_mixed(const f32x4* __restrict__ src, const int* __restrict__ srows, int64_t n,
                                               float* tab, const int* __restrict__ urows, int per_flush) {
  const int gl = threadIdx.x & 31;
  const int64_t grp = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
  const int64_t ngrp = ((int64_t)gridDim.x * 256) >> 5;
  const int64_t per = ((n + ngrp - 1) / ngrp + R - 1) / R * R;
  const int64_t lo = grp * per, hi = lo + per < n ? lo + per : n;
  f32x4 acc = {0, 0, 0, 0};
  int since = 0;
  for (int64_t i = lo; i < hi; i += R) {
    int r[R];
#pragma unroll
    for (int t = 0; t < R; ++t) r[t] = i + t < hi ? srows[i + t] : -1;
    f32x4 v[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
      v[t] = f32x4{0, 0, 0, 0};
      if (r[t] >= 0) v[t] = __builtin_nontemporal_load(src + (int64_t)r[t] * 32 + gl);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
    for (int t = 0; t < R; ++t) {
      acc += v[t];
      if (++since >= per_flush) {
        since = 0;
        if (MODE >= 0) row_update<(MODE < 0 ? 0 : MODE)>(tab + (int64_t)urows[(i + t) / per_flush] * 128, acc, gl);
        acc = f32x4{0, 0, 0, 0};
      }
    }
  }
  if (acc.x == 12345.678f) ((f32x4*)tab)[0] = acc;
}

// split roles: blocks with (blockIdx % split) == 0 only issue the row updates, the others only gather
template <int MODE, int R>
__global__ __launch_bounds__(256) void k_split(const f32x4* __restrict__ src, const int* __restrict__ srows, int64_t n,
                                               float* tab, const int* __restrict__ urows, int64_t nu, int split,
                                               f32x4* sink) {
  const int gl = threadIdx.x & 31;
  const bool upd = (blockIdx.x % split) == 0;
  const int nub = (gridDim.x + split - 1) / split, ngb = gridDim.x - nub;
  if (upd) {
    const int64_t grp = ((int64_t)(blockIdx.x / split) * 256 + threadIdx.x) >> 5;
    const int64_t ngrp = ((int64_t)nub * 256) >> 5;
    const f32x4 v = {1e-6f, 2e-6f, 3e-6f, 4e-6f};
    for (int64_t i = grp; i < nu; i += ngrp) row_update<MODE>(tab + (int64_t)urows[i] * 128, v, gl);
    return;
  }
  const int b = blockIdx.x - blockIdx.x / split - 1;
  const int64_t grp = ((int64_t)b * 256 + threadIdx.x) >> 5;
  const int64_t ngrp = ((int64_t)ngb * 256) >> 5;
  const int64_t per = ((n + ngrp - 1) / ngrp + R - 1) / R * R;
  const int64_t lo = grp * per, hi = lo + per < n ? lo + per : n;
  f32x4 acc = {0, 0, 0, 0};
  for (int64_t i = lo; i < hi; i += R) {
    int r[R];
#pragma unroll
    for (int t = 0; t < R; ++t) r[t] = i + t < hi ? srows[i + t] : -1;
    f32x4 v[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
      v[t] = f32x4{0, 0, 0, 0};
      if (r[t] >= 0) v[t] = __builtin_nontemporal_load(src + (int64_t)r[t] * 32 + gl);
    }
#pragma unroll
    for (int t = 0; t < R; ++t) acc += v[t];
  }
  if (acc.x == 12345.678f) sink[0] = acc;
}

static float time_it(const std::function<void()>& fn, int reps = 20) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) fn();
  CK(hipDeviceSynchronize());
  float best = 1e9, tot = 0;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0, 0));
    fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
    tot += ms;
  }
  CK(hipGetLastError());
  (void)tot;
  return best * 1e3f;      // us (best of reps, single launch incl. ~2 us of event overhead)
}

int main() {
  const int64_t ROWS = 425984;                 // B*F rows of 512 B = 218 MB
  const int64_t n4 = ROWS * 32;
  const int64_t TROWS = 1779442;               // cache table: 911 MB
  f32x4 *a, *b, *big, *sink;
  float* tab;
  CK(hipMalloc(&a, n4 * 16));
  CK(hipMalloc(&b, n4 * 16));
  CK(hipMalloc(&big, (size_t)1 << 30));
  CK(hipMalloc(&sink, 4096));
  CK(hipMalloc(&tab, TROWS * 512));
  CK(hipMemset(a, 0, n4 * 16));
  CK(hipMemset(b, 0, n4 * 16));
  CK(hipMemset(big, 0, (size_t)1 << 30));
  CK(hipMemset(tab, 0, TROWS * 512));
  const double MB = ROWS * 512 / 1e6;
  printf("== streaming read of %.0f MB (best of 20, us; TB/s)\n", MB);
#define RD(U, NT, G) { float t = time_it([&] { hipLaunchKernelGGL((k_read<U, NT>), dim3(G), dim3(256), 0, 0, a, n4, sink); }); printf("read  U=%2d nt=%d grid=%5d : %6.1f us  %.2f TB/s\n", U, NT, G, t, MB / t); }
  for (int g : {512, 1024, 2048, 4096}) { RD(4, 0, g); RD(8, 0, g); RD(16, 0, g); RD(8, 1, g); RD(16, 1, g); }
#define RC(U, NT, G) { float t = time_it([&] { hipLaunchKernelGGL((k_read_contig<U, NT>), dim3(G), dim3(256), 0, 0, a, n4, sink); }); printf("readc U=%2d nt=%d grid=%5d : %6.1f us  %.2f TB/s\n", U, NT, G, t, MB / t); }
  for (int g : {512, 1024, 2048}) { RC(8, 0, g); RC(16, 0, g); RC(16, 1, g); }
  {
    const int64_t bn4 = ((int64_t)1 << 30) / 16;
    float t = time_it([&] { hipLaunchKernelGGL((k_read<8, 0>), dim3(2048), dim3(256), 0, 0, big, bn4, sink); });
    printf("read 1 GiB U=8 grid=2048: %6.1f us  %.2f TB/s\n", t, 1073.74 / t);
    t = time_it([&] { hipLaunchKernelGGL((k_read<16, 1>), dim3(2048), dim3(256), 0, 0, big, bn4, sink); });
    printf("read 1 GiB U=16 nt grid=2048: %6.1f us  %.2f TB/s\n", t, 1073.74 / t);
  }
  printf("== fill of %.0f MB\n", MB);
#define FL(U, NT, G) { float t = time_it([&] { hipLaunchKernelGGL((k_fill<U, NT>), dim3(G), dim3(256), 0, 0, b, n4, 1.f); }); printf("fill  U=%2d nt=%d grid=%5d : %6.1f us  %.2f TB/s\n", U, NT, G, t, MB / t); }
  for (int g : {512, 1024, 2048, 4096}) { FL(4, 0, g); FL(4, 1, g); FL(8, 1, g); }
  printf("== copy %.0f -> %.0f MB\n", MB, MB);
#define CP(U, NT, G) { float t = time_it([&] { hipLaunchKernelGGL((k_copy<U, NT>), dim3(G), dim3(256), 0, 0, a, b, n4); }); printf("copy  U=%2d nt=%d grid=%5d : %6.1f us  %.2f TB/s (r+w)\n", U, NT, G, t, 2 * MB / t); }
  for (int g : {1024, 2048, 4096}) { CP(4, 0, g); CP(8, 1, g); CP(16, 1, g); }

  // ---- gathers
  std::vector<int> perm(ROWS), trow(ROWS);
  uint64_t x = 88172645463325252ull;
  auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
  for (int64_t i = 0; i < ROWS; ++i) perm[i] = (int)i;
  for (int64_t i = ROWS - 1; i > 0; --i) std::swap(perm[i], perm[rnd() % (i + 1)]);
  for (int64_t i = 0; i < ROWS; ++i) trow[i] = (int)(rnd() % TROWS);
  int *d_perm, *d_trow, *d_ident;
  CK(hipMalloc(&d_perm, ROWS * 4));
  CK(hipMalloc(&d_trow, ROWS * 4));
  CK(hipMalloc(&d_ident, ROWS * 4));
  CK(hipMemcpy(d_perm, perm.data(), ROWS * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_trow, trow.data(), ROWS * 4, hipMemcpyHostToDevice));
  { std::vector<int> id(ROWS); for (int64_t i = 0; i < ROWS; ++i) id[i] = (int)i; CK(hipMemcpy(d_ident, id.data(), ROWS * 4, hipMemcpyHostToDevice)); }
  printf("== gather of %lld rows x 512 B (32 lanes per row)\n", (long long)ROWS);
#define GA(R, NT, G, IDX, SRC, NAME) { float t = time_it([&] { hipLaunchKernelGGL((k_gather<R, NT>), dim3(G), dim3(256), 0, 0, (const f32x4*)SRC, IDX, ROWS, sink); }); printf("gather %-22s R=%2d nt=%d grid=%5d : %6.1f us  %.2f TB/s\n", NAME, R, NT, G, t, MB / t); }
  for (int g : {512, 1024, 2048}) {
    GA(16, 1, g, d_ident, a, "218MB in order");
    GA(16, 1, g, d_perm, a, "218MB permuted");
    GA(8, 1, g, d_perm, a, "218MB permuted");
    GA(16, 0, g, d_perm, a, "218MB permuted");
    GA(16, 0, g, d_trow, tab, "911MB random rows");
    GA(8, 0, g, d_trow, tab, "911MB random rows");
  }
  // ---- row updates: 49152 distinct random rows of the 911 MB table / of a 16 MB / 2 MB region
  const int64_t NU = 49152;
  std::vector<int> u_big(NU), u_16(NU), u_2(NU);
  {
    std::vector<int> all(TROWS);
    for (int64_t i = 0; i < TROWS; ++i) all[i] = (int)i;
    for (int64_t i = 0; i < NU; ++i) { std::swap(all[i], all[i + rnd() % (TROWS - i)]); u_big[i] = all[i]; }
    for (int64_t i = 0; i < NU; ++i) { u_16[i] = (int)(rnd() % 32768); u_2[i] = (int)(rnd() % 4096); }
  }
  int *d_ub, *d_u16, *d_u2;
  CK(hipMalloc(&d_ub, NU * 4)); CK(hipMalloc(&d_u16, NU * 4)); CK(hipMalloc(&d_u2, NU * 4));
  CK(hipMemcpy(d_ub, u_big.data(), NU * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_u16, u_16.data(), NU * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_u2, u_2.data(), NU * 4, hipMemcpyHostToDevice));
  printf("== %lld row updates of 512 B (25 MB): 0 atomic, 1 atomic sc1, 2 atomic nt, 3 atomic sc1 nt, 4 load-add-store, 5 store, 6 atomic untransposed\n", (long long)NU);
#define UP(M, G, IDX, NAME) { float t = time_it([&] { hipLaunchKernelGGL((k_update<M>), dim3(G), dim3(256), 0, 0, tab, IDX, NU); }); printf("update mode=%d %-18s grid=%5d : %6.1f us  %.2f TB/s of rows\n", M, NAME, G, t, NU * 512 / 1e6 / t); }
  for (int g : {512, 2048}) {
    UP(0, g, d_ub, "911MB distinct"); UP(1, g, d_ub, "911MB distinct"); UP(2, g, d_ub, "911MB distinct"); UP(3, g, d_ub, "911MB distinct");
    UP(4, g, d_ub, "911MB distinct"); UP(5, g, d_ub, "911MB distinct"); UP(6, g, d_ub, "911MB distinct");
    UP(0, g, d_u16, "16MB region"); UP(4, g, d_u16, "16MB region");
    UP(0, g, d_u2, "2MB region"); UP(4, g, d_u2, "2MB region");
  }
  // ---- mixed: gather 425,984 gradient rows (permuted) + one row update per 8.67 rows
  printf("== mixed (streaming-backward shape): gather %lld rows + 1 update per 9 rows; MODE -1 = no update\n", (long long)ROWS);
#define MX(M, R, G) { float t = time_it([&] { hipLaunchKernelGGL((k_mixed<M, R>), dim3(G), dim3(256), 0, 0, (const f32x4*)a, d_perm, ROWS, tab, d_ub, 9); }); printf("mixed mode=%2d R=%2d grid=%5d : %6.1f us\n", M, R, G, t); }
  for (int g : {512, 1024}) { MX(-1, 16, g); MX(0, 16, g); MX(1, 16, g); MX(2, 16, g); MX(4, 16, g); MX(5, 16, g); MX(0, 8, g); MX(-1, 8, g); }
  printf("== split roles: 1 of `split` blocks only updates (47 k rows), the others only gather\n");
#define SP(M, R, G, S) { float t = time_it([&] { hipLaunchKernelGGL((k_split<M, R>), dim3(G), dim3(256), 0, 0, (const f32x4*)a, d_perm, ROWS, tab, d_ub, (int64_t)47331, S, sink); }); printf("split mode=%d R=%2d grid=%5d split=%2d : %6.1f us\n", M, R, G, S, t); }
  for (int g : {512, 1024}) for (int s : {4, 8, 16}) { SP(0, 16, g, s); SP(4, 16, g, s); }
  return 0;
}
