// Prototype of the owner-exclusive streaming backward (round 3) on a host-built Criteo-1TB-shaped batch:
//   cur : the round-2 kernel's logic (every run end = transposed fp32 atomics)
//   new : runs that one lane group owns entirely start with a "W key" (gather the weight row itself, scale 1) and end
//         with a plain 512-B store; only long / shared runs keep the atomics.
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics probe_bwd.hip -o probe_bwd
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
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void flush_atomic(float* row_base, f32x4 v, int gl) {
  const int G = 32, q = 8;
  const int kb = gl / q, m = gl - kb * q;
  const bool b1 = kb & 2, b0 = kb & 1;
  float r0 = v.x, r1 = v.y, r2 = v.z, r3 = v.w;
  float x = b1 ? r0 : r2, y = __shfl_xor(x, 2 * q);
  if (b1) r0 = y; else r2 = y;
  x = b1 ? r1 : r3; y = __shfl_xor(x, 2 * q);
  if (b1) r1 = y; else r3 = y;
  x = b0 ? r0 : r1; y = __shfl_xor(x, q);
  if (b0) r0 = y; else r1 = y;
  x = b0 ? r2 : r3; y = __shfl_xor(x, q);
  if (b0) r2 = y; else r3 = y;
  float* d = row_base + 4 * m + kb;
  __hip_atomic_fetch_add(d, r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(d + G, r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(d + 2 * G, r2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(d + 3 * G, r3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// MODE 0: atomics for every run (no W keys expected); 1: W-headed runs end with a plain store; 2: no update traffic
// NTG: gradient rows loaded non-temporally; W rows always with the default policy (MODE 1 selects per key)
template <int R, int MODE, int NTG>
__global__ __launch_bounds__(256) void k_bwd(const f32x4* __restrict__ GO, float* __restrict__ W, const u64* __restrict__ keys,
                                             const int* __restrict__ starts, int64_t total, float alpha) {
  __shared__ u64 lk[8 * 128];
  const int tid = threadIdx.x, grp = tid >> 5, gl = tid & 31;
  const int g = blockIdx.x * 8 + grp;
  int64_t s0, s1;
  if (starts) { s0 = starts[g]; s1 = starts[g + 1]; }
  else {
    const int64_t all = (int64_t)gridDim.x * 8;
    const int64_t share = ((total + all - 1) / all + R - 1) / R * R;
    s0 = g * share; s1 = s0 + share < total ? s0 + share : total;
  }
  if (s0 >= s1) return;
  u64* mylk = lk + grp * 128;
  f32x4 acc = {0, 0, 0, 0};
  uint32_t cur = 0xffffffffu;
  bool cur_excl = false;
  const f32x4* __restrict__ WV = (const f32x4*)W;
  for (int64_t c0 = s0; c0 < s1; c0 += 128) {
    for (int k = gl; k < 128; k += 32) mylk[k] = c0 + k < s1 ? keys[c0 + k] : ~0ull;
    const int64_t c1 = s1 < c0 + 128 ? s1 : c0 + 128;
    for (int64_t q = c0; q < c1; q += R) {
      f32x4 v[R];
      uint32_t rw[R];
      uint32_t isw = 0;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const u64 k = mylk[(int)(q - c0) + t];
        const bool on = k != ~0ull;
        rw[t] = on ? (uint32_t)(k >> 32) : 0xffffffffu;
        const uint32_t low = (uint32_t)k;
        const bool w = (MODE == 1 || MODE == 3) && (low >> 31);
        if (w) isw |= 1u << t;
        const int64_t src = (int64_t)(low & 0x7fffffffu);
        const f32x4* p = (w ? WV : GO) + src * 32 + gl;
        v[t] = f32x4{0, 0, 0, 0};
        if (on) v[t] = (NTG && !w) ? __builtin_nontemporal_load(p) : *p;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
      for (int t = 0; t < R; ++t) {
        if (rw[t] != 0xffffffffu) {
          if (rw[t] != cur) {
            if (cur != 0xffffffffu) {
              if (MODE >= 2) { if (acc.x == 12345.678f) ((f32x4*)(W + (int64_t)cur * 128))[gl] = acc; }
              else if (cur_excl) ((f32x4*)(W + (int64_t)cur * 128))[gl] = acc;
              else flush_atomic(W + (int64_t)cur * 128, acc, gl);
            }
            acc = f32x4{0, 0, 0, 0};
            cur = rw[t];
            cur_excl = (isw >> t) & 1;
          }
          const float sc = ((isw >> t) & 1) ? 1.f : alpha;
          acc = acc + v[t] * sc;
        }
      }
    }
  }
  if (cur != 0xffffffffu) {
    if (MODE >= 2) { if (acc.x == 12345.678f) ((f32x4*)(W + (int64_t)cur * 128))[gl] = acc; }
    else if (cur_excl) ((f32x4*)(W + (int64_t)cur * 128))[gl] = acc;
    else flush_atomic(W + (int64_t)cur * 128, acc, gl);
  }
}

// flagged keys: bit 31 of the low word = this key heads a run that the lane group owns entirely (never crosses a
// 16-position block): the weight row is loaded beside the gradient row, the run ends with a plain store.
template <int R, int NTG, int FL>
__global__ __launch_bounds__(256) void k_bwd_flag(const f32x4* __restrict__ GO, float* __restrict__ W, const u64* __restrict__ keys,
                                                  int64_t total, float alpha) {
  __shared__ u64 lk[8 * 128];
  const int tid = threadIdx.x, grp = tid >> 5, gl = tid & 31;
  const int g = blockIdx.x * 8 + grp;
  const int64_t all = (int64_t)gridDim.x * 8;
  const int64_t share = ((total + all - 1) / all + 15) / 16 * 16;
  const int64_t s0 = g * share, s1 = s0 + share < total ? s0 + share : total;
  if (s0 >= s1) return;
  u64* mylk = lk + grp * 128;
  f32x4 acc = {0, 0, 0, 0};
  uint32_t cur = 0xffffffffu;
  bool cur_excl = false;
  const f32x4* __restrict__ WV = (const f32x4*)W;
  for (int64_t c0 = s0; c0 < s1; c0 += 128) {
    for (int k = gl; k < 128; k += 32) mylk[k] = c0 + k < s1 ? keys[c0 + k] : ~0ull;
    const int64_t c1 = s1 < c0 + 128 ? s1 : c0 + 128;
    for (int64_t q = c0; q < c1; q += R) {
      f32x4 v[R], w2[R];
      uint32_t rw[R];
      uint32_t isw = 0;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const u64 k = mylk[(int)(q - c0) + t];
        const bool on = k != ~0ull;
        rw[t] = on ? (uint32_t)(k >> 32) : 0xffffffffu;
        const uint32_t low = (uint32_t)k;
        const bool w = on && (low >> 31);
        if (w) isw |= 1u << t;
        const int64_t src = (int64_t)(low & 0x7fffffffu);
        const f32x4* p = GO + src * 32 + gl;
        v[t] = f32x4{0, 0, 0, 0};
        w2[t] = f32x4{0, 0, 0, 0};
        if (on) v[t] = NTG ? __builtin_nontemporal_load(p) : *p;
        if (w) w2[t] = WV[(int64_t)rw[t] * 32 + gl];
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
      for (int t = 0; t < R; ++t) {
        if (rw[t] != 0xffffffffu) {
          if (rw[t] != cur || ((isw >> t) & 1)) {
            if (cur != 0xffffffffu) {
              if (FL == 0) { if (acc.x == 12345.678f) ((f32x4*)(W + (int64_t)cur * 128))[gl] = acc; }
              else if (cur_excl) ((f32x4*)(W + (int64_t)cur * 128))[gl] = acc;
              else flush_atomic(W + (int64_t)cur * 128, acc, gl);
            }
            acc = w2[t];
            cur = rw[t];
            cur_excl = (isw >> t) & 1;
          }
          acc = acc + v[t] * alpha;
        }
      }
    }
  }
  if (cur != 0xffffffffu) {
    if (FL == 0) { if (acc.x == 12345.678f) ((f32x4*)(W + (int64_t)cur * 128))[gl] = acc; }
    else if (cur_excl) ((f32x4*)(W + (int64_t)cur * 128))[gl] = acc;
    else flush_atomic(W + (int64_t)cur * 128, acc, gl);
  }
}

// ---- two passes: (A) gather + fold; a flagged run (its row occurs nowhere else in the launch, the run lies inside one
// lane group's share) leaves its folded, scaled gradient in a scratch row (plain 512-byte store, scratch rows in key
// order: a nearly sequential write stream) instead of touching W; everything else keeps the atomics.  (B) every
// flagged run head: W[row] += scratch row, a plain read-modify-write with nothing else going on.
// SIDX: 0 = scratch row = key position (sparse over nnz rows), 1 = compact index from cidx[position]
template <int R, int NTG, int SIDX>
__global__ __launch_bounds__(256) void k_bwd_2pA(const f32x4* __restrict__ GO, float* __restrict__ W, const u64* __restrict__ keys,
                                                 const int* __restrict__ cidx, f32x4* __restrict__ scratch, int64_t total, float alpha) {
  __shared__ u64 lk[8 * 128];
  const int tid = threadIdx.x, grp = tid >> 5, gl = tid & 31;
  const int g = blockIdx.x * 8 + grp;
  const int64_t all = (int64_t)gridDim.x * 8;
  const int64_t share = ((total + all - 1) / all + 15) / 16 * 16;
  const int64_t s0 = g * share, s1 = s0 + share < total ? s0 + share : total;
  if (s0 >= s1) return;
  u64* mylk = lk + grp * 128;
  f32x4 acc = {0, 0, 0, 0};
  uint32_t cur = 0xffffffffu;
  int64_t cur_pos = -1;          // >= 0: the run in flight is a flagged one, its head sits at this position
  auto flush = [&]() {
    if (cur == 0xffffffffu) return;
    if (cur_pos >= 0) scratch[(SIDX ? (int64_t)cidx[cur_pos] : cur_pos) * 32 + gl] = acc;
    else flush_atomic(W + (int64_t)cur * 128, acc, gl);
  };
  for (int64_t c0 = s0; c0 < s1; c0 += 128) {
    for (int k = gl; k < 128; k += 32) mylk[k] = c0 + k < s1 ? keys[c0 + k] : ~0ull;
    const int64_t c1 = s1 < c0 + 128 ? s1 : c0 + 128;
    for (int64_t q = c0; q < c1; q += R) {
      f32x4 v[R];
      uint32_t rw[R];
      uint32_t isw = 0;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const u64 k = mylk[(int)(q - c0) + t];
        const bool on = k != ~0ull;
        rw[t] = on ? (uint32_t)(k >> 32) : 0xffffffffu;
        const uint32_t low = (uint32_t)k;
        if (on && (low >> 31)) isw |= 1u << t;
        const int64_t src = (int64_t)(low & 0x7fffffffu);
        const f32x4* p = GO + src * 32 + gl;
        v[t] = f32x4{0, 0, 0, 0};
        if (on) v[t] = NTG ? __builtin_nontemporal_load(p) : *p;
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
      for (int t = 0; t < R; ++t) {
        if (rw[t] != 0xffffffffu) {
          if (rw[t] != cur || ((isw >> t) & 1)) {
            flush();
            acc = f32x4{0, 0, 0, 0};
            cur = rw[t];
            cur_pos = ((isw >> t) & 1) ? q + t : -1;
          }
          acc = acc + v[t] * alpha;
        }
      }
    }
  }
  flush();
}

template <int RB, int SIDX>
__global__ __launch_bounds__(256) void k_bwd_2pB(float* __restrict__ W, const u64* __restrict__ keys, const int* __restrict__ cidx,
                                                 const f32x4* __restrict__ scratch, int64_t total) {
  const int tid = threadIdx.x, grp = tid >> 5, gl = tid & 31;
  const int64_t all = (int64_t)gridDim.x * 8;
  const int64_t g = (int64_t)blockIdx.x * 8 + grp;
  const int64_t share = ((total + all - 1) / all + 31) / 32 * 32;
  const int64_t s0 = g * share, s1 = s0 + share < total ? s0 + share : total;
  f32x4* WV = (f32x4*)W;
  for (int64_t c0 = s0; c0 < s1; c0 += 32) {
    const u64 k = c0 + gl < s1 ? keys[c0 + gl] : ~0ull;
    const bool fl = k != ~0ull && (((uint32_t)k) >> 31);
    uint32_t m = (uint32_t)__ballot(fl) >> 0;       // lanes 0..31 or 32..63 of the wave: take this group's half
    unsigned long long mb = __ballot(fl);
    m = (uint32_t)(mb >> ((tid & 32) ? 32 : 0));
    while (m) {
      f32x4 d[RB], w[RB];
      int64_t row[RB];
      int n = 0;
#pragma unroll
      for (int t = 0; t < RB; ++t) {
        row[t] = -1;
        if (m) {
          const int b = __ffs(m) - 1;
          m &= m - 1;
          const u64 kk = __shfl(k, (tid & 32) + b);
          row[t] = (int64_t)(kk >> 32);
          const int64_t si = SIDX ? (int64_t)cidx[c0 + b] : c0 + b;
          d[t] = scratch[si * 32 + gl];
          w[t] = WV[row[t] * 32 + gl];
          ++n;
        }
      }
#pragma unroll
      for (int t = 0; t < RB; ++t)
        if (row[t] >= 0) WV[row[t] * 32 + gl] = w[t] + d[t];
    }
  }
}

// decoupled flush: waves 0-2 of a workgroup only gather and fold (their vmcnt never holds a store or an atomic), wave 3
// only applies finished rows, handed over through 2-slot LDS rings (one per producer lane group)
template <int R, int NTG, int STORE>
__global__ __launch_bounds__(256) void k_bwd_split(const f32x4* __restrict__ GO, float* __restrict__ W, const u64* __restrict__ keys,
                                                   int64_t total, float alpha) {
  __shared__ u64 lk[6 * 128];
  __shared__ float ring[6][2][128];
  __shared__ uint32_t ring_row[6][2];
  __shared__ int ring_full[6][2];
  __shared__ int done_cnt;
  const int tid = threadIdx.x, grp = tid >> 5, gl = tid & 31;
  if (tid < 12) (&ring_full[0][0])[tid] = 0;
  if (tid == 0) done_cnt = 0;
  __syncthreads();
  if (grp >= 6) {
    // ---- writer wave: lane group c serves producers 3c .. 3c + 2
    const int c = grp - 6;
    int nxt[3] = {0, 0, 0};
    for (int guard = 0; guard < (1 << 24); ++guard) {
      bool any = false;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int p = 3 * c + j;
        volatile int* f = &ring_full[p][nxt[j]];
        if (*f) {
          const uint32_t row = *(volatile uint32_t*)&ring_row[p][nxt[j]];
          const volatile float* src = ring[p][nxt[j]];
          const float x0 = src[gl], x1 = src[gl + 32], x2 = src[gl + 64], x3 = src[gl + 96];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the slot is in registers
          if (gl == 0) *f = 0;
          float* d = W + (int64_t)row * 128 + gl;
          if (STORE) { d[0] = x0; d[32] = x1; d[64] = x2; d[96] = x3; }
          else {
            __hip_atomic_fetch_add(d, x0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(d + 32, x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(d + 64, x2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(d + 96, x3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          nxt[j] ^= 1;
          any = true;
        }
      }
      if (!any) {
        if (*(volatile int*)&done_cnt == 6) {
          bool left = false;
#pragma unroll
          for (int j = 0; j < 3; ++j) left |= (*(volatile int*)&ring_full[3 * c + j][nxt[j]]) != 0;
          if (!left) break;
        } else {
          __builtin_amdgcn_s_sleep(4);
        }
      }
    }
    return;
  }
  // ---- gather waves
  const int g = blockIdx.x * 6 + grp;
  const int64_t all = (int64_t)gridDim.x * 6;
  const int64_t share = ((total + all - 1) / all + 15) / 16 * 16;
  const int64_t s0 = g * share, s1 = s0 + share < total ? s0 + share : total;
  u64* mylk = lk + grp * 128;
  f32x4 acc = {0, 0, 0, 0};
  uint32_t cur = 0xffffffffu;
  int slot = 0;
  auto push = [&](uint32_t row, f32x4 a) {
    volatile int* f = &ring_full[grp][slot];
    for (int spin = 0; *f && spin < (1 << 22); ++spin) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    ((f32x4*)ring[grp][slot])[gl] = a;
    if (gl == 0) *(volatile uint32_t*)&ring_row[grp][slot] = row;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (gl == 0) *f = 1;
    slot ^= 1;
  };
  for (int64_t c0 = s0; c0 < s1; c0 += 128) {
    for (int k = gl; k < 128; k += 32) mylk[k] = c0 + k < s1 ? keys[c0 + k] : ~0ull;
    const int64_t c1 = s1 < c0 + 128 ? s1 : c0 + 128;
    for (int64_t q = c0; q < c1; q += R) {
      f32x4 v[R];
      uint32_t rw[R];
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const u64 k = mylk[(int)(q - c0) + t];
        const bool on = k != ~0ull;
        rw[t] = on ? (uint32_t)(k >> 32) : 0xffffffffu;
        const int64_t src = (int64_t)((uint32_t)k & 0x7fffffffu);
        v[t] = f32x4{0, 0, 0, 0};
        if (on) v[t] = NTG ? __builtin_nontemporal_load(GO + src * 32 + gl) : GO[src * 32 + gl];
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
      for (int t = 0; t < R; ++t) {
        if (rw[t] != 0xffffffffu) {
          if (rw[t] != cur) {
            if (cur != 0xffffffffu) push(cur, acc);
            acc = f32x4{0, 0, 0, 0};
            cur = rw[t];
          }
          acc = acc + v[t] * alpha;
        }
      }
    }
  }
  if (cur != 0xffffffffu) push(cur, acc);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (gl == 0) atomicAdd(&done_cnt, 1);
}

// ---- vmcnt ordering test: an older load, younger slow atomics (all lanes of all waves on ONE address), then
// s_waitcnt vmcnt(4): with in-order retirement the load has landed; if atomics could retire ahead of it the register
// would still hold the sentinel
__global__ __launch_bounds__(256) void k_vmcnt_order(const float* __restrict__ src, float* hot, float* out, int64_t n, int reps) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int bad = 0;
  for (int r = 0; r < reps; ++r) {
    const int64_t j = (i * 977 + (int64_t)r * 1315423911ll) % n;
    float v = -12345.f;
    const float* p = src + j * 32;              // a cold 128-byte line per lane
    asm volatile("global_load_dword %0, %1, off nt" : "+v"(v) : "v"(p) : "memory");
    float one = 1.f;
    float* h = hot + (threadIdx.x & 3) * 64;
    asm volatile("global_atomic_add_f32 %0, %1, off\n\tglobal_atomic_add_f32 %0, %1, off offset:4\n\t"
                 "global_atomic_add_f32 %0, %1, off offset:8\n\tglobal_atomic_add_f32 %0, %1, off offset:12"
                 :: "v"(h), "v"(one) : "memory");
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(v) :: "memory");
    if (v != (float)(j & 1023)) ++bad;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (bad) atomicAdd(out, (float)bad);
}

// ---- pipelined streaming backward: loads are inline asm (the compiler keeps no scoreboard for them), two buffers of
// R rows; the loads of chunk i+1 are issued BEFORE chunk i is folded, so the atomics of chunk i are YOUNGER than them,
// and the wait for chunk i+1 leaves those atomics outstanding: vmcnt(4 * flush groups issued since).  Relies on vmcnt
// retiring in issue order (k_vmcnt_order).
template <int NT>
__device__ __forceinline__ void ld16(f32x4& v, const f32x4* p) {
  if (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
}
#define WAIT8(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]) :: "memory")
__device__ __forceinline__ void wait_buf(f32x4 (&b)[8], int younger) {     // younger = vmem instructions issued after b's loads
  switch (younger) {
    case 0: WAIT8(0); break;   case 4: WAIT8(4); break;   case 8: WAIT8(8); break;   case 12: WAIT8(12); break;
    case 16: WAIT8(16); break; case 20: WAIT8(20); break; case 24: WAIT8(24); break; case 28: WAIT8(28); break;
    case 32: WAIT8(32); break; case 36: WAIT8(36); break; case 40: WAIT8(40); break; case 44: WAIT8(44); break;
    case 48: WAIT8(48); break; case 52: WAIT8(52); break; case 56: WAIT8(56); break; case 60: WAIT8(60); break;
    default: WAIT8(0); break;
  }
}

template <int NTG, int ATOM>
__global__ __launch_bounds__(256) void k_bwd_pipe(const f32x4* __restrict__ GO, float* __restrict__ W, const u64* __restrict__ keys,
                                                  int64_t total, float alpha) {
  constexpr int R = 8;
  __shared__ u64 lk[8 * 128];
  const int tid = threadIdx.x, grp = tid >> 5, gl = tid & 31;
  const int g = blockIdx.x * 8 + grp;
  const int64_t all = (int64_t)gridDim.x * 8;
  const int64_t share = ((total + all - 1) / all + 15) / 16 * 16;
  const int64_t s0 = g * share, s1 = s0 + share < total ? s0 + share : total;
  // both lane groups of a wave must run the same number of chunks (the vmcnt bookkeeping is per wave)
  const int64_t w0 = (int64_t)(blockIdx.x * 8 + (grp & ~1)) * share;
  int64_t nchunks = 0;
  if (w0 < total) nchunks = (share + R - 1) / R;
  u64* mylk = lk + grp * 128;
  f32x4 acc = {0, 0, 0, 0};
  uint32_t cur = 0xffffffffu;
  f32x4 bufA[R], bufB[R];
  uint32_t rwA[R], rwB[R];
  auto issue = [&](f32x4 (&b)[R], uint32_t (&rw)[R], int64_t q) {         // keys of chunk q are in LDS
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const u64 k = mylk[(int)((q - s0) & 127) + t];
      const bool on = (q + t < s1) && k != ~0ull;
      rw[t] = on ? (uint32_t)(k >> 32) : 0xffffffffu;
      const int64_t src = on ? (int64_t)((uint32_t)k & 0x7fffffffu) : 0;
      ld16<NTG>(b[t], GO + src * 32 + gl);                                 // always issued (count must be exact)
    }
  };
  int yA = 0, yB = 0;                  // vmem instructions issued after buffer A's / B's loads (wave-uniform)
  auto fold = [&](f32x4 (&b)[R], uint32_t (&rw)[R]) {
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const bool valid = rw[t] != 0xffffffffu;
      const bool fl = valid && rw[t] != cur && cur != 0xffffffffu;
      if (__builtin_amdgcn_ballot_w64(fl) != 0) { yA += 4; yB += 4; }    // this flush block issues 4 atomics / stores
      if (valid) {
        if (rw[t] != cur) {
          if (cur != 0xffffffffu) {
            if (ATOM) flush_atomic(W + (int64_t)cur * 128, acc, gl);
            else { float* d = W + (int64_t)cur * 128 + gl; d[0] = acc.x; d[32] = acc.y; d[64] = acc.z; d[96] = acc.w; }
          }
          acc = f32x4{0, 0, 0, 0};
          cur = rw[t];
        }
        acc = acc + b[t] * alpha;
      }
    }
  };
  auto refill = [&](int64_t c0) {      // 128 keys of the share -> LDS (compiler-managed loads: waited for with vmcnt(0))
    for (int k = gl; k < 128; k += 32) mylk[k] = c0 + k < s1 ? keys[c0 + k] : ~0ull;
  };
  if (nchunks == 0) return;
  refill(s0);
  issue(bufA, rwA, s0);
  yA = 0;
  for (int64_t c = 0; c < nchunks; c += 2) {
    // chunk c in A; issue chunk c + 1 into B, then fold A while B is in flight
    const int64_t q1 = s0 + (c + 1) * R, q2 = s0 + (c + 2) * R;
    if (c + 1 < nchunks) { issue(bufB, rwB, q1); yB = 0; yA += R; }
    wait_buf(bufA, yA);
    fold(bufA, rwA);
    if (c + 1 >= nchunks) break;
    // chunk c + 1 in B (its loads are older than the atomics just issued); refill boundary every 16 chunks
    if (c + 2 < nchunks) {
      if (((c + 2) & 15) == 0) {
        wait_buf(bufB, 0);                                   // one full stop per 128 keys (B's rows are needed next anyway)
        refill(q2);
        yA = yB = 0;
      }
      issue(bufA, rwA, q2);
      yA = 0;
      yB += R;
    }
    wait_buf(bufB, yB);
    fold(bufB, rwB);
  }
  if (cur != 0xffffffffu) {
    if (ATOM) flush_atomic(W + (int64_t)cur * 128, acc, gl);
    else { float* d = W + (int64_t)cur * 128 + gl; d[0] = acc.x; d[32] = acc.y; d[64] = acc.z; d[96] = acc.w; }
  }
}

// ---- 3-stage register pipeline, compiler-managed loads, atomics hidden from the compiler's wait-count scoreboard
// (inline asm): the loads of chunk i+2 are issued before chunk i is folded, so the counted wait the compiler emits
// for chunk i (vmcnt(16): two younger chunks) only ever waits for atomics that are a whole iteration old.
__device__ __forceinline__ void flush_atomic_asm(float* row_base, f32x4 v, int gl) {
  const int G = 32, q = 8;
  const int kb = gl / q, m = gl - kb * q;
  const bool b1 = kb & 2, b0 = kb & 1;
  float r0 = v.x, r1 = v.y, r2 = v.z, r3 = v.w;
  float x = b1 ? r0 : r2, y = __shfl_xor(x, 2 * q);
  if (b1) r0 = y; else r2 = y;
  x = b1 ? r1 : r3; y = __shfl_xor(x, 2 * q);
  if (b1) r1 = y; else r3 = y;
  x = b0 ? r0 : r1; y = __shfl_xor(x, q);
  if (b0) r0 = y; else r1 = y;
  x = b0 ? r2 : r3; y = __shfl_xor(x, q);
  if (b0) r2 = y; else r3 = y;
  float* d = row_base + 4 * m + kb;
  asm volatile("global_atomic_add_f32 %0, %1, off\n\tglobal_atomic_add_f32 %0, %2, off offset:128\n\t"
               "global_atomic_add_f32 %0, %3, off offset:256\n\tglobal_atomic_add_f32 %0, %4, off offset:384"
               :: "v"(d), "v"(r0), "v"(r1), "v"(r2), "v"(r3) : "memory");
  (void)G;
}

template <int NTG, int HIDE>
__global__ __launch_bounds__(256) void k_bwd_pipe3(const f32x4* __restrict__ GO, float* __restrict__ W, const u64* __restrict__ keys,
                                                   int64_t total, float alpha) {
  constexpr int R = 8;
  __shared__ u64 lk[8 * 256];
  const int tid = threadIdx.x, grp = tid >> 5, gl = tid & 31;
  const int g = blockIdx.x * 8 + grp;
  const int64_t all = (int64_t)gridDim.x * 8;
  const int64_t share = ((total + all - 1) / all + 15) / 16 * 16;      // <= 256 here
  const int64_t s0 = g * share, s1 = s0 + share < total ? s0 + share : total;
  if (s0 >= s1) return;
  u64* mylk = lk + grp * 256;
  for (int k = gl; k < 256; k += 32) mylk[k] = s0 + k < s1 ? keys[s0 + k] : ~0ull;
  const int nchunks = (int)((s1 - s0 + R - 1) / R);
  f32x4 acc = {0, 0, 0, 0};
  uint32_t cur = 0xffffffffu;
  f32x4 X[R], Y[R], Z[R];
  uint32_t rX[R], rY[R], rZ[R];
  auto issue = [&](f32x4 (&b)[R], uint32_t (&rw)[R], int c) {
#pragma unroll
    for (int t = 0; t < R; ++t) {
      const u64 k = c < nchunks ? mylk[c * R + t] : ~0ull;
      const bool on = k != ~0ull;
      rw[t] = on ? (uint32_t)(k >> 32) : 0xffffffffu;
      const int64_t src = on ? (int64_t)((uint32_t)k & 0x7fffffffu) : 0;
      b[t] = NTG ? __builtin_nontemporal_load(GO + src * 32 + gl) : GO[src * 32 + gl];     // unconditional: countable
    }
  };
  auto fold = [&](f32x4 (&b)[R], uint32_t (&rw)[R]) {
#pragma unroll
    for (int t = 0; t < R; ++t) {
      if (rw[t] != 0xffffffffu) {
        if (rw[t] != cur) {
          if (cur != 0xffffffffu) {
            if (HIDE) flush_atomic_asm(W + (int64_t)cur * 128, acc, gl);
            else flush_atomic(W + (int64_t)cur * 128, acc, gl);
          }
          acc = f32x4{0, 0, 0, 0};
          cur = rw[t];
        }
        acc = acc + b[t] * alpha;
      }
    }
  };
  issue(X, rX, 0);
  issue(Y, rY, 1);
  for (int c = 0; c < nchunks; c += 3) {
    issue(Z, rZ, c + 2);
    fold(X, rX);
    if (c + 1 >= nchunks) break;
    issue(X, rX, c + 3);
    fold(Y, rY);
    if (c + 2 >= nchunks) break;
    issue(Y, rY, c + 4);
    fold(Z, rZ);
  }
  if (cur != 0xffffffffu) {
    if (HIDE) flush_atomic_asm(W + (int64_t)cur * 128, acc, gl);
    else flush_atomic(W + (int64_t)cur * 128, acc, gl);
  }
}

static float time_it(const std::function<void()>& pre, const std::function<void()>& fn, int reps = 20) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) { pre(); fn(); }
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < reps; ++i) {
    pre();
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
  return ts[ts.size() / 2];      // median us
}

int main(int argc, char** argv) {
  const int F = 26, B = 16384, D = 128;
  const int64_t C = 1779442;
  const double s = 0.25;
  const int64_t sizes[F] = {45833188, 36746, 17245, 7413, 20243, 3, 7114, 1441, 62, 29275261, 1572176, 345138, 10, 2209, 11267,
                            128, 4, 974, 14, 48937457, 11316796, 40094537, 452104, 12606, 104, 35};
  std::mt19937_64 rng(1024);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  // slots: random distinct per unique (table, id)
  std::vector<int> pool(C);
  for (int64_t i = 0; i < C; ++i) pool[i] = (int)i;
  for (int64_t i = C - 1; i > 0; --i) std::swap(pool[i], pool[rng() % (i + 1)]);
  int64_t next_slot = 0;
  std::vector<std::vector<std::pair<uint32_t, uint32_t>>> seg(F);      // (slot, src row of grad_out)
  int64_t uniq = 0;
  for (int f = 0; f < F; ++f) {
    std::unordered_map<int64_t, int> slot_of;
    const double lo = pow(1.0 / (double)sizes[f], s);
    for (int b = 0; b < B; ++b) {
      const double x = U(rng) * (1.0 - lo) + lo;
      int64_t id = (int64_t)floor(1.0 / pow(x, 1.0 / s)) - 1;
      id = std::max<int64_t>(0, std::min<int64_t>(id, sizes[f] - 1));
      auto it = slot_of.find(id);
      int sl;
      if (it == slot_of.end()) { sl = pool[next_slot++]; slot_of[id] = sl; ++uniq; } else sl = it->second;
      seg[f].push_back({(uint32_t)sl, (uint32_t)(b * F + f)});
    }
    // exact grouping: by (bucket, slot), lookups of a row in batch order
    std::stable_sort(seg[f].begin(), seg[f].end(), [](auto& a, auto& b) {
      const uint32_t ba = a.first & 8191, bb = b.first & 8191;
      return ba != bb ? ba < bb : a.first < b.first;
    });
  }
  printf("batch: %d lookups, %lld unique rows\n", F * B, (long long)uniq);
  const int LMAX = argc > 1 ? atoi(argv[1]) : 32;
  std::vector<u64> k_cur, k_exp;
  std::vector<char> excl_start;      // per expanded key: 1 = a group share may start here
  int64_t n_excl = 0, n_runs = 0;
  for (int f = 0; f < F; ++f) {
    auto& v = seg[f];
    for (size_t i = 0; i < v.size();) {
      size_t j = i;
      while (j < v.size() && v[j].first == v[i].first) ++j;
      ++n_runs;
      const bool ex = (int)(j - i) <= LMAX;
      if (ex) { k_exp.push_back(((u64)v[i].first << 32) | 0x80000000u | v[i].first); excl_start.push_back(1); ++n_excl; }
      for (size_t t = i; t < j; ++t) {
        k_cur.push_back(((u64)v[t].first << 32) | v[t].second);
        k_exp.push_back(((u64)v[t].first << 32) | v[t].second);
        excl_start.push_back(ex ? (t == i ? 0 : 0) : 1);      // inside an exclusive run: not a legal start
      }
      i = j;
    }
  }
  printf("runs %lld, exclusive (len <= %d) %lld; keys %zu -> %zu\n", (long long)n_runs, LMAX, (long long)n_excl, k_cur.size(), k_exp.size());
  const int64_t T0 = k_cur.size(), T1 = k_exp.size();
  std::vector<u64> k_flag = k_cur;
  int64_t n_flag = 0;
  for (int64_t i = 0; i < T0;) {
    int64_t j = i;
    while (j < T0 && (k_cur[j] >> 32) == (k_cur[i] >> 32) && (j >> 14) == (i >> 14)) ++j;
    if (j - i <= LMAX && (i >> 4) == ((j - 1) >> 4)) { k_flag[i] |= 0x80000000ull; ++n_flag; }
    i = j;
  }
  printf("flagged run heads (len <= %d, inside one 16-block): %lld of %lld runs\n", LMAX, (long long)n_flag, (long long)n_runs);
  u64* d_flag;
  CK(hipMalloc(&d_flag, T0 * 8));
  CK(hipMemcpy(d_flag, k_flag.data(), T0 * 8, hipMemcpyHostToDevice));
  // compact index of every flagged run head (two-pass scratch)
  std::vector<int> h_cidx(T0, -1);
  {
    int c = 0;
    for (int64_t i = 0; i < T0; ++i) if (k_flag[i] & 0x80000000ull) h_cidx[i] = c++;
  }
  int* d_cidx;
  CK(hipMalloc(&d_cidx, T0 * 4));
  CK(hipMemcpy(d_cidx, h_cidx.data(), T0 * 4, hipMemcpyHostToDevice));
  f32x4* scratch;
  CK(hipMalloc(&scratch, (size_t)T0 * 512));

  f32x4* go;
  float *W, *W0;
  u64 *d_cur, *d_exp;
  CK(hipMalloc(&go, (size_t)F * B * 512));
  CK(hipMalloc(&W, C * 512));
  CK(hipMalloc(&W0, C * 512));
  CK(hipMalloc(&d_cur, T0 * 8));
  CK(hipMalloc(&d_exp, T1 * 8));
  CK(hipMemcpy(d_cur, k_cur.data(), T0 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_exp, k_exp.data(), T1 * 8, hipMemcpyHostToDevice));
  {
    std::vector<float> h((size_t)F * B * D);
    for (auto& x : h) x = (float)(U(rng) - 0.5) * 1e-2f;
    CK(hipMemcpy(go, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> w((size_t)C * D);
    for (size_t i = 0; i < w.size(); ++i) w[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
    CK(hipMemcpy(W0, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, W0, w.size() * 4, hipMemcpyDeviceToDevice));
    // ---- correctness of the new kernel against the atomic one on a few thousand rows
  }
  const float alpha = -0.5f;
  auto make_starts = [&](int grid, int R) {
    const int ng = grid * 8;
    std::vector<int> st(ng + 1);
    const int64_t share = ((T1 + ng - 1) / ng + R - 1) / R * R;
    for (int g = 0; g <= ng; ++g) {
      int64_t p = std::min<int64_t>(T1, (int64_t)g * share);
      while (p < T1 && !excl_start[p]) ++p;      // never cut an exclusive run
      st[g] = (int)p;
    }
    st[ng] = (int)T1;
    return st;
  };
  int* d_starts;
  CK(hipMalloc(&d_starts, (2048 * 8 + 1) * 4));
  auto nop = [] {};
  // correctness
  {
    const int grid = 512;
    auto st = make_starts(grid, 16);
    CK(hipMemcpy(d_starts, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, W0, C * 512, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL((k_bwd<16, 0, 1>), dim3(grid), dim3(256), 0, 0, go, W, d_cur, (const int*)nullptr, T0, alpha);
    std::vector<float> a((size_t)C * D), b((size_t)C * D);
    CK(hipMemcpy(a.data(), W, a.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(W, W0, C * 512, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL((k_bwd<16, 1, 1>), dim3(grid), dim3(256), 0, 0, go, W, d_exp, (const int*)d_starts, T1, alpha);
    CK(hipMemcpy(b.data(), W, b.size() * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0;
    for (size_t i = 0; i < a.size(); ++i) { maxd = std::max(maxd, (double)fabsf(a[i] - b[i])); maxv = std::max(maxv, (double)fabsf(a[i])); }
    printf("new vs atomic kernel: max |diff| = %.3g (max |w| %.3g)\n", maxd, maxv);
    CK(hipMemcpy(W, W0, C * 512, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL((k_bwd_flag<16, 1, 1>), dim3(grid), dim3(256), 0, 0, go, W, d_flag, T0, alpha);
    CK(hipMemcpy(b.data(), W, b.size() * 4, hipMemcpyDeviceToHost));
    maxd = 0;
    for (size_t i = 0; i < a.size(); ++i) maxd = std::max(maxd, (double)fabsf(a[i] - b[i]));
    printf("flag vs atomic kernel: max |diff| = %.3g\n", maxd);
    for (int sidx = 0; sidx < 2; ++sidx) {
      CK(hipMemcpy(W, W0, C * 512, hipMemcpyDeviceToDevice));
      if (sidx) {
        hipLaunchKernelGGL((k_bwd_2pA<16, 1, 1>), dim3(grid), dim3(256), 0, 0, go, W, d_flag, d_cidx, scratch, T0, alpha);
        hipLaunchKernelGGL((k_bwd_2pB<4, 1>), dim3(grid), dim3(256), 0, 0, W, d_flag, d_cidx, scratch, T0);
      } else {
        hipLaunchKernelGGL((k_bwd_2pA<16, 1, 0>), dim3(grid), dim3(256), 0, 0, go, W, d_flag, d_cidx, scratch, T0, alpha);
        hipLaunchKernelGGL((k_bwd_2pB<4, 0>), dim3(grid), dim3(256), 0, 0, W, d_flag, d_cidx, scratch, T0);
      }
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(b.data(), W, b.size() * 4, hipMemcpyDeviceToHost));
      maxd = 0;
      for (size_t i = 0; i < a.size(); ++i) maxd = std::max(maxd, (double)fabsf(a[i] - b[i]));
      printf("two-pass (scratch %s) vs atomic kernel: max |diff| = %.3g\n", sidx ? "compact" : "by position", maxd);
    }
    CK(hipMemcpy(W, W0, C * 512, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL((k_bwd_split<16, 1, 0>), dim3(683), dim3(256), 0, 0, go, W, d_cur, T0, alpha);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(b.data(), W, b.size() * 4, hipMemcpyDeviceToHost));
    maxd = 0;
    for (size_t i = 0; i < a.size(); ++i) maxd = std::max(maxd, (double)fabsf(a[i] - b[i]));
    printf("split vs atomic kernel: max |diff| = %.3g\n", maxd);
    {
      // vmcnt ordering
      float* srcv; float* hot; float* bad;
      const int64_t nl = 1 << 22;
      CK(hipMalloc(&srcv, nl * 128)); CK(hipMalloc(&hot, 4096)); CK(hipMalloc(&bad, 4));
      std::vector<float> hs(nl * 32);
      for (int64_t i = 0; i < nl; ++i) hs[i * 32] = (float)(i & 1023);
      CK(hipMemcpy(srcv, hs.data(), nl * 128, hipMemcpyHostToDevice));
      CK(hipMemset(hot, 0, 4096)); CK(hipMemset(bad, 0, 4));
      hipLaunchKernelGGL(k_vmcnt_order, dim3(2048), dim3(256), 0, 0, srcv, hot, bad, nl, 64);
      CK(hipDeviceSynchronize());
      float hb = -1; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
      printf("vmcnt ordering: %.0f of %lld loads were not there after s_waitcnt vmcnt(4) behind 4 younger atomics\n", hb, 2048ll * 256 * 64);
    }
    CK(hipMemcpy(W, W0, C * 512, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL((k_bwd_pipe<1, 1>), dim3(512), dim3(256), 0, 0, go, W, d_cur, T0, alpha);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(b.data(), W, b.size() * 4, hipMemcpyDeviceToHost));
    maxd = 0;
    for (size_t i = 0; i < a.size(); ++i) maxd = std::max(maxd, (double)fabsf(a[i] - b[i]));
    printf("pipe vs atomic kernel: max |diff| = %.3g\n", maxd);
    CK(hipMemcpy(W, W0, C * 512, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL((k_bwd_pipe3<1, 1>), dim3(512), dim3(256), 0, 0, go, W, d_cur, T0, alpha);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(b.data(), W, b.size() * 4, hipMemcpyDeviceToHost));
    maxd = 0;
    for (size_t i = 0; i < a.size(); ++i) maxd = std::max(maxd, (double)fabsf(a[i] - b[i]));
    printf("pipe3 vs atomic kernel: max |diff| = %.3g\n", maxd);
  }
#define RUN(NAME, R, MODE, NTG, GRID, KEYS, T, STARTS)                                                        \
  {                                                                                                           \
    float t = time_it(nop, [&] { hipLaunchKernelGGL((k_bwd<R, MODE, NTG>), dim3(GRID), dim3(256), 0, 0, go, W, KEYS, STARTS, T, alpha); }); \
    printf("%-34s R=%2d grid=%5d : %6.1f us\n", NAME, R, GRID, t);                                            \
  }
  for (int grid : {512, 768, 1024}) {
    RUN("cur  (atomics, nt grads)", 16, 0, 1, grid, d_cur, T0, (const int*)nullptr);
    RUN("cur  (atomics, plain loads)", 16, 0, 0, grid, d_cur, T0, (const int*)nullptr);
    RUN("cur  no update traffic", 16, 2, 1, grid, d_cur, T0, (const int*)nullptr);
    RUN("cur  R=8 atomics", 8, 0, 1, grid, d_cur, T0, (const int*)nullptr);
    auto st = make_starts(grid, 16);
    CK(hipMemcpy(d_starts, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    RUN("new  (W keys + stores, nt grads)", 16, 1, 1, grid, d_exp, T1, (const int*)d_starts);
    RUN("new  (W keys + stores, plain)", 16, 1, 0, grid, d_exp, T1, (const int*)d_starts);
    RUN("new  no update traffic", 16, 3, 1, grid, d_exp, T1, (const int*)d_starts);
#define RUNF(NAME, R, NTG, FL, GRID)                                                                         \
  {                                                                                                           \
    float t = time_it(nop, [&] { hipLaunchKernelGGL((k_bwd_flag<R, NTG, FL>), dim3(GRID), dim3(256), 0, 0, go, W, d_flag, T0, alpha); }); \
    printf("%-34s R=%2d grid=%5d : %6.1f us\n", NAME, R, GRID, t);                                            \
  }
#define RUNS(NAME, R, NTG, ST, GRID)                                                                         \
  {                                                                                                           \
    float t = time_it(nop, [&] { hipLaunchKernelGGL((k_bwd_split<R, NTG, ST>), dim3(GRID), dim3(256), 0, 0, go, W, d_cur, T0, alpha); }); \
    printf("%-34s R=%2d grid=%5d : %6.1f us\n", NAME, R, GRID, t);                                            \
  }
#define RUNP(NAME, NTG, AT, GRID)                                                                             \
  {                                                                                                           \
    float t = time_it(nop, [&] { hipLaunchKernelGGL((k_bwd_pipe<NTG, AT>), dim3(GRID), dim3(256), 0, 0, go, W, d_cur, T0, alpha); }); \
    printf("%-34s R= 8 grid=%5d : %6.1f us\n", NAME, GRID, t);                                            \
  }
#define RUNP3(NAME, NTG, HIDE, GRID)                                                                          \
  {                                                                                                           \
    float t = time_it(nop, [&] { hipLaunchKernelGGL((k_bwd_pipe3<NTG, HIDE>), dim3(GRID), dim3(256), 0, 0, go, W, d_cur, T0, alpha); }); \
    printf("%-34s R= 8 grid=%5d : %6.1f us\n", NAME, GRID, t);                                            \
  }
#define RUN2P(NAME, NTG, SIDX, RB, GRID, GRIDB)                                                                \
  {                                                                                                           \
    float ta = time_it(nop, [&] { hipLaunchKernelGGL((k_bwd_2pA<16, NTG, SIDX>), dim3(GRID), dim3(256), 0, 0, go, W, d_flag, d_cidx, scratch, T0, alpha); }); \
    float tb = time_it(nop, [&] { hipLaunchKernelGGL((k_bwd_2pB<RB, SIDX>), dim3(GRIDB), dim3(256), 0, 0, W, d_flag, d_cidx, scratch, T0); }); \
    float tab = time_it(nop, [&] { hipLaunchKernelGGL((k_bwd_2pA<16, NTG, SIDX>), dim3(GRID), dim3(256), 0, 0, go, W, d_flag, d_cidx, scratch, T0, alpha); \
                                   hipLaunchKernelGGL((k_bwd_2pB<RB, SIDX>), dim3(GRIDB), dim3(256), 0, 0, W, d_flag, d_cidx, scratch, T0); }); \
    printf("%-34s gridA=%5d gridB=%5d : A %6.1f  B %6.1f  A+B %6.1f us\n", NAME, GRID, GRIDB, ta, tb, tab);      \
  }
    RUN2P("2pass nt, scratch by position, RB4", 1, 0, 4, grid, grid);
    RUN2P("2pass nt, scratch compact, RB4", 1, 1, 4, grid, grid);
    RUN2P("2pass nt, compact, RB8, B 2x grid", 1, 1, 8, grid, 2 * grid);
    RUN2P("2pass nt, compact, RB2, B 4x grid", 1, 1, 2, grid, 4 * grid);
    RUN2P("2pass plain, compact, RB4", 0, 1, 4, grid, grid);
    RUNP3("pipe3 (3 x 8 rows, asm atomics, nt)", 1, 1, grid);
    RUNP3("pipe3 (3 x 8 rows, asm atomics, pl)", 0, 1, grid);
    RUNP3("pipe3 (3 x 8 rows, C++ atomics, nt)", 1, 0, grid);
    RUNP("pipe (2 x 8 rows, atomics, nt)", 1, 1, grid);
    RUNP("pipe (2 x 8 rows, atomics, plain)", 0, 1, grid);
    RUNP("pipe (2 x 8 rows, stores, nt)", 1, 0, grid);
    RUNF("flag (side load + store, nt)", 16, 1, 1, grid);
    RUNF("flag (side load + store, plain)", 16, 0, 1, grid);
    RUNF("flag no update traffic", 16, 1, 0, grid);
    RUNF("flag R=8 nt", 8, 1, 1, grid);
    RUNF("flag R=8 plain", 8, 0, 1, grid);
    auto st8 = make_starts(grid, 8);
    CK(hipMemcpy(d_starts, st8.data(), st8.size() * 4, hipMemcpyHostToDevice));
    RUN("new  R=8", 8, 1, 1, grid, d_exp, T1, (const int*)d_starts);
  }
  return 0;
}
