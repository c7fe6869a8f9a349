// EmbeddingBag gather-reduce forward (K12) and gradient scatter backward (K13/K14) for
// gfx950.  Replaces F.embedding_bag + autograd + SGD.step on the cache parameter
// (reference call sites: recsys/models/dlrm.py:99-110, benchmark/benchmark_cache.py:62-65,
// recsys/dlrm_main.py:274-279).  Bandwidth-bound gather / scatter: no MFMA.  A row is always moved by a
// lane group of G lanes (G * 16 B >= one row, G = 32 for D = 128) with 16-byte accesses.  What is in this file:
//   * k_bag_fwd -- the general forward (any offsets, mean, per-sample weights): a wave owns tiles of 64 consecutive
//     bags, descriptors in one VGPR per lane broadcast with ds_bpermute; single-id tiles (all Criteo / Avazu batches:
//     recsys/datasets/criteo.py:129-130) fetch 64 ids with one coalesced load and keep U row loads per lane in flight;
//     tiles with multi-id bags stage their contiguous index range in LDS.
//   * k_bag_presort_seg -- once per prefetch window: every 16384-lookup segment GROUPED by row with one LDS counting
//     pass (8192 bucket counters, a returning LDS atomic per lookup, a scan, a scatter); optionally resolves the
//     grad_out row of every lookup (SRC: source-row keys), the slot of every row (ROWS: the cache op's last kernel)
//     and owner-exclusive runs (EXCL: the whole segment's keys staged in 128 KB of LDS).
//   * k_bag_fwd_keys / k_bag_bwd_stream -- what the window pipelines and bench.py run: both walk the window's keys in
//     contiguous shares, keys staged through a group-private LDS slice, 16 rows in flight per lane group; the forward
//     loads a cache row once per run of equal rows, the backward folds a run and issues ONE transposed atomic row
//     update per run.  Shares are static but many (16 / 6 workgroups per CU: the dispatcher hands them out as CUs free up).
//   * k_bag_bwd_tile -- the backward for mean / per-sample weights / unsorted input (sorts 1024-lookup tiles in
//     registers + LDS itself, or walks presorted segments), k_bag_bwd_rows (COO values), k_rows_axpy (row-wise exchange).
// Output stores are non-temporal (never re-read here); row loads use the default policy so hot rows stay in
// L2 / Infinity Cache.
#include <stdlib.h>

#include <algorithm>

#include "ce_common.h"

namespace ce {

// Ablation switches of the backward kernels (CE_BWD_DEBUG: 1 = racy read-modify-write instead of atomics, 2 = no
// update traffic, 3 / 4 = tile prologue only, 5 = untransposed atomics, 6 = plain stores; DESIGN.md section 4 quotes
// them).  They produce WRONG results by design, so a product build compiles them out: -DCE_ABLATIONS brings them back.
#ifdef CE_ABLATIONS
#define CE_DBG(x) (x)
#else
#define CE_DBG(x) 0
#endif

struct BagParams {
  const float* weight;      // fwd: rows to gather from
  float* dst;               // fwd: out; bwd: grad_weight / weight / grad_rows
  const float* grad_out;    // bwd only
  const int64_t* indices;
  const void* offsets;
  const float* psw;
  int64_t nnz;
  int32_t num_bags;
  int32_t rowlen;           // vector chunks per row (dim/4 or dim)
  int32_t g_log2;           // log2(lanes per group)
  int32_t off64;
  int32_t include_last;
  int32_t mode;
  int32_t hookF;            // 0 = plain [num_bags, D]
  int32_t hookB;            // num_bags / hookF
  float alpha;              // bwd scale (1 or -lr)
  int32_t debug;            // ablation switch (CE_BWD_DEBUG): 0 = normal
  uint32_t num_rows;        // rows of the gathered / updated table: out-of-range indices are ignored
  int32_t tile_len;         // lookups per workgroup tile of the sorted scatter
  const unsigned long long* presorted;   // optional: segment-grouped keys from ce_bag_presort* (no sort in the kernel)
  int32_t policy;           // bit 0: fwd output stores non-temporal; bit 1: bwd gradient-row loads non-temporal
  int32_t interleave;       // key-walking kernels: lane group j of workgroup b takes share j * gridDim + b (see share_of)
};

// Which contiguous share of the key array a lane group walks.  The keys lie segment by segment, i.e. FEATURE by
// feature, and shares differ a lot in cost: a share of a small table's feature is one long run (gathers only, one
// row update / one row load), a share of a 45 M-row table's feature is mostly runs of one (an update / a load per
// key).  Consecutive shares per workgroup put eight alike shares into one workgroup, and with every workgroup resident
// at once the launch lasts as long as its slowest workgroup.  Interleaved, the groups of a workgroup take shares one
// eighth of the array apart: every workgroup gets a sample of the whole batch.  Streaming backward at the bench shape,
// back to back: 61.4 -> 53.5 us (means of four runs; 0.53 -> 0.62 of the HBM peak), 79 -> 75 us beside the cache op.
// (Keeping the two lane groups of a WAVE on adjacent shares and spreading only the waves: 61 us again -- it is the
// spreading itself that pays, not less divergence; a pseudo-random bijection of shares to lane groups: the same
// 52-57 us as this.  The key-driven forward does not gain: its cost per share is its
// stores, the same for every share.)
__device__ __forceinline__ int64_t share_of(const BagParams& p, int grp, int ngroups) {
  return p.interleave ? (int64_t)grp * gridDim.x + blockIdx.x : (int64_t)blockIdx.x * ngroups + grp;
}

// cache policies of the two big streams: both non-temporal (the sweeps of rounds 3-5 that said so
// are in profiles/ and docs/history.md, their switches are gone)
constexpr int kBagPolicy = 3;

// offsets == nullptr: the caller states one id per bag, in order (offsets = arange; ce_bag_forward, the presorts)
__device__ __forceinline__ int ld_off(const BagParams& p, int i) {
  if (p.offsets == nullptr) return i;
  return p.off64 ? (int)((const int64_t*)p.offsets)[i] : ((const int32_t*)p.offsets)[i];
}
__device__ __forceinline__ int bag_end(const BagParams& p, int b) {
  return (p.include_last || b + 1 < p.num_bags) ? ld_off(p, b + 1) : (int)p.nnz;
}
// row of the [B, F, D] output that feature-major bag g = f*B + b lands in
__device__ __forceinline__ int64_t out_row(const BagParams& p, int g) {
  if (p.hookF == 0) return g;
  int f = g / p.hookB;
  int b = g - f * p.hookB;
  return (int64_t)b * p.hookF + f;
}

constexpr int kIdxStage = 2048;   // indices of one 64-bag tile staged in LDS (8 KB per wave)

// Output store of the forward: non-temporal (SP = 1).  The output is written once and read by another kernel; what
// matters is how much of the L2 / Infinity Cache it takes from the cache rows the gather wants to find there (the
// sc1 / sc0 sc1 / sc1 nt forms measured the same or slower: profiles/r03_probe_fwd_xcd.txt, docs/history.md).
template <int SP>
__device__ __forceinline__ void store_out(f32x4* p, f32x4 v) {
  if (SP == 0) *p = v;
  else __builtin_nontemporal_store(v, p);
}
template <int SP>
__device__ __forceinline__ void store_out(float* p, float v) {
  if (SP == 0) *p = v;
  else __builtin_nontemporal_store(v, p);
}

// STAGE: tiles with multi-id bags stage their indices in LDS (32 KB per workgroup); the launcher picks the
// LDS-free variant when nnz == num_bags (single-id batches) so occupancy is set by registers alone.
template <typename VT, int NCH, bool STAGE, int U, int NTS>
__global__ __launch_bounds__(256) void k_bag_fwd(BagParams p) {
  __shared__ int lds_idx[STAGE ? 4 : 1][STAGE ? kIdxStage : 1];
  const int lane = threadIdx.x & 63;
  const int G = 1 << p.g_log2;
  const int gpw = 64 >> p.g_log2;
  const int grp = lane >> p.g_log2;
  const int gl = lane & (G - 1);
  const int wpb = blockDim.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * wpb;
  const VT* __restrict__ W = (const VT*)p.weight;
  VT* __restrict__ O = (VT*)p.dst;
  const int rowlen = p.rowlen;
  // tiles are dealt round-robin to the waves of an oversubscribed grid (an even contiguous split over a
  // resident-sized grid measured 15 % slower: 0.060 -> 0.069 ms)
  const int ntiles = (p.num_bags + 63) >> 6;

  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int b0 = (int)(tile << 6);
    const int nb = min(64, p.num_bags - b0);
    int lo = 0, hi = 0;
    if (lane < nb) {
      lo = ld_off(p, b0 + lane);
      hi = bag_end(p, b0 + lane);
    }
    const bool single = (hi - lo == 1) || (lane >= nb);
    if (__all(single)) {
      // ---- single-id tile: pure indexed row copy, U rows in flight per lane group
      int idx = 0;
      float w = 1.f;
      if (lane < nb) {
        idx = (int)p.indices[lo];
        if (p.psw) w = p.psw[lo];
      }
      for (int base = 0; base < nb; base += gpw * U) {
        VT v[U][NCH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int bi = base + u * gpw + grp;
          const int ri = __shfl(idx, bi & 63);
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const int ch = gl + c * G;
            v[u][c] = vzero<VT>();
            if (bi < nb && ch < rowlen && (uint32_t)ri < p.num_rows) v[u][c] = W[(int64_t)ri * rowlen + ch];
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int bi = base + u * gpw + grp;
          const float wi = __shfl(w, bi & 63);
          if (bi < nb) {
            const int64_t orow = out_row(p, b0 + bi);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              const int ch = gl + c * G;
              if (ch < rowlen) {
                VT val = p.psw ? v[u][c] * wi : v[u][c];
                store_out<NTS>(&O[orow * rowlen + ch], val);
              }
            }
          }
        }
      }
    } else {
      // ---- general tile.  The 64 bags' indices are one contiguous range: stage it in LDS with coalesced
      // loads (removes a dependent global round trip per step), then each lane group walks its bag with
      // 8 rows in flight.
      const int t_lo = __shfl(lo, 0);
      const int t_hi = __shfl(hi, nb - 1);
      const int t_n = t_hi - t_lo;
      const bool staged = STAGE && t_n <= kIdxStage;
      int* sidx = &lds_idx[STAGE ? (threadIdx.x >> 6) : 0][0];
      if (staged) {
        for (int k = lane; k < t_n; k += 64) sidx[k] = (int)p.indices[t_lo + k];
      }
      __builtin_amdgcn_wave_barrier();
      for (int base = 0; base < nb; base += gpw) {
        const int bi = base + grp;
        int blo = __shfl(lo, bi & 63);
        int bhi = __shfl(hi, bi & 63);
        if (bi >= nb) blo = bhi = 0;
        VT acc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] = vzero<VT>();
        for (int j = blo; j < bhi; j += 8) {
          int r[8];
          float w[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            r[t] = -1;
            w[t] = 1.f;
            if (j + t < bhi) {
              r[t] = staged ? sidx[j + t - t_lo] : (int)p.indices[j + t];
              if (p.psw) w[t] = p.psw[j + t];
            }
          }
          VT v[8][NCH];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              const int ch = gl + c * G;
              v[t][c] = vzero<VT>();
              if (ch < rowlen && (uint32_t)r[t] < p.num_rows) v[t][c] = W[(int64_t)r[t] * rowlen + ch];
            }
          }
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            if (j + t < bhi) {
#pragma unroll
              for (int c = 0; c < NCH; ++c) acc[c] = p.psw ? acc[c] + v[t][c] * w[t] : acc[c] + v[t][c];
            }
          }
        }
        if (p.mode == CE_MODE_MEAN && bhi - blo > 1) {
          const float inv = 1.f / (float)(bhi - blo);
#pragma unroll
          for (int c = 0; c < NCH; ++c) acc[c] = acc[c] * inv;
        }
        if (bi < nb) {
          const int64_t orow = out_row(p, b0 + bi);
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const int ch = gl + c * G;
            if (ch < rowlen) {
              store_out<NTS>(&O[orow * rowlen + ch], acc[c]);
            }
          }
        }
      }
    }
  }
}

__device__ __forceinline__ void atomic_add_vec(float* dst, float v) {
  __hip_atomic_fetch_add(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_add_vec(f32x4* dst, f32x4 v) {
  float* d = (float*)dst;
  __hip_atomic_fetch_add(d + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(d + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(d + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_fetch_add(d + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Row update of the tile-sorted scatter.  A lane holds 4 consecutive floats (16-B loads), but four
// scalar atomics issued that way each touch EVERY 128-B line of the row (16-B lane stride), and the L2
// atomic path is paid per line request.  A 4x4 butterfly transpose across the lane blocks of the group
// (2 x 2 xor-shuffles) regroups the data so that atomic instruction c covers one contiguous block of G
// floats: 4x fewer line requests per row (255 -> ~150 us/step measured).
__device__ __forceinline__ void flush_chunk(float* row_base, f32x4 v, int gl, int G, int c, int rowlen, int debug) {
  if (debug == 6) {                                          // ablation: plain store instead of the atomics (wrong result)
    const int ch6 = gl + c * G;
    if (ch6 < rowlen) ((f32x4*)row_base)[ch6] = v;
    return;
  }
  if (G >= 4 && debug == 0 && (c + 1) * G <= rowlen) {      // group-uniform: whole chunk present
    const int q = G >> 2;                     // lanes per lane block
    const int kb = gl / q;                    // my lane block 0..3
    const int m = gl - kb * q;
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
    float* d = row_base + c * G * 4 + 4 * m + kb;       // register k -> element k*G + 4m + kb of the chunk
    __hip_atomic_fetch_add(d, r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(d + G, r1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(d + 2 * G, r2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(d + 3 * G, r3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int ch = gl + c * G;
  if (ch >= rowlen) return;
  f32x4* dst = (f32x4*)row_base + ch;
  if (debug == 0 || debug == 5) atomic_add_vec(dst, v);
  else if (debug == 1) *dst = *dst + v;                      // ablation: plain read-modify-write (racy)
  else if (debug == 2) { if (v.x == 12345.678f) *dst = v; }  // ablation: no update traffic
}
__device__ __forceinline__ void flush_chunk(float* row_base, float v, int gl, int G, int c, int rowlen, int debug) {
  const int ch = gl + c * G;
  if (ch >= rowlen) return;
  float* dst = row_base + ch;
  if (debug == 0 || debug == 5) atomic_add_vec(dst, v);
  else if (debug == 1) *dst = *dst + v;
  else if (debug == 2) { if (v == 12345.678f) *dst = v; }
}

// Per-lookup gradient rows (COO values of the sparse=True backward; dest_index redirects row j):
// dst[indices ? indices[j] : j] = alpha*scale*psw[j]*grad_out[bag(j)].
template <typename VT, int NCH>
__global__ __launch_bounds__(256) void k_bag_bwd_rows(BagParams p) {
  constexpr int U = (NCH == 1) ? 4 : (NCH == 2 ? 2 : 1);
  const int lane = threadIdx.x & 63;
  const int G = 1 << p.g_log2;
  const int gpw = 64 >> p.g_log2;
  const int grp = lane >> p.g_log2;
  const int gl = lane & (G - 1);
  const int wpb = blockDim.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * wpb;
  const int ntiles = (p.num_bags + 63) >> 6;
  const VT* __restrict__ GO = (const VT*)p.grad_out;
  VT* __restrict__ DST = (VT*)p.dst;
  const int rowlen = p.rowlen;

  for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
    const int b0 = (int)(tile << 6);
    const int nb = min(64, p.num_bags - b0);
    int lo = 0, hi = 0;
    if (lane < nb) {
      lo = ld_off(p, b0 + lane);
      hi = bag_end(p, b0 + lane);
    }
    for (int base = 0; base < nb; base += gpw * U) {
      VT g[U][NCH];
      int blo[U], bhi[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int bi = base + u * gpw + grp;
        blo[u] = __shfl(lo, bi & 63);
        bhi[u] = __shfl(hi, bi & 63);
        if (bi >= nb) blo[u] = bhi[u] = 0;
        const int64_t orow = out_row(p, b0 + min(bi, nb - 1));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int ch = gl + c * G;
          g[u][c] = vzero<VT>();
          if (bhi[u] > blo[u] && ch < rowlen) g[u][c] = GO[orow * rowlen + ch];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float scale = p.alpha;
        if (p.mode == CE_MODE_MEAN && bhi[u] - blo[u] > 1) scale = scale / (float)(bhi[u] - blo[u]);
        for (int j = blo[u]; j < bhi[u]; ++j) {
          const float s = p.psw ? scale * p.psw[j] : scale;
          const int64_t r = p.indices ? p.indices[j] : (int64_t)j;
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const int ch = gl + c * G;
            if (ch < rowlen) {
              VT val = g[u][c] * s;
              __builtin_nontemporal_store(val, &DST[r * rowlen + ch]);
            }
          }
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------
// Tile-sorted scatter (K13/K14 main path).  fp32 atomics aimed at one row serialise, and Criteo-
// shaped batches hammer a few rows (tables of 3..100 rows get B lookups each; power-law heads).
// So a workgroup takes 1024 consecutive lookups, sorts (row, lookup) keys in LDS (bitonic, 8 KB),
// and folds runs of equal rows BEFORE touching HBM.  Gradient rows are read exactly once, coalesced
// per lane group, 8 in flight per lane.
constexpr int kBwdTile = 1024;

__device__ __forceinline__ int find_bag(const BagParams& p, int j) {
  // bag whose [offsets[b], end(b)) holds lookup j; single-id layouts (offsets == arange) hit the first test
  if (j < p.num_bags && ld_off(p, j) == j && bag_end(p, j) == j + 1) return j;
  int lo = 0, hi = p.num_bags - 1;
  while (lo < hi) {                       // last b with offsets[b] <= j
    const int mid = (lo + hi + 1) >> 1;
    if (ld_off(p, mid) <= j) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// sort keys: (row, lookup-in-tile).  32-bit keys (row < 2^22, i.e. caches up to 4 M rows) halve the LDS traffic
// of the bitonic network; wider tables fall back to 64-bit keys.
template <typename KT> struct KeyOps;
template <> struct KeyOps<uint32_t> {
  static __device__ __forceinline__ uint32_t make(uint32_t row, int i) { return (row << 10) | (uint32_t)i; }
  static __device__ __forceinline__ uint32_t row(uint32_t k) { return k >> 10; }
  static __device__ __forceinline__ int idx(uint32_t k) { return (int)(k & 1023u); }
  static __device__ __forceinline__ uint32_t invalid() { return 0xffffffffu; }
  static __device__ __forceinline__ uint32_t row_invalid() { return 0x3fffffu; }
};
template <> struct KeyOps<unsigned long long> {
  static __device__ __forceinline__ unsigned long long make(uint32_t row, int i) {
    return ((unsigned long long)row << 32) | (uint32_t)i;
  }
  static __device__ __forceinline__ uint32_t row(unsigned long long k) { return (uint32_t)(k >> 32); }
  static __device__ __forceinline__ int idx(unsigned long long k) { return (int)(uint32_t)k; }
  static __device__ __forceinline__ unsigned long long invalid() { return ~0ull; }
  static __device__ __forceinline__ uint32_t row_invalid() { return 0xffffffffu; }
};

// Bitonic sort of the tile's 1024 keys, four per thread (thread t holds positions 4t..4t+3).  Compare-exchange
// distances 1-2 stay inside a thread, 4-128 inside a wave (__shfl_xor, no LDS, no barrier); only the three stages
// with distance 256 / 512 cross waves and go through LDS.  (The all-LDS network paid a barrier on each of its 55
// stages: 13 us per tile; this one 3.)  Ends with the sorted keys in lds[0..1023] and a barrier.
template <typename KT>
__device__ __forceinline__ KT shfl_xor_key(KT v, int m) { return __shfl_xor(v, m); }
template <>
__device__ __forceinline__ unsigned long long shfl_xor_key<unsigned long long>(unsigned long long v, int m) {
  const unsigned lo = __shfl_xor((unsigned)v, m), hi = __shfl_xor((unsigned)(v >> 32), m);
  return ((unsigned long long)hi << 32) | lo;
}

template <typename KT>
__device__ __forceinline__ void tile_sort_1024(KT (&key)[4], KT* lds, int tid) {
  for (int k = 2; k <= kBwdTile; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 256) {                                  // partner lives in another wave
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[tid * 4 + r] = key[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = tid * 4 + r;
          const KT other = lds[i ^ j];
          const bool lower = (i & j) == 0, up = (i & k) == 0;
          const KT lo = key[r] < other ? key[r] : other, hi = key[r] < other ? other : key[r];
          key[r] = (lower == up) ? lo : hi;
        }
        __syncthreads();
      } else if (j >= 4) {                             // partner lane = lane ^ (j / 4)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = tid * 4 + r;
          const KT other = shfl_xor_key<KT>(key[r], j >> 2);
          const bool lower = (i & j) == 0, up = (i & k) == 0;
          const KT lo = key[r] < other ? key[r] : other, hi = key[r] < other ? other : key[r];
          key[r] = (lower == up) ? lo : hi;
        }
      } else {                                         // j = 1, 2: both elements in this thread
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (r & j) continue;
          const bool up = ((tid * 4 + r) & k) == 0;
          const KT a = key[r], b = key[r | j];
          if ((a > b) == up) {
            key[r] = b;
            key[r | j] = a;
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) lds[tid * 4 + r] = key[r];
  __syncthreads();
}

// The order the backward folds duplicates in does not depend on the gradient, only on the slots the cache op
// returns -- so it can be computed once per window on the cache-op stream (ce_bag_presort), and over a much wider
// scope than a workgroup can sort inside the backward: SEGMENTS of 16384 consecutive lookups (one Criteo feature
// at B = 16384).  The scope matters because the memory side retires fp32 atomics at only ~1.5 TB/s (one 64-byte
// line-op per ~5 ns and channel): 1024-lookup tiles leave 109 k row updates per batch, 16384-lookup segments ~55 k.
// The backward only needs EQUAL ROWS TO BE ADJACENT, not a total order, so the segment is not sorted (a 16 k
// bitonic network in registers / shuffles / LDS took 170 us per window, as much as it saved) but GROUPED with one
// counting pass: bucket = row & 8191; the lanes of a wave that hold the same bucket are matched with ballots and
// their leader reserves their places with ONE returning LDS atomic; a scan of the 8192 counters turns
// (bucket, place) into the output position.  ~2 keys share a bucket, so a hot row's lookups end up contiguous
// apart from the odd cold row of the same bucket.  One workgroup of 1024 threads per segment, 16 keys per thread.
// Key = row << 32 | lookup-in-segment; ignored lookups (out-of-range row) and padding = all ones, placed last.
constexpr int kSegLen = 16384;
constexpr int kSegKeys = 16;          // per thread
#ifndef CE_SEG_BUCKETS
#define CE_SEG_BUCKETS 8192
#endif
constexpr int kSegBuckets = CE_SEG_BUCKETS;
// Window form: n_segs = n_batches * segs_per_batch; batch b owns lookups [b * nnz_per_batch, (b + 1) * nnz_per_batch)
// and the keys [b * segs_per_batch * kSegLen, ...) -- segments never straddle two batches.
// SRC: the low word of a key is not the lookup's place in its segment but the row of grad_out the lookup reads
// (out_row(bag of the lookup), from the batch's offsets): everything the streaming backward needs, resolved here
// once per window instead of in every backward launch.  lay.offsets of batch b = offsets + b * off_stride elements.
// EXCL (with SRC): the keys are staged in LDS at their grouped positions, every bucket of at most kExclRun keys is
// sorted by row there (a handful of keys: insertion sort by the thread that owns the bucket's first position), and
// the first key of every run that lies inside ONE 16-position block of the segment gets bit 31 of its low word set:
// "all lookups of this row in this segment follow, and whoever reads this key reads all of them" -- shares of the
// streaming backward are multiples of 16 positions.  Together with the segment's id range (ids_minmax: no id is
// shared between two segments of a batch when their ranges are disjoint) that lets the backward update such a row
// with a plain read-modify-write instead of atomics.  The write-out is then one coalesced sweep.
constexpr int kExclRun = 32;
constexpr unsigned kExclFlag = 0x80000000u;

// ROWS (the cache op's fused form, presort_window_from_rows): slots_io holds the table ROW of every lookup; its slot
// is inverted[row] (one random 4-byte gather per lookup, all 16 of a thread in flight before anything else happens)
// and is written back in place.
// Every form stages: the keys go to their grouped positions in LDS first and leave with 16 coalesced 512-byte stores
// per wave instead of 16 x 64 scattered 8-byte ones -- one workgroup is one CU's store path, and 16384 single-line
// writes through it were most of the kernel's 43 us (it is latency-bound: 26-208 workgroups; the direct scatter it
// replaced: profiles/r05_ab_presort_staged.txt).  128 KB of LDS: the library needs gfx950's 160 KB per CU.
template <bool SRC, bool EXCL, bool ROWS>
__global__ __launch_bounds__(1024) void k_bag_presort_seg(const int64_t* __restrict__ indices, int64_t nnz_per_batch,
                                                         int32_t segs_per_batch, int64_t n_segs, uint32_t num_rows,
                                                         unsigned long long* __restrict__ keys_out, BagParams lay,
                                                         int64_t off_stride, const int64_t* __restrict__ ids,
                                                         int64_t* __restrict__ ids_minmax, int64_t* slots_io,
                                                         const int32_t* __restrict__ inverted,
                                                         const int* __restrict__ status) {
  __shared__ unsigned long long lk[kSegLen];          // the segment's keys by position (the counters first)
  static_assert(sizeof(unsigned long long) * kSegLen <= 160 * 1024 - 1024, "gfx950: 160 KB of LDS per CU");
  int* const cnt = (int*)lk;                            // [kSegBuckets + 1] bucket counters ([kSegBuckets] = ignored)
  __shared__ int wsum[16];
  __shared__ long long mm_s[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int64_t seg = blockIdx.x; seg < n_segs; seg += gridDim.x) {
    const int64_t batch = seg / segs_per_batch;
    const int sib = (int)(seg - batch * segs_per_batch);
    const int64_t in_base = batch * nnz_per_batch + (int64_t)sib * kSegLen;
    const int64_t base = seg * kSegLen;
    const int n_here = (int)min((int64_t)kSegLen, nnz_per_batch - (int64_t)sib * kSegLen);
    BagParams q = lay;
    if (SRC && lay.offsets) q.offsets = (const char*)lay.offsets + batch * off_stride * (lay.off64 ? 8 : 4);
    for (int i = tid; i <= kSegBuckets; i += 1024) cnt[i] = 0;
    __syncthreads();
    unsigned long long key[kSegKeys];
    int place[kSegKeys], bkt[kSegKeys];                  // place inside the bucket (a bucket can hold the whole segment)
    int32_t slot_r[ROWS ? kSegKeys : 1];
    if (ROWS) {
      const bool failed = *status != CE_OK;              // a failed cache op hands back -1 everywhere
      int32_t row_r[kSegKeys];
#pragma unroll
      for (int r = 0; r < kSegKeys; ++r) {
        const int e = r * 1024 + tid;
        row_r[r] = e < n_here ? (int32_t)slots_io[in_base + e] : -1;
      }
#pragma unroll
      for (int r = 0; r < kSegKeys; ++r) slot_r[r] = (row_r[r] >= 0 && !failed) ? inverted[row_r[r]] : -1;
#pragma unroll
      for (int r = 0; r < kSegKeys; ++r) {
        const int e = r * 1024 + tid;
        if (e < n_here) slots_io[in_base + e] = (int64_t)slot_r[r];
      }
    }
#pragma unroll
    for (int r = 0; r < kSegKeys; ++r) {
      // lane-interleaved ownership: lookup e = r * 1024 + tid (coalesced loads; the order inside a row's group
      // follows the lookup order up to the atomic race between waves)
      const int e = r * 1024 + tid;
      key[r] = ~0ull;
      bkt[r] = kSegBuckets;
      if (e < n_here) {
        const int64_t row = ROWS ? (int64_t)slot_r[ROWS ? r : 0] : indices[in_base + e];
        const bool valid = (uint64_t)row < (uint64_t)num_rows;
        if (valid || SRC) {
          unsigned low = (unsigned)e;
          // lay.offsets == nullptr: one id per bag, in order (bag of lookup j = j) -- what every Criteo / Avazu
          // batch is; saves the two offset loads per lookup that find_bag needs to establish it (-12 us per window)
          if (SRC && !(CE_DBG(lay.debug) & 2))
            low = (unsigned)out_row(q, lay.offsets ? find_bag(q, sib * kSegLen + e) : sib * kSegLen + e);
          // SRC: an ignored lookup (slot -1 / out of range) keeps its output row under the row 0xffffffff -- the
          // backward skips it like padding (row >= num_rows), the key-driven forward writes its zero row; it goes to
          // the last bucket with the padding
          key[r] = ((unsigned long long)(valid ? (uint32_t)row : 0xffffffffu) << 32) | low;
          if (valid) bkt[r] = (int)(row & (kSegBuckets - 1));
        }
      }
      if (CE_DBG(lay.debug) & 8) {       // ablation: no LDS atomic
        place[r] = 0;
        continue;
      }
      // Places inside the bucket: one returning LDS atomic per lane.  (Round 2 matched the lanes of a wave that hold
      // the same bucket with 13 ballots and let a leader reserve for all of them: on the bench's slots the ballots
      // cost 13 us per window more than the bank conflicts they avoid.)  Only a wave whose 64 lanes ALL hold one
      // bucket -- a feature with one hot row -- reserves its 64 places with a single atomic.
      const int b0 = __builtin_amdgcn_readfirstlane(bkt[r]);
      if (__all(bkt[r] == b0)) {
        int first = 0;
        if (lane == 0) first = atomicAdd(&cnt[b0], 64);
        place[r] = __shfl(first, 0) + lane;
      } else {
        place[r] = atomicAdd(&cnt[bkt[r]], 1);
      }
    }
    __syncthreads();
    // exclusive scan of the 4097 counters (thread t owns 4t..4t+3; the ignored bucket follows everything)
    constexpr int kPer = kSegBuckets / 1024;
    int c4[kPer], sum = 0;
#pragma unroll
    for (int q = 0; q < kPer; ++q) { c4[q] = cnt[tid * kPer + q]; sum += c4[q]; }
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int pre = inc - sum;
    for (int k = 0; k < wv; ++k) pre += wsum[k];
    int total = 0;
    for (int k = 0; k < 16; ++k) total += wsum[k];
#pragma unroll
    for (int q = 0; q < kPer; ++q) { cnt[tid * kPer + q] = pre; pre += c4[q]; }
    if (tid == 0) cnt[kSegBuckets] = total;
    __syncthreads();
    if (!EXCL) {
      int pos[kSegKeys];
#pragma unroll
      for (int r = 0; r < kSegKeys; ++r) pos[r] = cnt[bkt[r]] + place[r];
      __syncthreads();                   // the counters are dead: their LDS becomes the key buffer
#pragma unroll
      for (int r = 0; r < kSegKeys; ++r) lk[pos[r]] = key[r];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < kSegKeys; ++r) keys_out[base + r * 1024 + tid] = lk[r * 1024 + tid];
      __syncthreads();
      continue;
    }
    // ---- EXCL: keys to their positions in LDS (the counters are dead once every thread has its positions)
    int pos[kSegKeys];
#pragma unroll
    for (int r = 0; r < kSegKeys; ++r) pos[r] = cnt[bkt[r]] + place[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSegKeys; ++r) lk[pos[r]] = key[r];
    __syncthreads();
    // thread t owns the buckets that START in positions [16 t, 16 t + 16).  A neighbour sorting a bucket that reaches
    // into this range only permutes keys of one bucket, so "is p a bucket start" (high words of p - 1 and p) does not
    // depend on how far it got.
    {
      const uint32_t* hi = (const uint32_t*)lk;          // hi[2 p + 1] = row of position p
      const int p0 = tid * kSegKeys;
      for (int p = p0; p < p0 + kSegKeys; ++p) {
        const uint32_t row = hi[2 * p + 1];
        if (row == 0xffffffffu) break;                   // ignored lookups are last
        const uint32_t b = row & (kSegBuckets - 1);
        if (p > 0 && (hi[2 * p - 1] & (kSegBuckets - 1)) == b && hi[2 * p - 1] != 0xffffffffu) continue;   // not a start
        int e = p + 1;
        while (e < kSegLen && e - p <= kExclRun && hi[2 * e + 1] != 0xffffffffu && (hi[2 * e + 1] & (kSegBuckets - 1)) == b) ++e;
        if (e - p > kExclRun) continue;                  // a hot bucket: left as it is, flushed with atomics
        for (int i = p + 1; i < e; ++i) {                // insertion sort by row (stable)
          const unsigned long long k = lk[i];
          int j = i;
          while (j > p && (uint32_t)(lk[j - 1] >> 32) > (uint32_t)(k >> 32)) { lk[j] = lk[j - 1]; --j; }
          lk[j] = k;
        }
        for (int a = p; a < e;) {                        // runs: flag the head of a run inside one 16-position block
          int z = a + 1;
          while (z < e && (uint32_t)(lk[z] >> 32) == (uint32_t)(lk[a] >> 32)) ++z;
          if ((a >> 4) == ((z - 1) >> 4)) lk[a] |= kExclFlag;
          a = z;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSegKeys; ++r) keys_out[base + r * 1024 + tid] = lk[r * 1024 + tid];
    // the segment's id range (ids of ignored lookups included: only ever makes the range wider)
    if (ids_minmax) {
      long long lo = INT64_MAX, hi2 = INT64_MIN;
      if (ids) {
#pragma unroll
        for (int r = 0; r < kSegKeys; ++r) {
          const int e = r * 1024 + tid;
          if (e < n_here) {
            const long long v = ids[in_base + e];
            lo = v < lo ? v : lo;
            hi2 = v > hi2 ? v : hi2;
          }
        }
      } else {
        lo = INT64_MIN; hi2 = INT64_MAX;                 // no ids: the range says "everything" (never disjoint)
      }
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const long long ol = __shfl_xor(lo, d), oh = __shfl_xor(hi2, d);
        lo = ol < lo ? ol : lo;
        hi2 = oh > hi2 ? oh : hi2;
      }
      if (lane == 0) { mm_s[0][wv] = lo; mm_s[1][wv] = hi2; }
      __syncthreads();
      if (tid == 0) {
        for (int k = 1; k < 16; ++k) {
          lo = mm_s[0][k] < lo ? mm_s[0][k] : lo;
          hi2 = mm_s[1][k] > hi2 ? mm_s[1][k] : hi2;
        }
        ids_minmax[2 * seg] = lo;
        ids_minmax[2 * seg + 1] = hi2;
      }
    }
    __syncthreads();
  }
}

template <typename VT, int NCH, typename KT, int R>
__global__ __launch_bounds__(256) void k_bag_bwd_tile(BagParams p) {
  using K = KeyOps<KT>;
  __shared__ KT keys[kBwdTile];
  __shared__ int bagl[kBwdTile];
  __shared__ float scl[kBwdTile];

  const int tid = threadIdx.x;
  const int G = 1 << p.g_log2;
  const int ngroups = 256 >> p.g_log2;
  const int grp = tid >> p.g_log2;
  const int gl = tid & (G - 1);
  const int rowlen = p.rowlen;
  const int dim = rowlen * (int)(sizeof(VT) / 4);
  const VT* __restrict__ GO = (const VT*)p.grad_out;
  // tile_len <= kBwdTile is chosen by the launcher so that the tile count is a multiple of the CU count
  // (425,984 lookups -> 512 tiles of 832: two per CU, instead of 416 tiles = 1 or 2 per CU).
  // presorted: the tiles are tile_len consecutive positions of the segment-padded key array (any alignment)
  const int tile_len = p.tile_len;
  const int64_t total = p.presorted ? ((p.nnz + kSegLen - 1) / kSegLen) * kSegLen : p.nnz;
  const int ntiles = (int)((total + tile_len - 1) / tile_len);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int j0 = tile * tile_len;
    const int nv = min(tile_len, (int)(total - j0));
    // ---- a. keys, bag and scale of every lookup of the tile (out-of-range rows become invalid keys);
    //         thread t owns lookups 4t..4t+3 of the tile
    const bool presorted = p.presorted != nullptr;
    KT kr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = tid * 4 + r;
      kr[r] = K::invalid();
      int j = j0 + i;                                          // the lookup behind position i
      uint32_t row = 0xffffffffu;
      if (presorted) {
        const unsigned long long sk = i < nv ? p.presorted[(int64_t)j0 + i] : ~0ull;
        j = -1;
        if (sk != ~0ull) {
          row = (uint32_t)(sk >> 32);
          j = (int)((((int64_t)j0 + i) / kSegLen) * kSegLen) + (int)(uint32_t)sk;
        }
      } else if (i < nv) {
        const int64_t r64 = p.indices[j];
        if ((uint64_t)r64 < (uint64_t)p.num_rows) row = (uint32_t)r64;
      } else {
        j = -1;
      }
      if (j >= 0) {
        const int bag = find_bag(p, j);
        float sc = p.alpha;
        if (p.psw) sc *= p.psw[j];
        if (p.mode == CE_MODE_MEAN) {
          const int len = bag_end(p, bag) - ld_off(p, bag);
          if (len > 1) sc = sc / (float)len;
        }
        bagl[i] = (int)out_row(p, bag);      // row of grad_out this lookup reads (one division per lookup, not per lane)
        scl[i] = sc;
        if (row != 0xffffffffu) kr[r] = K::make(row, i);
      }
    }
    if (CE_DBG(p.debug) == 3) { __syncthreads(); continue; }
    // ---- b. sort (row major, lookup minor) -> runs are in lookup order, invalid keys last
    //         (skipped when the cache op already sorted this tile: ce_bag_presort)
    if (presorted) {
#pragma unroll
      for (int r = 0; r < 4; ++r) keys[tid * 4 + r] = kr[r];
      __syncthreads();
    } else {
      tile_sort_1024<KT>(kr, keys, tid);
    }
    // ---- c. reduce: every lane group walks ONE contiguous, equal share of the tile's sorted positions (13 chunks
    // of 64 over 8 groups cost two rounds -- the same as 16; 104 positions each cost 6.5/8 of that), R gradient
    // rows in flight, folding equal rows in lookup order; one atomic row update per (row, share).
    if (CE_DBG(p.debug) == 4) continue;
    const int kChunk = (nv + ngroups - 1) / ngroups;
    const int nchunks = (nv + kChunk - 1) / kChunk;
    for (int ck = grp; ck < nchunks; ck += ngroups) {
      const int s0 = ck * kChunk, s1 = min(nv, s0 + kChunk);
      VT acc[NCH];
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = vzero<VT>();
      uint32_t cur = K::row(keys[s0]);
      for (int q = s0; q < s1; q += R) {
        VT v[R][NCH];
        float sc[R];
        uint32_t rw[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
          const KT key = (q + t < s1) ? keys[q + t] : K::invalid();
          const bool on = key != K::invalid();
          const int li = on ? K::idx(key) : 0;
          rw[t] = on ? K::row(key) : K::row_invalid();
          sc[t] = on ? scl[li] : 0.f;
          const int64_t orow = bagl[li];
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const int ch = gl + c * G;
            v[t][c] = vzero<VT>();
            if (on && ch < rowlen) v[t][c] = __builtin_nontemporal_load(&GO[orow * rowlen + ch]);
          }
        }
        // one wait for all R gathers (see k_bag_bwd_stream): keeps the flushes' atomics fire-and-forget
        __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0)
#pragma unroll
        for (int t = 0; t < R; ++t) {
          if (rw[t] != K::row_invalid()) {          // group-uniform
            if (rw[t] != cur) {
#pragma unroll
              for (int c = 0; c < NCH; ++c) {
                if (cur != K::row_invalid())
                  flush_chunk(p.dst + (int64_t)cur * dim, acc[c], gl, G, c, rowlen, CE_DBG(p.debug));
                acc[c] = vzero<VT>();
              }
              cur = rw[t];
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = acc[c] + v[t][c] * sc[t];
          }
        }
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (cur != K::row_invalid()) flush_chunk(p.dst + (int64_t)cur * dim, acc[c], gl, G, c, rowlen, CE_DBG(p.debug));
      }
    }
    __syncthreads();
  }
}

// Streaming form of the sorted scatter for keys that already carry everything a lookup needs (ce_bag_presort_window_src:
// key = row << 32 | row of grad_out to read; sum mode, no per-sample weights, so the scale is alpha for every lookup).
// No tile, no barrier, no bag search: every lane group owns ONE contiguous, equal share of the key array and walks it
// R keys at a time -- gathers the R gradient rows, folds equal rows and flushes a row's sum with the transposed atomics
// when the row changes.  The grid is a few workgroups per CU whatever the size, so there is no per-tile prologue
// (12 of the tile kernel's 74 us).
// The keys travel through LDS, 4 per lane at a time (a group-private slice, written and read by the same wave, so no
// barrier): a GLOBAL key load inside the loop would share the vmcnt counter with the gathers and the fire-and-forget
// atomics, which retire in order -- the wait for the next keys would then wait for the atomics just issued
// (measured: 85 us with register-prefetched keys, 49 us of it gather), and a key load between two gathers serialises
// them (97 us).
// EXCL (keys from ce_bag_presort_window_src_excl + the batch's segment id ranges): a key with kExclFlag set heads a run
// that holds every lookup of its row in its segment and lies inside one 16-position block, i.e. inside ONE lane
// group's share (shares are multiples of 16).  If, in addition, no id occurs in two segments of the batch -- checked
// here, by every workgroup, on the id ranges the presort recorded: pairwise disjoint [min, max] -- that lane group is
// the only writer of the row in this launch: the row's old value is loaded together with the gradient rows (one more
// load in flight, no exposed latency), the run is folded on top of it and the result goes back with ONE plain
// 512-byte store.  Everything else (long runs, rows of hot buckets, runs that straddle a block, batches whose
// segments share ids) keeps the transposed fire-and-forget atomics.  Why: the L2 atomic units retire ~1 float per
// clock and channel -- 6.3 M lane-ops per launch = 24 us that do not overlap with the gather (profiles/r03_probe_*).
template <typename VT, int NCH, int R, bool NTG, bool EXCL>
__global__ __launch_bounds__(256) void k_bag_bwd_stream(BagParams p, int64_t total, const long long* __restrict__ seg_ranges) {
  __shared__ unsigned long long lk[256 * (R > 4 ? R : 4)];     // ngroups * kc, worst case G = 1
  const int tid = threadIdx.x;
  const int G = 1 << p.g_log2;
  const int ngroups = 256 >> p.g_log2;
  const int grp = tid >> p.g_log2;
  const int gl = tid & (G - 1);
  const int rowlen = p.rowlen;
  const int dim = rowlen * (int)(sizeof(VT) / 4);
  const VT* __restrict__ GO = (const VT*)p.grad_out;
  const VT* __restrict__ WV = (const VT*)p.dst;
  const unsigned long long* __restrict__ keys = p.presorted;
  bool excl = false;
  if (EXCL) {
    // no id in two segments <= the segments' id ranges are pairwise disjoint (empty segments: min > max)
    const int nseg = (int)(total / kSegLen);
    int clash = nseg > 64 || seg_ranges == nullptr;
    if (!clash) {
      for (int pr = tid; pr < nseg * nseg; pr += 256) {
        const int a = pr / nseg, b2 = pr - a * nseg;
        if (a >= b2) continue;
        const long long alo = seg_ranges[2 * a], ahi = seg_ranges[2 * a + 1];
        const long long blo = seg_ranges[2 * b2], bhi = seg_ranges[2 * b2 + 1];
        if (alo <= ahi && blo <= bhi && !(ahi < blo || bhi < alo)) clash = 1;
      }
    }
    excl = __syncthreads_or(clash) == 0;
  }
  const int64_t all_groups = (int64_t)gridDim.x * ngroups;
  const int64_t round = R > 16 ? R : 16;        // shares are whole 16-position blocks (see EXCL)
  const int64_t share = ((total + all_groups - 1) / all_groups + round - 1) / round * round;
  const int64_t s0 = share_of(p, grp, ngroups) * share;
  const int64_t s1 = min(total, s0 + share);
  if (s0 >= s1) return;
  const float alpha = p.alpha;
  const int kc = 4 * G > R ? 4 * G : R;         // keys staged per refill (G is a power of two: a multiple of R)
  unsigned long long* mylk = lk + grp * kc;
  VT acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = vzero<VT>();
  uint32_t cur = 0xffffffffu;
  bool cur_excl = false;
  for (int64_t c0 = s0; c0 < s1; c0 += kc) {
    for (int k = gl; k < kc; k += G) mylk[k] = c0 + k < s1 ? keys[c0 + k] : ~0ull;
    const int64_t c1 = min(s1, c0 + kc);
    for (int64_t q = c0; q < c1; q += R) {
      VT v[R][NCH];
      VT w2[EXCL ? R : 1][NCH];
      uint32_t rw[R];
      uint32_t heads = 0;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const unsigned long long k = mylk[(int)(q - c0) + t];
        const bool on = k != ~0ull && (uint32_t)(k >> 32) < p.num_rows;
        rw[t] = on ? (uint32_t)(k >> 32) : 0xffffffffu;
        const uint32_t low = (uint32_t)k;
        const int64_t src = (int64_t)(low & ~kExclFlag);
        const bool head = EXCL && excl && on && (low & kExclFlag);
        if (head) heads |= 1u << t;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int ch = gl + c * G;
          v[t][c] = vzero<VT>();
          if (on && ch < rowlen) v[t][c] = NTG ? __builtin_nontemporal_load(&GO[src * rowlen + ch]) : GO[src * rowlen + ch];
          if (EXCL) {
            w2[t][c] = vzero<VT>();
            if (head && ch < rowlen) w2[t][c] = WV[(int64_t)rw[t] * rowlen + ch];
          }
        }
      }
      // ONE wait for all R gathers here: left to the compiler, the wait for v[t] lands after the flush of v[t-1]'s
      // row, where the vmcnt counter also holds the atomics just issued -- every flush would be synchronous
      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt/lgkmcnt untouched
#pragma unroll
      for (int t = 0; t < R; ++t) {
        if (rw[t] != 0xffffffffu) {            // group-uniform
          const bool head = EXCL && ((heads >> t) & 1);
          if (rw[t] != cur || head) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              if (cur != 0xffffffffu) {
                if (EXCL && cur_excl) {
                  const int ch = gl + c * G;
                  if (ch < rowlen) ((VT*)(p.dst + (int64_t)cur * dim))[ch] = acc[c];
                } else {
                  flush_chunk(p.dst + (int64_t)cur * dim, acc[c], gl, G, c, rowlen, CE_DBG(p.debug));
                }
              }
              acc[c] = EXCL ? w2[EXCL ? t : 0][c] : vzero<VT>();
            }
            cur = rw[t];
            cur_excl = head;
          }
#pragma unroll
          for (int c = 0; c < NCH; ++c) acc[c] = acc[c] + v[t][c] * alpha;
        }
      }
    }
  }
  if (cur != 0xffffffffu) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (EXCL && cur_excl) {
        const int ch = gl + c * G;
        if (ch < rowlen) ((VT*)(p.dst + (int64_t)cur * dim))[ch] = acc[c];
      } else {
        flush_chunk(p.dst + (int64_t)cur * dim, acc[c], gl, G, c, rowlen, CE_DBG(p.debug));
      }
    }
  }
}

// Forward from the window's source-row keys (one id per bag, mode sum, no per-sample weights: out[bag] = W[slot], a
// row copy per lookup).  The keys of a batch hold, per lookup, the cache row and the row of the [B, F, D] output it
// lands in, GROUPED BY ROW inside every 16384-lookup segment -- so a lane group that walks its share of the key array
// loads a cache row ONCE per run of equal rows and stores it to every output row of the run: ~50 k row loads per
// Criteo batch instead of 425,984 (the other 375 k were L2 / Infinity-Cache hits, but every one of them crossed the
// fabric: the gather-shaped forward ran at the speed of a COPY of its output, 52 us; this one is bound by the FILL of
// its output).  Same walk as k_bag_bwd_stream: contiguous equal shares, keys staged through a group-private LDS slice,
// R keys per step -- the loads of the step's run heads in flight together, one wait, then the stores.
// An ignored lookup (row 0xffffffff, see k_bag_presort_seg) gets a zero row; padding keys (~0) are skipped.
template <typename VT, int NCH, int R, int NTS>
__global__ __launch_bounds__(256, 4) void k_bag_fwd_keys(BagParams p, int64_t total) {      // 4 waves per SIMD: <= 128 VGPRs
  __shared__ unsigned long long lk[256 * (R > 4 ? R : 4)];
  const int tid = threadIdx.x;
  const int G = 1 << p.g_log2;
  const int ngroups = 256 >> p.g_log2;
  const int grp = tid >> p.g_log2;
  const int gl = tid & (G - 1);
  const int rowlen = p.rowlen;
  const VT* __restrict__ W = (const VT*)p.weight;
  VT* __restrict__ O = (VT*)p.dst;
  const unsigned long long* __restrict__ keys = p.presorted;
  const int64_t all_groups = (int64_t)gridDim.x * ngroups;
  const int64_t round = R > 16 ? R : 16;
  const int64_t share = ((total + all_groups - 1) / all_groups + round - 1) / round * round;
  const int64_t s0 = share_of(p, grp, ngroups) * share;
  const int64_t s1 = min(total, s0 + share);
  if (s0 >= s1) return;
  const int kc = 4 * G > R ? 4 * G : R;
  unsigned long long* mylk = lk + grp * kc;
  VT prev[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) prev[c] = vzero<VT>();
  uint32_t prev_row = 0xffffffffu;          // "ignored": a zero row, which is what prev holds
  for (int64_t c0 = s0; c0 < s1; c0 += kc) {
    for (int k = gl; k < kc; k += G) mylk[k] = c0 + k < s1 ? keys[c0 + k] : ~0ull;
    const int64_t c1 = min(s1, c0 + kc);
    for (int64_t q = c0; q < c1; q += R) {
      VT v[R][NCH];
      uint32_t heads = 0;
      uint32_t last = prev_row;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        const unsigned long long k = mylk[(int)(q - c0) + t];
        const bool on = k != ~0ull;
        const uint32_t rw = (uint32_t)(k >> 32);
        const bool head = on && rw != last;
        if (on) last = rw;
        if (head) heads |= 1u << t;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int ch = gl + c * G;
          v[t][c] = vzero<VT>();
          if (head && ch < rowlen && rw < p.num_rows) v[t][c] = W[(int64_t)rw * rowlen + ch];
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): every head of the step has arrived
#pragma unroll
      for (int t = 0; t < R; ++t) {
        // (the output row comes out of LDS again rather than out of 16 more registers: the kernel sits at the
        // 128-VGPR line that separates 4 waves per SIMD from 3)
        const unsigned long long k = mylk[(int)(q - c0) + t];
        if (k == ~0ull) continue;              // padding (group-uniform)
        if ((heads >> t) & 1) {
#pragma unroll
          for (int c = 0; c < NCH; ++c) prev[c] = v[t][c];       // a new run: its row stays in `prev` until the next head
        }
        // (owner-exclusive keys -- presort_window(ids=...) -- carry kExclFlag in bit 31 of the low word: not part of the row)
        const int64_t orow = (int64_t)((uint32_t)k & ~kExclFlag);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int ch = gl + c * G;
          if (ch < rowlen) store_out<NTS>(&O[orow * rowlen + ch], prev[c]);
        }
      }
      prev_row = last;
    }
  }
}

// dst[index[i]] += alpha * src[i] for whole rows (the owner-side SGD of the row-wise exchange: `src` holds one
// already-folded gradient row per requested row, so there is nothing to sort or fold; rows requested by several
// peers repeat, hence atomics).  Lane group per row, U rows in flight, lane-block transposed atomics as above.
template <typename VT, int NCH>
__global__ __launch_bounds__(256) void k_rows_axpy(float* __restrict__ dst, uint32_t num_rows,
                                                   const int64_t* __restrict__ index, int64_t n,
                                                   const VT* __restrict__ src, int rowlen, int g_log2, int dim,
                                                   float alpha) {
  constexpr int U = (NCH == 1) ? 8 : (NCH == 2 ? 4 : 2);
  const int lane = threadIdx.x & 63;
  const int G = 1 << g_log2;
  const int gpw = 64 >> g_log2;
  const int grp = lane >> g_log2;
  const int gl = lane & (G - 1);
  const int wpb = blockDim.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * wpb;
  const int64_t per_wave = (int64_t)gpw * U;
  for (int64_t r0 = wave * per_wave; r0 < n; r0 += nwaves * per_wave) {
    VT v[U][NCH];
    int64_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = r0 + u * gpw + grp;
      row[u] = -1;
      if (i < n) {
        row[u] = index[i];
        if ((unsigned long long)row[u] >= (unsigned long long)num_rows) row[u] = -1;
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = gl + c * G;
        v[u][c] = vzero<VT>();
        if (row[u] >= 0 && ch < rowlen) v[u][c] = __builtin_nontemporal_load(&src[i * rowlen + ch]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (row[u] < 0) continue;
#pragma unroll
      for (int c = 0; c < NCH; ++c) flush_chunk(dst + row[u] * dim, v[u][c] * alpha, gl, G, c, rowlen, 0);
    }
  }
}

static int fill_params(BagParams& p, int32_t dim, const int64_t* indices, int64_t nnz, const void* offsets,
                       int32_t off64, int64_t num_bags, int32_t include_last, const float* psw, int32_t mode,
                       int64_t hookF, bool* vec, int* nch, const void* a0, const void* a1, const void* a2) {
  CE_REQUIRE(dim > 0, CE_ERR_INVALID, "embedding dim must be positive");
  CE_REQUIRE(num_bags >= 0 && nnz >= 0, CE_ERR_INVALID, "negative sizes");
  CE_REQUIRE(num_bags < (int64_t)INT32_MAX - 64 && nnz < (int64_t)INT32_MAX, CE_ERR_UNSUPPORTED,
             "more than 2^31 bags / lookups in one launch");
  CE_REQUIRE(mode == CE_MODE_SUM || mode == CE_MODE_MEAN, CE_ERR_UNSUPPORTED, "mode must be sum or mean");
  CE_REQUIRE(!(mode == CE_MODE_MEAN && psw), CE_ERR_INVALID, "per_sample_weights needs mode='sum'");
  CE_REQUIRE(hookF >= 0 && (hookF == 0 || num_bags % hookF == 0), CE_ERR_INVALID,
             "hook_features must divide num_bags");
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  *vec = (dim % 4 == 0) && al16(a0) && al16(a1) && al16(a2);
  const int rowlen = *vec ? dim / 4 : dim;
  int g = 1, gl2 = 0;
  while (g < rowlen && g < 64) { g <<= 1; ++gl2; }
  const int need = (rowlen + g - 1) / g;
  int n = 1;
  while (n < need) n <<= 1;
  CE_REQUIRE(n <= 4, CE_ERR_UNSUPPORTED, "embedding dim %d too large for this build", dim);
  *nch = n;
  p.indices = indices;
  p.offsets = offsets;
  p.psw = psw;
  p.nnz = nnz;
  p.num_bags = (int32_t)num_bags;
  p.rowlen = rowlen;
  p.g_log2 = gl2;
  p.off64 = off64;
  p.include_last = include_last;
  p.mode = mode;
  p.hookF = (int32_t)hookF;
  p.hookB = hookF ? (int32_t)(num_bags / hookF) : 0;
  p.alpha = 1.f;
  p.num_rows = 0xffffffffu;
  p.policy = kBagPolicy;
  return CE_OK;
}

static int bag_grid(int64_t num_bags) {
  int64_t tiles = cdiv(num_bags, 64);
  return grid_for(tiles, 4);
}

// grad accumulation / fused SGD by target row: grouped segments (ce_bag_presort*) are walked as they are, otherwise
// the kernel sorts 1024-lookup tiles itself
static int launch_bwd_scatter(const BagParams& p, bool vec, int nch, hipStream_t s) {
  BagParams q = p;
#ifdef CE_ABLATIONS
  { const char* dbg = getenv("CE_BWD_DEBUG"); q.debug = dbg ? atoi(dbg) : 0; }
#endif
  // sorted keys: a tile count that is a multiple of the CU count (425,984 keys -> 512 tiles of 832, two per CU,
  // instead of 416 tiles = one or two per CU): 80 -> 72.5 us.  More, smaller tiles lose to the per-tile prologue
  // (3/CU 77 us, 4/CU 85 us, 8/CU 94 us), and the self-sorting path does not gain (92 us either way).
  q.tile_len = kBwdTile;
  if (p.presorted) {
    constexpr int per_cu = 2;
    const int64_t total = cdiv(p.nnz, kSegLen) * kSegLen;
    if (per_cu > 0 && total > (int64_t)kNumCU * 256)
      q.tile_len = (int)std::min<int64_t>(kBwdTile, (cdiv(total, (int64_t)kNumCU * per_cu) + 15) & ~15ll);
  }
  const int ntiles = (int)cdiv(p.presorted ? cdiv(p.nnz, kSegLen) * kSegLen : p.nnz, q.tile_len);
  dim3 grid(std::min(ntiles, kMaxBlocks)), block(256);
  const bool k32 = q.num_rows <= (1u << 22) - 2;
#define CE_BWT(VT, N, R)                                                                              \
  do {                                                                                                \
    if (k32) hipLaunchKernelGGL((k_bag_bwd_tile<VT, N, uint32_t, R>), grid, block, 0, s, q);          \
    else hipLaunchKernelGGL((k_bag_bwd_tile<VT, N, unsigned long long, R>), grid, block, 0, s, q);    \
  } while (0)
  if (vec) {
    if (nch == 1) CE_BWT(f32x4, 1, 16);
    else if (nch == 2) CE_BWT(f32x4, 2, 4); else CE_BWT(f32x4, 4, 2);
  } else {
    if (nch == 1) CE_BWT(float, 1, 8); else if (nch == 2) CE_BWT(float, 2, 4); else CE_BWT(float, 4, 2);
  }
#undef CE_BWT
  CE_LAUNCH_CHECK();
  return CE_OK;
}

// per-lookup gradient rows (COO values of the sparse=True backward)
static int launch_bwd_rows(const BagParams& p, bool vec, int nch, hipStream_t s) {
  dim3 grid(bag_grid(p.num_bags)), block(256);
#define CE_BWD(VT, N) hipLaunchKernelGGL((k_bag_bwd_rows<VT, N>), grid, block, 0, s, p)
  if (vec) {
    if (nch == 1) CE_BWD(f32x4, 1); else if (nch == 2) CE_BWD(f32x4, 2); else CE_BWD(f32x4, 4);
  } else {
    if (nch == 1) CE_BWD(float, 1); else if (nch == 2) CE_BWD(float, 2); else CE_BWD(float, 4);
  }
#undef CE_BWD
  CE_LAUNCH_CHECK();
  return CE_OK;
}

}  // namespace ce

using namespace ce;

extern "C" int ce_bag_forward(const float* weight, int64_t num_rows, int32_t dim, const int64_t* indices,
                              int64_t nnz, const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                              int32_t include_last_offset, const float* per_sample_weights, int32_t mode,
                              int64_t hook_features, float* out, ce_stream_t stream) {
  if (num_bags == 0) return CE_OK;
  CE_REQUIRE(weight && out && (indices || nnz == 0), CE_ERR_INVALID, "null pointer");
  CE_REQUIRE(offsets || num_bags == nnz, CE_ERR_INVALID,
             "offsets == NULL states one id per bag (offsets = arange): num_bags must equal nnz");
  BagParams p{};
  bool vec;
  int nch;
  int rc = fill_params(p, dim, indices, nnz, offsets, offsets_are_i64, num_bags, include_last_offset,
                       per_sample_weights, mode, hook_features, &vec, &nch, weight, out, nullptr);
  if (rc) return rc;
  p.weight = weight;
  p.dst = out;
  CE_REQUIRE(num_rows > 0 && num_rows < (int64_t)INT32_MAX, CE_ERR_INVALID, "num_rows out of range");
  p.num_rows = (uint32_t)num_rows;
  dim3 grid(bag_grid(num_bags)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool stage = nnz != num_bags;      // multi-id bags possible
#define CE_FWD(VT, N, U)                                                                          \
  do {                                                                                            \
    if (stage) hipLaunchKernelGGL((k_bag_fwd<VT, N, true, U, 1>), grid, block, 0, s, p);          \
    else hipLaunchKernelGGL((k_bag_fwd<VT, N, false, U, 1>), grid, block, 0, s, p);               \
  } while (0)
  if (vec) {
    if (nch == 1) CE_FWD(f32x4, 1, 16);
    else if (nch == 2) CE_FWD(f32x4, 2, 4); else CE_FWD(f32x4, 4, 2);
  } else {
    if (nch == 1) CE_FWD(float, 1, 8); else if (nch == 2) CE_FWD(float, 2, 4); else CE_FWD(float, 4, 2);
  }
#undef CE_FWD
  CE_LAUNCH_CHECK();
  return CE_OK;
}

static int backward_dense_impl(float* grad_weight, int64_t num_rows, int32_t dim, const int64_t* indices, int64_t nnz,
                               const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                               int32_t include_last_offset, const float* per_sample_weights, int32_t mode,
                               int64_t hook_features, const float* grad_out, const unsigned long long* presorted,
                               ce_stream_t stream) {
  if (num_bags == 0 || nnz == 0) return CE_OK;
  CE_REQUIRE(grad_weight && grad_out && offsets && indices, CE_ERR_INVALID, "null pointer");
  BagParams p{};
  bool vec;
  int nch;
  int rc = fill_params(p, dim, indices, nnz, offsets, offsets_are_i64, num_bags, include_last_offset,
                       per_sample_weights, mode, hook_features, &vec, &nch, grad_weight, grad_out, nullptr);
  if (rc) return rc;
  p.dst = grad_weight;
  p.grad_out = grad_out;
  p.alpha = 1.f;
  CE_REQUIRE(num_rows > 0 && num_rows < (int64_t)INT32_MAX, CE_ERR_INVALID, "num_rows out of range");
  p.num_rows = (uint32_t)num_rows;
  p.presorted = presorted;
  return launch_bwd_scatter(p, vec, nch, (hipStream_t)stream);
}

extern "C" int ce_bag_backward_dense(float* grad_weight, int64_t num_rows, int32_t dim, const int64_t* indices,
                                     int64_t nnz, const void* offsets, int32_t offsets_are_i64,
                                     int64_t num_bags, int32_t include_last_offset,
                                     const float* per_sample_weights, int32_t mode, int64_t hook_features,
                                     const float* grad_out, ce_stream_t stream) {
  return backward_dense_impl(grad_weight, num_rows, dim, indices, nnz, offsets, offsets_are_i64, num_bags,
                             include_last_offset, per_sample_weights, mode, hook_features, grad_out, nullptr, stream);
}

extern "C" int ce_bag_backward_dense_presorted(float* grad_weight, int64_t num_rows, int32_t dim,
                                               const int64_t* indices, int64_t nnz, const void* offsets,
                                               int32_t offsets_are_i64, int64_t num_bags,
                                               int32_t include_last_offset, const float* per_sample_weights,
                                               int32_t mode, int64_t hook_features, const float* grad_out,
                                               const uint64_t* presorted_keys, ce_stream_t stream) {
  CE_REQUIRE(presorted_keys, CE_ERR_INVALID, "null presorted_keys");
  return backward_dense_impl(grad_weight, num_rows, dim, indices, nnz, offsets, offsets_are_i64, num_bags,
                             include_last_offset, per_sample_weights, mode, hook_features, grad_out,
                             (const unsigned long long*)presorted_keys, stream);
}

static int backward_sgd_impl(float* weight, int64_t num_rows, int32_t dim, const int64_t* indices, int64_t nnz,
                             const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                             int32_t include_last_offset, const float* per_sample_weights, int32_t mode,
                             int64_t hook_features, const float* grad_out, float lr,
                             const unsigned long long* presorted, ce_stream_t stream) {
  if (num_bags == 0 || nnz == 0) return CE_OK;
  CE_REQUIRE(weight && grad_out && offsets && indices, CE_ERR_INVALID, "null pointer");
  BagParams p{};
  bool vec;
  int nch;
  int rc = fill_params(p, dim, indices, nnz, offsets, offsets_are_i64, num_bags, include_last_offset,
                       per_sample_weights, mode, hook_features, &vec, &nch, weight, grad_out, nullptr);
  if (rc) return rc;
  p.dst = weight;
  p.grad_out = grad_out;
  p.alpha = -lr;
  CE_REQUIRE(num_rows > 0 && num_rows < (int64_t)INT32_MAX, CE_ERR_INVALID, "num_rows out of range");
  p.num_rows = (uint32_t)num_rows;
  p.presorted = presorted;
  return launch_bwd_scatter(p, vec, nch, (hipStream_t)stream);
}

extern "C" int ce_bag_backward_sgd(float* weight, int64_t num_rows, int32_t dim, const int64_t* indices,
                                   int64_t nnz, const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                                   int32_t include_last_offset, const float* per_sample_weights, int32_t mode,
                                   int64_t hook_features, const float* grad_out, float lr, ce_stream_t stream) {
  return backward_sgd_impl(weight, num_rows, dim, indices, nnz, offsets, offsets_are_i64, num_bags,
                           include_last_offset, per_sample_weights, mode, hook_features, grad_out, lr, nullptr, stream);
}

extern "C" int ce_bag_backward_sgd_presorted(float* weight, int64_t num_rows, int32_t dim, const int64_t* indices,
                                             int64_t nnz, const void* offsets, int32_t offsets_are_i64,
                                             int64_t num_bags, int32_t include_last_offset,
                                             const float* per_sample_weights, int32_t mode, int64_t hook_features,
                                             const float* grad_out, float lr, const uint64_t* presorted_keys,
                                             ce_stream_t stream) {
  CE_REQUIRE(presorted_keys, CE_ERR_INVALID, "null presorted_keys");
  return backward_sgd_impl(weight, num_rows, dim, indices, nnz, offsets, offsets_are_i64, num_bags,
                           include_last_offset, per_sample_weights, mode, hook_features, grad_out, lr,
                           (const unsigned long long*)presorted_keys, stream);
}

// keys = row << 32 | grad_out row (ce_bag_presort_window_src); alpha * grad_out rows are folded into dst.
// seg_ranges (fused SGD only): the batch's segment id ranges from ce_bag_presort_window_src_excl -> owner-exclusive
// rows are updated with plain read-modify-writes (k_bag_bwd_stream<EXCL>)
static int launch_bwd_stream(float* dst, int64_t num_rows, int32_t dim, int64_t nnz, const float* grad_out,
                             float alpha, const unsigned long long* keys, const int64_t* seg_ranges, hipStream_t s) {
  if (nnz == 0) return CE_OK;
  CE_REQUIRE(dst && grad_out && keys, CE_ERR_INVALID, "null pointer");
  CE_REQUIRE(num_rows > 0 && num_rows < (int64_t)INT32_MAX, CE_ERR_INVALID, "num_rows out of range");
  BagParams p{};
  bool vec;
  int nch;
  int rc = fill_params(p, dim, nullptr, nnz, nullptr, 0, 0, 1, nullptr, CE_MODE_SUM, 0, &vec, &nch, dst, grad_out,
                       nullptr);
  if (rc) return rc;
  p.dst = dst;
  p.grad_out = grad_out;
  p.alpha = alpha;
  p.num_rows = (uint32_t)num_rows;
  p.presorted = keys;
#ifdef CE_ABLATIONS
  { const char* dbg = getenv("CE_BWD_DEBUG"); p.debug = dbg ? atoi(dbg) : 0; }
#endif
  p.interleave = 1;
  const int64_t total = cdiv(nnz, kSegLen) * kSegLen;
  // 6 workgroups per CU, i.e. 12288 shares of ~35 keys at the bench shape.  Alone on the GPU the share size matters
  // little (2 / 3 / 4 / 5 / 6 / 8 / 12 per CU: 55.5 / 54 / 58.5 / 56 / 53.7 / 55.8 / 58.6 us); beside the cache op's
  // kernels it does -- a workgroup that gets its CU late holds the whole launch back by its share: 77 / 74 / 70.5 /
  // 67.5 / 67 / 68 / 70.5 us (profiles/r04_late/grid_sweep_*.txt: forward 16/CU + backward 6 or 8/CU take the bench
  // line from 2.91 to 3.02-3.03 G over 10 runs each; 6 is the one that is also fastest alone)
  // (re-checked in the one-stream arrangement: profiles/r05_ab_grid_interleaved.txt)
  constexpr int per_cu = 6;
  const int ngroups = 256 >> p.g_log2;
  // small inputs: one share of >= 16 keys per lane group
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)kNumCU * per_cu, cdiv(total, (int64_t)ngroups * 16)));
  dim3 g(grid), b(256);
  const long long* rg = (const long long*)seg_ranges;
  const bool excl = seg_ranges != nullptr && vec && nch == 1;
  if (vec) {
    if (nch == 1 && excl) hipLaunchKernelGGL((k_bag_bwd_stream<f32x4, 1, 16, true, true>), g, b, 0, s, p, total, rg);
    else if (nch == 1) hipLaunchKernelGGL((k_bag_bwd_stream<f32x4, 1, 16, true, false>), g, b, 0, s, p, total, rg);
    else if (nch == 2) hipLaunchKernelGGL((k_bag_bwd_stream<f32x4, 2, 8, true, false>), g, b, 0, s, p, total, rg);
    else hipLaunchKernelGGL((k_bag_bwd_stream<f32x4, 4, 4, true, false>), g, b, 0, s, p, total, rg);
  } else {
    if (nch == 1) hipLaunchKernelGGL((k_bag_bwd_stream<float, 1, 16, true, false>), g, b, 0, s, p, total, rg);
    else if (nch == 2) hipLaunchKernelGGL((k_bag_bwd_stream<float, 2, 8, true, false>), g, b, 0, s, p, total, rg);
    else hipLaunchKernelGGL((k_bag_bwd_stream<float, 4, 4, true, false>), g, b, 0, s, p, total, rg);
  }
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int ce_bag_backward_sgd_presorted_src(float* weight, int64_t num_rows, int32_t dim, int64_t nnz,
                                                 const float* grad_out, float lr, const uint64_t* src_keys,
                                                 ce_stream_t stream) {
  return launch_bwd_stream(weight, num_rows, dim, nnz, grad_out, -lr, (const unsigned long long*)src_keys, nullptr,
                           (hipStream_t)stream);
}

extern "C" int ce_bag_backward_sgd_presorted_src_excl(float* weight, int64_t num_rows, int32_t dim, int64_t nnz,
                                                      const float* grad_out, float lr, const uint64_t* src_keys,
                                                      const int64_t* seg_id_ranges, ce_stream_t stream) {
  return launch_bwd_stream(weight, num_rows, dim, nnz, grad_out, -lr, (const unsigned long long*)src_keys,
                           seg_id_ranges, (hipStream_t)stream);
}

extern "C" int ce_bag_backward_dense_presorted_src(float* grad_weight, int64_t num_rows, int32_t dim, int64_t nnz,
                                                   const float* grad_out, const uint64_t* src_keys,
                                                   ce_stream_t stream) {
  return launch_bwd_stream(grad_weight, num_rows, dim, nnz, grad_out, 1.f, (const unsigned long long*)src_keys, nullptr,
                           (hipStream_t)stream);
}

extern "C" int ce_bag_forward_src_keys(const float* weight, int64_t num_rows, int32_t dim, int64_t nnz,
                                       const uint64_t* src_keys, float* out, ce_stream_t stream) {
  if (nnz == 0) return CE_OK;
  CE_REQUIRE(weight && src_keys && out, CE_ERR_INVALID, "null pointer");
  CE_REQUIRE(num_rows > 0 && num_rows < (int64_t)INT32_MAX, CE_ERR_INVALID, "num_rows out of range");
  BagParams p{};
  bool vec;
  int nch;
  int rc = fill_params(p, dim, nullptr, nnz, nullptr, 0, 0, 1, nullptr, CE_MODE_SUM, 0, &vec, &nch, weight, out, nullptr);
  if (rc) return rc;
  p.weight = weight;
  p.dst = out;
  p.num_rows = (uint32_t)num_rows;
  p.presorted = (const unsigned long long*)src_keys;
  p.interleave = 0;
  const int64_t total = cdiv(nnz, kSegLen) * kSegLen;
  // 16 workgroups per CU: more than fit at once (97 VGPRs: 5), so the dispatcher hands the shares out as CUs free up
  // (measured at the bench shape beside the cache op: 4/CU 72-74 us per launch, 8/CU 66-70, 16/CU 63-64, 32/CU 62-65;
  // alone: 8/CU 48, 16/CU 46.7, 5/CU -- exactly resident -- 43 but 66 beside the cache op)
  constexpr int per_cu = 16;
  const int ngroups = 256 >> p.g_log2;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)kNumCU * per_cu, cdiv(total, (int64_t)ngroups * 16)));
  dim3 g(grid), b(256);
  hipStream_t s = (hipStream_t)stream;
#define CE_FWK(VT, N, R) hipLaunchKernelGGL((k_bag_fwd_keys<VT, N, R, 1>), g, b, 0, s, p, total)
  if (vec) {
    if (nch == 1) CE_FWK(f32x4, 1, 16);
    else if (nch == 2) CE_FWK(f32x4, 2, 8); else CE_FWK(f32x4, 4, 4);
  } else {
    if (nch == 1) CE_FWK(float, 1, 16); else if (nch == 2) CE_FWK(float, 2, 8); else CE_FWK(float, 4, 4);
  }
#undef CE_FWK
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int64_t ce_bag_presort_len(int64_t nnz) { return nnz <= 0 ? 0 : cdiv(nnz, kSegLen) * kSegLen; }

static int presort_window_impl(const int64_t* indices, int64_t nnz_per_batch, int64_t n_batches, int64_t num_rows,
                               uint64_t* keys_out, const BagParams* lay, int64_t off_stride, const int64_t* ids,
                               int64_t* ids_minmax, ce_stream_t stream) {
  if (nnz_per_batch == 0 || n_batches == 0) return CE_OK;
  CE_REQUIRE(indices && keys_out && nnz_per_batch > 0 && n_batches > 0, CE_ERR_INVALID, "bad arguments");
  CE_REQUIRE(nnz_per_batch < (int64_t)INT32_MAX - kSegLen, CE_ERR_INVALID, "batch too large");
  CE_REQUIRE(num_rows > 0 && num_rows < (int64_t)INT32_MAX, CE_ERR_INVALID, "num_rows out of range");
  const int64_t spb = cdiv(nnz_per_batch, kSegLen);
  const int64_t nseg = spb * n_batches;
  CE_REQUIRE(spb < (int64_t)INT32_MAX && nseg < (int64_t)INT32_MAX, CE_ERR_INVALID, "too many segments");
  const dim3 grid((unsigned)std::min<int64_t>(nseg, kMaxBlocks)), block(1024);
  int64_t* const no_io = nullptr;
  const int32_t* const no_inv = nullptr;
  const int* const no_st = nullptr;
  if (lay && ids_minmax)
    hipLaunchKernelGGL((k_bag_presort_seg<true, true, false>), grid, block, 0, (hipStream_t)stream, indices, nnz_per_batch,
                       (int32_t)spb, nseg, (uint32_t)num_rows, (unsigned long long*)keys_out, *lay, off_stride, ids,
                       ids_minmax, no_io, no_inv, no_st);
  else if (lay)
    hipLaunchKernelGGL((k_bag_presort_seg<true, false, false>), grid, block, 0, (hipStream_t)stream, indices, nnz_per_batch,
                       (int32_t)spb, nseg, (uint32_t)num_rows, (unsigned long long*)keys_out, *lay, off_stride,
                       (const int64_t*)nullptr, (int64_t*)nullptr, no_io, no_inv, no_st);
  else
    hipLaunchKernelGGL((k_bag_presort_seg<false, false, false>), grid, block, 0, (hipStream_t)stream, indices, nnz_per_batch,
                       (int32_t)spb, nseg, (uint32_t)num_rows, (unsigned long long*)keys_out, BagParams{}, 0ll,
                       (const int64_t*)nullptr, (int64_t*)nullptr, no_io, no_inv, no_st);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

namespace ce {
int presort_window_from_rows(int64_t* slots_io, int64_t nnz_per_batch, int64_t n_batches, int64_t num_rows,
                             const int32_t* inverted, const int* status, int32_t src_keys, const void* offsets,
                             int32_t offsets_are_i64, int64_t offsets_batch_stride, int64_t num_bags,
                             int32_t include_last_offset, int64_t hook_features, uint64_t* keys_out, hipStream_t stream) {
  if (nnz_per_batch == 0 || n_batches == 0) return CE_OK;
  CE_REQUIRE(slots_io && keys_out && inverted && status && nnz_per_batch > 0 && n_batches > 0, CE_ERR_INVALID, "bad arguments");
  CE_REQUIRE(nnz_per_batch < (int64_t)INT32_MAX - kSegLen, CE_ERR_INVALID, "batch too large");
  CE_REQUIRE(num_rows > 0 && num_rows < (int64_t)INT32_MAX, CE_ERR_INVALID, "num_rows out of range");
  const int64_t spb = cdiv(nnz_per_batch, kSegLen);
  const int64_t nseg = spb * n_batches;
  CE_REQUIRE(spb < (int64_t)INT32_MAX && nseg < (int64_t)INT32_MAX, CE_ERR_INVALID, "too many segments");
  const dim3 grid((unsigned)std::min<int64_t>(nseg, kMaxBlocks)), block(1024);
  if (src_keys) {
    CE_REQUIRE(num_bags > 0 && offsets_batch_stride >= 0, CE_ERR_INVALID, "bad offsets");
    CE_REQUIRE(offsets || num_bags == nnz_per_batch, CE_ERR_INVALID,
               "offsets may only be NULL for the one-id-per-bag layout (num_bags == nnz_per_batch)");
    BagParams lay{};
    bool vec;
    int nch;
    int rc = fill_params(lay, 4, nullptr, nnz_per_batch, offsets, offsets_are_i64, num_bags, include_last_offset, nullptr,
                         CE_MODE_SUM, hook_features, &vec, &nch, nullptr, nullptr, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL((k_bag_presort_seg<true, false, true>), grid, block, 0, stream, (const int64_t*)nullptr,
                       nnz_per_batch, (int32_t)spb, nseg, (uint32_t)num_rows, (unsigned long long*)keys_out, lay,
                       offsets_batch_stride, (const int64_t*)nullptr, (int64_t*)nullptr, slots_io, inverted, status);
  } else {
    hipLaunchKernelGGL((k_bag_presort_seg<false, false, true>), grid, block, 0, stream, (const int64_t*)nullptr,
                       nnz_per_batch, (int32_t)spb, nseg, (uint32_t)num_rows, (unsigned long long*)keys_out, BagParams{},
                       0ll, (const int64_t*)nullptr, (int64_t*)nullptr, slots_io, inverted, status);
  }
  CE_LAUNCH_CHECK();
  return CE_OK;
}
}  // namespace ce

extern "C" int ce_bag_presort_window(const int64_t* indices, int64_t nnz_per_batch, int64_t n_batches,
                                     int64_t num_rows, uint64_t* keys_out, ce_stream_t stream) {
  return presort_window_impl(indices, nnz_per_batch, n_batches, num_rows, keys_out, nullptr, 0, nullptr, nullptr, stream);
}

static int presort_window_src_impl(const int64_t* indices, int64_t nnz_per_batch, int64_t n_batches, int64_t num_rows,
                                   const void* offsets, int32_t offsets_are_i64, int64_t offsets_batch_stride,
                                   int64_t num_bags, int32_t include_last_offset, int64_t hook_features,
                                   const int64_t* ids, uint64_t* keys_out, int64_t* seg_id_ranges, ce_stream_t stream) {
  if (nnz_per_batch == 0 || n_batches == 0) return CE_OK;
  CE_REQUIRE(num_bags > 0 && offsets_batch_stride >= 0, CE_ERR_INVALID, "bad offsets");
  CE_REQUIRE(offsets || num_bags == nnz_per_batch, CE_ERR_INVALID,
             "offsets may only be NULL for the one-id-per-bag layout (num_bags == nnz_per_batch)");
  BagParams lay{};
  bool vec;
  int nch;
  int rc = fill_params(lay, 4, indices, nnz_per_batch, offsets, offsets_are_i64, num_bags, include_last_offset,
                       nullptr, CE_MODE_SUM, hook_features, &vec, &nch, nullptr, nullptr, nullptr);
  if (rc) return rc;
#ifdef CE_ABLATIONS
  { const char* dbg = getenv("CE_PRESORT_DEBUG"); lay.debug = dbg ? atoi(dbg) : 0; }
#endif
  return presort_window_impl(indices, nnz_per_batch, n_batches, num_rows, keys_out, &lay, offsets_batch_stride, ids,
                             seg_id_ranges, stream);
}

extern "C" int ce_bag_presort_window_src(const int64_t* indices, int64_t nnz_per_batch, int64_t n_batches,
                                         int64_t num_rows, const void* offsets, int32_t offsets_are_i64,
                                         int64_t offsets_batch_stride, int64_t num_bags, int32_t include_last_offset,
                                         int64_t hook_features, uint64_t* keys_out, ce_stream_t stream) {
  return presort_window_src_impl(indices, nnz_per_batch, n_batches, num_rows, offsets, offsets_are_i64,
                                 offsets_batch_stride, num_bags, include_last_offset, hook_features, nullptr, keys_out,
                                 nullptr, stream);
}

extern "C" int ce_bag_presort_window_src_excl(const int64_t* indices, int64_t nnz_per_batch, int64_t n_batches,
                                              int64_t num_rows, const void* offsets, int32_t offsets_are_i64,
                                              int64_t offsets_batch_stride, int64_t num_bags,
                                              int32_t include_last_offset, int64_t hook_features, const int64_t* ids,
                                              uint64_t* keys_out, int64_t* seg_id_ranges, ce_stream_t stream) {
  CE_REQUIRE(seg_id_ranges, CE_ERR_INVALID, "null seg_id_ranges");
  return presort_window_src_impl(indices, nnz_per_batch, n_batches, num_rows, offsets, offsets_are_i64,
                                 offsets_batch_stride, num_bags, include_last_offset, hook_features, ids, keys_out,
                                 seg_id_ranges, stream);
}

extern "C" int ce_bag_presort(const int64_t* indices, int64_t nnz, int64_t num_rows, uint64_t* keys_out,
                              ce_stream_t stream) {
  return ce_bag_presort_window(indices, nnz, 1, num_rows, keys_out, stream);
}

extern "C" int ce_rows_axpy(float* weight, int64_t num_rows, int32_t dim, const int64_t* index, int64_t n,
                            const float* src_rows, float alpha, ce_stream_t stream) {
  if (n == 0) return CE_OK;
  CE_REQUIRE(weight && index && src_rows, CE_ERR_INVALID, "null pointer");
  CE_REQUIRE(dim > 0 && n > 0 && n < (int64_t)INT32_MAX, CE_ERR_INVALID, "bad sizes");
  CE_REQUIRE(num_rows > 0 && num_rows < (int64_t)INT32_MAX, CE_ERR_INVALID, "num_rows out of range");
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  const bool vec = (dim % 4 == 0) && al16(weight) && al16(src_rows);
  const int rowlen = vec ? dim / 4 : dim;
  int g = 1, gl2 = 0;
  while (g < rowlen && g < 64) { g <<= 1; ++gl2; }
  const int need = (rowlen + g - 1) / g;
  int nch = 1;
  while (nch < need) nch <<= 1;
  CE_REQUIRE(nch <= 4, CE_ERR_UNSUPPORTED, "embedding dim %d too large for this build", dim);
  const int u = nch == 1 ? 8 : (nch == 2 ? 4 : 2);
  const int64_t per_block = (int64_t)(64 / g) * u * 4;
  dim3 grid(grid_for(n, (int)per_block)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define CE_AXPY(VT, N)                                                                                      \
  hipLaunchKernelGGL((k_rows_axpy<VT, N>), grid, block, 0, s, weight, (uint32_t)num_rows, index, n,         \
                     (const VT*)src_rows, rowlen, gl2, dim, alpha)
  if (vec) {
    if (nch == 1) CE_AXPY(f32x4, 1); else if (nch == 2) CE_AXPY(f32x4, 2); else CE_AXPY(f32x4, 4);
  } else {
    if (nch == 1) CE_AXPY(float, 1); else if (nch == 2) CE_AXPY(float, 2); else CE_AXPY(float, 4);
  }
#undef CE_AXPY
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int ce_bag_backward_rows(float* grad_rows, const int64_t* dest_index, int32_t dim, int64_t nnz,
                                    const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                                    int32_t include_last_offset, const float* per_sample_weights, int32_t mode,
                                    int64_t hook_features, const float* grad_out, ce_stream_t stream) {
  if (num_bags == 0 || nnz == 0) return CE_OK;
  CE_REQUIRE(grad_rows && grad_out && offsets, CE_ERR_INVALID, "null pointer");
  BagParams p{};
  bool vec;
  int nch;
  int rc = fill_params(p, dim, dest_index, nnz, offsets, offsets_are_i64, num_bags, include_last_offset,
                       per_sample_weights, mode, hook_features, &vec, &nch, grad_rows, grad_out, nullptr);
  if (rc) return rc;
  p.dst = grad_rows;
  p.grad_out = grad_out;
  p.alpha = 1.f;
  return launch_bwd_rows(p, vec, nch, (hipStream_t)stream);
}
