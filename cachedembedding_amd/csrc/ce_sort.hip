// Stable multi-way split on gfx950 and the two things built on it:
//  * ce_bucketize_rows: owner-bucketing of looked-up rows for row-wise sharding (the build's
//    replacement for KJTAllToAll's all-gather, recsys/datasets/utils.py:20-54);
//  * ce_bag_backward_sgd_sorted: deterministic K13+K14 -- lookups are stably radix-sorted by
//    target row, each row's gradients are summed in lookup order and applied once, which is
//    the coalesce()-then-add order torch uses for sparse grads (recsys/dlrm_main.py:274-279).
//
// One split pass = histogram per 4096-lookup tile (digit-major), one-block exclusive scan,
// stable scatter.  Ranks inside a tile come from wave64 ballots (8 ballots give each lane
// the mask of lanes holding the same digit) plus per-wave digit counters in LDS.
#include <stdlib.h>

#include <algorithm>

#include "ce_common.h"

namespace ce {

constexpr int kTile = 4096;      // lookups per block per pass
constexpr int kRounds = kTile / 256;

__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// in-place exclusive scan of n ints by one 1024-thread block
__global__ __launch_bounds__(1024) void k_scan1024(int32_t* a, int64_t n, int32_t* total_out) {
  __shared__ long long carry;
  __shared__ int wtot[16];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int v = (i < n) ? a[i] : 0;
    const int inc = wave_incl_scan_i(v, lane);
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    int pre = 0, tot = 0;
    for (int k = 0; k < 16; ++k) {
      if (k < w) pre += wtot[k];
      tot += wtot[k];
    }
    const long long c = carry;
    if (i < n) a[i] = (int32_t)(c + pre + inc - v);
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = (int32_t)carry;
}

// DIGIT: 0 = (key >> shift) & 255 ; 1 = key % world
template <int DIGIT>
__device__ __forceinline__ int digit_of(int32_t key, int shift, int world) {
  return DIGIT == 0 ? ((key >> shift) & 255) : (key % world);
}

// KEYSRC: 0 = keys_in[i] ; 1 = idx_map[ids[i]] (or ids[i]) ; 2 = (int32)ids64[i]
struct SplitArgs {
  const int32_t* keys_in;
  const int32_t* vals_in;     // nullptr: value = i
  const int64_t* ids;
  const int32_t* idx_map;
  int32_t* keys_out;
  int32_t* vals_out;
  int64_t* rows_out64;        // bucketize: local row = key / world
  int64_t* perm_out64;        // bucketize: position of lookup i
  int32_t* hist;              // [nb][ntiles] digit-major
  int64_t n;
  int ntiles;
  int shift;
  int world;
  int nb;
};

template <int KEYSRC>
__device__ __forceinline__ int32_t load_key(const SplitArgs& a, int64_t i) {
  if (KEYSRC == 0) return a.keys_in[i];
  const int64_t id = a.ids[i];
  if (KEYSRC == 1 && a.idx_map) return a.idx_map[id];
  return (int32_t)id;
}

template <int DIGIT, int KEYSRC>
__global__ __launch_bounds__(256) void k_split_hist(SplitArgs a) {
  __shared__ int cnt[256];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kTile;
  for (int r = 0; r < kRounds; ++r) {
    const int64_t i = base + r * 256 + threadIdx.x;
    if (i < a.n) atomicAdd(&cnt[digit_of<DIGIT>(load_key<KEYSRC>(a, i), a.shift, a.world)], 1);
  }
  __syncthreads();
  if ((int)threadIdx.x < a.nb) a.hist[(int64_t)threadIdx.x * a.ntiles + blockIdx.x] = cnt[threadIdx.x];
}

template <int DIGIT, int KEYSRC, bool BUCKETIZE>
__global__ __launch_bounds__(256) void k_split_scatter(SplitArgs a) {
  __shared__ int run[256];         // running position of each digit for this tile
  __shared__ int wcnt[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if ((int)threadIdx.x < a.nb) run[threadIdx.x] = a.hist[(int64_t)threadIdx.x * a.ntiles + blockIdx.x];
  const int64_t base = (int64_t)blockIdx.x * kTile;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < kRounds; ++r) {
    for (int d = threadIdx.x; d < 4 * 256; d += 256) (&wcnt[0][0])[d] = 0;
    __syncthreads();
    const int64_t i = base + r * 256 + threadIdx.x;
    const bool valid = i < a.n;
    int32_t key = 0;
    int dg = 0;
    if (valid) {
      key = load_key<KEYSRC>(a, i);
      dg = digit_of<DIGIT>(key, a.shift, a.world);
    }
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned long long m = __ballot((dg >> b) & 1);
      peers &= ((dg >> b) & 1) ? m : ~m;
    }
    const int rank = __popcll(peers & lt);
    if (valid && rank == 0) wcnt[w][dg] = __popcll(peers);
    __syncthreads();
    if (valid) {
      int pre = 0;
      for (int k = 0; k < w; ++k) pre += wcnt[k][dg];
      const int pos = run[dg] + pre + rank;
      if (BUCKETIZE) {
        a.rows_out64[pos] = (int64_t)(key / a.world);
        a.perm_out64[i] = pos;
      } else {
        a.keys_out[pos] = key;
        a.vals_out[pos] = a.vals_in ? a.vals_in[i] : (int32_t)i;
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < a.nb)
      run[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
    // next round's zeroing of wcnt is ordered by the __syncthreads at its top
    __syncthreads();
  }
}

__global__ void k_bucket_counts(const int32_t* hist_scanned, int32_t total, int ntiles, int world, int64_t* counts) {
  const int w = threadIdx.x;
  if (w < world) {
    const int32_t lo = hist_scanned[(int64_t)w * ntiles];
    const int32_t hi = (w + 1 < world) ? hist_scanned[(int64_t)(w + 1) * ntiles] : total;
    counts[w] = (int64_t)(hi - lo);
  }
}

// lookup -> bag id
__global__ __launch_bounds__(256) void k_expand_bags(const void* offsets, int off64, int64_t num_bags, int64_t nnz,
                                                     int include_last, int32_t* bag_of) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < num_bags; b += stride) {
    const int64_t lo = off64 ? ((const int64_t*)offsets)[b] : ((const int32_t*)offsets)[b];
    const int64_t hi = (include_last || b + 1 < num_bags)
                           ? (off64 ? ((const int64_t*)offsets)[b + 1] : ((const int32_t*)offsets)[b + 1])
                           : nnz;
    for (int64_t j = lo; j < hi; ++j) bag_of[j] = (int32_t)b;
  }
}

struct SegArgs {
  float* weight;
  const float* grad_out;
  const int32_t* rows_sorted;
  const int32_t* lookup_sorted;
  const int32_t* bag_of;
  const void* offsets;
  const float* psw;
  int64_t nnz;
  int64_t num_bags;
  int rowlen, g_log2, off64, include_last, mode, hookF, hookB;
  float lr;
};

// a lane group owns a sorted position; only segment heads work: sum the segment in order, update once
template <typename VT>
__global__ __launch_bounds__(256) void k_seg_sgd(SegArgs a) {
  const int G = 1 << a.g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> a.g_log2;
  VT* W = (VT*)a.weight;
  const VT* GO = (const VT*)a.grad_out;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> a.g_log2; i < a.nnz; i += gstride) {
    const int32_t row = a.rows_sorted[i];
    if (i > 0 && a.rows_sorted[i - 1] == row) continue;
    for (int c = gl; c < a.rowlen; c += G) {
      VT acc = vzero<VT>();
      for (int64_t t = i; t < a.nnz && a.rows_sorted[t] == row; ++t) {
        const int32_t j = a.lookup_sorted[t];
        const int32_t bag = a.bag_of[j];
        float s = a.psw ? a.psw[j] : 1.f;
        if (a.mode == CE_MODE_MEAN) {
          const int64_t lo = a.off64 ? ((const int64_t*)a.offsets)[bag] : ((const int32_t*)a.offsets)[bag];
          const int64_t hi = (a.include_last || bag + 1 < a.num_bags)
                                 ? (a.off64 ? ((const int64_t*)a.offsets)[bag + 1] : ((const int32_t*)a.offsets)[bag + 1])
                                 : a.nnz;
          if (hi - lo > 1) s = s / (float)(hi - lo);
        }
        int64_t orow = bag;
        if (a.hookF) {
          const int f = bag / a.hookB;
          orow = (int64_t)(bag - f * a.hookB) * a.hookF + f;
        }
        const VT g = GO[orow * a.rowlen + c];
        acc = (s == 1.f) ? acc + g : acc + g * s;
      }
      W[(int64_t)row * a.rowlen + c] = W[(int64_t)row * a.rowlen + c] - acc * a.lr;
    }
  }
}

// ---- fused dedupe + owner bucketing (row-wise exchange, requester side).  Three passes, no sort, no histogram /
// scan chain, no host sync and NO returning atomics (a random device-scope atomic with return costs a
// ~2 us round trip to the memory side and a CU keeps only a few dozen in flight: 426 k of them took 90 us):
//   mark:   stamp[row] = lookup index, plain stores (any ONE of the racing lookups of a row survives -- which
//           one does not matter); equal rows of a wave are merged first so hot rows cost one store per wave;
//   claim:  the lookup whose index survived owns the row: it numbers the row WITHIN ITS OWNER'S BUCKET
//           (wave ballots -> one counter bump per owner per workgroup);
//   finish: (owner, index in bucket) -> position in the owner-major list once the bucket sizes are final.
__global__ __launch_bounds__(256) void k_dedupe_mark_w(const int64_t* __restrict__ ids, int64_t n,
                                                       const int32_t* __restrict__ idx_map, int64_t num_rows,
                                                       int row_bits, int32_t* __restrict__ stamp,
                                                       int32_t* __restrict__ rows32, int64_t scratch_stride) {
  // blockIdx.y = batch of a window: every batch has its own ids / stamp array / scratch (see dedupe_bucket_impl)
  ids += (int64_t)blockIdx.y * n;
  stamp += (int64_t)blockIdx.y * num_rows;
  rows32 += (int64_t)blockIdx.y * scratch_stride;
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // wave-uniform trip count (ballots inside)
  for (int64_t b0 = (int64_t)blockIdx.x * blockDim.x; b0 < n; b0 += stride) {
    const int64_t i = b0 + threadIdx.x;
    bool on = i < n;
    int32_t row = -1;
    if (on) {
      const int64_t id = ids[i];
      on = (unsigned long long)id < (unsigned long long)num_rows;
      if (on) row = idx_map ? idx_map[id] : (int32_t)id;
      rows32[i] = row;
    }
    unsigned long long pm = __ballot(on);
    if (!on) pm = 0;
    for (int b = 0; b < row_bits; ++b) {
      const unsigned long long m = __ballot((row >> b) & 1);
      pm &= ((row >> b) & 1) ? m : ~m;
    }
    if (on && (__ffsll((long long)pm) - 1) == lane) stamp[row] = (int32_t)i;
  }
}

template <int IT>
__global__ __launch_bounds__(256) void k_dedupe_claim_w(const int32_t* __restrict__ rows32, int64_t n,
                                                        int32_t world, const int32_t* stamp,
                                                        int32_t* slot_of_row,      // (may alias stamp: no __restrict__)
                                                        int32_t* __restrict__ bucket_rows,
                                                        unsigned long long* counts, int64_t num_rows,
                                                        int64_t scratch_stride) {
  rows32 += (int64_t)blockIdx.y * scratch_stride;
  bucket_rows += (int64_t)blockIdx.y * scratch_stride;
  stamp += (int64_t)blockIdx.y * num_rows;
  slot_of_row += (int64_t)blockIdx.y * num_rows;
  counts += (int64_t)blockIdx.y * world;
  __shared__ int wave_cnt[4][64];
  __shared__ unsigned long long blk_base[64];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int64_t per_block = 256 * IT;
  for (int64_t b0 = (int64_t)blockIdx.x * per_block; b0 < n; b0 += (int64_t)gridDim.x * per_block) {
    int32_t row[IT];
    bool claim[IT];
#pragma unroll
    for (int t = 0; t < IT; ++t) {
      const int64_t i = b0 + t * 256 + threadIdx.x;
      row[t] = (i < n) ? rows32[i] : -1;
    }
#pragma unroll
    for (int t = 0; t < IT; ++t) {
      const int64_t i = b0 + t * 256 + threadIdx.x;
      claim[t] = row[t] >= 0 && stamp[row[t]] == (int32_t)i;
    }
    // rank of every claimed row inside (wave, owner); lane o carries the wave's running count for owner o
    int own[IT], rank[IT];
    int mycnt = 0;
#pragma unroll
    for (int t = 0; t < IT; ++t) {
      own[t] = claim[t] ? row[t] % world : 0;
      rank[t] = 0;
      for (int o = 0; o < world; ++o) {
        const bool mine = claim[t] && own[t] == o;
        const unsigned long long m = __ballot(mine);
        const int prev = __shfl(mycnt, o);
        if (mine) rank[t] = prev + __popcll(m & lt);
        if (lane == o) mycnt += __popcll(m);
      }
    }
    wave_cnt[wv][lane] = mycnt;
    __syncthreads();
    if (threadIdx.x < world) {
      const int tot = wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x] + wave_cnt[2][threadIdx.x] +
                      wave_cnt[3][threadIdx.x];
      blk_base[threadIdx.x] = tot ? atomicAdd(&counts[threadIdx.x], (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < IT; ++t) {
      if (claim[t]) {
        unsigned long long p = blk_base[own[t]] + rank[t];
        for (int k = 0; k < wv; ++k) p += wave_cnt[k][own[t]];
        bucket_rows[(int64_t)own[t] * n + p] = row[t] / world;
        // the place is stored as -1 - p: `slot_of_row` may BE the stamp array (one int32 per row instead of two), and a
        // negative entry can never pass another lookup's claim test `stamp[row] == lookup index`
        slot_of_row[row[t]] = -1 - (int32_t)p;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_dedupe_finish_w(const int32_t* __restrict__ rows32, int64_t n, int32_t world,
                                                         const int32_t* __restrict__ slot_of_row,
                                                         const int32_t* __restrict__ bucket_rows,
                                                         const unsigned long long* __restrict__ counts,
                                                         int64_t* __restrict__ local_rows_out,
                                                         int64_t* __restrict__ pos_out) {
  __shared__ long long pre[65];
  if (threadIdx.x == 0) {
    long long acc = 0;
    for (int o = 0; o < world; ++o) { pre[o] = acc; acc += (long long)counts[o]; }
    pre[world] = acc;
  }
  __syncthreads();
  const long long n_u = pre[world];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t row = rows32[i];
    pos_out[i] = row >= 0 ? pre[row % world] + (-1 - slot_of_row[row]) : -1;
    if (i < n_u) {                       // owner-major compaction of the bucket lists
      int o = 0;
      while (o + 1 < world && pre[o + 1] <= i) ++o;
      local_rows_out[i] = bucket_rows[(int64_t)o * n + (i - pre[o])];
    }
  }
}

// Fixed-capacity form for the graphed row-wise exchange: bucket o occupies rows_out[o * cap, (o + 1) * cap) (unused
// places = -1) and pos = o * cap + place, so every tensor of a step has a size that does not depend on the data and a
// window's steps (padded all-to-alls included) can be replayed as one hipGraph.  A bucket that does not fit sets
// *overflow (the caller re-plans that window on the variable-size path); its surplus lookups get pos = -1.
__global__ __launch_bounds__(256) void k_dedupe_finish_pad(const int32_t* __restrict__ rows32, int64_t n, int32_t world,
                                                           int64_t cap, const int32_t* __restrict__ slot_of_row,
                                                           const int32_t* __restrict__ bucket_rows,
                                                           const unsigned long long* __restrict__ counts,
                                                           int64_t* __restrict__ rows_out, int64_t* __restrict__ pos_out,
                                                           int32_t* overflow, int64_t num_rows,
                                                           int64_t scratch_stride) {
  rows32 += (int64_t)blockIdx.y * scratch_stride;
  bucket_rows += (int64_t)blockIdx.y * scratch_stride;
  slot_of_row += (int64_t)blockIdx.y * num_rows;
  counts += (int64_t)blockIdx.y * world;
  rows_out += (int64_t)blockIdx.y * world * cap;
  pos_out += (int64_t)blockIdx.y * n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid0 < world && (long long)counts[tid0] > cap) *overflow = 1;
  for (int64_t i = tid0; i < n; i += stride) {
    const int32_t row = rows32[i];
    int64_t p = -1;
    if (row >= 0) {
      const int32_t place = -1 - slot_of_row[row];
      if (place < cap) p = (int64_t)(row % world) * cap + place;
    }
    pos_out[i] = p;
  }
  for (int64_t i = tid0; i < (int64_t)world * cap; i += stride) {
    const int64_t o = i / cap, j = i - o * cap;
    rows_out[i] = j < (long long)counts[o] ? (int64_t)bucket_rows[o * n + j] : -1;
  }
}

// Requester side of the row-wise exchange, local bypass.  The rows a rank asks ITSELF for never travel: the place
// of such a lookup is replaced by the cache slot its own owner-side cache op resolved, every other place p becomes
// tail_base + p, a row of the exchange buffer that lies right behind the cache in the same allocation -- so the
// pooling and the fused update address "cache + received rows" as ONE table with one index.
__global__ __launch_bounds__(256) void k_exchange_local_index(const int64_t* __restrict__ pos, int64_t n_per_batch,
                                                              int64_t total, const int64_t* __restrict__ slots,
                                                              int64_t slots_batch_stride, int64_t lo, int64_t hi,
                                                              int64_t tail_base, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t p = pos[i];
    int64_t r = -1;
    if (p >= lo && p < hi) r = slots[(i / n_per_batch) * slots_batch_stride + p];
    else if (p >= 0) r = tail_base + p;
    out[i] = r;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Early / late split of the row-wise exchange (round 5).  A row of step t must carry the update
// of step t-1 only if SOME rank looked it up in step t-1; every other row can leave its owner while step t-1 still
// computes.  The owner sees every rank's requests of the whole window when it plans, so it classifies them:
//   bit 0 LATE    the row is requested, by any peer, in the batch before  -> travels after that step's update
//   bit 1 URGENT  the row is requested, by any peer, in the batch after   -> its returned gradient must be applied
//                                                                             before that step's rows leave
// One uint64 per local row holds "requested in batch b" in bit b (bit 63: in the last batch of the window before);
// three passes over the window's requests: set, read, clear -- the scratch is all zero again afterwards.
constexpr int kSplitPrevBit = 63;

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

__global__ __launch_bounds__(256) void k_split_set(const int64_t* __restrict__ serve, int world, int P, int64_t cap,
                                                   int64_t n_rows, const int64_t* __restrict__ prev, int64_t n_prev,
                                                   unsigned long long* mask) {
  const int64_t total = (int64_t)world * P * cap;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total + n_prev; e += stride) {
    int64_t row;
    int bit;
    if (e < total) {
      row = serve[e];
      bit = (int)((e / cap) % P);
    } else {
      row = prev[e - total];
      bit = kSplitPrevBit;
    }
    if (row < 0 || row >= n_rows) continue;
    const unsigned long long b = 1ull << bit;
    if (!(__hip_atomic_load(&mask[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & b)) atomicOr(&mask[row], b);
  }
}

__global__ __launch_bounds__(256) void k_split_read(const int64_t* __restrict__ serve, int world, int P, int64_t cap,
                                                    int64_t n_rows, int have_prev, const unsigned long long* __restrict__ mask,
                                                    uint8_t* __restrict__ flags) {
  const int64_t total = (int64_t)world * P * cap;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const int64_t row = serve[e];
    uint8_t f = 0;
    if (row >= 0 && row < n_rows) {
      const int b = (int)((e / cap) % P);
      const unsigned long long m = mask[row];
      // no window before this one (have_prev == 0): nothing is in flight that batch 0 could depend on -> all EARLY
      const bool late = b == 0 ? (have_prev && ((m >> kSplitPrevBit) & 1)) : ((m >> (b - 1)) & 1);
      // the window after this one is not planned yet: the last batch returns everything at once
      const bool urgent = b == P - 1 ? true : ((m >> (b + 1)) & 1);
      f = (uint8_t)((late ? 1 : 0) | (urgent ? 2 : 0));
    }
    flags[e] = f;
  }
}

__global__ __launch_bounds__(256) void k_split_clear(const int64_t* __restrict__ serve, int64_t total, int64_t n_rows,
                                                     const int64_t* __restrict__ prev, int64_t n_prev,
                                                     unsigned long long* mask) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total + n_prev; e += stride) {
    const int64_t row = e < total ? serve[e] : prev[e - total];
    if (row >= 0 && row < n_rows) mask[row] = 0ull;
  }
}

// Places inside the split exchange buffers: one workgroup per (batch, peer) chunk of `cap` requests.  ids / flags are
// batch-major [P][W][cap].  Forward buffer: EARLY rows of peer w at [w * cap_e, ...), LATE rows at W * cap_e + [w * cap_l,
// ...), each in request order; an early row beyond cap_e spills behind the chunk's late rows (sending a row later is
// always correct).  Backward buffer the same with DEFERRED (not urgent) first.  caps[b] = {cap_e, cap_l, cap_d, cap_u}
// of batch b.  counts[b][w] = {early, late, deferred, urgent} as classified (before any spill).
__global__ __launch_bounds__(1024) void k_split_places(const int64_t* __restrict__ ids, const uint8_t* __restrict__ flags,
                                                       int world, int64_t cap, int skip_peer,
                                                       const int32_t* __restrict__ caps, int32_t* __restrict__ place_fwd,
                                                       int32_t* __restrict__ place_bwd, int32_t* counts, int32_t* overflow) {
  __shared__ int run[2];          // late / urgent rows of the chunk seen so far
  __shared__ int tot[2];          // ... in the whole chunk
  __shared__ int red[16][3];
  __shared__ unsigned long long wsum[16];
  const int b = (int)(blockIdx.x / world), w = (int)(blockIdx.x % world);
  const int64_t base = (int64_t)blockIdx.x * cap;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ce = caps[4 * b + 0], cl = caps[4 * b + 1], cd = caps[4 * b + 2], cu = caps[4 * b + 3];
  const bool skip = w == skip_peer;
  // pass 1: the chunk's class totals (the spill positions need the number of late / urgent rows of the WHOLE chunk)
  int nl = 0, nu = 0, nv = 0;
  for (int64_t j = tid; j < cap; j += 1024) {
    if (ids[base + j] < 0) continue;
    const uint8_t f = flags[base + j];
    nv += 1;
    nl += f & 1;
    nu += (f >> 1) & 1;
  }
  nl = wave_sum_i(nl); nu = wave_sum_i(nu); nv = wave_sum_i(nv);
  if (lane == 0) { red[wv][0] = nl; red[wv][1] = nu; red[wv][2] = nv; }
  __syncthreads();
  if (tid == 0) {
    int a = 0, c = 0, v = 0;
    for (int k = 0; k < 16; ++k) { a += red[k][0]; c += red[k][1]; v += red[k][2]; }
    tot[0] = a; tot[1] = c;
    run[0] = run[1] = 0;
    if (counts) {
      int32_t* o = counts + 4 * (int64_t)blockIdx.x;
      o[0] = v - a; o[1] = a; o[2] = v - c; o[3] = c;
    }
    if (!skip) {
      const int spill_l = max(0, (v - a) - ce), spill_u = max(0, (v - c) - cd);
      if (a + spill_l > cl || c + spill_u > cu) atomicOr(overflow, 1);
    }
  }
  __syncthreads();
  const int n_late = tot[0], n_urg = tot[1];
  // pass 2: ranks in request order, tile by tile
  int done_v = 0;                  // valid rows of the tiles before this one (uniform)
  for (int64_t j0 = 0; j0 < cap; j0 += 1024) {
    const int64_t j = j0 + tid;
    const bool valid = j < cap && ids[base + j] >= 0;
    const uint8_t f = valid ? flags[base + j] : 0;
    const int isl = valid ? (f & 1) : 0, isu = valid ? ((f >> 1) & 1) : 0, isv = valid ? 1 : 0;
    // three inclusive wave scans packed into one: late (10 bits would do, 21 each to be safe)
    unsigned long long pk = (unsigned long long)isl | ((unsigned long long)isu << 21) | ((unsigned long long)isv << 42);
    unsigned long long inc = pk;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long o = __shfl_up(inc, d);
      if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned long long pre = inc - pk, all = 0;
    for (int k = 0; k < 16; ++k) { if (k < wv) pre += wsum[k]; all += wsum[k]; }
    const int r_l = (int)(pre & 0x1fffff) + run[0], r_u = (int)((pre >> 21) & 0x1fffff) + run[1];
    const int r_v = (int)(pre >> 42) + done_v;
    if (j < cap) {
      int pf = -1, pb = -1;
      if (valid && !skip) {
        const int r_e = r_v - r_l, r_d = r_v - r_u;        // rank among the early / deferred rows
        if (isl) pf = r_l < cl ? world * ce + w * cl + r_l : -1;
        else if (r_e < ce) pf = w * ce + r_e;
        else { const int q = n_late + (r_e - ce); pf = q < cl ? world * ce + w * cl + q : -1; }
        if (isu) pb = r_u < cu ? world * cd + w * cu + r_u : -1;
        else if (r_d < cd) pb = w * cd + r_d;
        else { const int q = n_urg + (r_d - cd); pb = q < cu ? world * cd + w * cu + q : -1; }
      }
      place_fwd[base + j] = pf;
      place_bwd[base + j] = pb;
    }
    __syncthreads();
    if (tid == 0) {
      run[0] += (int)(all & 0x1fffff);
      run[1] += (int)((all >> 21) & 0x1fffff);
    }
    done_v += (int)(all >> 42);
    __syncthreads();
  }
}

// Requester side: lookup -> row of "cache + exchange buffers" for the pooling (forward buffers) and for the fused
// fold + SGD (backward buffers).  A place in this rank's own chunk [local_lo, local_hi) is the cache slot its own
// owner-side cache op resolved; any other place p goes through place_fwd / place_bwd (k_split_places).  The EARLY and
// the DEFERRED region exist twice (a step's early rows arrive while the step before still reads its own; a step's
// deferred gradients leave while the next step already folds): batch b uses copy b & 1.
// fwd: tail_base + [E0 | E1 | L], bwd: tail_base + bwd_base + [D0 | U | D1] (the urgent region in the middle: the two
// regions a step's fold writes -- its deferred copy and U -- are then ONE contiguous range to zero, whichever copy it is);
// n_e = W * cap_e etc. are region sizes.
__global__ __launch_bounds__(256) void k_exchange_local_index_split(
    const int64_t* __restrict__ pos, int64_t n_per_batch, int64_t total, const int64_t* __restrict__ slots,
    const int32_t* __restrict__ place_fwd, const int32_t* __restrict__ place_bwd, int64_t chunk_stride, int64_t lo,
    int64_t hi, int64_t tail_base, int64_t bwd_base, const int32_t* __restrict__ caps, int world, int64_t n_e,
    int64_t n_u, int64_t n_d, int64_t* __restrict__ idx_fwd, int64_t* __restrict__ idx_bwd) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t p = pos[i];
    const int64_t b = i / n_per_batch;
    int64_t f = -1, g = -1;
    if (p >= lo && p < hi) {
      f = g = slots[b * chunk_stride + p];
    } else if (p >= 0) {
      const int pf = place_fwd[b * chunk_stride + p], pb = place_bwd[b * chunk_stride + p];
      const int64_t we = (int64_t)world * caps[4 * b + 0], wd = (int64_t)world * caps[4 * b + 2];
      if (pf >= 0) f = tail_base + (pf < we ? (b & 1) * n_e + pf : 2 * n_e + (pf - we));
      if (pb >= 0) g = tail_base + bwd_base + (pb < wd ? (b & 1) * (n_d + n_u) + pb : n_d + (pb - wd));
    }
    idx_fwd[i] = f;
    idx_bwd[i] = g;
  }
}

struct SortWs {
  int32_t *keys[2], *vals[2], *bag_of, *hist, *total;
  size_t bytes;
};

static SortWs carve(void* ws, int64_t nnz) {
  SortWs s{};
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const int64_t ntiles = std::max<int64_t>(1, cdiv(nnz, kTile));
  char* p = (char*)ws;
  size_t o = 0;
  for (int i = 0; i < 2; ++i) { s.keys[i] = (int32_t*)(p + o); o = al(o + (size_t)nnz * 4); }
  for (int i = 0; i < 2; ++i) { s.vals[i] = (int32_t*)(p + o); o = al(o + (size_t)nnz * 4); }
  s.bag_of = (int32_t*)(p + o); o = al(o + (size_t)nnz * 4);
  s.hist = (int32_t*)(p + o);   o = al(o + (size_t)256 * ntiles * 4);
  s.total = (int32_t*)(p + o);  o = al(o + 256);
  s.bytes = o;
  return s;
}

}  // namespace ce

using namespace ce;

extern "C" size_t ce_bucketize_workspace(int64_t n, int32_t world) {
  if (n < 0 || world < 1) return 0;
  const int64_t ntiles = std::max<int64_t>(1, cdiv(n, kTile));
  return (size_t)world * ntiles * 4 + 512;
}

extern "C" int ce_bucketize_rows(const int64_t* ids, int64_t n, const int32_t* idx_map, int32_t world,
                                 int64_t* local_rows_out, int64_t* perm_out, int64_t* counts_out, void* workspace,
                                 size_t workspace_bytes, ce_stream_t stream) {
  CE_REQUIRE(world >= 1 && world <= 256, CE_ERR_UNSUPPORTED, "world size must be in [1, 256]");
  CE_REQUIRE(n >= 0 && n < (int64_t)INT32_MAX, CE_ERR_UNSUPPORTED, "too many lookups");
  CE_REQUIRE(counts_out && workspace, CE_ERR_INVALID, "null pointer");
  CE_REQUIRE(workspace_bytes >= ce_bucketize_workspace(n, world), CE_ERR_INVALID, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    CE_HIP_CHECK(hipMemsetAsync(counts_out, 0, sizeof(int64_t) * world, s));
    return CE_OK;
  }
  CE_REQUIRE(ids && local_rows_out && perm_out, CE_ERR_INVALID, "null pointer");
  const int ntiles = (int)cdiv(n, kTile);
  SplitArgs a{};
  a.ids = ids;
  a.idx_map = idx_map;
  a.rows_out64 = local_rows_out;
  a.perm_out64 = perm_out;
  a.hist = (int32_t*)workspace;
  int32_t* total = a.hist + (int64_t)world * ntiles;
  a.n = n;
  a.ntiles = ntiles;
  a.world = world;
  a.nb = world;
  hipLaunchKernelGGL((k_split_hist<1, 1>), dim3(ntiles), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_scan1024, dim3(1), dim3(1024), 0, s, a.hist, (int64_t)world * ntiles, total);
  hipLaunchKernelGGL(k_bucket_counts, dim3(1), dim3(256), 0, s, (const int32_t*)a.hist, n > 0 ? (int32_t)n : 0,
                     ntiles, world, counts_out);
  hipLaunchKernelGGL((k_split_scatter<1, 1, true>), dim3(ntiles), dim3(256), 0, s, a);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

// n_batches > 1 (fixed-capacity form only): the batches of a window in ONE launch per pass, batch b = blockIdx.y with
// its own ids (ids + b * n), stamp / slot_of_row arrays (+ b * num_rows: the passes of different batches race on a
// row otherwise), scratch (+ b * (world + 1) * n) and outputs.
static int dedupe_bucket_impl(const int64_t* ids, int64_t n, const int32_t* idx_map, int64_t num_rows, int32_t world,
                              int64_t cap, int32_t* stamp, int32_t* slot_of_row, int32_t* scratch,
                              int64_t* local_rows_out, int64_t* pos_out, int64_t* counts_out, int32_t* overflow,
                              ce_stream_t stream, int64_t n_batches = 1) {
  CE_REQUIRE(n_batches >= 1 && n_batches <= 65535 && (n_batches == 1 || cap > 0), CE_ERR_INVALID, "bad batch count");
  const unsigned nb = (unsigned)n_batches;
  const int64_t sstride = (int64_t)(world + 1) * n;
  CE_REQUIRE(n >= 0 && n < (int64_t)INT32_MAX && num_rows > 0 && num_rows < (int64_t)INT32_MAX, CE_ERR_INVALID,
             "bad sizes");
  CE_REQUIRE(world >= 1 && world <= 64, CE_ERR_UNSUPPORTED, "world size must be in [1, 64]");
  CE_REQUIRE(stamp && counts_out, CE_ERR_INVALID, "null pointer");
  if (!slot_of_row) slot_of_row = stamp;      // one scratch array: the claim pass reuses the entry it has just tested
  hipStream_t s = (hipStream_t)stream;
  CE_HIP_CHECK(hipMemsetAsync(counts_out, 0, sizeof(int64_t) * world * n_batches, s));
  if (n == 0 && cap <= 0) return CE_OK;
  CE_REQUIRE(scratch && local_rows_out && (n == 0 || (ids && pos_out)), CE_ERR_INVALID, "null pointer");
  int bits = 1;
  while ((1ll << bits) < num_rows && bits < 31) ++bits;
  int32_t* rows32 = scratch;                 // int32[n]
  int32_t* bucket_rows = scratch + n;        // int32[world][n]
  unsigned long long* counts = (unsigned long long*)counts_out;
  if (n > 0) {
    hipLaunchKernelGGL(k_dedupe_mark_w, dim3(grid_for(n, 256), nb), dim3(256), 0, s, ids, n, idx_map, num_rows, bits,
                       stamp, rows32, sstride);
    // 4 lookups per thread in the claim pass (1 and 2 measured slower)
    hipLaunchKernelGGL((k_dedupe_claim_w<4>), dim3(grid_for(n, 1024), nb), dim3(256), 0, s, (const int32_t*)rows32, n,
                         world, (const int32_t*)stamp, slot_of_row, bucket_rows, counts, num_rows, sstride);
  }
  if (cap > 0)
    hipLaunchKernelGGL(k_dedupe_finish_pad, dim3(grid_for(std::max<int64_t>(n, world * cap), 256), nb), dim3(256), 0, s,
                       (const int32_t*)rows32, n, world, cap, (const int32_t*)slot_of_row, (const int32_t*)bucket_rows,
                       (const unsigned long long*)counts, local_rows_out, pos_out, overflow, num_rows, sstride);
  else
    hipLaunchKernelGGL(k_dedupe_finish_w, dim3(grid_for(n, 256)), dim3(256), 0, s, (const int32_t*)rows32, n, world,
                       (const int32_t*)slot_of_row, (const int32_t*)bucket_rows, (const unsigned long long*)counts,
                       local_rows_out, pos_out);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int ce_dedupe_bucket_rows(const int64_t* ids, int64_t n, const int32_t* idx_map, int64_t num_rows,
                                     int32_t world, int32_t* stamp, int32_t* slot_of_row, int32_t* scratch,
                                     int64_t* local_rows_out, int64_t* pos_out, int64_t* counts_out,
                                     ce_stream_t stream) {
  return dedupe_bucket_impl(ids, n, idx_map, num_rows, world, 0, stamp, slot_of_row, scratch, local_rows_out, pos_out,
                            counts_out, nullptr, stream);
}

extern "C" int ce_dedupe_bucket_rows_padded(const int64_t* ids, int64_t n, const int32_t* idx_map, int64_t num_rows,
                                            int32_t world, int64_t capacity, int32_t* stamp, int32_t* slot_of_row,
                                            int32_t* scratch, int64_t* local_rows_out, int64_t* pos_out,
                                            int64_t* counts_out, int32_t* overflow_flag, ce_stream_t stream) {
  CE_REQUIRE(capacity > 0 && overflow_flag, CE_ERR_INVALID, "capacity must be positive, overflow_flag non-null");
  return dedupe_bucket_impl(ids, n, idx_map, num_rows, world, capacity, stamp, slot_of_row, scratch, local_rows_out,
                            pos_out, counts_out, overflow_flag, stream);
}

extern "C" int ce_dedupe_bucket_rows_padded_window(const int64_t* ids, int64_t n, int64_t n_batches,
                                                   const int32_t* idx_map, int64_t num_rows, int32_t world,
                                                   int64_t capacity, int32_t* stamp, int32_t* slot_of_row,
                                                   int32_t* scratch, int64_t* local_rows_out, int64_t* pos_out,
                                                   int64_t* counts_out, int32_t* overflow_flag, ce_stream_t stream) {
  CE_REQUIRE(capacity > 0 && overflow_flag, CE_ERR_INVALID, "capacity must be positive, overflow_flag non-null");
  if (n_batches == 0) return CE_OK;
  return dedupe_bucket_impl(ids, n, idx_map, num_rows, world, capacity, stamp, slot_of_row, scratch, local_rows_out,
                            pos_out, counts_out, overflow_flag, stream, n_batches);
}

extern "C" int ce_exchange_local_index(const int64_t* pos, int64_t n_per_batch, int64_t n_batches,
                                       const int64_t* slots, int64_t slots_batch_stride, int64_t local_lo,
                                       int64_t local_hi, int64_t tail_base, int64_t* index_out, ce_stream_t stream) {
  const int64_t total = n_per_batch * n_batches;
  if (total <= 0) return CE_OK;
  CE_REQUIRE(pos && index_out && (slots || local_hi <= local_lo), CE_ERR_INVALID, "null pointer");
  CE_REQUIRE(local_lo >= 0 && local_hi <= slots_batch_stride && tail_base >= 0, CE_ERR_INVALID,
             "the local range must lie inside one batch's slots");
  hipLaunchKernelGGL(k_exchange_local_index, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, pos,
                     n_per_batch, total, slots, slots_batch_stride, local_lo, local_hi, tail_base, index_out);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int ce_split_classify(const int64_t* serve, int32_t world, int32_t n_batches, int64_t capacity,
                                 int64_t n_local_rows, const int64_t* prev, int64_t n_prev, uint64_t* mask,
                                 uint8_t* flags_out, ce_stream_t stream) {
  const int64_t total = (int64_t)world * n_batches * capacity;
  if (total <= 0) return CE_OK;
  CE_REQUIRE(serve && mask && flags_out && n_local_rows > 0, CE_ERR_INVALID, "null pointer");
  CE_REQUIRE(world >= 1 && n_batches >= 1 && n_batches < kSplitPrevBit, CE_ERR_UNSUPPORTED,
             "a window of at most %d batches", kSplitPrevBit - 1);
  CE_REQUIRE(n_prev >= 0 && (prev || n_prev == 0), CE_ERR_INVALID, "bad prev");
  hipStream_t s = (hipStream_t)stream;
  const dim3 g(grid_for(total + n_prev, 256)), b(256);
  hipLaunchKernelGGL(k_split_set, g, b, 0, s, serve, (int)world, (int)n_batches, capacity, n_local_rows, prev, n_prev,
                     (unsigned long long*)mask);
  hipLaunchKernelGGL(k_split_read, g, b, 0, s, serve, (int)world, (int)n_batches, capacity, n_local_rows,
                     n_prev > 0 ? 1 : 0, (const unsigned long long*)mask, flags_out);
  hipLaunchKernelGGL(k_split_clear, g, b, 0, s, serve, total, n_local_rows, prev, n_prev, (unsigned long long*)mask);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int ce_split_places(const int64_t* ids, const uint8_t* flags, int32_t n_batches, int32_t world,
                               int64_t capacity, int32_t skip_peer, const int32_t* caps, int32_t* place_fwd,
                               int32_t* place_bwd, int32_t* counts_out, int32_t* overflow_flag, ce_stream_t stream) {
  if ((int64_t)n_batches * world * capacity <= 0) return CE_OK;
  CE_REQUIRE(ids && flags && caps && place_fwd && place_bwd && overflow_flag, CE_ERR_INVALID, "null pointer");
  CE_REQUIRE(capacity < (1 << 21), CE_ERR_UNSUPPORTED, "capacity beyond 2^21 rows per bucket");
  hipLaunchKernelGGL(k_split_places, dim3((unsigned)(n_batches * world)), dim3(1024), 0, (hipStream_t)stream, ids, flags,
                     (int)world, capacity, (int)skip_peer, caps, place_fwd, place_bwd, counts_out, overflow_flag);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int ce_exchange_local_index_split(const int64_t* pos, int64_t n_per_batch, int64_t n_batches,
                                             const int64_t* slots, const int32_t* place_fwd, const int32_t* place_bwd,
                                             int64_t chunk_stride, int64_t local_lo, int64_t local_hi, int64_t tail_base,
                                             int64_t bwd_base, const int32_t* caps, int32_t world, int64_t n_early,
                                             int64_t n_urgent, int64_t n_deferred, int64_t* index_fwd, int64_t* index_bwd,
                                             ce_stream_t stream) {
  const int64_t total = n_per_batch * n_batches;
  if (total <= 0) return CE_OK;
  CE_REQUIRE(pos && slots && place_fwd && place_bwd && caps && index_fwd && index_bwd, CE_ERR_INVALID, "null pointer");
  CE_REQUIRE(local_lo >= 0 && local_hi <= chunk_stride && tail_base >= 0 && bwd_base >= 0, CE_ERR_INVALID, "bad ranges");
  hipLaunchKernelGGL(k_exchange_local_index_split, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, pos,
                     n_per_batch, total, slots, place_fwd, place_bwd, chunk_stride, local_lo, local_hi, tail_base,
                     bwd_base, caps, (int)world, n_early, n_urgent, n_deferred, index_fwd, index_bwd);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" size_t ce_bag_backward_sgd_sorted_workspace(int64_t num_rows, int64_t nnz) {
  (void)num_rows;
  if (nnz < 0) return 0;
  return carve(nullptr, std::max<int64_t>(nnz, 1)).bytes;
}

extern "C" int ce_bag_backward_sgd_sorted(float* weight, int64_t num_rows, int32_t dim, const int64_t* indices,
                                          int64_t nnz, const void* offsets, int32_t offsets_are_i64,
                                          int64_t num_bags, int32_t include_last_offset,
                                          const float* per_sample_weights, int32_t mode, int64_t hook_features,
                                          const float* grad_out, float lr, void* workspace, size_t workspace_bytes,
                                          ce_stream_t stream) {
  if (num_bags == 0 || nnz == 0) return CE_OK;
  CE_REQUIRE(weight && grad_out && offsets && indices && workspace, CE_ERR_INVALID, "null pointer");
  CE_REQUIRE(num_rows > 0 && num_rows < (int64_t)INT32_MAX && nnz < (int64_t)INT32_MAX &&
                 num_bags < (int64_t)INT32_MAX,
             CE_ERR_UNSUPPORTED, "sizes beyond 2^31");
  CE_REQUIRE(mode == CE_MODE_SUM || mode == CE_MODE_MEAN, CE_ERR_UNSUPPORTED, "mode must be sum or mean");
  CE_REQUIRE(hook_features >= 0 && (hook_features == 0 || num_bags % hook_features == 0), CE_ERR_INVALID,
             "hook_features must divide num_bags");
  CE_REQUIRE(workspace_bytes >= ce_bag_backward_sgd_sorted_workspace(num_rows, nnz), CE_ERR_INVALID,
             "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  SortWs ws = carve(workspace, nnz);
  const int ntiles = (int)cdiv(nnz, kTile);
  int bits = 1;
  while ((1ll << bits) < num_rows) ++bits;
  const int passes = (bits + 7) / 8;
  hipLaunchKernelGGL(k_expand_bags, dim3(grid_for(num_bags, 256)), dim3(256), 0, s, offsets, offsets_are_i64,
                     num_bags, nnz, include_last_offset, ws.bag_of);
  int cur = 0;
  for (int p = 0; p < passes; ++p) {
    SplitArgs a{};
    a.n = nnz;
    a.ntiles = ntiles;
    a.shift = 8 * p;
    a.world = 1;
    a.nb = 256;
    a.hist = ws.hist;
    a.keys_out = ws.keys[cur ^ 1];
    a.vals_out = ws.vals[cur ^ 1];
    if (p == 0) {
      a.ids = indices;
      hipLaunchKernelGGL((k_split_hist<0, 2>), dim3(ntiles), dim3(256), 0, s, a);
      hipLaunchKernelGGL(k_scan1024, dim3(1), dim3(1024), 0, s, a.hist, (int64_t)256 * ntiles, ws.total);
      hipLaunchKernelGGL((k_split_scatter<0, 2, false>), dim3(ntiles), dim3(256), 0, s, a);
    } else {
      a.keys_in = ws.keys[cur];
      a.vals_in = ws.vals[cur];
      hipLaunchKernelGGL((k_split_hist<0, 0>), dim3(ntiles), dim3(256), 0, s, a);
      hipLaunchKernelGGL(k_scan1024, dim3(1), dim3(1024), 0, s, a.hist, (int64_t)256 * ntiles, ws.total);
      hipLaunchKernelGGL((k_split_scatter<0, 0, false>), dim3(ntiles), dim3(256), 0, s, a);
    }
    cur ^= 1;
  }
  SegArgs g{};
  g.weight = weight;
  g.grad_out = grad_out;
  g.rows_sorted = ws.keys[cur];
  g.lookup_sorted = ws.vals[cur];
  g.bag_of = ws.bag_of;
  g.offsets = offsets;
  g.psw = per_sample_weights;
  g.nnz = nnz;
  g.num_bags = num_bags;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  const bool vec = (dim % 4 == 0) && al16(weight) && al16(grad_out);
  g.rowlen = vec ? dim / 4 : dim;
  int gl = 1, gl2 = 0;
  while (gl < g.rowlen && gl < 64) { gl <<= 1; ++gl2; }
  g.g_log2 = gl2;
  g.off64 = offsets_are_i64;
  g.include_last = include_last_offset;
  g.mode = mode;
  g.hookF = (int)hook_features;
  g.hookB = hook_features ? (int)(num_bags / hook_features) : 0;
  g.lr = lr;
  const int gpb = 256 >> gl2;
  if (vec) hipLaunchKernelGGL((k_seg_sgd<f32x4>), dim3(grid_for(nnz, gpb)), dim3(256), 0, s, g);
  else hipLaunchKernelGGL((k_seg_sgd<float>), dim3(grid_for(nnz, gpb)), dim3(256), 0, s, g);
  CE_LAUNCH_CHECK();
  return CE_OK;
}
