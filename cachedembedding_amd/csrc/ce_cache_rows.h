// Cache op, row movement and map updates: evict / stage, free list, admit, slots, flush, pack / unpack
// (SURVEY App. A.4, A.6, A.7).
// Part of the one translation unit ce_cache.hip (included there, in this order: ce_cache_index.h, ce_cache_select.h,
// ce_cache_rows.h, ce_cache_fused.h, ce_cache_worker.h); not a stand-alone header.
#pragma once

namespace ce {

// group of G lanes per row, 16 B per lane (or 4 B when the row is not 16-B sized)
template <typename VT>
__device__ __forceinline__ void copy_row(const VT* __restrict__ src, VT* __restrict__ dst, int rowlen, int gl, int G) {
  for (int c = gl; c < rowlen; c += G) dst[c] = src[c];
}

constexpr int kSwapRows = 16;   // rows in flight per lane group in the PCIe swap kernels

template <typename VT>
__global__ __launch_bounds__(1024) void k_evict(const int32_t* __restrict__ victims, int32_t* cached_idx_map,
                                               int32_t* inverted, const VT* __restrict__ cache, VT* host,
                                               long long first, int rowlen, int g_log2, const Ctl* ctl) {
  const long long k = (ctl->status == CE_OK) ? ctl->k_evict : 0;
  if (k <= first) return;
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> g_log2;
  if (!host) return;
  // kSwapRows rows in flight per lane group: the kernel is PCIe-latency bound, so it is launched on a SMALL grid
  // (it must not occupy the wave slots of the training kernels it overlaps with) and gets its
  // memory-level parallelism from unrolling instead
  for (int64_t i = first + (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2) * kSwapRows; i < k;
       i += gstride * kSwapRows) {
    if (rowlen <= G) {
      VT v[kSwapRows];
      int64_t dst[kSwapRows];
#pragma unroll
      for (int t = 0; t < kSwapRows; ++t) {
        dst[t] = -1;
        if (i + t < k) {
          const int32_t slot = victims[i + t];
          dst[t] = cached_idx_map[slot];
          if (gl < rowlen) v[t] = cache[(int64_t)slot * rowlen + gl];
        }
      }
#pragma unroll
      for (int t = 0; t < kSwapRows; ++t) {
        if (dst[t] < 0) continue;
        if (gl < rowlen) host[dst[t] * rowlen + gl] = v[t];
        if (gl == 0) {                      // maps of the victims this kernel moves (k_evict_stage does its own)
          inverted[dst[t]] = -1;
          cached_idx_map[victims[i + t]] = -1;
        }
      }
    } else {
      for (int t = 0; t < kSwapRows && i + t < k; ++t) {
        const int32_t slot = victims[i + t];
        const int32_t row = cached_idx_map[slot];
        copy_row(cache + (int64_t)slot * rowlen, host + (int64_t)row * rowlen, rowlen, gl, G);
        __builtin_amdgcn_wave_barrier();
        if (gl == 0) {
          inverted[row] = -1;
          cached_idx_map[slot] = -1;
        }
      }
    }
  }
}
// Full-duplex swap: victims' rows are first copied cache -> HBM staging (fast); k_swap then writes them to the
// host table from the staging buffer while its other workgroups read the missed rows -- PCIe carries both
// directions at once.  Victims beyond the staging capacity (rare) are
// written back directly by k_evict (`first` = staging capacity).
constexpr int kStageRowsInFlight = 4;   // rows in flight per lane group of the HBM-to-HBM row movers

// Steady state (the call began with no free slot: every missing row takes a victim's place): the free-slot list IS
// the victim list in ascending slot order.  Workgroup j of this role owns the 4096 slots k_victims' workgroup j
// counted (blk_vic[j]), adds up the counts before its own (<= a few hundred values out of L2), finds its victims again
// from the keys (key <= the threshold k_victims recorded) and writes their slots at base + rank: no scan of
// cached_idx_map, no launch of its own -- it rides in k_evict_stage's grid (which clears cached_idx_map meanwhile:
// nothing here reads it).
__device__ __forceinline__ void free_list_from_victims(const unsigned long long* __restrict__ keys, int64_t C,
                                                       const int32_t* __restrict__ blk_vic, int32_t* free_list,
                                                       const Ctl* ctl, int j) {
  if (ctl->status != CE_OK || ctl->k_evict == 0) return;
  const unsigned long long T = ctl->sel_prefix_after[0];
  const long long need = ctl->n_miss;
  __shared__ int part_s[4];
  __shared__ unsigned long long mask_s[64];
  __shared__ int pre_s[65];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int part = 0;
  for (int i = threadIdx.x; i < j; i += 256) part += blk_vic[i];
  part = wave_sum(part);
  if (lane == 0) part_s[wv] = part;
  // victims among 64 consecutive slots -> one 64-bit mask (wave wv takes the 64-slot groups wv, wv + 4, ...)
  const int64_t s0 = (int64_t)j * 4096;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int g = wv + 4 * u;
    const int64_t sl = s0 + (int64_t)g * 64 + lane;
    const unsigned long long key = sl < C ? keys[sl] : ~0ull;
    const unsigned long long m = __ballot(key <= T && key != ~0ull);
    if (lane == 0) mask_s[g] = m;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = __popcll(mask_s[threadIdx.x]);
    const int inc = wave_incl_scan(c, lane);
    pre_s[threadIdx.x] = inc - c;
  }
  __syncthreads();
  const long long base = (long long)part_s[0] + part_s[1] + part_s[2] + part_s[3];
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int g = wv + 4 * u;
    const unsigned long long m = mask_s[g];
    if ((m >> lane) & 1) {
      const long long pos = base + pre_s[g] + __popcll(m & lt);
      if (pos < need) free_list[pos] = (int32_t)(s0 + (int64_t)g * 64 + lane);
    }
  }
}

// Worker transport: "which rows did write-back job j stage, and where" -- so that the NEXT call's admission can take a
// row that job j evicted out of j's staging buffer (still intact in HBM) instead of waiting until the host has
// scattered it into the table.  One open-addressing table per job parity; an entry is tag << 32 | row with tag = the
// low 32 bits of the job number (never 0), its staging position in a parallel array.  A job treats every entry of
// another tag as free, so the tables are never cleared: job j's entries form gap-free probe runs (j only ever skips
// entries of its own) until job j + 2 starts overwriting them, by which time job j + 1's admission -- their only
// reader -- has finished.  At most stage_rows entries per job in >= 4 x stage_rows places.
struct EvTable {
  unsigned long long* keys;
  int32_t* pos;
  uint32_t mask;
};
__device__ __forceinline__ uint32_t evt_hash(int32_t row, uint32_t mask) {
  return (((uint32_t)row * 2654435761u) >> 7) & mask;
}
__device__ __forceinline__ void evt_insert(const EvTable t, uint32_t tag, int32_t row, int32_t pos) {
  const unsigned long long mine = ((unsigned long long)tag << 32) | (uint32_t)row;
  uint32_t h = evt_hash(row, t.mask);
  for (;;) {
    const unsigned long long cur = __hip_atomic_load(&t.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(cur >> 32) == tag) {          // taken by this job
      h = (h + 1) & t.mask;
      continue;
    }
    if (atomicCAS(&t.keys[h], cur, mine) == cur) {
      t.pos[h] = pos;
      return;
    }
  }
}
__device__ __forceinline__ int32_t evt_find(const unsigned long long* __restrict__ keys,
                                            const int32_t* __restrict__ pos, uint32_t mask, uint32_t tag,
                                            int32_t row) {
  uint32_t h = evt_hash(row, mask);
  for (;;) {
    const unsigned long long cur = keys[h];
    if ((uint32_t)(cur >> 32) != tag) return -1;
    if ((uint32_t)cur == (uint32_t)row) return pos[h];
    h = (h + 1) & mask;
  }
}

template <typename VT>
__global__ __launch_bounds__(256) void k_evict_stage(const int32_t* __restrict__ victims,
                                                     int32_t* cached_idx_map, int32_t* inverted,
                                                     const VT* __restrict__ cache, VT* stage, int32_t* stage_rows_idx,
                                                     long long cap, int rowlen, int g_log2, const Ctl* ctl,
                                                     WbMail* mail, long long job, int stage_grid,
                                                     const unsigned long long* __restrict__ keys, int64_t C,
                                                     const int32_t* __restrict__ blk_vic, int32_t* free_list,
                                                     EvTable evt, VT* host_overflow) {
  if ((int)blockIdx.x >= stage_grid) {       // (only launched with these workgroups in the steady-state form)
    free_list_from_victims(keys, C, blk_vic, free_list, ctl, (int)blockIdx.x - stage_grid);
    return;
  }
  const long long k_all = (ctl->status == CE_OK) ? ctl->k_evict : 0;
  long long k = k_all;
  if (k > cap) k = cap;
  if (host_overflow && k_all > cap) {
    // More victims than the staging holds (a cache larger than 262144 slots turning over in one call: never at the
    // bench sizes): the rest goes to the host table directly, a row per lane group at a time -- a loop in this grid
    // instead of a kernel of its own that every call launched to find nothing to do (round 5: -1 launch per call).
    const int G0 = 1 << g_log2;
    const int gl0 = threadIdx.x & (G0 - 1);
    const int64_t gs0 = ((int64_t)stage_grid * blockDim.x) >> g_log2;
    for (int64_t i = cap + (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2); i < k_all; i += gs0) {
      const int32_t slot = victims[i];
      const int32_t row = cached_idx_map[slot];
      if (row < 0) continue;             // (group-uniform; an unmapped victim slot writes nothing: ADVICE r5)
      copy_row(cache + (int64_t)slot * rowlen, host_overflow + (int64_t)row * rowlen, rowlen, gl0, G0);
      __builtin_amdgcn_wave_barrier();
      if (gl0 == 0) {
        inverted[row] = -1;
        cached_idx_map[slot] = -1;
      }
    }
  }
  if (mail && blockIdx.x == 0 && threadIdx.x == 0) {      // read by the worker after this kernel's event
    mail->count = k;
    mail->job = job;
  }
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)stage_grid * blockDim.x) >> g_log2;
  constexpr int R = kStageRowsInFlight;
  for (int64_t i = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2) * R; i < k; i += gstride * R) {
    if (rowlen <= G) {          // R rows in flight per lane group (one row at a time left this kernel latency bound)
      int32_t slot[R];
      VT v[R];
#pragma unroll
      for (int t = 0; t < R; ++t) slot[t] = i + t < k ? victims[i + t] : -1;
#pragma unroll
      for (int t = 0; t < R; ++t) {
        if (slot[t] < 0) continue;
        if (gl == 0) {                      // the victim's host row, then both maps cleared (was k_evict_maps)
          const int32_t row = cached_idx_map[slot[t]];
          stage_rows_idx[i + t] = row;
          inverted[row] = -1;
          cached_idx_map[slot[t]] = -1;
          if (evt.keys) evt_insert(evt, (uint32_t)job, row, (int32_t)(i + t));
        }
        if (gl < rowlen) v[t] = cache[(int64_t)slot[t] * rowlen + gl];
      }
#pragma unroll
      for (int t = 0; t < R; ++t)
        if (slot[t] >= 0 && gl < rowlen) stage[(i + t) * rowlen + gl] = v[t];
    } else {
      for (int t = 0; t < R && i + t < k; ++t) {
        const int32_t slot = victims[i + t];
        if (gl == 0) {
          const int32_t row = cached_idx_map[slot];
          stage_rows_idx[i + t] = row;
          inverted[row] = -1;
          cached_idx_map[slot] = -1;
          if (evt.keys) evt_insert(evt, (uint32_t)job, row, (int32_t)(i + t));
        }
        copy_row(cache + (int64_t)slot * rowlen, stage + (i + t) * rowlen, rowlen, gl, G);
      }
    }
  }
}

template <typename VT, int R>
__device__ __forceinline__ void writeback_rows(const int32_t* __restrict__ stage_rows_idx,
                                               const VT* __restrict__ stage, VT* host, long long cap, int rowlen,
                                               int g_log2, const Ctl* ctl, int block, int nblocks) {
  long long k = (ctl->status == CE_OK) ? ctl->k_evict : 0;
  if (k > cap) k = cap;
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)nblocks * blockDim.x) >> g_log2;
  for (int64_t i = (((int64_t)block * blockDim.x + threadIdx.x) >> g_log2) * R; i < k;
       i += gstride * R) {
    if (rowlen <= G) {
      VT v[R];
      int64_t dst[R];
#pragma unroll
      for (int t = 0; t < R; ++t) {
        dst[t] = -1;
        if (i + t < k) {
          dst[t] = stage_rows_idx[i + t];
          if (gl < rowlen) v[t] = stage[(i + t) * rowlen + gl];
        }
      }
#pragma unroll
      for (int t = 0; t < R; ++t)
        if (dst[t] >= 0 && gl < rowlen) host[dst[t] * rowlen + gl] = v[t];
    } else {
      for (int t = 0; t < R && i + t < k; ++t)
        copy_row(stage + (i + t) * rowlen, host + (int64_t)stage_rows_idx[i + t] * rowlen, rowlen, gl, G);
    }
  }
}

// map updates run after the payload pass (rows read cached_idx_map above)
__global__ __launch_bounds__(256) void k_evict_maps(const int32_t* __restrict__ victims, int32_t* cached_idx_map,
                                                    int32_t* inverted, int32_t* evicted_rows, const Ctl* ctl) {
  const long long k = (ctl->status == CE_OK) ? ctl->k_evict : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += stride) {
    const int32_t slot = victims[i];
    const int32_t row = cached_idx_map[slot];
    if (evicted_rows) evicted_rows[i] = row;
    inverted[row] = -1;
    cached_idx_map[slot] = -1;
  }
}

__global__ __launch_bounds__(256) void k_free_count(const int32_t* __restrict__ cached_idx_map, int64_t C,
                                                    int32_t* blk_free, const Ctl* ctl) {
  if (ctl->status != CE_OK || ctl->n_miss == 0) return;
  const int64_t s0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  int f = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (s0 + t < C) f += (cached_idx_map[s0 + t] < 0);
  __shared__ int sf[4];
  f = wave_sum(f);
  if ((threadIdx.x & 63) == 0) sf[threadIdx.x >> 6] = f;
  __syncthreads();
  if (threadIdx.x == 0) blk_free[blockIdx.x] = sf[0] + sf[1] + sf[2] + sf[3];
}

// every workgroup adds up the free counts of the blocks before its own (a few KB out of L2): no scan kernel
__global__ __launch_bounds__(256) void k_free_emit(const int32_t* __restrict__ cached_idx_map, int64_t C,
                                                   const int32_t* __restrict__ blk_free, int32_t* free_list,
                                                   const Ctl* ctl) {
  if (ctl->status != CE_OK || ctl->n_miss == 0) return;
  const long long need = ctl->n_miss;
  __shared__ int wsum4[4];
  int part = 0;                       // free slots before this block: < cuda_row_num < 2^31
  for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) part += blk_free[i];
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) wsum4[threadIdx.x >> 6] = part;
  __syncthreads();
  const long long before = (long long)wsum4[0] + wsum4[1] + wsum4[2] + wsum4[3];
  if (before >= need) return;   // block-uniform
  const int64_t s0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  int fl[4];
  int f = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    fl[t] = (s0 + t < C) && (cached_idx_map[s0 + t] < 0);
    f += fl[t];
  }
  int tot;
  long long pos = block_excl_scan_256(f, &tot) + before;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (fl[t]) {
      if (pos < need) free_list[pos] = (int32_t)(s0 + t);
      ++pos;
    }
  }
}

// small caches: count, scan and emit of the free-slot list in ONE workgroup (three launches otherwise; a
// prefetch_num = 1 step is a chain of such launches).  Walks the slots in order, 4096 per round, and stops as soon
// as the first n_miss free slots are out.
__global__ __launch_bounds__(1024) void k_free_single(const int32_t* __restrict__ cached_idx_map, int64_t C,
                                                      int32_t* free_list, const Ctl* ctl) {
  if (ctl->status != CE_OK || ctl->n_miss == 0) return;
  const long long need = ctl->n_miss;
  __shared__ int wtot[16];
  __shared__ long long base_s;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int64_t s0 = 0; s0 < C; s0 += 4096) {
    const long long base = base_s;
    if (base >= need) break;                       // block-uniform
    const int64_t i0 = s0 + (int64_t)threadIdx.x * 4;
    int fl[4], f = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fl[t] = (i0 + t < C) && (cached_idx_map[i0 + t] < 0);
      f += fl[t];
    }
    const int inc = wave_incl_scan(f, lane);
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    int pre = 0, tot = 0;
    for (int k = 0; k < 16; ++k) {
      if (k < w) pre += wtot[k];
      tot += wtot[k];
    }
    long long pos = base + pre + inc - f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (fl[t]) {
        if (pos < need) free_list[pos] = (int32_t)(i0 + t);
        ++pos;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) base_s = base + tot;
    __syncthreads();
  }
}

// rows[i] -> slots[i] for first <= i < n (slots == nullptr: slot i; rows == nullptr: row i)
template <typename VT, int R>
__device__ __forceinline__ void admit_rows(const int32_t* __restrict__ rows, const int32_t* __restrict__ slots,
                                           const long long* n_ptr, long long n_imm, const VT* __restrict__ host,
                                           VT* cache, int rowlen, int g_log2, const Ctl* ctl, int block, int nblocks,
                                           long long first = 0) {
  if (ctl && ctl->status != CE_OK) return;
  const long long n = n_ptr ? *n_ptr : n_imm;
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)nblocks * blockDim.x) >> g_log2;
  for (int64_t i = first + (((int64_t)block * blockDim.x + threadIdx.x) >> g_log2) * R; i < n; i += gstride * R) {
    if (rowlen <= G) {          // R host rows in flight per lane group (see k_evict)
      VT v[R];
      int64_t dst[R];
#pragma unroll
      for (int t = 0; t < R; ++t) {
        dst[t] = -1;
        if (i + t < n) {
          const int64_t row = rows ? rows[i + t] : i + t;
          dst[t] = slots ? slots[i + t] : i + t;
          if (gl < rowlen) v[t] = host[row * rowlen + gl];
        }
      }
#pragma unroll
      for (int t = 0; t < R; ++t)
        if (dst[t] >= 0 && gl < rowlen) cache[dst[t] * rowlen + gl] = v[t];
    } else {
      for (int t = 0; t < R && i + t < n; ++t) {
        const int64_t row = rows ? rows[i + t] : i + t;
        const int64_t slot = slots ? slots[i + t] : i + t;
        copy_row(host + row * rowlen, cache + slot * rowlen, rowlen, gl, G);
      }
    }
  }
}

template <typename VT, int R = kSwapRows>
__global__ __launch_bounds__(1024) void k_admit(const int32_t* __restrict__ rows, const int32_t* __restrict__ slots,
                                               const long long* n_ptr, long long n_imm,
                                               const VT* __restrict__ host, VT* cache, int rowlen, int g_log2,
                                               const Ctl* ctl, long long first) {
  admit_rows<VT, R>(rows, slots, n_ptr, n_imm, host, cache, rowlen, g_log2, ctl, (int)blockIdx.x,
                    (int)gridDim.x, first);
}

// Admission kernel of the worker transport: the missed rows of a call, host table -> in_stage, on the admission stream.
// The previous call's write-back need not have landed: row i comes out of THAT job's staging buffer if the job
// evicted it (EvTable above; evt_keys == NULL: there is no such job), out of the host table otherwise.  n_ptr: the
// count the call's plan left on the device (the launch thread never learns it).
template <typename VT, int R = kSwapRows>
__global__ __launch_bounds__(1024) void k_admit_probe(const int32_t* __restrict__ rows, const long long* n_ptr,
                                                     const VT* __restrict__ host, VT* dst, int rowlen, int g_log2,
                                                     const unsigned long long* __restrict__ evt_keys,
                                                     const int32_t* __restrict__ evt_pos, uint32_t evt_mask,
                                                     uint32_t tag, const VT* __restrict__ prev_stage) {
  const long long n = *n_ptr;
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> g_log2;
  for (int64_t i = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2) * R; i < n; i += gstride * R) {
    if (rowlen <= G) {
      VT v[R];
#pragma unroll
      for (int t = 0; t < R; ++t) {
        if (i + t < n) {
          const int32_t row = rows[i + t];
          const int32_t p = evt_keys ? evt_find(evt_keys, evt_pos, evt_mask, tag, row) : -1;
          const VT* src = p >= 0 ? prev_stage + (int64_t)p * rowlen : host + (int64_t)row * rowlen;
          if (gl < rowlen) v[t] = src[gl];
        }
      }
#pragma unroll
      for (int t = 0; t < R; ++t)
        if (i + t < n && gl < rowlen) dst[(i + t) * rowlen + gl] = v[t];
    } else {
      for (int t = 0; t < R && i + t < n; ++t) {
        const int32_t row = rows[i + t];
        const int32_t p = evt_keys ? evt_find(evt_keys, evt_pos, evt_mask, tag, row) : -1;
        copy_row(p >= 0 ? prev_stage + (int64_t)p * rowlen : host + (int64_t)row * rowlen, dst + (i + t) * rowlen,
                 rowlen, gl, G);
      }
    }
  }
}

// worker transport, host-gather admission: rows [0, min(n_miss, cap)) arrived contiguously in `in_stage`; move them to
// their slots.  (The chained admission's form is k_unpack_chained, ce_cache_fused.h.)
template <typename VT>
__global__ __launch_bounds__(256) void k_unpack_admitted(const int32_t* __restrict__ slots, const long long* n_ptr,
                                                         long long cap, const VT* __restrict__ in_stage, VT* cache,
                                                         int rowlen, int g_log2, Ctl* ctl,
                                                         const unsigned long long* fail_word, long long job,
                                                         const int32_t* __restrict__ rows,
                                                         const VT* __restrict__ host_overflow) {
  const bool ok = ctl->status == CE_OK;         // (nothing in this kernel writes ctl->status)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // The admission worker flags a job whose rows did not arrive (a HIP call of its own failed or timed out) in a
    // word of pinned host memory.  ONE thread fetches it over PCIe and leaves the verdict in the control block for
    // k_admit_maps, which then marks nothing resident; whatever this kernel copies into the (free) slots meanwhile
    // is never looked at.
    ctl->lost = *(volatile const unsigned long long*)fail_word == (unsigned long long)job;
  }
  if (!ok) return;
  long long n = *n_ptr;
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> g_log2;
  if (host_overflow && n > cap) {
    // more misses than the staging holds (rare): the rest is read zero-copy out of the host table -- here, behind the
    // parked wait (the host gather has waited for every earlier write-back), in this grid instead of a launch of its own
    for (int64_t i = cap + (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2); i < n; i += gstride)
      copy_row(host_overflow + (int64_t)rows[i] * rowlen, cache + (int64_t)slots[i] * rowlen, rowlen, gl, G);
  }
  if (n > cap) n = cap;
  constexpr int R = kStageRowsInFlight;
  for (int64_t i = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2) * R; i < n; i += gstride * R) {
    if (rowlen <= G) {
      int32_t slot[R];
      VT v[R];
#pragma unroll
      for (int t = 0; t < R; ++t) {
        slot[t] = i + t < n ? slots[i + t] : -1;
        if (slot[t] >= 0 && gl < rowlen) v[t] = in_stage[(i + t) * rowlen + gl];
      }
#pragma unroll
      for (int t = 0; t < R; ++t)
        if (slot[t] >= 0 && gl < rowlen) cache[(int64_t)slot[t] * rowlen + gl] = v[t];
    } else {
      for (int t = 0; t < R && i + t < n; ++t)
        copy_row(in_stage + (i + t) * rowlen, cache + (int64_t)slots[i + t] * rowlen, rowlen, gl, G);
    }
  }
}

// Full-duplex swap in ONE launch: the first wb_blocks workgroups stream the staged victims to the host table,
// the others read the missed rows from it.  (An earlier version ran the write-back on an auxiliary stream; HIP
// multiplexes streams onto a few hardware queues and that stream could land on the TRAINING stream's queue,
// stalling training for the whole write-back -- seen in a rocprofv3 timeline.  One kernel needs no extra stream.)
template <typename VT, int R>
__global__ __launch_bounds__(1024) void k_swap(const int32_t* __restrict__ stage_rows_idx,
                                              const VT* __restrict__ stage, long long cap, int wb_blocks,
                                              const int32_t* __restrict__ rows, const int32_t* __restrict__ slots,
                                              const long long* n_ptr, VT* host, VT* cache, int rowlen, int g_log2,
                                              const Ctl* ctl) {
  if ((int)blockIdx.x < wb_blocks)
    writeback_rows<VT, R>(stage_rows_idx, stage, host, cap, rowlen, g_log2, ctl, (int)blockIdx.x, wb_blocks);
  else
    admit_rows<VT, R>(rows, slots, n_ptr, 0ll, (const VT*)host, cache, rowlen, g_log2, ctl,
                      (int)blockIdx.x - wb_blocks, (int)gridDim.x - wb_blocks);
}

__global__ __launch_bounds__(256) void k_admit_maps(const int32_t* __restrict__ rows,
                                                    const int32_t* __restrict__ slots, const long long* n_ptr,
                                                    long long n_imm, int32_t* cached_idx_map, int32_t* inverted,
                                                    int64_t* freq, const int64_t* freq_vals, int32_t* slot_epoch,
                                                    int32_t epoch_imm, Ctl* ctl, ce_call_stats_t* ring,
                                                    long long seq_arg, const unsigned long long* fail_word,
                                                    long long job, long long* n_unpack_out = nullptr) {
  // ring == NULL (preload): no record to publish, the epoch is the caller's constant
  const long long seq = ring ? call_seq(ctl, seq_arg) : 0;
  const int32_t epoch = ring ? call_epoch(seq) : epoch_imm;
  ce_call_stats_t* const ring_slot = ring ? ring + (seq % kRing) : nullptr;
  // host-gather admission: the worker reports a job it could not complete (failed / timed-out HIP call): the rows
  // never arrived, so nothing may be marked resident.
  const bool lost = fail_word && ctl->lost != 0;      // left by k_unpack_admitted (the kernel before this one)
  // last kernel of prepare_ids that can change the call's record: publish it (a slot whose seq matches is complete)
  if (ring_slot && blockIdx.x == 0 && threadIdx.x == 0) {
    // chained admission: the rows the unpack kernel moves for this call (it runs on the admission stream, possibly
    // while the next call's front rewrites the control block)
    if (n_unpack_out) *n_unpack_out = (ctl->status == CE_OK && n_ptr) ? *n_ptr : 0;
    if (lost && ctl->status == CE_OK) {
      // the victims are gone (written back) but their slots stay free: undo the plan's share of the free count
      ctl->n_free = ctl->n_free + ctl->n_miss;
      ring_slot->status = CE_ERR_HIP;
      ring_slot->n_free_after = ctl->n_free;
      __hip_atomic_store(&ctl->status, CE_ERR_HIP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence_system();
    *(volatile long long*)&ring_slot->seq = seq;
  }
  if (lost) return;
  if (ctl && __hip_atomic_load(&ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != CE_OK) return;
  const long long n = n_ptr ? *n_ptr : n_imm;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int32_t row = rows ? rows[i] : (int32_t)i;
    const int32_t slot = slots ? slots[i] : (int32_t)i;
    cached_idx_map[slot] = row;
    inverted[row] = slot;
    if (freq) freq[slot] = freq_vals ? freq_vals[i] : 0;
    slot_epoch[slot] = epoch;
  }
}

// _id_to_cached_cuda_id alone (ce_cache_lookup_slots): inverted[idx_map[id]]
__global__ __launch_bounds__(256) void k_lookup(const int64_t* __restrict__ ids, int64_t n,
                                                const int32_t* __restrict__ idx_map,
                                                const int32_t* __restrict__ inverted, int64_t N, int64_t* slots_out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t id = ids[i];
    int64_t slot = -1;
    if ((unsigned long long)id < (unsigned long long)N) slot = inverted[idx_map ? idx_map[id] : (int32_t)id];
    slots_out[i] = slot;
  }
}

// Last kernel of prepare_ids: k_mark left the ROW of every id in `slots` (-1 = bad id); turn it into the slot in
// place -- one random 4-byte gather per id instead of the two dependent ones of inverted[idx_map[id]] [A.6].
// A failed call (overflow / bad id) changes nothing but still hands back well-defined slots (-1): callers that skip
// the status check (strict=False) then gather zero rows instead of garbage.
__global__ __launch_bounds__(256) void k_slots(int64_t* slots, int64_t n, const int32_t* __restrict__ inverted,
                                               const Ctl* ctl) {
  const bool failed = ctl->status != CE_OK;
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * U;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x * U + threadIdx.x; i0 < n; i0 += stride) {
    int64_t row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      row[u] = i < n ? slots[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      if (i < n) slots[i] = (failed || row[u] < 0) ? -1 : (int64_t)inverted[row[u]];
    }
  }
}

// LFU form of k_slots: slots + `freq[slot] += multiplicity` [A.3-7].  A hot slot collects >100 k lookups per
// window and same-address device atomics serialise at the memory side (~7 ns each): with one merged atomic per
// WAVE the hottest counter alone still took 53 k of them (k_slots 381 us per window).  Here every workgroup owns
// a contiguous range of lookups and counts them in an LDS hash table (slot -> count, open addressing); only the
// table's entries go to memory, so a counter sees at most one atomic per workgroup.  Four lookups per thread are
// in flight to cover the random load.
constexpr int kSlotsHashBits = 13;
constexpr int kSlotsHash = 1 << kSlotsHashBits;
__global__ __launch_bounds__(1024) void k_slots_lfu(int64_t* slots, int64_t n, const int32_t* __restrict__ inverted,
                                                    int64_t* freq, const Ctl* ctl) {
  __shared__ int hkey[kSlotsHash];
  __shared__ int hcnt[kSlotsHash];
  for (int i = threadIdx.x; i < kSlotsHash; i += blockDim.x) { hkey[i] = -1; hcnt[i] = 0; }
  __syncthreads();
  const bool failed = ctl->status != CE_OK;
  const int64_t per_block = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per_block;
  const int64_t hi = lo + per_block < n ? lo + per_block : n;
  constexpr int U = 4;
  for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)blockDim.x * U) {
    int64_t row[U];
    int slot[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      row[u] = (i < hi && !failed) ? slots[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) slot[u] = row[u] >= 0 ? inverted[row[u]] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      if (i < hi) slots[i] = slot[u];
      if (slot[u] < 0) continue;
      unsigned h = ((unsigned)slot[u] * 2654435761u) >> (32 - kSlotsHashBits);
      bool done = false;
      for (int p = 0; p < 16 && !done; ++p) {
        const int old = atomicCAS(&hkey[h], -1, slot[u]);
        if (old == -1 || old == slot[u]) {
          atomicAdd(&hcnt[h], 1);
          done = true;
        } else {
          h = (h + 1) & (kSlotsHash - 1);
        }
      }
      if (!done) atomicAdd((unsigned long long*)&freq[slot[u]], 1ull);      // table crowded: count directly
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kSlotsHash; i += blockDim.x)
    if (hkey[i] >= 0) atomicAdd((unsigned long long*)&freq[hkey[i]], (unsigned long long)hcnt[i]);
}

template <typename VT>
__global__ __launch_bounds__(256) void k_flush_rows(const int32_t* __restrict__ cached_idx_map, int64_t C,
                                                    const VT* __restrict__ cache, VT* host, int rowlen,
                                                    int g_log2) {
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> g_log2;
  for (int64_t s = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2; s < C; s += gstride) {
    const int32_t row = cached_idx_map[s];
    if (row >= 0) copy_row(cache + s * rowlen, host + (int64_t)row * rowlen, rowlen, gl, G);
  }
}

__global__ __launch_bounds__(256) void k_flush_maps(int32_t* cached_idx_map, int64_t C, int32_t* inverted,
                                                    int64_t* freq, int32_t* slot_epoch, Ctl* ctl) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int n = 0;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < C; s += stride) {
    const int32_t row = cached_idx_map[s];
    if (row >= 0) {
      inverted[row] = -1;
      cached_idx_map[s] = -1;
      ++n;
    }
    if (freq) freq[s] = INT64_MAX;
    slot_epoch[s] = kEpochNever;
  }
  n = wave_sum(n);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd((unsigned long long*)&ctl->k_evict, (unsigned long long)n);
}

__global__ void k_flush_end(int64_t C, Ctl* ctl, ce_call_stats_t* ring_slot, long long seq) {
  ctl->seq = seq;
  ctl->n_free = C;
  ring_slot->n_ids = 0;
  ring_slot->n_unique = 0;
  ring_slot->n_miss = 0;
  ring_slot->n_evict = ctl->k_evict;
  ring_slot->miss_lookups = 0;
  ring_slot->n_free_after = C;
  ring_slot->status = CE_OK;
  ring_slot->kind = CE_CALL_FLUSH;
  __threadfence_system();
  ring_slot->seq = seq;
}

__global__ void k_preload_end(long long n, Ctl* ctl, ce_call_stats_t* ring_slot, long long seq) {
  ctl->seq = seq;
  ctl->n_free -= n;
  ring_slot->n_ids = 0;
  ring_slot->n_unique = n;
  ring_slot->n_miss = n;
  ring_slot->n_evict = 0;
  ring_slot->miss_lookups = 0;
  ring_slot->n_free_after = ctl->n_free;
  ring_slot->status = CE_OK;
  ring_slot->kind = CE_CALL_PRELOAD;
  __threadfence_system();
  ring_slot->seq = seq;
}

__global__ __launch_bounds__(256) void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__(256) void k_fill_i64(int64_t* p, int64_t n, int64_t v) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// staged transport helpers: pack rows (by slot list) into / out of a contiguous device buffer
template <typename VT>
__global__ __launch_bounds__(256) void k_pack_rows(const int32_t* __restrict__ slots, long long n,
                                                   const VT* __restrict__ cache, VT* staging, int rowlen,
                                                   int g_log2) {
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> g_log2;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2; i < n; i += gstride)
    copy_row(cache + (int64_t)slots[i] * rowlen, staging + i * rowlen, rowlen, gl, G);
}
template <typename VT>
__global__ __launch_bounds__(256) void k_unpack_rows(const int32_t* __restrict__ slots, long long n,
                                                     const VT* __restrict__ staging, VT* cache, int rowlen,
                                                     int g_log2) {
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> g_log2;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2; i < n; i += gstride)
    copy_row(staging + i * rowlen, cache + (int64_t)slots[i] * rowlen, rowlen, gl, G);
}

}  // namespace ce
