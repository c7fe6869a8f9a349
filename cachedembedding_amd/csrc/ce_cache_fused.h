// Merged forms of the cache op's index kernels (round 6).  Included by ce_cache.hip behind the per-phase kernels whose
// helpers they use (miss masks, select_level, EvTable, copy_row).
//
// One prepare_ids used to be 11-13 dependent launches (reference: one torch op + a device sync per phase,
// recsys/dlrm_main.py:259 -> upstream CachedParamMgr.prepare_ids).  At prefetch_num = 1 and in the
// reference's own micro-benchmark shape (benchmark/benchmark_cache.py:58-72) the chain -- not the bag kernels -- is
// the step, and what a kernel of the chain costs there is its number of DEPENDENT memory round trips (2-3 us each
// beside the training kernels), not its bytes.  Two merges take round trips out without putting anything in:
//
//   k_touch         the front per LOOKUP instead of per table row, for calls of at most cuda_row_num ids: see below
//   k_miss_rank     (round 6, first form: k_emit_scan, count + emit in one pass over the bitmap with decoupled
//                   look-back -- still N / 8 bytes and one workgroup per 32768 rows per call; replaced)
//   k_rank_victims  for a FULL cache (the call evicts exactly as many rows as it misses and the slots to fill are its
//   k_stage_maps    victims): victims per 4096-slot block counted and ranked by the same look-back, written as the
//                   ascending free-slot list; then, over that list, their rows staged for the write-back AND both maps
//                   rewritten for the rows that take their slots -- k_victims + k_evict_stage + its free-list
//                   workgroups + k_admit_maps in two launches, no returning atomic.
//
// (Tried first and measured slower, alone and beside training: the same phases as ONE launch each with a grid barrier
// between co-resident workgroups -- begin + mark + count + emit 107 us against 69 for the four launches, keys + passes
// + victims + staging 244 against 87 at Kaggle 5 %: a software barrier costs a release (the XCD's L2 written back), a
// returning atomic, a poll and an acquire -- two kernel boundaries' worth -- and the few fat workgroups that can be
// co-resident keep fewer loads in flight than the launch-sized grids.  docs/history.md.)
#pragma once

namespace ce {

// ---- decoupled look-back.  One 64-bit word per workgroup: tag (which call wrote it) | state | two counts.  A word of
// another tag is "not there yet"; every call of this form rewrites every word, so no kernel ever clears them.  Written
// and polled with relaxed agent-scope accesses: the payload IS the word, nothing else is handed over.
constexpr unsigned long long kLbAggregate = 1ull, kLbPrefix = 2ull;
__device__ __forceinline__ unsigned long long lb_pack(unsigned tag, unsigned long long state, unsigned a, unsigned b) {
  return ((unsigned long long)(tag & 0xffffu) << 48) | (state << 46) | ((unsigned long long)(a & 0x7fffffu) << 23) |
         (unsigned long long)(b & 0x7fffffu);
}
// Called by the 64 lanes of ONE wave of workgroup j after the workgroup's counts (a, b) are final: publishes them, adds
// up the counts of the workgroups before it -- 256 words per round trip, four loads in flight per lane: all workgroups of
// these launches are resident at once, so what a late workgroup finds are mostly aggregates, hundreds of them -- and
// publishes the inclusive prefix.  Workgroups are dispatched in index order, so every predecessor is running or done;
// the spin is bounded all the same (a trap instead of a hung device).  Every lane returns the exclusive prefix.
__device__ __forceinline__ void lb_scan_wave(unsigned long long* words, int j, unsigned tag, unsigned a, unsigned b,
                                             unsigned* base_a, unsigned* base_b) {
  const int lane = threadIdx.x & 63;
  unsigned sa = 0, sb = 0;
  if (j > 0) {
    if (lane == 0)
      __hip_atomic_store(&words[j], lb_pack(tag, kLbAggregate, a, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = wall_clock64();
    int i = j - 1;                       // nearest predecessor not yet consumed
    bool done = false;
    while (!done) {
      unsigned long long w[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = i - 64 * q - lane;
        // (below index 0: "an inclusive prefix of zero" ends the walk)
        w[q] = idx >= 0 ? __hip_atomic_load(&words[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                        : lb_pack(tag, kLbPrefix, 0u, 0u);
      }
      bool again = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (done || again) continue;
        const unsigned long long st = (w[q] >> 46) & 3ull;
        const bool valid = (unsigned)(w[q] >> 48) == (tag & 0xffffu) && st != 0;
        const unsigned long long m_inv = __ballot(!valid), m_pre = __ballot(valid && st == kLbPrefix);
        const unsigned long long m_stop = m_inv | m_pre;
        const int stop = m_stop ? __ffsll((long long)m_stop) - 1 : 64;      // nearest lane that ends or stalls the walk
        const bool stop_is_prefix = m_stop && ((m_pre >> stop) & 1ull);
        const bool take = lane < stop || (lane == stop && stop_is_prefix);
        unsigned xa = take ? (unsigned)(w[q] >> 23) & 0x7fffffu : 0u, xb = take ? (unsigned)w[q] & 0x7fffffu : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
          xa += __shfl_xor(xa, d);
          xb += __shfl_xor(xb, d);
        }
        sa += xa;
        sb += xb;
        if (stop_is_prefix) {
          done = true;
        } else if (m_stop) {             // a word that is not there yet: come back for it
          i = i - 64 * q - stop;
          again = true;
        }
      }
      if (!done && !again) i -= 256;
      if (again) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 400000000ull) __builtin_trap();
      }
    }
  }
  if (lane == 0)
    __hip_atomic_store(&words[j], lb_pack(tag, kLbPrefix, sa + a, sb + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *base_a = sa;
  *base_b = sb;
}

// ---- the per-lookup front: calls of at most cuda_row_num ids (prefetch_num 1-2, the reference's micro-benchmark shape)
//
// k_mark + k_count / k_emit read and rewrite a bitmap of the whole TABLE to learn which rows a call names: N / 8 bytes
// scanned per call whatever the call's size, one workgroup per 32768 rows with a look-back chain through all of them
// (83 us of a 426 k-id call on the Kaggle table, most of it waiting).  A small call is better served per LOOKUP:
//
//   k_touch      id -> row -> slot.  A resident row's slot gets the call's stamp (a plain store: repeats are harmless, so
//                the hot rows cost nothing); a missing row is test-and-set in the bitmap -- the thread that finds the
//                bit clear owns the row, appends it to an unordered list (one returning atomic per WORKGROUP) and
//                counts it in the 1024-row and 32768-row counters of its place in the table.  Misses are a few per
//                cent of the lookups and rarely repeat, so the returning atomics that sank "count in k_mark" (round 3:
//                every lookup of every hot row) are not on this path.
//   k_miss_rank  the list in ascending order = every row at its RANK: (missing rows in the chunks before its chunk: a
//                prefix over at most 16384 counters, recomputed by every workgroup in LDS) + (in the 1024-row groups
//                before its group: <= 31 counters) + (bits below it in its group: <= 32 bitmap words) -- sixteen
//                independent 16-byte loads per missing row, no sort, no pass over the table.  The distinct resident rows
//                are counted off the stamps (a pass over the cache's 4-byte stamps), the radix histograms cleared, and
//                the workgroup that finishes last writes the plan and the record -- what k_begin, k_emit_scan's
//                look-back and its last workgroup did.
//
// The bits, counters of the missing rows are cleared by the next kernel of the call (k_keys' prologue) along the list.
// A failed call (bad id) keeps its stamps: with protect_depth > 0 its resident rows stay protected for that many
// calls more, which only ever keeps rows.

// One pass, one iteration: workgroup b owns the ids [b * 1024 * U, (b + 1) * 1024 * U), a wave U * 64 consecutive ones.
template <int U>
__global__ __launch_bounds__(1024) void k_touch(const int64_t* __restrict__ ids, int64_t n,
                                               const int32_t* __restrict__ idx_map,
                                               const int32_t* __restrict__ inverted, int64_t N, uint32_t* bitmap,
                                               int32_t* fine, int32_t* coarse, int32_t* slot_epoch, long long seq_arg,
                                               FrontWords* fw, int32_t* miss_tmp, unsigned miss_cap, int64_t* rows_out,
                                               int allow_pad) {
  __shared__ int wcnt[16], wcold[16];
  __shared__ unsigned base_s;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
  const int32_t epoch = call_epoch(seq_arg);
  const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63)) * U;
  int32_t row[U], inv[U];
  bool valid[U], first[U];
  bool bad = false;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = i0 + u * 64 + lane;
    valid[u] = i < n;
    row[u] = 0;
    if (valid[u]) {
      const int64_t id = ids[i];
      if ((unsigned long long)id >= (unsigned long long)N) {
        if (!(allow_pad && id == -1)) bad = true;       // (k_mark: -1 is padding on the padded entry point only)
        valid[u] = false;
        rows_out[i] = -1;
      } else {
        row[u] = idx_map ? idx_map[id] : (int32_t)id;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    inv[u] = 0;
    if (valid[u]) {
      rows_out[i0 + u * 64 + lane] = row[u];
      inv[u] = inverted[row[u]];
    }
  }
  int cold = 0, mine = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    first[u] = false;
    if (!valid[u]) continue;
    if (inv[u] >= 0) {
      slot_epoch[inv[u]] = epoch;                       // evict_backlist membership [A.3-3]
    } else {
      ++cold;
      const uint32_t bit = 1u << (row[u] & 31);
      const uint32_t old = atomicOr(bitmap + (row[u] >> 5), bit);
      first[u] = (old & bit) == 0;
      mine += first[u] ? 1 : 0;
    }
  }
  const int winc = wave_incl_scan(mine, lane);
  cold = wave_sum(cold);
  if (lane == 63) wcnt[wv] = winc;
  if (lane == 0) wcold[wv] = cold;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0, tc = 0;
    for (int k = 0; k < nw; ++k) {
      tot += wcnt[k];
      tc += wcold[k];
    }
    base_s = tot ? atomicAdd(&fw->n_miss, (unsigned)tot) : 0u;
    if (tc) atomicAdd(&fw->miss_lookups, (unsigned long long)tc);
  }
  __syncthreads();
  unsigned pos = base_s + (unsigned)(winc - mine);
  for (int k = 0; k < wv; ++k) pos += (unsigned)wcnt[k];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (!first[u]) continue;
    if (pos < miss_cap) miss_tmp[pos] = row[u];        // (beyond the list: the call fails, k_miss_rank says so)
    ++pos;
    atomicAdd(fine + (row[u] >> kFineShift), 1);
    atomicAdd(coarse + (row[u] >> kChunkShift), 1);
  }
  if (bad) fw->bad = 1;
}

// (only when a call failed on the host between its front and its k_keys: the next call's front clears up first)
__global__ __launch_bounds__(256) void k_front_cleanup(const int32_t* __restrict__ miss_tmp, const FrontWords* fw,
                                                       uint32_t* bitmap, int32_t* fine, int32_t* coarse, int n_chunks) {
  front_cleanup(miss_tmp, fw, bitmap, fine, coarse, n_chunks);
}

__global__ __launch_bounds__(256) void k_miss_rank(const int32_t* __restrict__ miss_tmp, FrontWords* fw,
                                                   FrontWords* fw_next, const uint32_t* __restrict__ bitmap,
                                                   const int32_t* __restrict__ fine,
                                                   const int32_t* __restrict__ coarse, int n_chunks,
                                                   unsigned miss_cap, int32_t* miss_list,
                                                   const int32_t* __restrict__ slot_epoch,
                                                   int64_t C, uint32_t* hist, long long seq_arg, Ctl* ctl,
                                                   int64_t n_ids, ce_call_stats_t* ring, long long in_cap,
                                                   int assume_free0, long long* n_admit_out) {
  extern __shared__ int cbase[];                       // [n_chunks]: missing rows in the chunks before
  __shared__ int wtot[4];
  __shared__ int last_s;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned gtid = blockIdx.x * 256u + threadIdx.x, gsize = gridDim.x * 256u;
  const unsigned m_all = fw->n_miss;                   // (final: k_touch is a launch ago)
  const unsigned m = m_all <= miss_cap ? m_all : 0u;   // beyond the list: the call fails, nothing is ranked
  const int32_t epoch = call_epoch(seq_arg);
  // the stamps first: their loads are in flight while the prefix is worked out
  int hits = 0;
  {
    const int64_t c4 = C >> 2;
    const int4* const e4 = (const int4*)slot_epoch;
    for (int64_t q = gtid; q < c4; q += gsize) {
      const int4 x = e4[q];
      hits += (x.x == epoch) + (x.y == epoch) + (x.z == epoch) + (x.w == epoch);
    }
    for (int64_t s_ = (c4 << 2) + gtid; s_ < C; s_ += gsize) hits += slot_epoch[s_] == epoch;
  }
  for (unsigned i = gtid; i < (unsigned)kHistWords; i += gsize) hist[i] = 0;
  if (m) {                                             // (uniform over the grid)
    const int per = (n_chunks + 255) >> 8;
    const int c0 = (int)threadIdx.x * per;
    int sum = 0;
    for (int j = 0; j < per; ++j)
      if (c0 + j < n_chunks) sum += coarse[c0 + j];
    const int inc = wave_incl_scan(sum, lane);
    if (lane == 63) wtot[wv] = inc;
    __syncthreads();
    int pre = inc - sum;
    for (int k = 0; k < wv; ++k) pre += wtot[k];
    for (int j = 0; j < per; ++j)
      if (c0 + j < n_chunks) {
        cbase[c0 + j] = pre;
        pre += coarse[c0 + j];
      }
    __syncthreads();
    for (unsigned i = gtid; i < m; i += gsize) {
      const int32_t r = miss_tmp[i];
      const int c = r >> kChunkShift, f = r >> kFineShift;
      const int nf = f - c * kFinePerChunk;            // 1024-row groups of the chunk before the row's own
      const int4* const fp = (const int4*)(fine + (int64_t)c * kFinePerChunk);
      const uint4* const bp = (const uint4*)(bitmap + ((int64_t)f << (kFineShift - 5)));
      int4 fc[kFinePerChunk / 4];
      uint4 bw[8];
#pragma unroll
      for (int j = 0; j < kFinePerChunk / 4; ++j) fc[j] = fp[j];
#pragma unroll
      for (int j = 0; j < 8; ++j) bw[j] = bp[j];
      int cnt = cbase[c];
#pragma unroll
      for (int j = 0; j < kFinePerChunk / 4; ++j) {
        cnt += (4 * j + 0 < nf ? fc[j].x : 0) + (4 * j + 1 < nf ? fc[j].y : 0) + (4 * j + 2 < nf ? fc[j].z : 0) +
               (4 * j + 3 < nf ? fc[j].w : 0);
      }
      const int w = (r >> 5) & 31;
      const uint32_t below = (1u << (r & 31)) - 1u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t wd[4] = {bw[j].x, bw[j].y, bw[j].z, bw[j].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int wi = 4 * j + q;
          cnt += wi < w ? __popc(wd[q]) : (wi == w ? __popc(wd[q] & below) : 0);
        }
      }
      miss_list[cnt] = r;
    }
  }
  hits = wave_sum(hits);
  __syncthreads();                                     // (wtot is reused)
  if (lane == 0) wtot[wv] = hits;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (tot) __hip_atomic_fetch_add(&fw->hit_unique, (unsigned)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned ticket = __hip_atomic_fetch_add(&fw->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last_s = ticket == gridDim.x - 1;
    if (last_s) {
      // ---- the plan (k_begin's reset + k_emit's workgroup 0): every workgroup's count is in
      const long long hu = __hip_atomic_load(&fw->hit_unique, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ce_call_stats_t* const ring_slot = ring + (seq_arg % kRing);
      const long long free0 = ctl->n_free;
      const long long tu = hu + m_all;
      int status = fw->bad ? CE_ERR_RANGE : CE_OK;
      if (status == CE_OK && (tu > C || m_all > miss_cap)) status = CE_ERR_CAPACITY;    // [A.3-2]
      if (status == CE_OK && assume_free0 && free0 != 0) status = CE_ERR_HIP;      // (k_emit: the steady form's premise)
      const long long tm = status == CE_OK ? (long long)m : 0;                   // a failed call admits nothing
      if (m_all > miss_cap) fw->overflow = 1;
      long long k = 0;
      if (status == CE_OK) {
        k = tm - free0;
        if (k < 0) k = 0;
        ctl->n_free = free0 + k - tm;
      }
      ctl->seq = seq_arg;
      ctl->n_free_start = free0;
      ctl->n_unique = tu;
      ctl->n_miss = tm;
      ctl->k_evict = k;
      ctl->sel_krem = k;
      ctl->sel_prefix = 0;
      ctl->miss_lookups = (long long)fw->miss_lookups;
      ctl->n_eligible = 0;
      ctl->victims_count = 0;
      ctl->lost = 0;
      __hip_atomic_store(&ctl->status, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ring_slot->n_ids = n_ids;
      ring_slot->n_unique = tu;
      ring_slot->n_miss = tm;
      ring_slot->n_evict = k;
      ring_slot->miss_lookups = (status == CE_OK) ? (long long)fw->miss_lookups : 0;
      ring_slot->n_free_after = ctl->n_free;
      ring_slot->status = status;
      ring_slot->kind = CE_CALL_PREPARE;
      if (n_admit_out) *n_admit_out = tm < in_cap ? tm : in_cap;
      fw_next->miss_lookups = 0;                       // the other parity's set, for the call after this one
      fw_next->n_miss = 0;
      fw_next->hit_unique = 0;
      fw_next->done = 0;
      fw_next->bad = 0;
      fw_next->overflow = 0;
    }
  }
}

struct StageArgs {
  int32_t* cached_idx_map;
  int32_t* inverted;
  int64_t* freq;
  int32_t* slot_epoch;
  int64_t C;
  long long seq_arg;
  int top_pass;
  const unsigned long long* keys;
  const uint32_t* hist;
  Ctl* ctl;
  unsigned long long* lb;
  unsigned tag;
  int32_t* free_list;            // out: the victims ascending = the slots the missing rows take
  const int32_t* miss_list;
  ce_call_stats_t* ring;
  long long* n_unpack_out;
  const void* cache;
  void* stage;
  int32_t* stage_rows_idx;
  long long scap;
  int rowlen, g_log2, vec;
  WbMail* mail;
  long long job;
  EvTable evt;
  void* host_overflow;
};

constexpr int kRemapSlots = 4096;      // slots per workgroup of k_rank_victims (k_victims' blocks)

// Victims of a FULL cache in ascending slot order = the call's free-slot list.  Workgroup j owns the slots
// [4096 j, 4096 j + 4096): its victims (key <= the threshold the level-0 histogram gives, like k_victims), their count
// handed round by look-back, their slots written at (victims of the blocks before) + rank -- no returning atomic, no
// second pass over the keys (k_victims + k_evict_stage's free-list workgroups).
// (One kernel for ranks AND staging AND maps, every workgroup moving the victims of its own 4096 slots, was 147 us
// against 20: the slots a full cache turns over are the ones its coldest rows sit in, a few blocks hold most of them.)
__global__ __launch_bounds__(256) void k_rank_victims(const StageArgs a) {
  __shared__ unsigned long long prefix_s;
  __shared__ int fail_s, go_s;
  __shared__ unsigned long long mask_s[64];
  __shared__ int pre_s[65];
  __shared__ unsigned base_s;
  Ctl* const ctl = a.ctl;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = (int)blockIdx.x;
  // wave wv holds the 64-slot groups wv, wv + 4, ... (in flight while wave 0 works out the threshold)
  const int64_t s0 = (int64_t)j * kRemapSlots;
  unsigned long long key[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int64_t sl = s0 + (int64_t)(wv + 4 * u) * 64 + lane;
    key[u] = sl < a.C ? a.keys[sl] : ~0ull;
  }
  if (threadIdx.x < 64) {
    const SelState st = select_level(a.hist, 0, a.top_pass, ctl, threadIdx.x, false);
    if (threadIdx.x == 0) {
      prefix_s = st.prefix;
      fail_s = st.fail;
      go_s = (ctl->k_evict != 0 && ctl->status == CE_OK) ? 1 : 0;
    }
  }
  __syncthreads();
  if (!go_s) return;                     // no miss / a failed call: k_stage_maps publishes the record
  if (fail_s) {
    // fewer evictable slots than rows to admit -- the capacity overflow of the overlapped pipeline: nothing is evicted
    // or admitted, the record says so.  Every workgroup sees the same; ONE thread records it.
    if (j == 0 && threadIdx.x == 0) {
      ce_call_stats_t* const ring_slot = a.ring + (call_seq(ctl, a.seq_arg) % kRing);
      ctl->n_free = ctl->n_free - ctl->k_evict + ctl->n_miss;
      ctl->k_evict = 0;
      ctl->status = CE_ERR_CAPACITY;
      ring_slot->status = CE_ERR_CAPACITY;
      ring_slot->n_evict = 0;
      ring_slot->n_free_after = ctl->n_free;
    }
    return;
  }
  const unsigned long long T = prefix_s;       // the k-th smallest key; keys are unique
  const long long k = ctl->k_evict;
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const unsigned long long mk = __ballot(key[u] <= T && key[u] != ~0ull);
    if (lane == 0) mask_s[wv + 4 * u] = mk;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int cgrp = __popcll(mask_s[threadIdx.x]);
    const int inc = wave_incl_scan(cgrp, lane);
    pre_s[threadIdx.x] = inc - cgrp;
    unsigned b0 = 0, b1 = 0;
    lb_scan_wave(a.lb, j, a.tag, (unsigned)__shfl(inc, 63), 0u, &b0, &b1);
    if (threadIdx.x == 0) base_s = b0;
  }
  __syncthreads();
  const long long base = base_s;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int g = wv + 4 * u;
    const unsigned long long mk = mask_s[g];
    if ((mk >> lane) & 1) {
      const long long pos = base + pre_s[g] + __popcll(mk & lt);
      if (pos < k) a.free_list[pos] = (int32_t)(s0 + (int64_t)g * 64 + lane);
    }
  }
}

// victims [0, k) of the call, grid-stride, a lane group per R of them
template <typename VT>
__device__ __forceinline__ void stage_and_remap(const StageArgs& a, long long k, int32_t epoch) {
  const VT* const cache = (const VT*)a.cache;
  VT* const stage = (VT*)a.stage;
  VT* const host = (VT*)a.host_overflow;
  const int rowlen = a.rowlen;
  const int Gl = 1 << a.g_log2;
  const int gl = threadIdx.x & (Gl - 1);
  const long long gstride = ((long long)gridDim.x * blockDim.x) >> a.g_log2;
  constexpr int R = kStageRowsInFlight;
  for (long long i0 = ((((long long)blockIdx.x * blockDim.x + threadIdx.x)) >> a.g_log2) * R; i0 < k; i0 += gstride * R) {
    int32_t slot[R], old[R], in_row[R];
    VT v[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
      slot[t] = i0 + t < k ? a.free_list[i0 + t] : -1;
      in_row[t] = (i0 + t < k && gl == 0) ? a.miss_list[i0 + t] : -1;
    }
#pragma unroll
    for (int t = 0; t < R; ++t) {            // every victim's old row and payload in flight before the first store
      old[t] = (slot[t] >= 0 && gl == 0) ? a.cached_idx_map[slot[t]] : -1;
      if (slot[t] >= 0 && rowlen <= Gl && gl < rowlen) v[t] = cache[(int64_t)slot[t] * rowlen + gl];
    }
#pragma unroll
    for (int t = 0; t < R; ++t) {
      if (slot[t] < 0) continue;
      const long long i = i0 + t;
      const bool staged = i < a.scap;
      if (gl == 0) {
        // the victim's row leaves both maps, the missing row of the same rank takes its slot (A.4 pairs the missing
        // rows with the free slots in ascending order of both; a full cache's free slots are its victims)
        a.inverted[old[t]] = -1;
        a.cached_idx_map[slot[t]] = in_row[t];
        a.inverted[in_row[t]] = slot[t];
        a.slot_epoch[slot[t]] = epoch;
        if (a.freq) a.freq[slot[t]] = 0;
        if (staged) {
          a.stage_rows_idx[i] = old[t];
          if (a.evt.keys) evt_insert(a.evt, (uint32_t)a.job, old[t], (int32_t)i);
        }
      }
      int32_t row = 0;
      if (!staged) row = __shfl(old[t], (threadIdx.x & 63) & ~(Gl - 1));      // the group's lane 0 has it
      if (rowlen <= Gl) {
        if (gl < rowlen) {
          if (staged) stage[i * rowlen + gl] = v[t];
          else if (host) host[(int64_t)row * rowlen + gl] = v[t];      // beyond the staging: straight to the host table
        }
      } else {
        VT* const dst = staged ? stage + i * rowlen : (host ? host + (int64_t)row * rowlen : nullptr);
        if (dst) copy_row(cache + (int64_t)slot[t] * rowlen, dst, rowlen, gl, Gl);
      }
    }
  }
}

// ... and what happens to them, over the whole list, a lane group per four victims: rows staged for the write-back, the
// victims' rows out of both maps, the missing rows of the same rank in (k_evict_stage + k_admit_maps; the payload of
// the admitted rows follows on the admission stream: k_unpack_chained).  Publishes the call's record.
__global__ __launch_bounds__(256) void k_stage_maps(const StageArgs a) {
  Ctl* const ctl = a.ctl;
  const long long seq = call_seq(ctl, a.seq_arg);
  const int32_t epoch = call_epoch(seq);
  const long long k = (ctl->status == CE_OK) ? ctl->k_evict : 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ce_call_stats_t* const ring_slot = a.ring + (seq % kRing);
    *a.n_unpack_out = k;
    if (a.mail) {                              // read by the write-back worker after this kernel's event
      a.mail->count = k < a.scap ? k : a.scap;
      a.mail->job = a.job;
    }
    __threadfence_system();
    *(volatile long long*)&ring_slot->seq = seq;
  }
  if (a.vec) stage_and_remap<f32x4>(a, k, epoch);
  else stage_and_remap<float>(a, k, epoch);
}

// The rows of call w arrived in in_stage (the admission kernel, on the admission stream); move them to the slots the
// selection gave them.  Rows beyond the staging capacity (a call that misses more than 262144 rows) come straight
// from the host table -- or, like the admission kernel's, out of the previous write-back job's staging buffer when
// that job evicted them (EvTable): the write-back before THAT one has landed before the call was enqueued.
template <typename VT>
__global__ __launch_bounds__(256) void k_unpack_chained(const int32_t* __restrict__ slots, const long long* n_ptr,
                                                        long long cap, const VT* __restrict__ in_stage, VT* cache,
                                                        int rowlen, int g_log2, const int32_t* __restrict__ rows,
                                                        const VT* __restrict__ host,
                                                        const unsigned long long* __restrict__ evt_keys,
                                                        const int32_t* __restrict__ evt_pos, uint32_t evt_mask,
                                                        uint32_t tag, const VT* __restrict__ prev_stage) {
  const long long n_all = *n_ptr;
  const int G = 1 << g_log2;
  const int gl = threadIdx.x & (G - 1);
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> g_log2;
  if (host && n_all > cap) {
    for (int64_t i = cap + (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> g_log2); i < n_all; i += gstride) {
      const int32_t row = rows[i];
      const int32_t p = evt_keys ? evt_find(evt_keys, evt_pos, evt_mask, tag, row) : -1;
      copy_row(p >= 0 ? prev_stage + (int64_t)p * rowlen : host + (int64_t)row * rowlen,
               cache + (int64_t)slots[i] * rowlen, rowlen, gl, G);
    }
  }
  const long long n = n_all < cap ? n_all : cap;
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

}  // namespace ce
