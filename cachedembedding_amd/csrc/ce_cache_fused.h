// Merged forms of the cache op's index kernels (round 6).  Included by ce_cache.hip behind the per-phase kernels whose
// helpers they use (miss masks, select_level, EvTable, copy_row).
//
// One prepare_ids used to be 11-13 dependent launches (reference: one torch op + a device sync per phase,
// recsys/dlrm_main.py:259 -> upstream CachedParamMgr.prepare_ids).  At prefetch_num = 1 and in the
// reference's own micro-benchmark shape (benchmark/benchmark_cache.py:58-72) the chain -- not the bag kernels -- is
// the step, and what a kernel of the chain costs there is its number of DEPENDENT memory round trips (2-3 us each
// beside the training kernels), not its bytes.  Two merges take round trips out without putting anything in:
//
//   k_emit_scan     count + emit in ONE pass over the bitmap and the map entries of the rows seen: a workgroup's place
//                   in the miss list is the number of missing rows in the chunks before it -- single-pass scan with
//                   decoupled look-back (every workgroup publishes its count in a tagged 64-bit word and adds up its
//                   predecessors' words), so there is no count kernel and no second gather of the map.  Needs the
//                   call's verdict before the first stamp, i.e. "unique rows <= cache rows" known up front: calls of at
//                   most cuda_row_num ids (prefetch_num 1-2; a window of 8 batches takes k_count + k_emit).
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

// count + emit + plan in one pass (see the header).  One workgroup per 32768-row chunk of the bitmap; the LAST one
// holds the call's totals when its look-back ends and records the plan (what k_emit's workgroup 0 does from k_count's
// sums).  The bad-id verdict is k_mark's, a launch ago; unique <= n <= C by the caller's choice of this kernel.
// (One workgroup of 1024 threads per four chunks -- a quarter of the pollers, look-backs of one or two rounds -- was
// slower: 44 against 37 us at the Kaggle table.  What the kernel waits for is the workgroups of the table's dense head,
// whose threads fetch 128 map entries each before they can publish a count.)
__global__ __launch_bounds__(256) void k_emit_scan(uint4* bitmap4, const int32_t* __restrict__ inverted, int64_t N,
                                                   int n_chunks, unsigned long long* lb, unsigned tag,
                                                   int32_t* miss_list, int32_t* slot_epoch, long long seq_arg, Ctl* ctl,
                                                   int64_t n_ids, ce_call_stats_t* ring, long long in_cap,
                                                   int assume_free0, long long* n_admit_out) {
  __shared__ int su[4], sm[4], wsub[4];
  __shared__ unsigned base_s[2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = (int)blockIdx.x;
  const long long seq_ = call_seq(ctl, seq_arg);
  const int32_t epoch = call_epoch(seq_);
  const int st_in = ctl->status;                                     // (k_mark's, a launch ago)
  const bool stale = assume_free0 && ctl->n_free_start != 0;
  const bool ok = st_in == CE_OK && !stale;
  const int64_t v = (int64_t)c * 256 + threadIdx.x;
  const uint4 q = bitmap4[v];
  const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
  uint32_t mm[4];
  int u = 0, m = 0;
  if (ok) miss_masks4_stamp(inverted, v * 128, wds, N, slot_epoch, epoch, mm);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!ok) mm[k] = 0;
    u += __popc(wds[k]);
    m += __popc(mm[k]);
  }
  const int inc = wave_incl_scan(m, lane);
  const int uw = wave_sum(u);
  if (lane == 63) {
    su[wv] = uw;
    sm[wv] = inc;
    wsub[wv] = inc;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const unsigned tu_c = (unsigned)(su[0] + su[1] + su[2] + su[3]), tm_c = (unsigned)(sm[0] + sm[1] + sm[2] + sm[3]);
    unsigned bu = 0, bm = 0;
    lb_scan_wave(lb, c, tag, tu_c, tm_c, &bu, &bm);
    if (threadIdx.x == 0) {
      base_s[0] = bu;
      base_s[1] = bm;
    }
    if (threadIdx.x == 0 && c == n_chunks - 1) {
      // ---- the plan: this workgroup's inclusive prefix is the call's total
      const long long tu = (long long)bu + tu_c, tm = (long long)bm + tm_c;
      ce_call_stats_t* const ring_slot = ring + (seq_ % kRing);
      int status = st_in;
      if (status == CE_OK && stale) status = CE_ERR_HIP;
      long long k = 0;
      if (status == CE_OK) {
        k = tm - ctl->n_free;
        if (k < 0) k = 0;
        ctl->n_free = ctl->n_free + k - tm;
      }
      __hip_atomic_store(&ctl->status, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ctl->n_unique = tu;
      ctl->n_miss = tm;
      ctl->k_evict = k;
      ctl->sel_krem = k;
      ring_slot->n_ids = n_ids;
      ring_slot->n_unique = tu;
      ring_slot->n_miss = tm;
      ring_slot->n_evict = k;
      ring_slot->miss_lookups = (status == CE_OK) ? ctl->miss_lookups : 0;
      ring_slot->n_free_after = ctl->n_free;
      ring_slot->status = status;
      ring_slot->kind = CE_CALL_PREPARE;
      if (n_admit_out) {
        const long long mrows = (status == CE_OK) ? tm : 0;
        *n_admit_out = mrows < in_cap ? mrows : in_cap;
      }
    }
  }
  __syncthreads();
  if (ok) {
    int pos = (int)base_s[1] + inc - m;
    for (int k = 0; k < wv; ++k) pos += wsub[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t bits = mm[k];
      const int64_t row0 = v * 128 + k * 32;
      while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        miss_list[pos++] = (int32_t)(row0 + b);
      }
    }
  }
  if (q.x | q.y | q.z | q.w) bitmap4[v] = make_uint4(0, 0, 0, 0);
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
