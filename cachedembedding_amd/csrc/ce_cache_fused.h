// Fused forms of the cache op's index kernels (round 6).  Included by ce_cache.hip behind the per-phase kernels it
// re-uses helpers from (miss_mask, miss_mask_stamp, mark_pass, select_digit, EvTable, copy_row).
//
// One prepare_ids used to be 11-13 dependent launches (reference: one torch op + a device sync per phase,
// recsys/dlrm_main.py:259 -> upstream CachedParamMgr.prepare_ids).  At prefetch_num = 1 and in the
// reference's own micro-benchmark shape (benchmark/benchmark_cache.py:58-72) every one of those launches is a few
// microseconds of work behind a 5-12 us dependent boundary, and the chain -- not the bag kernels -- is the step.  Here
// the phases that only depend on each other through grid-wide sums run as ONE launch each, the boundaries replaced by
// a grid barrier between co-resident workgroups:
//
//   k_front    ctl reset | mark ids -> bitmap | count unique / missing per workgroup range | plan + ordered emission
//              of the missing rows + epoch stamps + bitmap clear                                     (2 barriers)
//   k_select   keys + top digit | (levels - 1) x histogram pass | victim counts | ascending victim list = the free
//              slots of a full cache, victims staged, BOTH maps rewritten for the rows that take their slots, the
//              call's record published                                                          (levels + 1 barriers)
//
// Residency: <= kCoopMaxG workgroups of 1024 threads, <= 64 VGPRs and <= 48 KB of LDS each, i.e. two fit a CU and a
// grid is at most a quarter of the chip's resident capacity -- whatever else runs (the bag kernels' workgroups
// retire by themselves) all workgroups of a launch become resident, and up to four such launches (other managers,
// other processes sharing the GPU) can wait for each other's barriers at once without starving one another.  The
// barrier's spin is bounded all the same: a grid that never becomes resident traps instead of hanging the device.
#pragma once

namespace ce {

constexpr int kCoopMaxG = 128;
constexpr int kCoopThreads = 1024;
constexpr unsigned long long kBarrierTimeoutTicks = 400000000ull;      // wall_clock64 runs at 100 MHz: 4 s

// device scratch of the fused kernels, one per manager; the barrier words and the hand-over counts on lines of their own
struct Coop {
  unsigned bar_count;
  unsigned pad0[31];
  unsigned bar_gen;
  unsigned pad1[31];
  // rows the admission kernel reads / the unpack kernel moves for the call of either parity: written by the plan
  // (k_front) and by the kernel that knows whether the selection held (k_select / k_admit_maps); the control block's
  // own per-call fields are rewritten by the NEXT call's front while these two kernels may still be running
  long long n_admit[2];
  long long n_unpack[2];
  long long pad2[12];
  // per-workgroup partial sums (the grid-wide totals every workgroup adds up for itself behind a barrier)
  int32_t part_unique[kCoopMaxG];
  int32_t part_miss[kCoopMaxG];
  int32_t part_cold[kCoopMaxG];
  int32_t part_bad[kCoopMaxG];
  int32_t part_elig[kCoopMaxG];
  int32_t part_vic[kCoopMaxG];
};
static_assert(sizeof(Coop) <= kCoopBytes, "Coop must fit the workspace region make_layout reserves for it");

// Grid barrier between the co-resident workgroups of one launch: arrival counter + generation word; the last arriver
// resets the counter and bumps the generation (self-resetting: no per-launch base, so a launch needs no argument that
// depends on the launches before it).  Release by the arriving lane before its arrival (its workgroup's stores leave
// the XCD's L2), acquire after the generation moved (this CU's L1 dropped); the poll itself is a relaxed agent-scope
// load (an acquire per poll would invalidate the L1 of every waiting CU per iteration).
__device__ __forceinline__ void grid_sync(Coop* co, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned gen = __hip_atomic_load(&co->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(&co->bar_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nblocks - 1) {
      __hip_atomic_store(&co->bar_count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // (release: the reset is performed before the generation moves -- a workgroup that sees the new generation may
      // arrive at the NEXT barrier at once)
      __hip_atomic_fetch_add(&co->bar_gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(&co->bar_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > kBarrierTimeoutTicks) __builtin_trap();      // not all workgroups resident: loud, not a hang
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__device__ __forceinline__ int coop_load(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum of one int per thread over the 1024-thread workgroup (every thread gets it); `tmp` = 16 ints of LDS
__device__ __forceinline__ int block_sum_1024(int v, int* tmp) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) tmp[threadIdx.x >> 6] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += tmp[i];
  return t;
}

// exclusive scan of one int per thread over the 1024-thread workgroup; *total = workgroup sum.  `tmp` = 16 ints.
__device__ __forceinline__ int block_excl_scan_1024(int v, int* tmp, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int inc = wave_incl_scan(v, lane);
  __syncthreads();
  if (lane == 63) tmp[w] = inc;
  __syncthreads();
  int pre = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i < w) pre += tmp[i];
    tot += tmp[i];
  }
  *total = tot;
  return pre + inc - v;
}

struct FrontArgs {
  const int64_t* ids;
  int64_t n;
  const int32_t* idx_map;
  const int32_t* inverted;
  int64_t N, C;
  int word_bits, hot_words;
  uint32_t* bitmap;
  int64_t n_vec;               // uint4 words of the bitmap (n_chunks * 256)
  Ctl* ctl;
  Coop* coop;
  int64_t* rows_out;
  int allow_pad, assume_free0, parity;
  int32_t* miss_list;
  int32_t* slot_epoch;
  uint32_t* hist;
  long long seq_arg, in_cap;
  ce_call_stats_t* ring;
};

// begin + mark + count + emit.  Workgroup g owns the uint4 words [g * per, (g + 1) * per) of the bitmap (per a multiple
// of the workgroup size): contiguous and ascending in g, so the place of its first missing row in the miss list is the
// number of missing rows of the workgroups before it -- G <= 128 partial counts, added up by every workgroup for
// itself behind the second barrier, together with the call's totals: every workgroup reaches the same verdict,
// workgroup 0 records it (what k_emit's workgroup 0 did).
template <bool MERGE, int U>
__global__ __launch_bounds__(kCoopThreads) void k_front(const FrontArgs a) {
  extern __shared__ uint32_t hot[];
  __shared__ int tmp_s[16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int g = (int)blockIdx.x, G = (int)gridDim.x;
  Ctl* const ctl = a.ctl;
  Coop* const co = a.coop;
  // ---- per-call reset (k_begin): nothing below reads these before the first barrier
  if (g == 0 && tid == 0) {
    ctl->seq = a.seq_arg ? a.seq_arg : ctl->seq + 1;
    ctl->n_unique = 0;
    ctl->n_miss = 0;
    ctl->k_evict = 0;
    ctl->miss_lookups = 0;
    ctl->sel_prefix = 0;
    ctl->sel_krem = 0;
    ctl->n_eligible = 0;
    ctl->victims_count = 0;
    ctl->status = CE_OK;
    ctl->lost = 0;
    ctl->n_free_start = ctl->n_free;
  }
  for (int i = g * kCoopThreads + tid; i < kHistWords; i += G * kCoopThreads) a.hist[i] = 0;
  // ---- mark
  int cold = 0;
  bool bad = false;
  mark_pass<MERGE, U>(a.ids, a.n, a.idx_map, a.inverted, a.N, a.word_bits, a.hot_words, hot, a.bitmap, a.rows_out,
                      a.allow_pad, &cold, &bad);
  cold = block_sum_1024(cold, tmp_s);
  const int any_bad = __syncthreads_or(bad ? 1 : 0);
  if (tid == 0) {
    co->part_cold[g] = cold;
    co->part_bad[g] = any_bad;
  }
  grid_sync(co, (unsigned)G);
  // ---- count
  const int64_t per = ((a.n_vec + G - 1) / G + kCoopThreads - 1) / kCoopThreads * kCoopThreads;
  const int64_t v0 = (int64_t)g * per, v1 = v0 + per < a.n_vec ? v0 + per : a.n_vec;
  const uint4* const bitmap4 = (const uint4*)a.bitmap;
  int u = 0, m = 0;
  for (int64_t v = v0 + tid; v < v1; v += kCoopThreads) {
    const uint4 q = bitmap4[v];
    const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!wds[k]) continue;
      u += __popc(wds[k]);
      m += __popc(miss_mask(a.inverted, v * 128 + k * 32, wds[k], a.N));
    }
  }
  u = block_sum_1024(u, tmp_s);
  m = block_sum_1024(m, tmp_s);
  if (tid == 0) {
    co->part_unique[g] = u;
    co->part_miss[g] = m;
  }
  grid_sync(co, (unsigned)G);
  // ---- totals, this workgroup's base, the verdict
  int pu = 0, pm = 0, pc = 0, pb = 0, pbase = 0;
  if (tid < G) {
    pu = coop_load(&co->part_unique[tid]);
    pm = coop_load(&co->part_miss[tid]);
    pc = coop_load(&co->part_cold[tid]);
    pb = coop_load(&co->part_bad[tid]);
    if (tid < g) pbase = pm;
  }
  const long long tu = block_sum_1024(pu, tmp_s);
  const long long tm = block_sum_1024(pm, tmp_s);
  const long long tcold = block_sum_1024(pc, tmp_s);
  const int tbad = block_sum_1024(pb, tmp_s);
  const long long base = block_sum_1024(pbase, tmp_s);
  const long long n_free = __hip_atomic_load(&ctl->n_free, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // assume_free0: the host launched the call in its steady-state form (the slots to fill are the call's victims)
  // because the last record it has seen said "no free slot left"; a call launched before the record of a lost
  // admission arrived fails here, state untouched.  (Workgroup 0's plan below leaves n_free alone in exactly the two
  // cases this test tells apart, so it does not matter whether it has run yet.)
  const bool stale = a.assume_free0 && n_free != 0;
  int status = tbad ? CE_ERR_RANGE : CE_OK;
  if (status == CE_OK && tu > a.C) status = CE_ERR_CAPACITY;
  if (status == CE_OK && stale) status = CE_ERR_HIP;
  const bool ok = status == CE_OK;
  const long long seq_ = call_seq(ctl, a.seq_arg);
  const int32_t epoch = call_epoch(seq_);
  if (g == 0 && tid == 0) {
    ce_call_stats_t* const ring_slot = a.ring + (seq_ % kRing);
    long long k = 0;
    if (ok) {
      k = tm - n_free;
      if (k < 0) k = 0;
      ctl->n_free = n_free + k - tm;
    }
    __hip_atomic_store(&ctl->status, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ctl->n_unique = tu;
    ctl->n_miss = tm;
    ctl->k_evict = k;
    ctl->sel_krem = k;
    ctl->miss_lookups = ok ? tcold : 0;
    ring_slot->n_ids = a.n;
    ring_slot->n_unique = tu;
    ring_slot->n_miss = tm;
    ring_slot->n_evict = k;
    ring_slot->miss_lookups = ok ? tcold : 0;
    ring_slot->n_free_after = ok ? n_free + k - tm : n_free;
    ring_slot->status = status;
    ring_slot->kind = CE_CALL_PREPARE;
    const long long rows = ok ? tm : 0;
    co->n_admit[a.parity] = rows < a.in_cap ? rows : a.in_cap;
    // (the record's seq -- "complete" -- is published by the last kernel that can amend it: k_select / k_admit_maps)
  }
  // ---- ordered emission, epoch stamps of the resident rows, bitmap clear
  uint4* const bitmap4w = (uint4*)a.bitmap;
  long long run = base;
  for (int64_t vb = v0; vb < v1; vb += kCoopThreads) {
    const int64_t v = vb + tid;
    uint4 q = make_uint4(0, 0, 0, 0);
    if (v < v1) q = bitmap4[v];
    const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
    uint32_t mm[4];
    int mt = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mm[k] = 0;
      if (ok && wds[k]) {
        mm[k] = miss_mask_stamp(a.inverted, v * 128 + k * 32, wds[k], a.N, a.slot_epoch, epoch);
        mt += __popc(mm[k]);
      }
    }
    int tot;
    long long pos = run + block_excl_scan_1024(mt, tmp_s, &tot);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t bits = mm[k];
      const int64_t row0 = v * 128 + k * 32;
      while (bits) {
        const int b = __ffs(bits) - 1;
        bits &= bits - 1;
        a.miss_list[pos++] = (int32_t)(row0 + b);
      }
    }
    if (q.x | q.y | q.z | q.w) bitmap4w[v] = make_uint4(0, 0, 0, 0);
    run += tot;
  }
}

struct SelectArgs {
  int32_t* cached_idx_map;
  int32_t* inverted;
  int64_t* freq;
  int32_t* slot_epoch;
  int64_t C, N;
  long long seq_arg;
  int32_t depth;
  int slot_bits, lfu, top_pass, parity;
  unsigned long long* keys;
  uint32_t* hist;
  Ctl* ctl;
  Coop* coop;
  int32_t* free_list;            // out: the victims ascending = the slots the missing rows take
  const int32_t* miss_list;
  ce_call_stats_t* ring;
  // staging of the victims (k_evict_stage's arguments)
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

template <typename VT>
__device__ __forceinline__ void stage_and_remap(const SelectArgs& a, long long first, long long end, int32_t epoch) {
  const VT* const cache = (const VT*)a.cache;
  VT* const stage = (VT*)a.stage;
  VT* const host = (VT*)a.host_overflow;
  const int rowlen = a.rowlen;
  const int Gl = 1 << a.g_log2;
  const int gl = threadIdx.x & (Gl - 1);
  const int grp = threadIdx.x >> a.g_log2, ngrp = kCoopThreads >> a.g_log2;
  constexpr int R = kStageRowsInFlight;
  for (long long i = first + (long long)grp * R; i < end; i += (long long)ngrp * R) {
    int32_t slot[R], old[R];
    VT v[R];
#pragma unroll
    for (int t = 0; t < R; ++t) slot[t] = i + t < end ? a.free_list[i + t] : -1;
#pragma unroll
    for (int t = 0; t < R; ++t) {
      old[t] = -1;
      if (slot[t] < 0) continue;
      if (gl == 0) {
        // the victim's row leaves both maps, the missing row of the same rank takes its slot (A.4 pairs the missing
        // rows with the free slots in ascending order of both; a full cache's free slots are its victims)
        const int32_t row = a.cached_idx_map[slot[t]];
        const int32_t in_row = a.miss_list[i + t];
        old[t] = row;
        a.inverted[row] = -1;
        a.cached_idx_map[slot[t]] = in_row;
        a.inverted[in_row] = slot[t];
        a.slot_epoch[slot[t]] = epoch;
        if (a.freq) a.freq[slot[t]] = 0;
        if (i + t < a.scap) {
          a.stage_rows_idx[i + t] = row;
          if (a.evt.keys) evt_insert(a.evt, (uint32_t)a.job, row, (int32_t)(i + t));
        }
      }
      if (rowlen <= Gl) {
        if (gl < rowlen) v[t] = cache[(int64_t)slot[t] * rowlen + gl];
      }
    }
#pragma unroll
    for (int t = 0; t < R; ++t) {
      if (slot[t] < 0) continue;
      const bool staged = i + t < a.scap;
      int32_t row = 0;
      if (!staged || rowlen > Gl) row = __shfl(old[t], (threadIdx.x & 63) & ~(Gl - 1));      // the group's lane 0 has it
      if (rowlen <= Gl) {
        if (gl < rowlen) {
          if (staged) stage[(i + t) * rowlen + gl] = v[t];
          else if (host) host[(int64_t)row * rowlen + gl] = v[t];      // beyond the staging: straight to the host table
        }
      } else {
        VT* const dst = staged ? stage + (i + t) * rowlen : (host ? host + (int64_t)row * rowlen : nullptr);
        if (dst) copy_row(cache + (int64_t)slot[t] * rowlen, dst, rowlen, gl, Gl);
      }
    }
  }
}

// keys + radix select + victims + staging + maps, for a FULL cache (the call began with no free slot, so it evicts
// exactly as many rows as it misses and the slots to fill are its victims).  Workgroup g owns the slots
// [g * per, (g + 1) * per): the victims it finds, written in slot order at (victims of the workgroups before it) +
// rank, ARE the ascending list k_victims + free_list_from_victims produced with an atomic and a second pass.
__global__ __launch_bounds__(kCoopThreads) void k_select(const SelectArgs a) {
  __shared__ uint32_t sh[kBins];
  __shared__ int tmp_s[16];
  __shared__ unsigned long long prefix_s;
  __shared__ int krem_s;
  const int tid = threadIdx.x;
  const int g = (int)blockIdx.x, G = (int)gridDim.x;
  Ctl* const ctl = a.ctl;
  Coop* const co = a.coop;
  const long long seq = call_seq(ctl, a.seq_arg);
  const int32_t epoch = call_epoch(seq);
  ce_call_stats_t* const ring_slot = a.ring + (seq % kRing);
  const long long k = ctl->k_evict;                       // (written by the launch before this one)
  const bool call_ok = ctl->status == CE_OK;
  if (!call_ok || k == 0) {                               // grid-uniform: nothing to select
    if (g == 0 && tid == 0) {
      co->n_unpack[a.parity] = 0;
      if (a.mail) {
        a.mail->count = 0;
        a.mail->job = a.job;
      }
      __threadfence_system();
      *(volatile long long*)&ring_slot->seq = seq;
    }
    return;
  }
  const int64_t per = ((a.C + G - 1) / G + kCoopThreads - 1) / kCoopThreads * kCoopThreads;
  const int64_t s0 = (int64_t)g * per, s1 = s0 + per < a.C ? s0 + per : a.C;
  // ---- keys (k_keys) + the top digit's histogram
  for (int i = tid; i < kBins; i += kCoopThreads) sh[i] = 0;
  __syncthreads();
  {
    const int shift = a.top_pass * kDigitBits;
    const int key_bits = (a.top_pass + 1) * kDigitBits;
    const unsigned long long fmax = (1ull << ((key_bits < 63 ? key_bits : 63) - a.slot_bits)) - 1;
    int elig = 0;
    for (int64_t s = s0 + tid; s < s1; s += kCoopThreads) {
      const int32_t row = a.cached_idx_map[s];
      const bool prot = (epoch - a.slot_epoch[s]) <= a.depth;
      unsigned long long key = ~0ull;
      if (row >= 0 && !prot) {
        if (a.lfu) {
          const long long f = a.freq[s];
          unsigned long long uf = f < 0 ? 0ull : (unsigned long long)f;
          if (uf > fmax) uf = fmax;
          key = (uf << a.slot_bits) | (unsigned long long)s;
        } else {
          key = (unsigned long long)(a.N - 1 - row);
        }
        ++elig;
      }
      a.keys[s] = key;
      atomicAdd(&sh[(key >> shift) & (kBins - 1)], 1u);
    }
    elig = block_sum_1024(elig, tmp_s);
    if (tid == 0) co->part_elig[g] = elig;
    uint32_t* const mine = a.hist + a.top_pass * kBins;
    for (int i = tid; i < kBins; i += kCoopThreads)
      if (sh[i]) atomicAdd(&mine[i], sh[i]);
  }
  grid_sync(co, (unsigned)G);
  const long long n_elig = block_sum_1024(tid < G ? coop_load(&co->part_elig[tid]) : 0, tmp_s);
  if (n_elig < k) {
    // fewer evictable slots than rows to admit (the protected window holds them): the capacity overflow of the
    // overlapped pipeline.  Nothing is evicted or admitted; the record says so (k_victims' failure path).
    if (g == 0 && tid == 0) {
      ctl->n_free = ctl->n_free - ctl->k_evict + ctl->n_miss;
      ctl->k_evict = 0;
      ctl->status = CE_ERR_CAPACITY;
      ring_slot->status = CE_ERR_CAPACITY;
      ring_slot->n_evict = 0;
      ring_slot->n_free_after = ctl->n_free;
      co->n_unpack[a.parity] = 0;
      if (a.mail) {
        a.mail->count = 0;
        a.mail->job = a.job;
      }
      __threadfence_system();
      *(volatile long long*)&ring_slot->seq = seq;
    }
    return;
  }
  // ---- the digits below the top one: every workgroup resolves the level above from its (complete) histogram, then
  // adds its slots' share to the next one
  if (tid == 0) {
    prefix_s = 0;
    krem_s = (int)k;
  }
  __syncthreads();
  for (int pass = a.top_pass - 1; pass >= 0; --pass) {
    for (int i = tid; i < kBins; i += kCoopThreads) sh[i] = 0;
    if (tid < 64) {
      const SelState st = select_digit(a.hist, pass + 1, prefix_s, krem_s, tid);
      if (tid == 0) {
        prefix_s = st.prefix;
        krem_s = st.krem;
      }
    }
    __syncthreads();
    const unsigned long long prefix = prefix_s;
    const int shift = pass * kDigitBits;
    constexpr int UK = 4;
    for (int64_t sb = s0 + tid; sb < s1; sb += (int64_t)kCoopThreads * UK) {
      unsigned long long key[UK];
#pragma unroll
      for (int q = 0; q < UK; ++q) {
        const int64_t s = sb + (int64_t)q * kCoopThreads;
        key[q] = s < s1 ? a.keys[s] : 0ull;
      }
#pragma unroll
      for (int q = 0; q < UK; ++q) {
        const int64_t s = sb + (int64_t)q * kCoopThreads;
        const bool match = (key[q] >> (shift + kDigitBits)) == (prefix >> (shift + kDigitBits));
        if (s < s1 && match) atomicAdd(&sh[(key[q] >> shift) & (kBins - 1)], 1u);
      }
    }
    __syncthreads();
    uint32_t* const mine = a.hist + pass * kBins;
    for (int i = tid; i < kBins; i += kCoopThreads)
      if (sh[i]) atomicAdd(&mine[i], sh[i]);
    grid_sync(co, (unsigned)G);
  }
  if (tid < 64) {
    const SelState st = select_digit(a.hist, 0, prefix_s, krem_s, tid);
    if (tid == 0) prefix_s = st.prefix;
  }
  __syncthreads();
  const unsigned long long T = prefix_s;       // the k-th smallest key; keys are unique
  // ---- victims of this workgroup's slots
  int cnt = 0;
  for (int64_t s = s0 + tid; s < s1; s += kCoopThreads) {
    const unsigned long long key = a.keys[s];
    cnt += (key <= T && key != ~0ull);
  }
  cnt = block_sum_1024(cnt, tmp_s);
  if (tid == 0) co->part_vic[g] = cnt;
  grid_sync(co, (unsigned)G);
  const long long base = block_sum_1024(tid < g ? coop_load(&co->part_vic[tid]) : 0, tmp_s);
  long long run = base;
  for (int64_t sb = s0; sb < s1; sb += kCoopThreads) {
    const int64_t s = sb + tid;
    const unsigned long long key = s < s1 ? a.keys[s] : ~0ull;
    const int hit = (key <= T && key != ~0ull);
    int tot;
    const long long pos = run + block_excl_scan_1024(hit, tmp_s, &tot);
    if (hit && pos < k) a.free_list[pos] = (int32_t)s;
    run += tot;
  }
  __syncthreads();
  long long end = base + cnt;
  if (end > k) end = k;
  if (a.vec) stage_and_remap<f32x4>(a, base, end, epoch);
  else stage_and_remap<float>(a, base, end, epoch);
  if (g == 0 && tid == 0) {
    co->n_unpack[a.parity] = k;
    if (a.mail) {                              // read by the write-back worker after this kernel's event
      a.mail->count = k < a.scap ? k : a.scap;
      a.mail->job = a.job;
    }
    __threadfence_system();
    *(volatile long long*)&ring_slot->seq = seq;
  }
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
