// Cache op, victim selection: keys, radix histograms, victims (canonical ties of App. B#1).
// Part of the one translation unit ce_cache.hip (included there, in this order: ce_cache_index.h, ce_cache_select.h,
// ce_cache_rows.h, ce_cache_fused.h, ce_cache_worker.h); not a stand-alone header.
#pragma once

namespace ce {

// selection keys: smaller = evicted first.  Ineligible (empty / protected) = all ones.  The histogram of the TOP digit
// is taken here too (the keys are in registers): one pass over the keys less.
__global__ __launch_bounds__(256) void k_keys(const int32_t* __restrict__ cached_idx_map,
                                              const int64_t* __restrict__ freq,
                                              const int32_t* __restrict__ slot_epoch, int64_t C, int64_t N,
                                              long long seq_arg, int32_t depth, int slot_bits, int lfu, int top_pass,
                                              unsigned long long* keys, uint32_t* hist, Ctl* ctl,
                                              const int32_t* __restrict__ miss_tmp, const FrontWords* fw,
                                              uint32_t* bitmap, int32_t* fine, int32_t* coarse_cnt, int n_chunks) {
  // behind the per-lookup front (miss_tmp != NULL): what it left in the bitmap and its counters goes first
  if (miss_tmp) front_cleanup(miss_tmp, fw, bitmap, fine, coarse_cnt, n_chunks);
  if (ctl->k_evict == 0) return;          // (the histograms were cleared by k_begin / k_miss_rank)
  const int32_t epoch = call_epoch(call_seq(ctl, seq_arg));
  __shared__ uint32_t sh[kBins];
  for (int i = threadIdx.x; i < kBins; i += blockDim.x) sh[i] = 0;
  __syncthreads();
  const int shift = top_pass * kDigitBits;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // LFU: a counter is clamped so that the key stays inside the digits the select looks at (the host derives
  // top_pass from an upper bound of the counters; freq_cnter is the caller's tensor, so nothing else guarantees it)
  const int key_bits = (top_pass + 1) * kDigitBits;
  const unsigned long long fmax = (1ull << ((key_bits < 63 ? key_bits : 63) - slot_bits)) - 1;
  int elig = 0;
  // four slots per thread in flight (one after the other, a thread of the 512-workgroup grid walked 13 slots of a
  // 1.7 M-slot cache in 13 dependent round trips)
  constexpr int UK = 4;
  for (int64_t s0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s0 < C; s0 += stride * UK) {
    int32_t row[UK], ep[UK];
    long long fr[UK];
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      const int64_t s = s0 + (int64_t)u * stride;
      row[u] = s < C ? cached_idx_map[s] : -1;
      ep[u] = s < C ? slot_epoch[s] : 0;
      fr[u] = (lfu && s < C) ? freq[s] : 0;
    }
#pragma unroll
    for (int u = 0; u < UK; ++u) {
      const int64_t s = s0 + (int64_t)u * stride;
      if (s >= C) continue;
      const bool prot = (epoch - ep[u]) <= depth;
      unsigned long long key = ~0ull;
      if (row[u] >= 0 && !prot) {
        if (lfu) {
          unsigned long long uf = fr[u] < 0 ? 0ull : (unsigned long long)fr[u];
          if (uf > fmax) uf = fmax;
          key = (uf << slot_bits) | (unsigned long long)s;
        } else {
          key = (unsigned long long)(N - 1 - row[u]);
        }
        ++elig;
      }
      keys[s] = key;
      atomicAdd(&sh[(key >> shift) & (kBins - 1)], 1u);
    }
  }
  // evictable slots are counted here, not read off the top-digit histogram: a DATASET key N-1-row can share its
  // top digit with the all-ones key of an ineligible slot.
  // One atomic per WORKGROUP on a grid of at most 512: same-address device atomics serialise at ~7 ns each, and
  // one per wave of a 1738-workgroup grid cost this kernel 50 us.
  __shared__ int wsum[4];
  elig = wave_sum(elig);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = elig;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (tot) atomicAdd((unsigned long long*)&ctl->n_eligible, (unsigned long long)tot);
  }
  uint32_t* const mine = hist + top_pass * kBins;
  for (int i = threadIdx.x; i < kBins; i += blockDim.x)
    if (sh[i]) atomicAdd(&mine[i], sh[i]);
}

// Few, fat workgroups: every workgroup ends with one device atomic per non-empty bin and same-address atomics
// serialise (~7 ns each), so 1738 workgroups of 256 cost 12 us per pass in the histogram flush alone; 256
// workgroups of 1024 threads with 4 independent key loads per thread read the 14 MB of keys just as fast.
// Digits of the k-th smallest key decided so far, from the per-pass histograms hist[q][2048] of the passes q > lowest
// (what a single-thread pick kernel between two histogram passes would compute): one wave, 32 bins per lane, per
// level a wave scan, the first lane whose running count reaches k, then that lane's bins handed round with
// shuffles.  Every workgroup of the NEXT kernel recomputes it in its prologue (8 KB per level out of L2) -- that
// removes one launch per pass.  The inputs (k in ctl->sel_krem, the histograms) are read-only while it runs, so all
// workgroups agree.
struct SelState {
  unsigned long long prefix;
  int krem;
  int fail;      // fewer evictable slots than k: capacity overflow of the overlapped pipeline
};
// Resolves ONE level: the digit of the k-th smallest key at level q from that level's histogram, given the digits
// and the remaining rank of the levels above.  One wave, 32 bins per lane: a wave scan, the first lane whose running
// count reaches the rank, then that lane's bins handed round with shuffles.
__device__ __forceinline__ SelState select_digit(const uint32_t* hist, int q, unsigned long long prefix_in,
                                                 int krem_in, int lane) {
  SelState st;
  st.prefix = prefix_in;
  st.krem = krem_in;
  st.fail = 0;
  uint4 h[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) h[j] = ((const uint4*)(hist + q * kBins))[lane * 8 + j];
  int sum = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) sum += (int)(h[j].x + h[j].y + h[j].z + h[j].w);
  const int inc = wave_incl_scan(sum, lane);
  const unsigned long long m = __ballot(inc >= st.krem);
  const int L = m ? __ffsll((long long)m) - 1 : 63;
  const int r = st.krem - __shfl(inc - sum, L);          // rank inside lane L's 32 bins
  int dd = 31, before = 0, cum = 0;
  bool found = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int vals[4] = {(int)h[j].x, (int)h[j].y, (int)h[j].z, (int)h[j].w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int val = __shfl(vals[c], L);
      if (!found && r <= cum + val) {
        dd = 4 * j + c;
        before = cum;
        found = true;
      }
      cum += val;
    }
  }
  if (!found) before = cum - __shfl((int)h[7].w, L);      // (only on the failure path: k beyond the candidates)
  st.prefix |= ((unsigned long long)(32 * L + dd)) << (q * kDigitBits);
  st.krem = r - before;
  return st;
}

// The per-launch form: every workgroup of a kernel resolves the level above its own in its prologue (8 KB out of L2),
// from the state workgroup 0 of the kernel before left in the control block (the top level starts from k itself), and
// workgroup 0 records the result for the next kernel: no pick kernel between two passes, no dependency between
// workgroups, one histogram read per kernel.
__device__ __forceinline__ SelState select_level(const uint32_t* __restrict__ hist, int q, int top_pass, Ctl* ctl,
                                                 int lane, bool record) {
  const unsigned long long prefix_in = q == top_pass ? 0ull : ctl->sel_prefix_after[q + 1];
  const int krem_in = q == top_pass ? (int)ctl->sel_krem : (int)ctl->sel_krem_after[q + 1];
  SelState st = select_digit(hist, q, prefix_in, krem_in, lane);
  // With protect_depth > 0 the protected set can leave fewer than k candidates: that is the capacity overflow of
  // the overlapped pipeline (unique(window k u k+1) > cuda_row_num); evictable slots are counted by k_keys
  st.fail = ctl->n_eligible < ctl->sel_krem;
  if (record && lane == 0) {
    ctl->sel_prefix_after[q] = st.prefix;
    ctl->sel_krem_after[q] = st.krem;
  }
  return st;
}

__global__ __launch_bounds__(1024) void k_hist(const unsigned long long* __restrict__ keys, int64_t C, int pass,
                                               int top_pass, uint32_t* hist, Ctl* ctl) {
  if (ctl->k_evict == 0) return;
  __shared__ uint32_t sh[kBins];
  __shared__ unsigned long long prefix_s;
  __shared__ int fail_s;
  for (int i = threadIdx.x; i < kBins; i += blockDim.x) sh[i] = 0;
  if (threadIdx.x < 64) {
    const SelState st = select_level(hist, pass + 1, top_pass, ctl, threadIdx.x, blockIdx.x == 0);
    if (threadIdx.x == 0) {
      prefix_s = st.prefix;
      fail_s = st.fail;
    }
  }
  __syncthreads();
  if (fail_s) return;                  // k_victims records the failure
  const int shift = pass * kDigitBits;
  const unsigned long long prefix = prefix_s;
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * U;
  for (int64_t s0 = (int64_t)blockIdx.x * blockDim.x * U + threadIdx.x; s0 < C; s0 += stride) {
    unsigned long long key[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t s = s0 + (int64_t)u * blockDim.x;
      key[u] = s < C ? keys[s] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t s = s0 + (int64_t)u * blockDim.x;
      // (only passes below the top one get here: shift + kDigitBits <= 55)
      const bool match = (key[u] >> (shift + kDigitBits)) == (prefix >> (shift + kDigitBits));
      if (s < C && match) atomicAdd(&sh[(key[u] >> shift) & (kBins - 1)], 1u);
    }
  }
  __syncthreads();
  uint32_t* const mine = hist + pass * kBins;
  for (int i = threadIdx.x; i < kBins; i += blockDim.x)
    if (sh[i]) atomicAdd(&mine[i], sh[i]);
}

// blk_vic (steady-state calls only): victims among this workgroup's 4096 slots -- k_evict_stage's free-list
// workgroups turn them into the ascending list of slots to fill without another scan of cached_idx_map; the
// threshold key goes to ctl->sel_prefix_after[0] for them.
__global__ __launch_bounds__(256) void k_victims(const unsigned long long* __restrict__ keys, int64_t C,
                                                 int32_t* victims, int64_t cap, Ctl* ctl, const uint32_t* hist,
                                                 int top_pass, ce_call_stats_t* ring, long long seq_arg,
                                                 int32_t* blk_vic) {
  ce_call_stats_t* const ring_slot = ring + (call_seq(ctl, seq_arg) % kRing);
  __shared__ unsigned long long prefix_s;
  __shared__ int fail_s, go_s;
  // this workgroup's 4096 slots (16 per thread, strided by 256): in flight while wave 0 works out the threshold
  // (every workgroup recomputes it from the histograms: few, fat workgroups keep that redundant work small)
  constexpr int KV = 16;
  const int64_t s0 = (int64_t)blockIdx.x * (256 * KV) + threadIdx.x;
  unsigned long long key[KV];
#pragma unroll
  for (int u = 0; u < KV; ++u) {
    const int64_t sl = s0 + u * 256;
    key[u] = sl < C ? keys[sl] : ~0ull;
  }
  if (blk_vic && threadIdx.x == 0) blk_vic[blockIdx.x] = 0;
  if (threadIdx.x < 64) {
    const SelState st = select_level(hist, 0, top_pass, ctl, threadIdx.x, blk_vic != nullptr && blockIdx.x == 0);
    if (threadIdx.x == 0) {
      prefix_s = st.prefix;
      fail_s = st.fail;
      // k_evict is cleared by workgroup 0 of THIS kernel when the call fails: it is read once per workgroup, by one
      // thread, so that all threads of a workgroup take the same way around the barrier below
      go_s = ctl->k_evict != 0;
    }
  }
  __syncthreads();
  if (!go_s) return;
  if (fail_s) {
    // every workgroup sees the same failure (read-only inputs); ONE thread turns the call into a capacity failure:
    // nothing is evicted or admitted, the record says so.  The kernels that follow read k_evict / status.
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ctl->n_free = ctl->n_free - ctl->k_evict + ctl->n_miss;
      ctl->k_evict = 0;
      ctl->status = CE_ERR_CAPACITY;
      ring_slot->status = CE_ERR_CAPACITY;
      ring_slot->n_evict = 0;
      ring_slot->n_free_after = ctl->n_free;
      __threadfence_system();
    }
    return;
  }
  const unsigned long long T = prefix_s;   // k-th smallest key; keys are unique
  // the workgroup's victims are counted with a block scan and reserve their places with ONE returning atomic (a returning device atomic is a ~2 us round trip; one
  // per wave with a victim in it was ~24 k of them on one address per call)
  __shared__ int base_s;
  int hits = 0;
#pragma unroll
  for (int u = 0; u < KV; ++u) hits += (key[u] <= T && key[u] != ~0ull);
  int tot;
  int pos = block_excl_scan_256(hits, &tot);
  if (tot == 0) return;                     // block-uniform
  if (threadIdx.x == 0) {
    base_s = atomicAdd(&ctl->victims_count, tot);
    if (blk_vic) blk_vic[blockIdx.x] = tot;
  }
  __syncthreads();
  pos += base_s;
#pragma unroll
  for (int u = 0; u < KV; ++u) {
    if (key[u] <= T && key[u] != ~0ull) {
      if (pos < cap) victims[pos] = (int32_t)(s0 + u * 256);
      ++pos;
    }
  }
}

}  // namespace ce
