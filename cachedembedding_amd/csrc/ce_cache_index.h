// Cache op, index side: the device control block, workspace layout and the kernels from ids to the miss list
// (begin, mark, count, emit).
// Part of the one translation unit ce_cache.hip (included there, in this order: ce_cache_index.h, ce_cache_select.h,
// ce_cache_rows.h, ce_cache_fused.h, ce_cache_worker.h); not a stand-alone header.
#pragma once

namespace ce {

constexpr int kChunkShift = 15;
constexpr int kChunkRows = 1 << kChunkShift;   // rows covered by one 256-thread workgroup of the bitmap scan (uint4/thread)
constexpr int kCoarseShift = 6;                 // k_count also sums its chunk counts per 64 chunks (k_emit adds those up)
constexpr int kCoarsePad = 32;                  // ints between two coarse counters: a 128-byte line each (device atomics to one
                                                // line serialise whatever the address: packed counters cost k_count 15 us)
constexpr int kSlotsPerBlock = 1024;  // slots covered by one block of the slot-space scans (4/thread)
constexpr int kRing = 1024;           // pinned host ring of per-call stats
constexpr unsigned long long kGraphFreqHeadroom = 1ull << 31;   // ids a captured LFU call may see before a new capture
constexpr int kDigitBits = 11;         // radix select: 11-bit digits (28-bit DATASET keys of a 178 M-row table: 3 passes)
constexpr int kBins = 1 << kDigitBits;
constexpr int kLevels = 6;             // 6 x 11 >= 64 bits
constexpr int kHistWords = kLevels * kBins;   // one histogram per radix pass
constexpr int32_t kEpochNever = -(1 << 30);
constexpr int64_t kHistoryKeep = 1 << 16;   // per-call records kept on the host side

struct Ctl {                 // device control block (one per manager)
  long long n_free;          // persistent: free slots
  long long n_unique;        // per call
  long long n_miss;
  long long k_evict;
  long long miss_lookups;
  unsigned long long sel_prefix;   // (unused)
  long long sel_krem;              // k of the current select (set by the plan)
  long long n_eligible;      // slots that may be evicted in this call (resident and not protected)
  int victims_count;
  int status;
  int lost;                  // per call: the admission worker reported that the rows did not arrive (see k_admit_maps)
  int pad_;
  long long n_free_start;    // per call: n_free when the call began (k_begin; read-only for the rest of the call)
  // radix select: digits and remaining rank after level q was resolved (written by workgroup 0 of the kernel that
  // resolves level q -- every workgroup of that kernel computes the same thing for itself --, read by later kernels)
  unsigned long long sel_prefix_after[8];
  long long sel_krem_after[8];
  // number of the call in flight (= the host's h->seq).  A launched call brings it along (k_begin stores it); a call
  // replayed from a captured hipGraph has no per-launch arguments, so there k_begin counts it up itself -- the
  // epoch of the eviction backlist and the record slot in the stats ring are both derived from it on the device.
  long long seq;
};

// call number -> what the kernels need from it (seq_arg != 0: launched with its number; 0: replayed, see Ctl::seq)
__device__ __forceinline__ long long call_seq(const Ctl* ctl, long long seq_arg) { return seq_arg ? seq_arg : ctl->seq; }
__device__ __forceinline__ int32_t call_epoch(long long seq) { return (int32_t)(seq & 0x3fffffff); }

struct WbMail {              // pinned host mailbox: how many rows a worker job moves (written by the device)
  long long job;
  long long count;
};

struct Layout {              // byte offsets inside the caller-provided workspace
  size_t ctl, bitmap, blk_unique, blk_miss, coarse, miss_list, slot_epoch, keys, hist, victims, blk_free, free_list,
      chain, lb_emit, lb_remap, miss_list2, free_list2, front, fine_cnt, coarse_cnt, stage_idx, stage, stage_idx2,
      stage2, in_stage, total;
  int64_t n_chunks, n_slot_blocks, list_cap, bitmap_words, stage_rows;
};

constexpr int64_t kStageRowsMax = 262144;   // write-back staging: 128 MB at D = 128
// Rows the admission kernel reads / the unpack kernel moves for the call of either parity: written by the call's
// plan (k_emit / k_emit_scan) and by the kernel that knows whether the selection held (k_stage_remap / k_admit_maps).
// The control block's own per-call fields are rewritten by the NEXT call's front while those two kernels may still be
// running on the admission stream.
struct ChainWords {
  long long n_admit[2];
  long long n_unpack[2];
};

// Counters of the per-lookup front (k_touch / k_miss_rank, ce_cache_fused.h), one set per call parity: a call adds to
// its own set and clears the other one for the call after it, so no kernel has to run before the first kernel of a
// call just to zero them.
struct FrontWords {
  unsigned long long miss_lookups;   // lookups of rows that are not resident (with repeats)
  unsigned int n_miss;               // distinct missing rows appended to the unordered list
  unsigned int hit_unique;           // slots that carry the call's stamp = distinct resident rows it looked up
  unsigned int done;                 // workgroups of k_miss_rank that have added their share of hit_unique
  int bad;                           // an id outside [0, N) that is not accepted padding was met
  int overflow;                      // more distinct missing rows than the list holds (a call that fails: capacity)
  unsigned int pad_[9];
};
static_assert(sizeof(FrontWords) == 64, "one 64-byte line per parity");
constexpr int kFineShift = 10;       // rows per fine counter of the per-lookup front (32 bitmap words)
constexpr int kFinePerChunk = 1 << (kChunkShift - kFineShift);
constexpr int kRankMaxChunks = 16384;   // k_miss_rank keeps the chunk prefix in LDS (64 KB): tables up to 2^29 rows

static Layout make_layout(int64_t N, int64_t C, int64_t max_ids, int64_t D) {
  Layout L{};
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  L.n_chunks = cdiv(N, kChunkRows);
  L.bitmap_words = L.n_chunks * (kChunkRows / 32);
  L.n_slot_blocks = cdiv(C, kSlotsPerBlock);
  L.list_cap = std::max<int64_t>(1, std::min<int64_t>(C, std::max<int64_t>(max_ids, 1)));
  size_t o = 0;
  L.ctl = o;        o = al(o + sizeof(Ctl));
  L.bitmap = o;     o = al(o + (size_t)L.bitmap_words * 4);
  L.blk_unique = o; o = al(o + (size_t)(L.n_chunks + 1) * 4);
  L.blk_miss = o;   o = al(o + (size_t)(L.n_chunks + 1) * 4);
  L.coarse = o;     o = al(o + (size_t)((L.n_chunks >> kCoarseShift) + 1) * 2 * kCoarsePad * 4);   // [unique, missing] per 64 chunks
  L.miss_list = o;  o = al(o + (size_t)L.list_cap * 4);
  L.slot_epoch = o; o = al(o + (size_t)C * 4);
  L.keys = o;       o = al(o + (size_t)C * 8);
  L.hist = o;       o = al(o + (size_t)kHistWords * 4);
  L.victims = o;    o = al(o + (size_t)L.list_cap * 4);
  L.blk_free = o;   o = al(o + (size_t)(L.n_slot_blocks + 1) * 4);
  L.free_list = o;  o = al(o + (size_t)L.list_cap * 4);
  // chained admission (worker transport): the row counts handed to the admission stream, the look-back words of the
  // single-pass kernels (ce_cache_fused.h), and a second miss / free list -- the admission and unpack kernels of call w
  // read theirs on the admission stream while call w + 1's front fills the other pair
  L.chain = o;      o = al(o + sizeof(ChainWords));
  L.lb_emit = o;    o = al(o + (size_t)(L.n_chunks + 1) * 8);
  L.lb_remap = o;   o = al(o + (size_t)(cdiv(C, 4096) + 1) * 8);
  L.miss_list2 = o; o = al(o + (size_t)L.list_cap * 4);
  L.free_list2 = o; o = al(o + (size_t)L.list_cap * 4);
  // per-lookup front: its counters, and the distinct missing rows counted per 1024 / 32768 rows (zero between calls:
  // the call that raised them clears them along its own miss list)
  L.front = o;      o = al(o + 2 * sizeof(FrontWords));
  L.fine_cnt = o;   o = al(o + (size_t)(L.n_chunks * kFinePerChunk + 4) * 4);
  L.coarse_cnt = o; o = al(o + (size_t)(L.n_chunks + 1) * 4);
  L.stage_rows = std::min<int64_t>(L.list_cap, kStageRowsMax);
  L.stage_idx = o;  o = al(o + (size_t)L.stage_rows * 4);
  L.stage = o;      o = al(o + (size_t)L.stage_rows * (size_t)D * 4);
  // worker transport (CE_TRANSPORT_WORKER): the eviction staging is double-buffered -- the victims of call w stay
  // in HBM until the host worker has copied them out, while call w+1 stages into the other buffer -- and the
  // admitted rows arrive in `in_stage` (one pinned hipMemcpyAsync per chunk) before a kernel moves them to their slots
  L.stage_idx2 = o; o = al(o + (size_t)L.stage_rows * 4);
  L.stage2 = o;     o = al(o + (size_t)L.stage_rows * (size_t)D * 4);
  L.in_stage = o;   o = al(o + (size_t)L.stage_rows * (size_t)D * 4);
  L.total = o;
  return L;
}

// ----------------------------------------------------------------------------- device helpers

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan of one int per thread over a 256-thread block; returns exclusive prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan_256(int v, int* total) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = wave_incl_scan(v, lane);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < w) base += wsum[i];
    tot += wsum[i];
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}

// bits, counters of a call's missing rows back to zero, along its unordered list (the prologue of the kernel behind
// k_miss_rank: every workgroup a share)
__device__ __forceinline__ void front_cleanup(const int32_t* __restrict__ miss_tmp, const FrontWords* fw,
                                              uint32_t* bitmap, int32_t* fine, int32_t* coarse, int n_chunks) {
  const unsigned m = fw->n_miss;
  if (fw->overflow) {
    // the list does not hold every row that was marked (the call failed: more distinct rows than cache slots):
    // everything goes, at the price of a pass over the table's bitmap -- once per failed call
    const int64_t nw = (int64_t)n_chunks * (kChunkRows / 32), nf = (int64_t)n_chunks * kFinePerChunk;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, ts = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = t0; i < nw; i += ts) bitmap[i] = 0;
    for (int64_t i = t0; i < nf; i += ts) fine[i] = 0;
    for (int64_t i = t0; i < n_chunks; i += ts) coarse[i] = 0;
    return;
  }
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
    const int32_t r = miss_tmp[i];
    bitmap[r >> 5] = 0;                 // (every set bit of the word is a row of this list)
    fine[r >> kFineShift] = 0;
    coarse[r >> kChunkShift] = 0;
  }
}


// ----------------------------------------------------------------------------- kernels

// per-call reset: the control block's call fields, the coarse chunk sums k_count adds to, the radix histograms
__global__ __launch_bounds__(256) void k_begin(Ctl* ctl, int32_t* coarse, int n_coarse2, uint32_t* hist,
                                               long long seq_arg) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctl->seq = seq_arg ? seq_arg : ctl->seq + 1;
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
  // (a few workgroups: one of 256 threads spent 10 us on these 13-18 k stores -- a launch of the chain like any other)
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  if (coarse)
    for (int i = tid; i < n_coarse2; i += nth) coarse[i] = 0;
  if (hist)
    for (int i = tid; i < kHistWords; i += nth) hist[i] = 0;
}

// Bits of one bitmap word (32 consecutive rows from row0) whose row is not resident.  The frequency ranking packs
// the hot rows into the lowest words, where nearly every bit is set: a lookup per set bit would be up to 128
// dependent-latency loads in one thread (the tail of k_emit), so dense words fetch the 32 map entries
// as eight 16-byte loads instead.
__device__ __forceinline__ bool dense_word(uint32_t bits, int64_t row0, int64_t N) {
  return __popc(bits) >= 6 && row0 + 32 <= N;
}
__device__ __forceinline__ uint32_t miss_mask(const int32_t* __restrict__ inverted, int64_t row0, uint32_t bits,
                                              int64_t N) {
  uint32_t mm = 0;
  if (dense_word(bits, row0, N)) {
    const int4* p = (const int4*)(inverted + row0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int4 x = p[j];
      mm |= ((uint32_t)(x.x < 0) | ((uint32_t)(x.y < 0) << 1) | ((uint32_t)(x.z < 0) << 2) |
             ((uint32_t)(x.w < 0) << 3)) << (4 * j);
    }
    return mm & bits;
  }
  while (bits) {
    const int b = __ffs(bits) - 1;
    bits &= bits - 1;
    if (inverted[row0 + b] < 0) mm |= 1u << b;
  }
  return mm;
}

// The same, and every resident row's slot gets the call's stamp on the way (k_emit: the slot is in hand here, a second
// pass over the map just for the stamps was k_emit's tail).
__device__ __forceinline__ uint32_t miss_mask_stamp(const int32_t* __restrict__ inverted, int64_t row0, uint32_t bits,
                                                    int64_t N, int32_t* slot_epoch, int32_t epoch) {
  uint32_t mm = 0;
  if (dense_word(bits, row0, N)) {
    const int4* p = (const int4*)(inverted + row0);
    int4 x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = p[j];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int32_t sl[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (!((bits >> (4 * j + c)) & 1)) continue;
        if (sl[c] < 0) mm |= 1u << (4 * j + c);
        else slot_epoch[sl[c]] = epoch;       // evict_backlist membership [A.3-3]
      }
    }
    return mm;
  }
  while (bits) {
    const int b = __ffs(bits) - 1;
    bits &= bits - 1;
    const int32_t sl = inverted[row0 + b];
    if (sl < 0) mm |= 1u << b;
    else slot_epoch[sl] = epoch;
  }
  return mm;
}

// The four words of one uint4 of the bitmap at once (k_emit_scan): the map entries of ALL sparse words are fetched in
// one batch -- up to 20 loads in flight -- instead of word after word, bit after bit (a chain of up to 20 dependent
// round trips in miss_mask_stamp's tail loop: 2-3 us each beside the training kernels); dense words take the 16-byte
// path as before.
__device__ __forceinline__ void miss_masks4_stamp(const int32_t* __restrict__ inverted, int64_t row0,
                                                  const uint32_t (&wds)[4], int64_t N, int32_t* slot_epoch,
                                                  int32_t epoch, uint32_t (&mm)[4]) {
  int idx[4][5];
  int32_t val[4][5];
  bool dense[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    dense[k] = __popc(wds[k]) > 5;      // (miss_mask_stamp: sixteen-byte loads, or bit by bit in the table's last word)
    uint32_t b = dense[k] ? 0u : wds[k];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      idx[k][j] = b ? __ffs(b) - 1 : -1;
      b &= b - 1;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 5; ++j) val[k][j] = idx[k][j] >= 0 ? inverted[row0 + 32 * k + idx[k][j]] : 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    mm[k] = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (idx[k][j] < 0) continue;
      if (val[k][j] < 0) mm[k] |= 1u << idx[k][j];
      else slot_epoch[val[k][j]] = epoch;       // evict_backlist membership [A.3-3]
    }
    if (dense[k]) mm[k] = miss_mask_stamp(inverted, row0 + 32 * k, wds[k], N, slot_epoch, epoch);
  }
}

// ids -> rows -> bits in the row bitmap.
//
// Hot rows share bitmap words (rank order puts the hottest 32 rows in word 0) and a Criteo window sends >100k ids at
// a 3-row table, so a global atomicOr per id would serialise.  Rows in frequency order (idx_map present): the
// lowest `hot_words` words live in an LDS window per workgroup and are flushed once at the end; a cold id first
// LOOKS at its word and only issues the (fire-and-forget) atomic when its bit is still clear.  Rows in id order
// (MERGE): hot rows are scattered, so the lanes of a wave that still aim at the same word are merged with ballots
// and one lane issues the atomicOr for all of them.
// (Round 3 tried to count the unique / missing rows here as well -- the thread whose atomicOr sets a bit first owns
// the row -- so that the bitmap would be scanned once instead of twice: the returning atomics that needs cost 40 us,
// more than the k_count pass they replaced; round 5 folded the repeats of every 8192-id chunk in an LDS hash table
// first: 7x slower, the same-word atomics of the hot rows that the LDS window below absorbs.  docs/history.md.)
// U ids per thread are in flight (a chain of three dependent random accesses per id).
// rows_out: the row of every id (-1 = bad id), as int64 in the caller's slots buffer -- k_slots turns it into the
// slot in place, so idx_map is gathered once per id per call.
// The body of k_mark: a grid-stride pass of the calling grid over the ids.  *cold += this
// thread's lookups of rows that are not resident; *bad = it met an id outside [0, N) that is not accepted padding.
template <bool MERGE, int U>
__device__ __forceinline__ void mark_pass(const int64_t* __restrict__ ids, int64_t n,
                                          const int32_t* __restrict__ idx_map, const int32_t* __restrict__ inverted,
                                          int64_t N, int word_bits, int hot_words, uint32_t* hot, uint32_t* bitmap,
                                          int64_t* rows_out, int allow_pad, int* cold_out, bool* bad_out) {
  for (int w = threadIdx.x; w < hot_words; w += blockDim.x) hot[w] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * U;
  int cold = 0;
  bool bad = false;
  // wave-uniform trip count: a wave owns U * 64 consecutive ids per iteration
  for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63)) * U; i0 < n; i0 += stride) {
    int32_t row[U], inv[U];
    uint32_t cur[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * 64 + lane;
      valid[u] = i < n;
      row[u] = 0;
      if (valid[u]) {
        const int64_t id = ids[i];
        if ((unsigned long long)id >= (unsigned long long)N) {
          // ce_cache_prepare_ids_padded only: -1 = padding (fixed-capacity exchange), no lookup, slot -1.  On the
          // plain entry point a -1 is a bad id like any other (upstream's idx_map.index_select raises on it).
          if (!(allow_pad && id == -1)) bad = true;
          valid[u] = false;
          rows_out[i] = -1;
        } else {
          row[u] = idx_map ? idx_map[id] : (int32_t)id;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + u * 64 + lane;
      inv[u] = 0;
      cur[u] = ~0u;
      if (valid[u]) {
        rows_out[i] = row[u];
        inv[u] = inverted[row[u]];
        const int word = row[u] >> 5;
        // (looking at hot[word] first and skipping the LDS atomic when the bit is set -- what the cold path does with
        // the global bitmap -- measured in round 4: 67.2 against 65.1 us, no gain.  Also round 4: a RESIDENCY BITMAP
        // (bit r = row r is resident, kept by the admit / evict kernels) in place of the inverted[] gathers of this
        // kernel, k_count and k_emit: k_count 13.6 -> 7.8 us, but k_mark 65 -> 69, k_emit 31 -> 37 and the slots +
        // keys kernel 32 -> 38 -- every lookup needs inverted[row] once anyway (its slot), and this kernel's gather is
        // what has it in L2 when the later kernels ask; 328 GPU tests green, no net gain, not kept.)
        if (word < hot_words) atomicOr(&hot[word], 1u << (row[u] & 31));
        else cur[u] = *(volatile uint32_t*)(bitmap + word);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int word = row[u] >> 5;
      const int bidx = row[u] & 31;
      const bool need = valid[u] && (cur[u] & (1u << bidx)) == 0;
      if (valid[u]) cold += inv[u] < 0;
      if (!MERGE) {
        if (need) atomicOr(bitmap + word, 1u << bidx);
      } else if (__any(need)) {
        unsigned long long pm = __ballot(need);
        if (!need) pm = 0;
        for (int b = 0; b < word_bits; ++b) {
          const unsigned long long m = __ballot((word >> b) & 1);
          pm &= ((word >> b) & 1) ? m : ~m;
        }
        uint32_t orbits = 0;
#pragma unroll
        for (int b = 0; b < 32; ++b) {
          const unsigned long long m = __ballot(need && bidx == b);
          if (m & pm) orbits |= (1u << b);
        }
        if (need && (__ffsll((long long)pm) - 1) == lane) atomicOr(bitmap + word, orbits);
      }
    }
  }
  __syncthreads();
  // the window goes out with all of a thread's looks at the bitmap in flight together (one word after the other was up
  // to 16 dependent round trips: half of this kernel's time at 426 k ids)
  for (int w0 = threadIdx.x; w0 < hot_words; w0 += (int)blockDim.x * 8) {
    uint32_t v[8], cur[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int w = w0 + q * (int)blockDim.x;
      v[q] = w < hot_words ? hot[w] : 0u;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) cur[q] = v[q] ? *(volatile uint32_t*)(bitmap + w0 + q * (int)blockDim.x) : ~0u;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (v[q] && (cur[q] & v[q]) != v[q]) atomicOr(bitmap + w0 + q * (int)blockDim.x, v[q]);
  }
  *cold_out += cold;
  *bad_out = *bad_out || bad;
}

template <bool MERGE, int U>
__global__ __launch_bounds__(1024) void k_mark(const int64_t* __restrict__ ids, int64_t n,
                                              const int32_t* __restrict__ idx_map,
                                              const int32_t* __restrict__ inverted, int64_t N, int word_bits,
                                              int hot_words, uint32_t* bitmap, Ctl* ctl, int64_t* rows_out,
                                              int allow_pad) {
  extern __shared__ uint32_t hot[];
  int cold = 0;
  bool bad = false;
  mark_pass<MERGE, U>(ids, n, idx_map, inverted, N, word_bits, hot_words, hot, bitmap, rows_out, allow_pad, &cold, &bad);
  if (bad) ctl->status = CE_ERR_RANGE;
  cold = wave_sum(cold);
  if ((threadIdx.x & 63) == 0 && cold) atomicAdd((unsigned long long*)&ctl->miss_lookups, (unsigned long long)cold);
}

// unique / missing rows per 32768-row chunk of the bitmap (one uint4 = 128 rows per thread), and their sums per 64
// chunks (two device atomics per workgroup on ~85 addresses: k_emit adds those up instead of 5431 chunk counts)
__global__ __launch_bounds__(256) void k_count(const uint4* __restrict__ bitmap4,
                                               const int32_t* __restrict__ inverted, int64_t N,
                                               int32_t* blk_unique, int32_t* blk_miss, int32_t* coarse) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const uint4 q = bitmap4[v];
  const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
  int u = 0, m = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (!wds[k]) continue;
    u += __popc(wds[k]);
    m += __popc(miss_mask(inverted, v * 128 + k * 32, wds[k], N));
  }
  __shared__ int su[4], sm[4];
  u = wave_sum(u);
  m = wave_sum(m);
  if ((threadIdx.x & 63) == 0) {
    su[threadIdx.x >> 6] = u;
    sm[threadIdx.x >> 6] = m;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tu = su[0] + su[1] + su[2] + su[3], tm = sm[0] + sm[1] + sm[2] + sm[3];
    blk_unique[blockIdx.x] = tu;
    blk_miss[blockIdx.x] = tm;
    if (tu) atomicAdd(&coarse[(2 * (blockIdx.x >> kCoarseShift)) * kCoarsePad], tu);
    if (tm) atomicAdd(&coarse[(2 * (blockIdx.x >> kCoarseShift) + 1) * kCoarsePad], tm);
  }
}

// Ordered emission of the missing rows + the plan.  The bitmap is scanned in 32768-row chunks (one uint4 = 128 rows
// per thread of a 256-thread workgroup); a workgroup takes kEmitSub chunks, strided by the grid size, with all its
// loads in flight at once: the whole grid is then resident at the same time (1358 workgroups at N = 178 M) instead of
// running in 2.7 rounds of short latency-bound workgroups, and the dense chunks of the hot rows (the lowest ones)
// land in different workgroups.  A chunk's place in the miss list is the number of missing rows in the chunks
// before it; every workgroup adds that up itself -- k_count's sums per 64 chunks plus the chunk counts of the
// chunk's own group of 64: ~150 values out of L2 -- so there is neither a scan kernel (k_plan: 13.6 us + a launch)
// nor a chain between workgroups.  The same sums give every workgroup the call's totals and hence the same verdict;
// workgroup 0 also records it: capacity check, k = misses - free slots, the stats record, the mailbox.
constexpr int kEmitSub = 1;      // (4 chunks per workgroup, contiguous or strided, measured SLOWER: 111 / 56 us against 46)
__global__ __launch_bounds__(256) void k_emit(uint4* bitmap4, const int32_t* __restrict__ inverted, int64_t N,
                                              const int32_t* __restrict__ blk_miss, const int32_t* __restrict__ coarse,
                                              int n_chunks, int32_t* miss_list, int32_t* slot_epoch, long long seq_arg,
                                              Ctl* ctl, int64_t C, int64_t n_ids, ce_call_stats_t* ring,
                                              WbMail* mail_in, long long job, long long in_cap, int32_t* miss_host,
                                              int assume_free0, long long* n_admit_out = nullptr) {
  const long long seq_ = call_seq(ctl, seq_arg);
  const int32_t epoch = call_epoch(seq_);
  ce_call_stats_t* const ring_slot = ring + (seq_ % kRing);
  __shared__ long long red[2 + kEmitSub][4];
  __shared__ int wsub[kEmitSub][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int bid = (int)blockIdx.x, G = (int)gridDim.x;
  const int st_in = __hip_atomic_load(&ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int chunk[kEmitSub];
  uint4 q[kEmitSub];
#pragma unroll
  for (int j = 0; j < kEmitSub; ++j) {
    chunk[j] = bid + j * G;
    q[j] = make_uint4(0, 0, 0, 0);
    if (chunk[j] < n_chunks) q[j] = bitmap4[(int64_t)chunk[j] * 256 + threadIdx.x];
  }
  // totals, and every chunk's base, from the coarse sums + the chunk counts of the chunk's own group of 64
  const int n_coarse = (n_chunks >> kCoarseShift) + 1;
  long long tu_p = 0, tm_p = 0, base_p[kEmitSub];
#pragma unroll
  for (int j = 0; j < kEmitSub; ++j) base_p[j] = 0;
  for (int g = threadIdx.x; g < n_coarse; g += 256) {
    const int cu = coarse[(2 * g) * kCoarsePad], cm = coarse[(2 * g + 1) * kCoarsePad];
    tu_p += cu;
    tm_p += cm;
#pragma unroll
    for (int j = 0; j < kEmitSub; ++j)
      if (g < (chunk[j] >> kCoarseShift)) base_p[j] += cm;
  }
#pragma unroll
  for (int j = 0; j < kEmitSub; ++j) {
    const int i = ((chunk[j] >> kCoarseShift) << kCoarseShift) + (int)(threadIdx.x & 63);
    if ((int)(threadIdx.x >> 6) == j && i < chunk[j] && chunk[j] < n_chunks) base_p[j] += blk_miss[i];      // wave j: chunk j
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    tu_p += __shfl_xor(tu_p, d);
    tm_p += __shfl_xor(tm_p, d);
#pragma unroll
    for (int j = 0; j < kEmitSub; ++j) base_p[j] += __shfl_xor(base_p[j], d);
  }
  if (lane == 0) {
    red[0][wv] = tu_p;
    red[1][wv] = tm_p;
#pragma unroll
    for (int j = 0; j < kEmitSub; ++j) red[2 + j][wv] = base_p[j];
  }
  __syncthreads();
  const long long tu = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  const long long tm = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  // assume_free0: the host launched the call in its steady-state form (no free-list scan: the slots to fill are the
  // victims, see k_evict_stage) because the last record it has seen said "no free slot left".  Nothing but a flush or
  // a lost admission raises the count again and the host knows of both -- but a call launched before the record of
  // a lost admission arrived must not pair missing rows with a stale list: it fails, state untouched.
  const bool stale = assume_free0 && ctl->n_free_start != 0;
  // workgroup 0 may already have turned CE_OK into CE_ERR_CAPACITY below: the verdict is the same either way
  const bool ok = st_in == CE_OK && tu <= C && !stale;
  if (bid == 0 && threadIdx.x == 0) {
    int status = st_in;
    if (status == CE_OK && tu > C) status = CE_ERR_CAPACITY;
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
    if (mail_in) {      // rows the admission worker gathers for this call (read after the event behind this kernel)
      const long long mrows = (status == CE_OK) ? tm : 0;
      mail_in->count = mrows < in_cap ? mrows : in_cap;
      mail_in->job = job;
    }
    if (n_admit_out) {  // chained admission: the same count, for the admission kernel behind this kernel's event
      const long long mrows = (status == CE_OK) ? tm : 0;
      *n_admit_out = mrows < in_cap ? mrows : in_cap;
    }
    // seq (the "record complete" marker) is published by the last kernel of the call that may still amend the
    // record (k_victims can turn it into a capacity failure): k_admit_maps
  }
  // ONE pass over the map entries of the rows seen: which of them are missing, and the call's stamp on the slots of
  // the others (a failed call stamps nothing and emits nothing)
  uint32_t mm[kEmitSub][4];
  int m[kEmitSub], inc[kEmitSub];
#pragma unroll
  for (int j = 0; j < kEmitSub; ++j) {
    m[j] = 0;
    const uint32_t wds[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
    const int64_t v = (int64_t)chunk[j] * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mm[j][k] = 0;
      if (ok && wds[k]) {
        mm[j][k] = miss_mask_stamp(inverted, v * 128 + k * 32, wds[k], N, slot_epoch, epoch);
        m[j] += __popc(mm[j][k]);
      }
    }
    inc[j] = wave_incl_scan(m[j], lane);
    if (lane == 63) wsub[j][wv] = inc[j];
  }
  __syncthreads();
  if (ok) {
#pragma unroll
    for (int j = 0; j < kEmitSub; ++j) {
      if (chunk[j] >= n_chunks) continue;
      int pos = (int)(red[2 + j][0] + red[2 + j][1] + red[2 + j][2] + red[2 + j][3]) + inc[j] - m[j];
      for (int k = 0; k < wv; ++k) pos += wsub[j][k];
      const int64_t v = (int64_t)chunk[j] * 256 + threadIdx.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t bits = mm[j][k];
        const int64_t row0 = v * 128 + k * 32;
        while (bits) {
          const int b = __ffs(bits) - 1;
          bits &= bits - 1;
          if (miss_host && pos < in_cap) miss_host[pos] = (int32_t)(row0 + b);      // the admission worker's copy
          miss_list[pos++] = (int32_t)(row0 + b);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kEmitSub; ++j)
    if (chunk[j] < n_chunks && (q[j].x | q[j].y | q[j].z | q[j].w))
      bitmap4[(int64_t)chunk[j] * 256 + threadIdx.x] = make_uint4(0, 0, 0, 0);
}

}  // namespace ce
