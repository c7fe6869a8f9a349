/*
 * ce_api.h -- C ABI of the MI355X-native frequency-aware cached EmbeddingBag.
 *
 * The reference (hpcaitech/CachedEmbedding) has no FFI: its boundary for this path is a
 * Python nn.Module API implemented by ColossalAI's CachedParamMgr / CachedEmbeddingBag in
 * pure torch ops.  Each entry point below replaces one of those torch-op sequences; the
 * Python mirror in cachedembedding_amd/ (ctypes) is what a reference maintainer binds
 * (see INTEGRATION.md).  Citations are relative to /root/reference; "[A.x]" refers to the
 * upstream semantics recorded in SURVEY.md Appendix A.
 *
 * Conventions: plain pointers and sizes only (no torch types); every device pointer is a
 * raw HIP device address; `stream` is a hipStream_t passed as void*; all work is enqueued
 * asynchronously on that stream with NO hidden host synchronisation unless a function
 * says it blocks; every function returns an int status (CE_OK == 0) and
 * ce_last_error() gives the message of the last failure on the calling thread.
 * Handles are not thread-safe (the reference is single-threaded per process too).
 */
#ifndef CE_API_H
#define CE_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CE_API_VERSION 6

/* status codes */
#define CE_OK 0
#define CE_ERR_INVALID 1     /* bad argument */
#define CE_ERR_HIP 2         /* a HIP runtime call failed */
#define CE_ERR_CAPACITY 3    /* unique rows of one prepare_ids call exceed cuda_row_num:
                                the reference's AssertionError caught at
                                benchmark/benchmark_cache.py:106-108 */
#define CE_ERR_NOMEM 4
#define CE_ERR_UNSUPPORTED 5
#define CE_ERR_RANGE 6       /* an id outside [0, num_embeddings) */

/* EvictionStrategy (recsys/models/dlrm.py:66,80) */
#define CE_EVICT_DATASET 0
#define CE_EVICT_LFU 1

/* nn.EmbeddingBag mode (recsys/models/dlrm.py:74 passes 'sum') */
#define CE_MODE_SUM 0
#define CE_MODE_MEAN 1

/* how missed / evicted rows move between the host table and the HBM cache */
#define CE_TRANSPORT_ZEROCOPY 0  /* swap kernels read/write the mapped pinned host table over PCIe */
#define CE_TRANSPORT_STAGED 1    /* host threads gather/scatter through pinned staging + hipMemcpyAsync */
#define CE_TRANSPORT_WORKER 2    /* both directions leave the CUs: evictions are packed in HBM by the cache op, copied
                                    out with pinned hipMemcpyAsync (SDMA) on a private stream and scattered into the
                                    host table by a worker thread inside the library; missed rows are brought into
                                    an HBM staging block under the control of a second worker thread -- it waits for
                                    the miss list and for earlier write-backs to land, then either launches a
                                    small kernel on a private stream that reads them out of the mapped table
                                    (default; it does not wait for the write-back of the call just before: a row that
                                    call evicted is taken out of its staging block, still in HBM), or gathers them
                                    into pinned staging and copies that with
                                    hipMemcpyAsync (CE_WORKER_ADMIT=sdma, and tables without a device mapping) --
                                    and the cache-op stream waits for them in a hipStreamWaitValue64 after it has
                                    selected and staged the victims; the arrived rows are then copied into their
                                    slots, the maps updated and the call's slots (and keys) written (CE_EARLY_MAPS=1:
                                    maps, slots and keys before the wait).  An admission the worker reports lost (a
                                    HIP call of its own failed / timed out) gives the call CE_ERR_HIP with nothing
                                    marked resident; every later call fails.  prepare_ids stays
                                    one asynchronous call, but returns before the host table has the evicted rows
                                    (ce_cache_writeback_wait / ce_cache_flush make it current).  Not capture-safe.
                                    The library does not trust the environment for this: the first call on a stream
                                    self-tests that its copy streams run while that stream is parked (falls back to
                                    ZEROCOPY with a message otherwise), every wait on a worker has a deadline
                                    (CE_WORKER_TIMEOUT_S, default 30 s), and an admission that fails or times out
                                    releases the stream with the call flagged CE_ERR_HIP and nothing admitted. */

typedef void* ce_stream_t; /* hipStream_t */
typedef struct ce_cache ce_cache_t;

int ce_version(void);
/* CPUs this process may use: hardware threads capped by the affinity mask and the cgroup CPU quota (what the
 * library sizes its helper-thread pools against). */
int32_t ce_cpu_budget(void);
const char* ce_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Side stream for the cache manager (the "Stream1" lane of pics/prefetch.png).  The cache op is
 * PCIe/latency bound, the training kernels HBM bound; if both share all 256 CUs the cache op's
 * resident waves take wave slots from training.  ce_stream_create_cu_mask returns a HIP stream
 * restricted to the CUs whose bits are set in cu_mask (word w bit b = CU 32*w + b), so e.g. one XCD
 * (32 CUs) serves the cache manager and 7 XCDs keep training.  words == 0: an ordinary stream. */
int ce_stream_create_cu_mask(const uint32_t* cu_mask, int32_t words, ce_stream_t* out);
int ce_stream_destroy(ce_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Host table (the `weight` of CachedParamMgr, A.1; `pin_weight=True` at
 * benchmark/benchmark_fbgemm_uvm.py:98-105).  Pinned + device-mapped host memory.
 * ce_host_alloc pins `bytes` (first-touched by `threads` workers so a 91 GB table is
 * spread over the NUMA nodes; blocks of 64 MB and more are mapped 2 MB-aligned with
 * MADV_HUGEPAGE and then registered -- the swap workers touch one row per page -- unless
 * CE_HOST_THP=0 or the registration is refused); ce_host_register pins memory the caller already owns
 * (e.g. a `_weight` tensor).  Both return in *dev_ptr the address device code must use.
 */
int ce_host_alloc(size_t bytes, int threads, void** host_ptr, void** dev_ptr);
int ce_host_free(void* host_ptr);
int ce_host_register(void* host_ptr, size_t bytes, void** dev_ptr);
int ce_host_unregister(void* host_ptr);
/* weight init of CachedEmbeddingBag: uniform_(-1/N, 1/N) (A.7); counter-based RNG so the
 * result is independent of `threads`. */
int ce_host_fill_uniform(float* dst, int64_t n, float lo, float hi, uint64_t seed, int threads);
/* The values ce_host_fill_uniform(dst, N * dim, lo, hi, seed, .) gave the rows `rows[0..n)` of an [N, dim] table,
 * regenerated on the device into out[n, dim] (bit-identical: the generator is counter-based).  What a checker needs to
 * state "row r still holds its initial value" / "row r = initial value - lr * sum of its gradients" for a 91 GB table
 * without keeping a second copy of it. */
int ce_host_fill_uniform_rows(const int64_t* rows, int64_t n, int32_t dim, float lo, float hi, uint64_t seed,
                              float* out, ce_stream_t stream);
/* out[i] = table[rows[i]] (i < n) read through the device mapping `table_dev` of a pinned host table (the *dev_ptr of
 * ce_host_alloc / ce_host_register); rows outside [0, num_rows) give zeros.  `rows` and `out` are device memory.
 * (CachedParamMgr.cpu_weight_data for many rows at once; a PCIe-bound read.) */
int ce_host_rows_gather(const float* table_dev, int64_t num_rows, int32_t dim, const int64_t* rows, int64_t n,
                        float* out, ce_stream_t stream);
/* Memory-system probe of the box the process runs on: `reps` streaming reads, then `reps` fills, of `bytes` of
 * device scratch, each block of launches bracketed by hipEvents on `stream` (blocks until done).  A bench line carries
 * the two rates so that a reader can tell a slow box from slow code (allocations of the same GPU model differ by
 * several per cent). */
int ce_box_probe(void* scratch, size_t bytes, int32_t reps, double* read_GBps, double* fill_GBps, ce_stream_t stream);
/* How fast is THIS buffer when its rows are visited `fold` rows apart (API 5)?  The hook-folded forward stores -- and
 * the streaming backward reads -- 512-byte rows of a [B, F, D] tensor in an order that touches a different page with
 * every row (row b * F + f for consecutive b): how long that takes depends on how the allocation is mapped (the same
 * launch measures 38 or 45 us on two buffers of one process: profiles/r05_alloc_lottery.txt), not on its address or on
 * the table.  `reps` passes of row stores, then of row loads, over buf = device fp32 [rows, dim] (contents are
 * overwritten), hipEvent-bracketed on `stream`; blocks until done.  cachedembedding_amd.functional.pick_fast_buffer
 * allocates a few candidates and keeps the fastest. */
int ce_probe_rows(float* buf, int64_t rows, int32_t dim, int64_t fold, int32_t reps, double* write_us, double* read_us,
                  ce_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K12: F.embedding_bag(slots, cuda_cached_weight, offsets, mode, per_sample_weights,
 * include_last_offset) -- reached from recsys/models/dlrm.py:99-110 and
 * benchmark/benchmark_cache.py:62 [A.7].  fp32 rows, int64 indices, int32 or int64 offsets.
 *   out[bag] = sum_j psw[j] * weight[indices[j]]   (/ len for CE_MODE_MEAN)
 * hook_features == 0: out is [num_bags, dim].  hook_features == F > 0 folds
 * sparse_embedding_shape_hook (recsys/models/dlrm.py:26-27) into the store: bags are
 * feature-major (bag = f*B + b, B = num_bags/F) and out is written as [B, F, dim].
 * include_last_offset == 0: offsets has num_bags entries and the last bag ends at nnz.
 * offsets == NULL (only with num_bags == nnz) states the one-id-per-bag layout (offsets = arange: every Criteo /
 * Avazu batch, recsys/datasets/criteo.py:127-134): the kernel then reads no offsets at all.
 */
int ce_bag_forward(const float* weight, int64_t num_rows, int32_t dim,
                   const int64_t* indices, int64_t nnz,
                   const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                   int32_t include_last_offset, const float* per_sample_weights, int32_t mode,
                   int64_t hook_features, float* out, ce_stream_t stream);

/* K13 (dense form): grad_weight[indices[j]] += psw[j]*scale*grad_out[bag(j)] accumulated into a
 * caller-zeroed [num_rows, dim] buffer -- the `sparse=False` autograd backward of K12.  Duplicates are
 * folded per 1024-lookup tile (LDS sort) before one fp32 atomic row update per (row, chunk).
 * grad_out uses the same layout convention as `out` above. */
int ce_bag_backward_dense(float* grad_weight, int64_t num_rows, int32_t dim,
                          const int64_t* indices, int64_t nnz,
                          const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                          int32_t include_last_offset, const float* per_sample_weights, int32_t mode,
                          int64_t hook_features, const float* grad_out, ce_stream_t stream);

/* K13 (sparse form): values of the COO gradient that `sparse=True` produces
 * (scripts/kaggle.sh:71 --use_sparse_embed_grad): grad_rows[j] = psw[j]*scale*grad_out[bag(j)],
 * one row per lookup, paired with `indices` as the COO indices.  dest_index (device int64[nnz],
 * NULL = identity) redirects row j to grad_rows[dest_index[j]] -- the row-wise sharded backward
 * uses it to emit the rows already in owner-bucket order. */
int ce_bag_backward_rows(float* grad_rows, const int64_t* dest_index, int32_t dim, int64_t nnz,
                         const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                         int32_t include_last_offset, const float* per_sample_weights, int32_t mode,
                         int64_t hook_features, const float* grad_out, ce_stream_t stream);

/* K13+K14 fused: weight[indices[j]] -= lr * psw[j]*scale*grad_out[bag(j)] -- autograd
 * backward + torch.optim.SGD.step on the cache parameter (recsys/dlrm_main.py:274-279,
 * 455-461) in one pass.  Same tile-sorted scatter as the dense form with alpha = -lr; the order in which
 * DIFFERENT tiles update one row is not fixed (fp32 sums may differ in the last bits run to run). */
int ce_bag_backward_sgd(float* weight, int64_t num_rows, int32_t dim,
                        const int64_t* indices, int64_t nnz,
                        const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                        int32_t include_last_offset, const float* per_sample_weights, int32_t mode,
                        int64_t hook_features, const float* grad_out, float lr, ce_stream_t stream);

/* The (row, lookup) order the backward folds duplicates in depends only on the slots, so it is computed ahead of
 * the backward, once per prefetch window on the cache-op stream right after ce_cache_prepare_ids, and over a wider
 * scope than a workgroup can sort on the fly: every SEGMENT of 16384 consecutive lookups is grouped by row (one
 * counting pass in LDS) into keys (row << 32 | lookup-in-segment, all-ones = ignored / padding, placed last).
 * ce_bag_presort handles one batch: keys_out is device uint64[ce_bag_presort_len(nnz)] (nnz rounded up to whole
 * segments).  ce_bag_presort_window handles the n_batches equal-sized batches of a window in ONE launch (indices =
 * the window's slots, batch b at [b * nnz_per_batch, (b + 1) * nnz_per_batch); keys of batch b at keys_out +
 * b * ce_bag_presort_len(nnz_per_batch); a segment never straddles two batches).
 * The *_presorted backward entry points consume the keys: same result as the unsorted forms up to fp32 summation
 * order, about half the atomic row updates. */
int64_t ce_bag_presort_len(int64_t nnz);
int ce_bag_presort(const int64_t* indices, int64_t nnz, int64_t num_rows, uint64_t* keys_out, ce_stream_t stream);
int ce_bag_presort_window(const int64_t* indices, int64_t nnz_per_batch, int64_t n_batches, int64_t num_rows,
                          uint64_t* keys_out, ce_stream_t stream);
int ce_bag_backward_sgd_presorted(float* weight, int64_t num_rows, int32_t dim,
                                  const int64_t* indices, int64_t nnz,
                                  const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                                  int32_t include_last_offset, const float* per_sample_weights, int32_t mode,
                                  int64_t hook_features, const float* grad_out, float lr,
                                  const uint64_t* presorted_keys, ce_stream_t stream);
/* the same for the plain accumulation (ce_bag_backward_dense): grad_weight += folded gradients */
int ce_bag_backward_dense_presorted(float* grad_weight, int64_t num_rows, int32_t dim,
                                    const int64_t* indices, int64_t nnz,
                                    const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                                    int32_t include_last_offset, const float* per_sample_weights, int32_t mode,
                                    int64_t hook_features, const float* grad_out,
                                    const uint64_t* presorted_keys, ce_stream_t stream);
/* Source-row keys: for mode = sum without per-sample weights every lookup scales its gradient row by the same
 * factor, so the only per-lookup facts the backward needs are the target row and WHICH row of grad_out to read.
 * ce_bag_presort_window_src resolves the second one at window time too (the bag of the lookup from the batch's
 * offsets, then its place in the [B, F, D] / [num_bags, D] output): key = row << 32 | grad_out row, same grouping
 * and layout as ce_bag_presort_window.  offsets of batch b = offsets + b * offsets_batch_stride elements (0: all
 * batches share one offsets array); num_bags / include_last_offset / hook_features describe ONE batch, as in
 * ce_bag_forward.  offsets == NULL (only with num_bags == nnz_per_batch) states the one-id-per-bag layout
 * (offsets = arange: every Criteo / Avazu batch, recsys/datasets/criteo.py:127-134) without the kernel having to read
 * the offsets to establish it.  The *_presorted_src backward entry points stream over such keys with no per-tile set-up
 * (67 vs 74 us at the bench shape); they trust the keys (grad_out row < num_bags) and ignore rows >= num_rows.
 * An ignored lookup (index outside [0, num_rows), e.g. slot -1) keeps its grad_out / output row in the low word under
 * the row 0xffffffff (API 4; all ones before) -- the backward skips it like padding, ce_bag_forward_src_keys writes its
 * zero row; only the positions beyond nnz_per_batch in the last segment are all ones.
 * Replaces the same upstream call as the forms above (recsys/dlrm_main.py:274-279, loss.backward + optimizer.step). */
int ce_bag_presort_window_src(const int64_t* indices, int64_t nnz_per_batch, int64_t n_batches, int64_t num_rows,
                              const void* offsets, int32_t offsets_are_i64, int64_t offsets_batch_stride,
                              int64_t num_bags, int32_t include_last_offset, int64_t hook_features,
                              uint64_t* keys_out, ce_stream_t stream);
/* K12 from the same keys, for the one-id-per-bag layout (every Criteo / Avazu batch; mode sum, no per-sample
 * weights): out[low word of key] = weight[high word of key], a zero row for an ignored lookup.  The keys are grouped by
 * row, so a cache row is LOADED once per run of equal rows and stored to every output row of the run (~50 k row loads
 * per Criteo batch instead of 425,984): the forward becomes a fill of its output instead of a copy.  `out` is
 * [nnz, dim] (or [B, F, dim] when the keys were built with hook_features = F); only valid for keys built from a layout
 * in which every bag holds exactly one lookup. */
int ce_bag_forward_src_keys(const float* weight, int64_t num_rows, int32_t dim, int64_t nnz, const uint64_t* src_keys,
                            float* out, ce_stream_t stream);
int ce_bag_backward_sgd_presorted_src(float* weight, int64_t num_rows, int32_t dim, int64_t nnz,
                                      const float* grad_out, float lr, const uint64_t* src_keys, ce_stream_t stream);
int ce_bag_backward_dense_presorted_src(float* grad_weight, int64_t num_rows, int32_t dim, int64_t nnz,
                                        const float* grad_out, const uint64_t* src_keys, ce_stream_t stream);
/* Owner-exclusive form (API 3).  The fp32 atomics of the fused update retire at ~1 float per clock and L2 channel
 * (24 us for a Criteo-shaped batch, not overlapped with the gather), so rows that ONE lane group of the backward
 * can own are taken off them.  ce_bag_presort_window_src_excl additionally (a) sorts every bucket of <= 32 keys by
 * row and sets bit 31 of the low word of a key that heads a run holding ALL lookups of its row in its segment and
 * lying inside one 16-position block, and (b) records per segment the [min, max] of `ids` (device int64
 * [n_batches * nnz_per_batch], the ids `indices` was computed from, position by position; NULL = unknown) in
 * seg_id_ranges (device int64 [n_batches * segments_per_batch][2]).  ce_bag_backward_sgd_presorted_src_excl takes a
 * batch's keys and its ranges (seg_id_ranges + b * 2 * segments_per_batch): when the ranges are pairwise disjoint no
 * id -- hence no row -- occurs in two segments, a flagged run is the only writer of its row in the launch, and it is
 * applied as old row + folded gradients with one plain store; otherwise (or for unflagged runs) the atomics are
 * used, so the result never depends on the flags being usable.  Keys with flags are also valid input for the plain
 * *_presorted_src entry points (they ignore the flag). */
int ce_bag_presort_window_src_excl(const int64_t* indices, int64_t nnz_per_batch, int64_t n_batches, int64_t num_rows,
                                   const void* offsets, int32_t offsets_are_i64, int64_t offsets_batch_stride,
                                   int64_t num_bags, int32_t include_last_offset, int64_t hook_features,
                                   const int64_t* ids, uint64_t* keys_out, int64_t* seg_id_ranges,
                                   ce_stream_t stream);
int ce_bag_backward_sgd_presorted_src_excl(float* weight, int64_t num_rows, int32_t dim, int64_t nnz,
                                           const float* grad_out, float lr, const uint64_t* src_keys,
                                           const int64_t* seg_id_ranges, ce_stream_t stream);

/* Deterministic variant of the fused update: lookups are stably radix-sorted by target row
 * (workspace from ce_bag_backward_sgd_sorted_workspace), each row's gradients are summed in
 * lookup order and applied once:  W[r] -= lr * sum.  Matches the reference's coalesce-then-add
 * order for sparse grads; bit-reproducible, slower for very hot rows. */
size_t ce_bag_backward_sgd_sorted_workspace(int64_t num_rows, int64_t nnz);
int ce_bag_backward_sgd_sorted(float* weight, int64_t num_rows, int32_t dim,
                               const int64_t* indices, int64_t nnz,
                               const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                               int32_t include_last_offset, const float* per_sample_weights,
                               int32_t mode, int64_t hook_features, const float* grad_out, float lr,
                               void* workspace, size_t workspace_bytes, ce_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * CachedParamMgr [A.1-A.6].  Device state arrays are owned by the caller (so the Python
 * mirror can expose them as tensors: cached_idx_map, inverted_cached_idx, idx_map,
 * freq_cnter, cuda_cached_weight); the library owns only the opaque scratch it is handed.
 * Row ids are int32 on the device (num_embeddings < 2^31), counters int64.
 */
typedef struct ce_cache_config {
  int64_t num_embeddings;        /* N: rows of the host table                              */
  int64_t cuda_row_num;          /* C: rows resident in HBM (int(N*cache_ratio), A.7)       */
  int32_t embedding_dim;         /* D (fp32 elements per row)                              */
  int32_t evict_strategy;        /* CE_EVICT_*                                             */
  int32_t transport;             /* CE_TRANSPORT_*                                         */
  int32_t protect_depth;         /* 0 = reference semantics; d>0 also protects the rows of
                                    the previous d prepare_ids calls (overlapped pipeline)  */
  int64_t max_ids_per_call;      /* upper bound of n in ce_cache_prepare_ids                */
  float* host_weight;            /* [N, D] host table, HOST address                        */
  float* host_weight_dev;        /* same memory, DEVICE-visible address (ce_host_*)        */
  float* cache_weight;           /* device [C, D]  cuda_cached_weight                      */
  int32_t* idx_map;              /* device int32[N] id -> cpu_row_idx, NULL = identity     */
  int32_t* inverted_cached_idx;  /* device int32[N] cpu_row_idx -> slot, -1 = absent (16-B aligned) */
  int32_t* cached_idx_map;       /* device int32[C] slot -> cpu_row_idx, -1 = empty        */
  int64_t* freq_cnter;           /* device int64[C] (LFU) or NULL                          */
  void* workspace;               /* device scratch, ce_cache_workspace_bytes() bytes       */
  size_t workspace_bytes;
} ce_cache_config_t;

/* per-call statistics [A.3-4]: what num_hits_history / num_miss_history /
 * num_write_back_history record (recsys/dlrm_main.py:286-289) */
typedef struct ce_call_stats {
  int64_t seq;          /* call number, 1-based                      */
  int64_t n_ids;        /* ids in the call                           */
  int64_t n_unique;     /* unique rows                               */
  int64_t n_miss;       /* unique rows not resident                  */
  int64_t n_evict;      /* rows written back + evicted               */
  int64_t miss_lookups; /* sum of multiplicities of missed rows      */
  int64_t n_free_after; /* free slots after the call                 */
  int32_t status;       /* CE_OK, CE_ERR_CAPACITY or CE_ERR_RANGE    */
  int32_t kind;         /* CE_CALL_*                                 */
} ce_call_stats_t;

#define CE_CALL_PREPARE 0
#define CE_CALL_PRELOAD 1
#define CE_CALL_FLUSH 2

size_t ce_cache_workspace_bytes(int64_t num_embeddings, int64_t cuda_row_num, int64_t max_ids_per_call,
                                int32_t embedding_dim);

/* Builds the manager over caller-owned arrays and initialises them to the empty-cache
 * state of A.1 (maps = -1, freq_cnter = INT64_MAX, cache rows untouched).  Blocks. */
int ce_cache_create(const ce_cache_config_t* cfg, ce_stream_t stream, ce_cache_t** out);
int ce_cache_destroy(ce_cache_t* h);

/* Warm-up preload of reorder() [A.2-2]: rows[i] (device int32, NULL = i) -> slot i for
 * i < n, with freq_vals (device int64, NULL = 0) written to freq_cnter under LFU. */
int ce_cache_preload(ce_cache_t* h, const int32_t* rows, const int64_t* freq_vals, int64_t n,
                     ce_stream_t stream);

/* LFU only: tells the manager that no freq_cnter value exceeds `bound` (the largest value handed to
 * ce_cache_preload); together with the ids seen since, this bounds every counter and lets the victim select skip the
 * radix passes above it.  Optional: without it the select runs all 8 byte passes until the bound is known. */
int ce_cache_set_freq_bound(ce_cache_t* h, int64_t bound);

/* prepare_ids [A.3] -- recsys/dlrm_main.py:259: unique rows of `ids` (device int64[n]; any id outside
 * [0, num_embeddings) -- -1 included -- fails the call with CE_ERR_RANGE, as upstream's idx_map.index_select raises) are
 * made resident (victim selection A.5 with the canonical tie rule, write-back, admit A.4),
 * slots_out (device int64[n]) receives inverted_cached_idx[idx_map[ids]] [A.6], LFU
 * counters gain the multiplicities.  Fully asynchronous on `stream`.  On overflow or a bad
 * id the call leaves every piece of state untouched and flags the status in its stats.
 * slots_out is SCRATCH from the call's first kernel on (the row of every id is parked there and converted to its slot
 * in place by the last kernel): nothing may read or write it on another stream until the call has finished, and it
 * must not alias `ids`. */
int ce_cache_prepare_ids(ce_cache_t* h, const int64_t* ids, int64_t n, int64_t* slots_out,
                         ce_stream_t stream);
/* prepare_ids for a prefetch window of n_batches equal batches (ids = [n_batches, nnz_per_batch]) that also leaves
 * the window's keys in keys_out (device uint64[n_batches * ce_bag_presort_len(nnz_per_batch)]): what
 * ce_cache_prepare_ids followed by ce_bag_presort_window (src_keys == 0) or ce_bag_presort_window_src (src_keys != 0;
 * the offsets / num_bags / include_last_offset / hook_features arguments as there) produce, but the call's last kernel
 * converts the rows to slots AND writes the keys in ONE pass instead of writing 8 bytes of slot per id for the presort
 * to read straight back (the reference's `_train` window block, recsys/dlrm_main.py:243-262, plus this build's
 * window presort).  slots_out is still filled.  Capturable like ce_cache_prepare_ids. */
int ce_cache_prepare_ids_keys(ce_cache_t* h, const int64_t* ids, int64_t n_batches, int64_t nnz_per_batch,
                              int64_t* slots_out, int32_t src_keys, const void* offsets, int32_t offsets_are_i64,
                              int64_t offsets_batch_stride, int64_t num_bags, int32_t include_last_offset,
                              int64_t hook_features, uint64_t* keys_out, ce_stream_t stream);
/* ce_cache_prepare_ids_keys (keys_out != NULL) or ce_cache_prepare_ids over a [n_batches, nnz_per_batch] window
 * (keys_out == NULL) issued in TWO HALVES on one stream, so that a caller can put work of its own between them:
 *   _begin   unique rows, misses (the admission worker starts fetching them); with the zero-copy / staged transports
 *            also victim selection, staging of the victims and the free-slot list -- everything that needs nothing from
 *            the host table;
 *   _finish  (worker transport, API 5: victim selection, staging of the victims -- the write-back worker takes them --,
 *            free-slot list: beside the admission kernel's PCIe reads these kernels ran 2-3 x slower than alone, behind
 *            the caller's steps they run alone and the admission starts as early as before; CE_SPLIT_AFTER_EMIT=0
 *            restores the API-4 split point), then the wait for the admitted rows, their unpacking, the map updates,
 *            slots (and keys).
 * Why: run on a side stream beside the training kernels, the cache op's kernels and the bag kernels slow each other
 * down by MORE than the cache op's own kernel time (all of them are bound by the same memory system); issued on the
 * TRAINING stream as begin(window k+1) -> the steps of window k -> finish(window k+1), nothing runs beside anything,
 * and the one thing that does take wall time without using the GPU -- the PCIe admission -- still overlaps with the
 * steps in between.  Window k's rows must be protected while the cache op of window k+1 selects victims:
 * protect_depth >= 1.
 * No other call on the handle between the two; not capturable. */
int ce_cache_prepare_ids_begin(ce_cache_t* h, const int64_t* ids, int64_t n_batches, int64_t nnz_per_batch,
                               int64_t* slots_out, int32_t src_keys, const void* offsets, int32_t offsets_are_i64,
                               int64_t offsets_batch_stride, int64_t num_bags, int32_t include_last_offset,
                               int64_t hook_features, uint64_t* keys_out, ce_stream_t stream);
int ce_cache_prepare_ids_finish(ce_cache_t* h, ce_stream_t stream);
/* The same, for callers whose id lists are PADDED to a fixed capacity (the row-wise exchange's fixed-size buckets):
 * an entry of -1 is padding -- it takes no part in the call and gets slot -1.  Every other id outside the table still
 * fails the call.  Opt-in on purpose: on the plain entry point a -1 sentinel leaking out of a data pipeline must fail
 * loudly, not train on zero rows. */
int ce_cache_prepare_ids_padded(ce_cache_t* h, const int64_t* ids, int64_t n, int64_t* slots_out,
                                ce_stream_t stream);
/* First half of ce_cache_prepare_ids_padded (API 5): as ce_cache_prepare_ids_begin, for a padded id list and without
 * keys; ce_cache_prepare_ids_finish enqueues the second half.  The owner-side cache op of the row-wise exchange in its
 * one-stream arrangement (parallel.GraphedShardedWindow(arrangement="interleaved")). */
int ce_cache_prepare_ids_begin_padded(ce_cache_t* h, const int64_t* ids, int64_t n, int64_t* slots_out,
                                      ce_stream_t stream);

/* Worker transport, chained admission (API 6; the default when the host table has a device mapping): the missed rows
 * of a call travel on a private admission stream -- admission kernel behind the call's front, unpack kernel behind its
 * victim selection -- while the call's own stream goes on with selection, maps and slots.  By default a prepare_ids
 * call (or its _finish half) ENDS by making its stream wait for the rows, so "everything the call did is ordered
 * before whatever the caller enqueues next on that stream" holds as for every other transport.
 * A pipeline that issues the next cache op on the same stream before anything reads the cache (prefetch_num = 1 with
 * the cache op on a side stream) can take that wait out of the cache-op stream's chain:
 *   ce_cache_set_deferred_rows(h, 1)   calls no longer wait for their rows;
 *   ce_cache_rows_ticket(h)            ticket of the most recent call (0: no chained call yet);
 *   ce_cache_wait_rows(h, t, stream)   `stream` waits until the rows of call `t` (and of every call before it) are in
 *                                      their slots -- before the first kernel that reads the cache rows of that call's
 *                                      slots.  t <= 0: the most recent call.  A no-op for the other transports.
 * ce_cache_flush / _preload / _last_stats / _set_transport order themselves behind the rows whatever the setting. */
int ce_cache_set_deferred_rows(ce_cache_t* h, int32_t on);
int64_t ce_cache_rows_ticket(ce_cache_t* h);
int ce_cache_wait_rows(ce_cache_t* h, int64_t ticket, ce_stream_t stream);

/* Blocks until the most recent prepare_ids/preload/flush has finished on the device and
 * returns that call's statistics; returns its status (CE_ERR_CAPACITY ...). */
int ce_cache_last_stats(ce_cache_t* h, ce_call_stats_t* out);
/* Non-blocking totals accumulated from finished calls (call ce_cache_last_stats first for
 * exact values): _cpu_to_cuda_numel, _cuda_to_cpu_numel, _cache_miss, _total_cache. */
int ce_cache_totals(ce_cache_t* h, int64_t* cpu_to_cuda_numel, int64_t* cuda_to_cpu_numel,
                    int64_t* cache_miss, int64_t* total_cache, int64_t* n_calls);
/* Non-blocking: how many FINISHED calls ended with a status other than CE_OK (capacity overflow, bad id), and the
 * status / call number of the latest one.  A pipeline that never blocks on a call's record (strict = False) polls
 * this once per window and raises the reference's AssertionError one window late instead of never. */
int ce_cache_failures(ce_cache_t* h, int64_t* n_failed, int32_t* last_status, int64_t* last_seq);
/* Copies up to `cap` per-call records of finished calls starting at call `first_seq` (the library keeps the most
 * recent 65536 records). */
int64_t ce_cache_history(ce_cache_t* h, int64_t first_seq, ce_call_stats_t* out, int64_t cap);

/* _id_to_cached_cuda_id [A.6] alone (no cache maintenance): slots_out = inverted[idx_map[ids]] */
int ce_cache_lookup_slots(ce_cache_t* h, const int64_t* ids, int64_t n, int64_t* slots_out,
                          ce_stream_t stream);

/* flush() [A.7]: every resident row is written back to the host table, maps emptied,
 * freq_cnter reset.  Asynchronous on `stream`. */
int ce_cache_flush(ce_cache_t* h, ce_stream_t stream);

int ce_cache_set_protect_depth(ce_cache_t* h, int32_t depth);
/* Switching away from / to CE_TRANSPORT_WORKER blocks until the queued write-backs have landed. */
int ce_cache_set_transport(ce_cache_t* h, int32_t transport);
/* The transport in force (CE_TRANSPORT_*).  It can differ from what was set: the first prepare_ids on a stream runs a
 * self-test of the worker transport (do the library's copy streams make progress while that stream is parked in
 * hipStreamWaitValue64? -- a property of the process's HIP runtime settings, e.g. GPU_MAX_HW_QUEUES) and falls back
 * to CE_TRANSPORT_ZEROCOPY, with a message on stderr, when they do not. */
int32_t ce_cache_get_transport(ce_cache_t* h);
/* Move the cache to another allocation (API 3): `cache_weight` = device fp32 [>= cuda_row_num, D] whose first
 * cuda_row_num rows the caller has filled with the current cache contents.  Blocks until the device is idle, then
 * every later call addresses the new rows.  Lets a caller keep extra rows right BEHIND the cache in one allocation
 * (the row-wise exchange's receive buffer: ce_exchange_local_index).  No upstream counterpart: upstream's
 * cuda_cached_weight is a torch tensor the caller owns (cache_mgr.py:83-89). */
int ce_cache_set_cache_weight(ce_cache_t* h, float* cache_weight);
/* A cache op inside the caller's hipGraph (API 3).  ce_cache_prepare_ids issued while `stream` is being captured
 * (hipStreamBeginCapture) records its kernels into that capture instead of running them: zero-copy transport only,
 * phase timers off; ids / slots_out must stay valid for every replay.  The call number that dates the eviction
 * backlist and addresses the stats ring is then counted on the device, and the caller reports the replays --
 * n_calls captured calls of ids_per_call ids each, just launched on `stream` -- right after every hipGraphLaunch:
 * this keeps the host's call count, the stats history and (LFU) the counter bound in step.  CE_ERR_UNSUPPORTED from it
 * means an LFU capture has outlived the key width it was captured with (2^31 ids): capture again.  With it the
 * whole step of a prefetch_num = 1 loop -- cache op of batch k+1 beside forward + backward of batch k -- is ONE graph
 * launch instead of ~16 kernel launches (recsys/dlrm_main.py:256-279 is the loop this replaces). */
int ce_cache_graph_replayed(ce_cache_t* h, int64_t n_calls, int64_t ids_per_call, ce_stream_t stream);
/* Phase timers of prepare_ids (upstream's per-phase Timer / record_function ranges, recsys/dlrm_main.py:258,294):
 * when on, every call brackets its phases with hipEvents on its own stream (no host sync); ce_cache_phase_times
 * blocks until the calls issued so far have finished and returns the accumulated milliseconds per phase
 * (ce_cache_phase_count() values, named by ce_cache_phase_name) and the number of calls they cover. */
int ce_cache_set_profiling(ce_cache_t* h, int32_t on);
int32_t ce_cache_phase_count(void);
const char* ce_cache_phase_name(int32_t i);
int ce_cache_phase_times(ce_cache_t* h, double* ms_out, int32_t cap, int64_t* calls, int32_t reset);
/* CE_TRANSPORT_WORKER: blocks until every eviction of the calls issued so far has reached the host table (the
 * reference's `weight` is current the moment prepare_ids returns; with the worker transport it is current after
 * this call or after ce_cache_flush).  No-op for the other transports. */
int ce_cache_writeback_wait(ce_cache_t* h);
/* Worker-side accounting of the row swap (upstream's swap_out_bandwidth / swap_in_bandwidth, printed by
 * print_comm_stats, recsys/dlrm_main.py:294): seconds6 = {out: waiting for staging, out: copying + scattering,
 * in: waiting for the miss list and earlier write-backs, in: gathering + copying, in: of which gathering +
 * enqueueing the copies, (a count, not seconds) admission jobs that ran while the previous call's write-back was
 * still on its way -- the rows that call evicted were taken from its staging buffer in HBM}, counts4 = {rows out,
 * jobs out, rows in, jobs in}. */
int ce_cache_swap_stats(ce_cache_t* h, double* seconds6, int64_t* counts4);
/* upstream buffer_size / LimitBuffIndexCopyer: rows > 0 bounds the pinned + device staging of the STAGED
 * transport to `rows` rows; larger swaps walk it in chunks.  0 (default) = stage a whole swap at once. */
int ce_cache_set_buffer_rows(ce_cache_t* h, int64_t rows);
/* number of free slots as of the last finished call (blocks like ce_cache_last_stats) */
int ce_cache_free_rows(ce_cache_t* h, int64_t* out);

/* ---------------------------------------------------------------------------------------
 * Row-wise sharding helpers (BASELINE.json north_star): the build's
 * replacement for KJTAllToAll (recsys/datasets/utils.py:20-54).  Rows are owned by
 * rank = row % world, local row = row / world.
 * ce_bucketize_rows: ids (device int64[n]) -> rows via idx_map (NULL = identity), stable
 * counting sort by owner: local_rows_out[perm position] = row / world (int64),
 * perm_out[j] = position of lookup j in the bucketed order, counts_out[w] (device int64[world]).
 */
size_t ce_bucketize_workspace(int64_t n, int32_t world);
int ce_bucketize_rows(const int64_t* ids, int64_t n, const int32_t* idx_map, int32_t world,
                      int64_t* local_rows_out, int64_t* perm_out, int64_t* counts_out,
                      void* workspace, size_t workspace_bytes, ce_stream_t stream);

/* ce_dedupe_bucket_rows: what the row-wise exchange uses instead of ce_dedupe_rows + ce_bucketize_rows: the
 * batch's unique rows, grouped by owner (row % world, any order inside a bucket), as local rows
 * (row / world) in local_rows_out[0 .. sum(counts)), pos_out[j] = position of lookup j's row in that list
 * (-1 for an id outside [0, num_rows)), counts_out[w] = unique rows owned by w (device int64[world]).
 * stamp, slot_of_row: device int32[num_rows] scratch owned by the caller (no initialisation needed, contents
 * are meaningless between calls); slot_of_row may be NULL: the stamp array then serves both purposes (4 bytes of
 * scratch per table row instead of 8 -- per batch of a window in the _window form); scratch: device
 * int32[(world + 1) * n].  world <= 64.
 * No host sync: the bucket sizes stay on the device (the caller all-to-alls them as a fixed-size message). */
int ce_dedupe_bucket_rows(const int64_t* ids, int64_t n, const int32_t* idx_map, int64_t num_rows,
                          int32_t world, int32_t* stamp, int32_t* slot_of_row, int32_t* scratch,
                          int64_t* local_rows_out, int64_t* pos_out, int64_t* counts_out, ce_stream_t stream);

/* Fixed-capacity form (API 3) for the graphed exchange: bucket w occupies local_rows_out[w * capacity, (w + 1) *
 * capacity) (device int64[world * capacity], unused places = -1) and pos_out[j] = w * capacity + place, so nothing a
 * training step touches has a data-dependent size and a window's steps -- padded, equal-split all-to-alls included --
 * replay as one hipGraph.  A bucket larger than `capacity` sets *overflow_flag (device int32, caller-zeroed) and its
 * surplus lookups get pos -1: the caller re-plans that window on the variable-size path.  Padding rows (-1) are
 * accepted by ce_cache_prepare_ids as "no lookup" (slot -1). */
int ce_dedupe_bucket_rows_padded(const int64_t* ids, int64_t n, const int32_t* idx_map, int64_t num_rows,
                                 int32_t world, int64_t capacity, int32_t* stamp, int32_t* slot_of_row,
                                 int32_t* scratch, int64_t* local_rows_out, int64_t* pos_out, int64_t* counts_out,
                                 int32_t* overflow_flag, ce_stream_t stream);

/* The same for the n_batches batches of a window in one launch per pass (3 launches instead of 3 per batch): ids =
 * device int64 [n_batches, n] contiguous, outputs [n_batches, world * capacity] / [n_batches, n] / [n_batches, world];
 * stamp and slot_of_row = device int32 [n_batches * num_rows] EACH (one array per batch: concurrent batches would race
 * on a row's entry), scratch = device int32 [n_batches * (world + 1) * n]; one overflow flag for the window. */
int ce_dedupe_bucket_rows_padded_window(const int64_t* ids, int64_t n, int64_t n_batches, const int32_t* idx_map,
                                        int64_t num_rows, int32_t world, int64_t capacity, int32_t* stamp,
                                        int32_t* slot_of_row, int32_t* scratch, int64_t* local_rows_out,
                                        int64_t* pos_out, int64_t* counts_out, int32_t* overflow_flag,
                                        ce_stream_t stream);

/* Local bypass of the row-wise exchange (API 3).  pos: device int64 [n_batches, n_per_batch], the places
 * ce_dedupe_bucket_rows_padded returned (bucket * capacity + place, -1 = none); slots: device int64, batch b at
 * slots + b * slots_batch_stride: the cache slots this rank's owner-side cache op resolved for the rows requested
 * from it, requester-major.  A rank's request to ITSELF occupies places [local_lo, local_hi) = [rank * capacity,
 * (rank + 1) * capacity) on both sides, so for those places index_out = slots[b][pos] (the row is read and updated
 * in the cache, it never passes the exchange buffers); every other place p >= 0 becomes tail_base + p, a row of the
 * receive buffer the caller keeps tail_base rows behind the cache's first row in the same allocation
 * (ce_cache_set_cache_weight), so ce_bag_forward / ce_bag_backward_sgd_presorted_src see ONE table of tail_base +
 * world * capacity rows.  At world 1 the sharded step is then the unsharded one.  No upstream counterpart (upstream
 * has no row-wise sharding of a cached table). */
int ce_exchange_local_index(const int64_t* pos, int64_t n_per_batch, int64_t n_batches, const int64_t* slots,
                            int64_t slots_batch_stride, int64_t local_lo, int64_t local_hi, int64_t tail_base,
                            int64_t* index_out, ce_stream_t stream);

/* Early / late split of the row-wise exchange (API 5).  The reference exchanges the KJT
 * synchronously before every forward (recsys/datasets/utils.py:20-54) and the pooled embeddings after it; row-wise
 * sharding (baselines/dlrm_main.py:715-716 is the only place the reference can select it) exchanges ROWS, and a row of
 * step t depends on step t-1 only if some rank looked it up in step t-1.  The owner of a shard sees every rank's
 * requests of a whole prefetch window when it plans the window, so it can tell the two kinds apart:
 *   serve: device int64 [world][n_batches][capacity] -- the local rows peer w asks for in batch b (-1 = padding), as
 *          the id all-to-all of the window delivered them;
 *   prev / n_prev: the rows ANY peer asked for in the LAST batch of the window trained before this one (-1 entries
 *          are ignored), or NULL / 0 when no window precedes (then batch 0 depends on nothing in flight);
 *   mask:  scratch, device uint64 [n_local_rows], all zero on entry and all zero again on return;
 *   flags_out: device uint8, same shape as serve: bit 0 = LATE (requested by any peer in the batch before: the row
 *          must leave after that step's update has been applied), bit 1 = URGENT (requested by any peer in the batch
 *          after: the gradient returned for it must be applied before that step's late rows leave).  The last batch
 *          of the window is all URGENT (the next window is not planned yet).  Three launches, no host wait. */
int ce_split_classify(const int64_t* serve, int32_t world, int32_t n_batches, int64_t capacity, int64_t n_local_rows,
                      const int64_t* prev, int64_t n_prev, uint64_t* mask, uint8_t* flags_out, ce_stream_t stream);

/* Places of a window's requests inside the split exchange buffers (API 5), identical on both sides of the exchange:
 * the requester calls it on its requests and the flags the owners sent back, the owner on what it serves and its own
 * flags -- both batch-major, device int64 / uint8 [n_batches][world][capacity].  caps: device int32 [n_batches][4] =
 * {cap_early, cap_late, cap_deferred, cap_urgent} of every batch (the fixed message sizes of that step, rows per peer).
 * place_fwd (device int32, same shape): an EARLY row of peer w gets w * cap_early + its rank among the chunk's early
 * rows, a LATE row world * cap_early + w * cap_late + its rank among the late ones; an early row that does not fit
 * cap_early is placed behind the chunk's late rows (a row may always be sent later).  place_bwd the same with
 * DEFERRED (= not urgent) first.  Entries of peer `skip_peer` (a rank's requests to itself never travel: -1 = none),
 * padding and rows that fit nowhere get -1; the latter also set *overflow_flag (device int32, OR-ed): the caller
 * re-plans that window on the variable-size path.  counts_out (optional): device int32 [n_batches][world][4] = rows
 * classified {early, late, deferred, urgent} per chunk, for choosing the capacities. */
int ce_split_places(const int64_t* ids, const uint8_t* flags, int32_t n_batches, int32_t world, int64_t capacity,
                    int32_t skip_peer, const int32_t* caps, int32_t* place_fwd, int32_t* place_bwd,
                    int32_t* counts_out, int32_t* overflow_flag, ce_stream_t stream);

/* ce_exchange_local_index for the split exchange (API 5): TWO indices per lookup, one into "cache + forward buffers"
 * for the pooling, one into "cache + backward buffers" for the fused fold + SGD.  Behind the cache (tail_base rows
 * from its first row, one allocation: ce_cache_set_cache_weight) lie [E0 | E1 | L] -- n_early, n_early, n_late rows:
 * the early region exists twice because a step's early rows arrive while the step before still pools from its own,
 * batch b uses copy b & 1 -- and, bwd_base rows further, [D0 | U | D1] (n_deferred, n_urgent, n_deferred rows: with U
 * between the two deferred copies a step's fold writes ONE contiguous range, which is what is zeroed before it).
 * Places in [local_lo, local_hi) (this rank's own chunk) become the cache slot in both indices. */
int ce_exchange_local_index_split(const int64_t* pos, int64_t n_per_batch, int64_t n_batches, const int64_t* slots,
                                  const int32_t* place_fwd, const int32_t* place_bwd, int64_t chunk_stride,
                                  int64_t local_lo, int64_t local_hi, int64_t tail_base, int64_t bwd_base,
                                  const int32_t* caps, int32_t world, int64_t n_early, int64_t n_urgent,
                                  int64_t n_deferred, int64_t* index_fwd, int64_t* index_bwd, ce_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * The F.embedding_bag arguments the reference forwards (recsys/models/dlrm.py:99-110 -> upstream A.7) and none of its
 * scripts sets (API 3).  Plain kernels (any dim, one wave per bag / row), off the benchmarked path.
 *
 * mode = 'max': out[bag][d] = max over the bag's rows (0 for a bag without valid rows), written like ce_bag_forward
 * (hook_features folds the shape hook).  max_pos (device int32 [num_bags, dim], may be NULL for inference) receives the
 * lookup that supplied each maximum (-1: none; the first of equal maxima, as torch's CPU kernel).  ce_bag_backward_max
 * routes the gradient there: dst[indices[max_pos[bag][d]]][d] += alpha * grad_out[bag][d] -- alpha 1 into a zeroed
 * grad_weight, alpha -lr straight into the weight (fused SGD). */
int ce_bag_forward_max(const float* weight, int64_t num_rows, int32_t dim, const int64_t* indices, int64_t nnz,
                       const void* offsets, int32_t offsets_are_i64, int64_t num_bags, int32_t include_last_offset,
                       int64_t hook_features, float* out, int32_t* max_pos, ce_stream_t stream);
int ce_bag_backward_max(float* dst, int64_t num_rows, int32_t dim, const int64_t* indices, int64_t nnz,
                        int64_t num_bags, int64_t hook_features, const float* grad_out, const int32_t* max_pos,
                        float alpha, ce_stream_t stream);
/* Gradient w.r.t. per_sample_weights (mode 'sum'): grad_psw[j] = < grad_out[bag of j], weight[indices[j]] >, 0 for an
 * ignored lookup.  grad_psw: device fp32 [nnz]. */
int ce_bag_backward_psw(const float* weight, int64_t num_rows, int32_t dim, const int64_t* indices, int64_t nnz,
                        const void* offsets, int32_t offsets_are_i64, int64_t num_bags, int32_t include_last_offset,
                        int64_t hook_features, const float* grad_out, float* grad_psw, ce_stream_t stream);
/* max_norm / norm_type (torch.embedding_renorm_, applied by F.embedding_bag before the lookup): every row an index
 * names is scaled by max_norm / (norm + 1e-7) if its norm_type-norm exceeds max_norm -- once, however often it is named.
 * workspace: device memory of ce_rows_renorm_workspace(num_rows) bytes (a bitmap of the named rows). */
size_t ce_rows_renorm_workspace(int64_t num_rows);
int ce_rows_renorm(float* weight, int64_t num_rows, int32_t dim, const int64_t* indices, int64_t n, float max_norm,
                   float norm_type, void* workspace, size_t workspace_bytes, ce_stream_t stream);

/* weight[index[i]] += alpha * src_rows[i] for i < n (whole rows of `dim` floats; repeated / out-of-range
 * index entries are summed / skipped).  Owner-side update of the row-wise exchange: the requester has
 * already folded a batch's duplicates, so each received gradient row is applied as is (alpha = -lr). */
int ce_rows_axpy(float* weight, int64_t num_rows, int32_t dim, const int64_t* index, int64_t n,
                 const float* src_rows, float alpha, ce_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CE_API_H */
