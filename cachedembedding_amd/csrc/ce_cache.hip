// Device-resident restatement of CachedParamMgr (reference call
// sites recsys/dlrm_main.py:259, benchmark/benchmark_cache.py:62) for gfx950.
//
// The reference runs prepare_ids as a chain of torch ops with a device sync after every
// phase (unique -> isin -> topk -> index_select/.cpu()/.cuda() -> index_copy_).  Here one
// prepare_ids call is a fixed sequence of kernels on one stream with no host round trip:
// every count the host would need (unique rows, misses, victims) stays in a device-side
// control block, and the per-call statistics are stored straight into a pinned host ring.
//
//   mark      ids -> rows (idx_map) -> bits in a row bitmap (N/8 bytes); the row of every id is left in
//             slots_out for the last kernel                                                 [unique, K2/K3]
//   count     unique / missing rows per 32768-row chunk, and per 64 chunks; miss = inverted[row] < 0   [K4]
//   emit      every workgroup adds up the counts before its own (that IS the scan), workgroup 0 also does the
//             plan (capacity check, k = miss - free); miss rows in ascending order; hit slots stamped with the
//             call epoch; bitmap cleared
//   keys/hist x<=8/victims      exact k-smallest selection over all slots          [K5]
//   evict     victims' rows written back to the host table, maps cleared           [K6]
//   free      first n_miss free slots ascending (ordered compaction)              [K7]
//   admit     miss rows host -> cache rows, maps + counters updated               [K8/K9]
//   slots     inverted[idx_map[id]] for every id, LFU counters += multiplicity    [K10/K11]
//
// A bitmap replaces torch.unique's sort: it yields the unique rows already in ascending
// order (the order A.4 pairs missing rows with free slots in) for N/8 bytes of streaming
// traffic instead of a multi-pass radix sort of every id.  Victim selection is a radix
// select on 64-bit keys that encode the canonical order of SURVEY.md Appendix B#1
// (LFU: (freq asc, slot asc); DATASET: cpu_row_idx desc), so evict sets are id-exact
// against oracle/cache_oracle.py.  Row payloads move either by zero-copy kernels that
// address the mapped pinned host table directly over PCIe, or (CE_TRANSPORT_STAGED) through
// pinned staging + hipMemcpyAsync with host worker threads doing the table gather/scatter.
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "ce_common.h"

#include "ce_cache_index.h"
#include "ce_cache_select.h"
#include "ce_cache_rows.h"
#include "ce_cache_fused.h"
#include "ce_cache_worker.h"


struct ce_cache {
  ce_cache_config_t cfg;
  ce::Layout L;
  char* ws;
  ce::Ctl* ctl;
  uint32_t* bitmap;
  int32_t *blk_unique, *blk_miss, *coarse, *miss_list, *slot_epoch, *victims, *blk_free, *free_list;
  int32_t* miss_list_b[2];     // [0] = miss_list, [1] = the second pair (chained admission: by call parity)
  int32_t* free_list_b[2];
  ce::ChainWords* chain;       // row counts for the admission stream
  unsigned long long *lb_emit, *lb_remap;      // look-back words (ce_cache_fused.h)
  unsigned lb_tag;             // calls that used them (every such call rewrites every word under its own tag)
  ce::FrontWords* front;       // [2]: counters of the per-lookup front (k_touch / k_miss_rank), by parity of front_calls
  int32_t *fine_cnt, *coarse_cnt;
  long long front_calls;       // calls that took the per-lookup front
  bool front_cleanup_pending;  // ... and whose k_keys has not been launched yet (it clears the front's bits and counters)
  unsigned long long* keys;
  uint32_t* hist;
  ce_call_stats_t* ring;       // pinned host
  ce_call_stats_t* ring_dev;   // device-visible alias
  hipEvent_t ev;
  float* stage;                // device staging for evicted rows (inside the workspace)
  int32_t* stage_idx;          // host row of every staged victim
  long long seq;               // calls issued
  long long drained;           // calls whose stats were folded into history
  std::vector<ce_call_stats_t> history;
  int64_t cpu_to_cuda_numel, cuda_to_cpu_numel, cache_miss, total_cache;
  int vec;                     // rows moved as 16-B vectors
  int rowlen, g_log2, slot_bits, word_bits;
  // staged transport
  int host_threads;
  float* stage_dev;            // device staging [stage_rows, D]
  float* stage_host;           // pinned host staging
  int32_t* list_host;          // pinned host copy of row/slot lists
  ce::Ctl* ctl_host;           // pinned host copy of the control block
  int64_t stage_rows;
  int64_t list_rows;           // capacity of list_host
  int64_t buffer_rows;         // > 0: staged transfers go through at most this many staging rows at a time
  ce::RowPool* pool;           // host threads of the staged transport (created on first use)
  ce::SwapEngine* wb;          // CE_TRANSPORT_WORKER state (created on first use)
  unsigned long long* evt_keys[2];   // rows staged by the write-back job of either parity (ce::EvTable; worker transport)
  int32_t* evt_pos[2];
  uint32_t evt_mask;
  float* stage2;               // second eviction staging buffer + row list, admission staging (workspace)
  int32_t* stage_idx2;
  float* in_stage;
  long long hist_base;         // seq of history[0]
  ce::PhaseProf* prof;         // optional per-phase hipEvent timers (ce_cache_set_profiling)
  uint64_t freq_bound;         // LFU: upper bound of any freq_cnter value (shortens the radix select)
  uint64_t graph_freq_limit;   // LFU: freq_bound the most recently captured call's pass count allows for
  bool freq_bound_known;       // false after a preload with caller-supplied counters until the caller states their max
  long long n_failed;          // finished prepare_ids calls whose status was not CE_OK
  int last_fail_status;
  long long last_fail_seq;
  // "the cache has no free slot left": true once a finished prepare_ids record issued after the last flush says so,
  // until the next flush or failed call.  Calls launched while it holds take the steady-state form (free slots = this
  // call's victims: no free-list scan, see free_list_from_victims); the device re-checks the premise (k_emit).
  bool free_zero;
  long long free_reset_seq;    // records up to this call number say nothing about the present
  bool deferred_rows;          // chained admission: calls do not make their stream wait for the rows (ce_cache_wait_rows)
  // a prepare_ids call issued in two halves (ce_cache_prepare_ids_begin / _finish): what the second half needs
  struct Pending {
    bool active = false;
    int64_t n = 0;
    int64_t* slots_out = nullptr;
    hipStream_t s = nullptr;
    bool worker = false, capturing = false, has_tail = false;
    long long in_job = 0, seq_arg = 0;
    int cap_groups = 1, swap_threads = 256, pslot = 0, pmark = 0;
    bool chained = false;        // chained admission: the second half is chained_second_half
    long long call = 0;          // its ticket (SwapEngine::chain_calls)
    int parity = 0;
    bool sel_pending = false;    // the selection / staging part has not been launched yet (select_and_stage)
    hipStream_t sel_s = nullptr;
    int64_t sel_n = 0;
    bool sel_steady = false;
    long long sel_out_job = 0;
    int sel_wbuf = 0, sel_n_vblocks = 0;
    const void* prof = nullptr;      // the phase timers the first half recorded into (they may be switched off / on in between)
    struct {
      int64_t n_batches, nnz_per_batch;
      int32_t src_keys;
      const void* offsets;
      int32_t offsets_are_i64;
      int64_t offsets_batch_stride, num_bags;
      int32_t include_last_offset;
      int64_t hook_features;
      uint64_t* keys_out;
    } tail;
  } pend;
};

using namespace ce;

static int ensure_writeback(ce_cache* h);

static void drain(ce_cache* h) {
  while (h->drained < h->seq) {
    const long long s = h->drained + 1;
    const ce_call_stats_t& r = h->ring[s % kRing];
    if (r.seq != s) break;
    if (h->history.empty()) h->hist_base = s;
    h->history.push_back(r);
    if (h->history.size() > (size_t)2 * kHistoryKeep) {      // bounded: keep the most recent records
      h->history.erase(h->history.begin(), h->history.end() - kHistoryKeep);
      h->hist_base = h->history.front().seq;
    }
    if (r.status != CE_OK) {
      h->n_failed += 1;
      h->last_fail_status = r.status;
      h->last_fail_seq = s;
      h->free_zero = false;            // (a lost admission gives its slots back: n_free > 0 again)
      h->free_reset_seq = h->seq;
    } else if (r.kind == CE_CALL_PREPARE && s > h->free_reset_seq) {
      h->free_zero = r.n_free_after == 0;
    }
    if (r.status == CE_OK && r.kind != CE_CALL_PRELOAD) {   // warm-up preload is not counted upstream either
      h->cpu_to_cuda_numel += r.n_miss * h->cfg.embedding_dim;
      h->cuda_to_cpu_numel += r.n_evict * h->cfg.embedding_dim;
      h->cache_miss += r.miss_lookups;
      h->total_cache += r.n_ids;
      if (h->wb && h->wb->chained && r.kind == CE_CALL_PREPARE) {      // (no worker job counts them on this path)
        std::lock_guard<std::mutex> g(h->wb->m);
        h->wb->in_rows += r.n_miss;
      }
    }
    h->drained = s;
  }
}

// chained admission: the rows of the calls issued so far are moved into their slots on the admission stream
static hipEvent_t last_rows_event(ce_cache* h) {
  if (!h->wb || !h->wb->chained || h->wb->chain_calls == 0) return nullptr;
  return h->wb->ev_rows[h->wb->chain_calls % SwapEngine::kRowsRing];
}
// ... `s` waits for them (anything that reads or rewrites cache rows / the maps outside prepare_ids)
static int join_rows(ce_cache* h, hipStream_t s) {
  if (hipEvent_t ev = last_rows_event(h)) CE_HIP_CHECK(hipStreamWaitEvent(s, ev, 0));
  return CE_OK;
}

static int sync_and_drain(ce_cache* h) {
  CE_HIP_CHECK(hipEventSynchronize(h->ev));
  // (deferred rows: the call's own stream did not wait for them)
  if (hipEvent_t ev = last_rows_event(h)) CE_HIP_CHECK(hipEventSynchronize(ev));
  drain(h);
  return CE_OK;
}

extern "C" size_t ce_cache_workspace_bytes(int64_t num_embeddings, int64_t cuda_row_num, int64_t max_ids_per_call,
                                           int32_t embedding_dim) {
  if (num_embeddings <= 0 || cuda_row_num <= 0 || embedding_dim <= 0) return 0;
  return make_layout(num_embeddings, cuda_row_num, max_ids_per_call, embedding_dim).total;
}

extern "C" int ce_cache_create(const ce_cache_config_t* cfg, ce_stream_t stream, ce_cache_t** out) {
  CE_REQUIRE(cfg && out, CE_ERR_INVALID, "null config");
  CE_REQUIRE(cfg->num_embeddings > 0 && cfg->num_embeddings < (int64_t)INT32_MAX, CE_ERR_UNSUPPORTED,
             "num_embeddings must be in (0, 2^31)");
  if (cfg->cuda_row_num == 0) {
    set_error("cuda_row_num == 0 is not implemented (NotImplementedError upstream, A.1)");
    return CE_ERR_UNSUPPORTED;
  }
  CE_REQUIRE(cfg->cuda_row_num > 0 && cfg->cuda_row_num <= cfg->num_embeddings, CE_ERR_INVALID,
             "cuda_row_num must be in (0, num_embeddings]");
  CE_REQUIRE(cfg->embedding_dim > 0, CE_ERR_INVALID, "embedding_dim must be positive");
  CE_REQUIRE(cfg->evict_strategy == CE_EVICT_DATASET || cfg->evict_strategy == CE_EVICT_LFU, CE_ERR_INVALID,
             "unknown eviction strategy");
  CE_REQUIRE(cfg->cache_weight && cfg->inverted_cached_idx && cfg->cached_idx_map && cfg->workspace,
             CE_ERR_INVALID, "null device array");
  CE_REQUIRE((((uintptr_t)cfg->inverted_cached_idx) & 15) == 0, CE_ERR_INVALID,
             "inverted_cached_idx must be 16-byte aligned");
  CE_REQUIRE(cfg->evict_strategy != CE_EVICT_LFU || cfg->freq_cnter, CE_ERR_INVALID, "LFU needs freq_cnter");
  CE_REQUIRE(cfg->host_weight && cfg->host_weight_dev, CE_ERR_INVALID, "null host table");
  Layout L = make_layout(cfg->num_embeddings, cfg->cuda_row_num, cfg->max_ids_per_call, cfg->embedding_dim);
  CE_REQUIRE(cfg->workspace_bytes >= L.total, CE_ERR_INVALID, "workspace too small: need %zu bytes", L.total);
  CE_REQUIRE((((uintptr_t)cfg->workspace) & 255) == 0, CE_ERR_INVALID, "workspace must be 256-byte aligned");

  ce_cache* h = new ce_cache();
  h->cfg = *cfg;
  h->L = L;
  h->ws = (char*)cfg->workspace;
  h->ctl = (Ctl*)(h->ws + L.ctl);
  h->bitmap = (uint32_t*)(h->ws + L.bitmap);
  h->blk_unique = (int32_t*)(h->ws + L.blk_unique);
  h->blk_miss = (int32_t*)(h->ws + L.blk_miss);
  h->coarse = (int32_t*)(h->ws + L.coarse);
  h->miss_list = (int32_t*)(h->ws + L.miss_list);
  h->slot_epoch = (int32_t*)(h->ws + L.slot_epoch);
  h->keys = (unsigned long long*)(h->ws + L.keys);
  h->hist = (uint32_t*)(h->ws + L.hist);
  h->victims = (int32_t*)(h->ws + L.victims);
  h->blk_free = (int32_t*)(h->ws + L.blk_free);
  h->free_list = (int32_t*)(h->ws + L.free_list);
  h->miss_list_b[0] = h->miss_list;
  h->miss_list_b[1] = (int32_t*)(h->ws + L.miss_list2);
  h->free_list_b[0] = h->free_list;
  h->free_list_b[1] = (int32_t*)(h->ws + L.free_list2);
  h->chain = (ChainWords*)(h->ws + L.chain);
  h->lb_emit = (unsigned long long*)(h->ws + L.lb_emit);
  h->lb_remap = (unsigned long long*)(h->ws + L.lb_remap);
  h->lb_tag = 0;
  h->front = (FrontWords*)(h->ws + L.front);
  h->fine_cnt = (int32_t*)(h->ws + L.fine_cnt);
  h->coarse_cnt = (int32_t*)(h->ws + L.coarse_cnt);
  h->front_calls = 0;
  h->front_cleanup_pending = false;
  h->seq = h->drained = 0;
  h->cpu_to_cuda_numel = h->cuda_to_cpu_numel = h->cache_miss = h->total_cache = 0;
  const int D = cfg->embedding_dim;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  h->vec = (D % 4 == 0) && al16(cfg->cache_weight) && al16(cfg->host_weight_dev);
  h->rowlen = h->vec ? D / 4 : D;
  int g = 1, gl2 = 0;
  while (g < h->rowlen && g < 64) { g <<= 1; ++gl2; }
  h->g_log2 = gl2;
  int sb = 1;
  while ((1ll << sb) < cfg->cuda_row_num) ++sb;
  h->slot_bits = sb;
  int wb = 1;
  while ((1ll << wb) < cdiv(cfg->num_embeddings, 32)) ++wb;
  h->word_bits = wb;
  h->host_threads = (int)std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
  h->stage_dev = nullptr;
  h->stage_host = nullptr;
  h->list_host = nullptr;
  h->ctl_host = nullptr;
  h->stage_rows = 0;
  h->list_rows = 0;
  h->buffer_rows = 0;
  h->pool = nullptr;
  h->wb = nullptr;
  h->hist_base = 1;
  h->prof = nullptr;
  h->freq_bound = 0;
  h->graph_freq_limit = ~0ull;
  h->freq_bound_known = true;
  h->n_failed = 0;
  h->last_fail_status = CE_OK;
  h->last_fail_seq = 0;
  h->free_zero = false;
  h->free_reset_seq = 0;
  h->deferred_rows = false;
  h->stage2 = (float*)(h->ws + L.stage2);
  h->stage_idx2 = (int32_t*)(h->ws + L.stage_idx2);
  h->in_stage = (float*)(h->ws + L.in_stage);

  hipStream_t s = (hipStream_t)stream;
  void* ring_host = nullptr;
  if (hipHostMalloc(&ring_host, sizeof(ce_call_stats_t) * kRing, hipHostMallocMapped) != hipSuccess) {
    delete h;
    set_error("hipHostMalloc(stats ring) failed");
    return CE_ERR_HIP;
  }
  memset(ring_host, 0, sizeof(ce_call_stats_t) * kRing);
  h->ring = (ce_call_stats_t*)ring_host;
  void* ring_dev = nullptr;
  if (hipHostGetDevicePointer(&ring_dev, ring_host, 0) != hipSuccess) ring_dev = ring_host;
  h->ring_dev = (ce_call_stats_t*)ring_dev;
  h->stage = (float*)(h->ws + L.stage);
  h->stage_idx = (int32_t*)(h->ws + L.stage_idx);
  if (hipEventCreateWithFlags(&h->ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipHostFree(ring_host);
    delete h;
    set_error("hipEventCreate failed");
    return CE_ERR_HIP;
  }
  // empty-cache state of A.1
  const int64_t N = cfg->num_embeddings, C = cfg->cuda_row_num;
  (void)hipMemsetAsync(h->ws, 0, L.stage_idx, s);
  hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(N, 256)), dim3(256), 0, s, cfg->inverted_cached_idx, N, -1);
  hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(C, 256)), dim3(256), 0, s, cfg->cached_idx_map, C, -1);
  hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(C, 256)), dim3(256), 0, s, h->slot_epoch, C, kEpochNever);
  if (cfg->freq_cnter)
    hipLaunchKernelGGL(k_fill_i64, dim3(grid_for(C, 256)), dim3(256), 0, s, cfg->freq_cnter, C, (int64_t)INT64_MAX);
  Ctl init{};
  init.n_free = C;
  (void)hipMemcpyAsync(h->ctl, &init, sizeof(Ctl), hipMemcpyHostToDevice, s);
  hipError_t e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("cache state initialisation failed: %s", hipGetErrorString(e));
    (void)hipEventDestroy(h->ev);
    (void)hipHostFree(ring_host);
    delete h;
    return CE_ERR_HIP;
  }
  (void)hipEventRecord(h->ev, s);
  if (cfg->transport == CE_TRANSPORT_WORKER) {
    int rc = ensure_writeback(h);
    if (rc) {
      (void)hipEventDestroy(h->ev);
      (void)hipHostFree(ring_host);
      delete h;
      return rc;
    }
  }
  *out = h;
  return CE_OK;
}

extern "C" int ce_cache_destroy(ce_cache_t* h) {
  if (!h) return CE_OK;
  (void)hipEventSynchronize(h->ev);
  if (h->wb && h->wb->in_stream) (void)hipStreamSynchronize(h->wb->in_stream);
  delete h->wb;          // finishes the queued jobs, joins the workers
  for (int b = 0; b < 2; ++b) {
    if (h->evt_keys[b]) (void)hipFree(h->evt_keys[b]);
    if (h->evt_pos[b]) (void)hipFree(h->evt_pos[b]);
  }
  delete h->pool;
  delete h->prof;
  (void)hipEventDestroy(h->ev);
  (void)hipHostFree(h->ring);
  if (h->stage_dev) (void)hipFree(h->stage_dev);
  if (h->stage_host) (void)hipHostFree(h->stage_host);
  if (h->list_host) (void)hipHostFree(h->list_host);
  if (h->ctl_host) (void)hipHostFree(h->ctl_host);
  delete h;
  return CE_OK;
}

static int before_call(ce_cache* h) {
  // never let the device overwrite a ring slot the host has not folded into the history yet
  if (h->seq - h->drained >= kRing - 2) return sync_and_drain(h);
  drain(h);
  return CE_OK;
}

static int ensure_staging(ce_cache* h, int64_t rows, int64_t list_rows) {
  if (rows > h->stage_rows) {
    if (h->stage_dev) (void)hipFree(h->stage_dev);
    if (h->stage_host) (void)hipHostFree(h->stage_host);
    h->stage_dev = nullptr; h->stage_host = nullptr;
    h->stage_rows = 0;
    const size_t bytes = (size_t)rows * h->cfg.embedding_dim * sizeof(float);
    CE_HIP_CHECK(hipMalloc((void**)&h->stage_dev, bytes));
    CE_HIP_CHECK(hipHostMalloc((void**)&h->stage_host, bytes, hipHostMallocDefault));
    h->stage_rows = rows;
  }
  if (list_rows > h->list_rows) {
    if (h->list_host) (void)hipHostFree(h->list_host);
    h->list_host = nullptr;
    h->list_rows = 0;
    CE_HIP_CHECK(hipHostMalloc((void**)&h->list_host, (size_t)list_rows * sizeof(int32_t), hipHostMallocDefault));
    h->list_rows = list_rows;
  }
  return CE_OK;
}

// CPUs this process may actually burn: hardware threads, capped by the affinity mask and by the cgroup CPU quota
// (the bench boxes show 256 hardware threads behind a 16-CPU quota: more helper threads than that only fight)
static int cpu_budget() {
  long n = (long)std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min<long>(n, std::max(1, CPU_COUNT(&set)));
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                 // cgroup v2: "<quota|max> <period>"
    char q[32] = {0};
    long long per = 0;
    if (fscanf(f, "%31s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0)
      n = std::min<long>(n, std::max<long long>(1, atoll(q) / per));
    fclose(f);
  } else {
    long long quota = -1, per = 0;                                       // cgroup v1
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &per) != 1) per = 0; fclose(g); }
    if (quota > 0 && per > 0) n = std::min<long>(n, std::max<long long>(1, quota / per));
  }
  return (int)std::max<long>(1, n);
}

extern "C" int32_t ce_cpu_budget(void) { return cpu_budget(); }

static RowPool* host_pool(ce_cache* h) {
  if (!h->pool) h->pool = new RowPool(std::max(1, std::min(h->host_threads, 32)));
  return h->pool;
}

// CE_TRANSPORT_WORKER state: pinned landing / gather buffers, mailboxes, the two copy streams, the two workers
static int ensure_writeback(ce_cache* h) {
  if (h->wb) return CE_OK;
  const Layout& L = h->L;
  SwapEngine* w = new SwapEngine();
  int rc = CE_OK;
  do {
    int can = 0;
    if (hipGetDevice(&w->device) != hipSuccess) { rc = CE_ERR_HIP; break; }
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, w->device);
    if (!can) {
      delete w;
      set_error("the worker transport needs hipStreamWaitValue64, which this device does not support");
      return CE_ERR_UNSUPPORTED;
    }
    w->D = h->cfg.embedding_dim;
    w->stage_rows = L.stage_rows;
    w->table = h->cfg.host_weight;
    w->stage_dev[0] = h->stage;
    w->stage_dev[1] = h->stage2;
    w->idx_dev[0] = h->stage_idx;
    w->idx_dev[1] = h->stage_idx2;
    w->in_stage_dev = h->in_stage;
    {
      // admission: chained on the admission stream (a small kernel reads the missed rows out of the mapped table) unless
      // CE_WORKER_ADMIT=sdma asks for the host gather + staged copy, or the table has no device-visible mapping
      const char* e = getenv("CE_WORKER_ADMIT");
      w->chained = !(e && !strcmp(e, "sdma")) && h->cfg.host_weight_dev != nullptr;
    }
    {
      if (w->chained && !h->evt_keys[0]) {
        uint32_t places = 4096;
        while ((int64_t)places < 4 * L.stage_rows && places < (1u << 30)) places <<= 1;
        bool ok = (int64_t)places >= 4 * L.stage_rows;
        for (int b = 0; b < 2 && ok; ++b) {
          ok = hipMalloc((void**)&h->evt_keys[b], (size_t)places * 8) == hipSuccess &&
               hipMalloc((void**)&h->evt_pos[b], (size_t)places * 4) == hipSuccess &&
               hipMemset(h->evt_keys[b], 0, (size_t)places * 8) == hipSuccess;
        }
        if (ok) {
          h->evt_mask = places - 1;
        } else {
          (void)hipGetLastError();
          for (int b = 0; b < 2; ++b) {
            if (h->evt_keys[b]) (void)hipFree(h->evt_keys[b]);
            if (h->evt_pos[b]) (void)hipFree(h->evt_pos[b]);
            h->evt_keys[b] = nullptr;
            h->evt_pos[b] = nullptr;
          }
          rc = CE_ERR_NOMEM;
          break;
        }
      }
#ifdef CE_TEST_HOOKS
      if (const char* e = getenv("CE_WORKER_OUT_DELAY_US")) w->out_delay_us = std::max(0, atoi(e));
      if (const char* e = getenv("CE_WORKER_FAIL_IN_JOB")) w->fail_in_job = atoll(e);
#endif
      for (int b = 0; b < 2; ++b) {
        w->evt_keys[b] = h->evt_keys[b];
        w->evt_pos[b] = h->evt_pos[b];
      }
      w->evt_mask = h->evt_mask;
    }
    const size_t rows_bytes = (size_t)L.stage_rows * w->D * 4, idx_bytes = (size_t)L.stage_rows * 4;
    // The copy streams must never share a hardware queue with the parked cache-op stream: a hipMemcpyAsync is not
    // queue-free (the runtime brackets the SDMA copy with barrier packets in the stream's queue), so a copy queued
    // behind the parked hipStreamWaitValue64 never starts and the wait is never released -- seen as a hang once
    // enough streams were alive for HIP to double them up on its (by default 4) hardware queues.  HIP keeps a
    // separate pool of hardware queues per stream priority, so the copy streams are created at the HIGHEST
    // priority: they only ever share queues with each other (and nothing of theirs ever waits).
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (prio_lo == prio_hi) {
      delete w;
      set_error("the worker transport needs stream priorities (a hardware-queue pool of its own for the copy streams)");
      return CE_ERR_UNSUPPORTED;
    }
    if (hipStreamCreateWithPriority(&w->out_stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&w->out_stream2, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&w->in_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) {
      rc = CE_ERR_HIP;
      break;
    }
    void* p = nullptr;
    void* pd = nullptr;
    if (hipHostMalloc(&p, sizeof(WbMail) * 3, hipHostMallocMapped) != hipSuccess) { rc = CE_ERR_NOMEM; break; }
    memset(p, 0, sizeof(WbMail) * 3);
    w->mail = (WbMail*)p;
    if (hipHostGetDevicePointer(&pd, p, 0) != hipSuccess) pd = p;
    w->mail_dev = (WbMail*)pd;
    if (hipHostMalloc(&p, 64, hipHostMallocMapped) != hipSuccess) { rc = CE_ERR_NOMEM; break; }
    memset(p, 0, 64);
    w->sig = (unsigned long long*)p;
    if (hipHostGetDevicePointer(&pd, p, 0) != hipSuccess) pd = p;
    w->sig_dev = (unsigned long long*)pd;
    if (hipHostMalloc(&p, idx_bytes, hipHostMallocMapped) != hipSuccess) { rc = CE_ERR_NOMEM; break; }
    w->miss_host = (int32_t*)p;
    if (hipHostGetDevicePointer(&pd, p, 0) != hipSuccess) pd = p;
    w->miss_host_dev = (int32_t*)pd;
    if (!w->chained &&
        hipHostMalloc((void**)&w->in_host, rows_bytes, hipHostMallocDefault) != hipSuccess) { rc = CE_ERR_NOMEM; break; }
    if (w->chained) {
      for (auto& ev : w->ev_miss)
        if (rc == CE_OK && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) rc = CE_ERR_HIP;
      for (auto& ev : w->ev_rows)
        if (rc == CE_OK && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) rc = CE_ERR_HIP;
      if (rc) break;
    }
    // the workers sleep in hipEventSynchronize (blocking-sync events): spinning threads would eat the CPU quota
    // the helpers need
    const unsigned evf = hipEventDisableTiming | hipEventBlockingSync;
    if (hipEventCreateWithFlags(&w->in_ev[0], evf) != hipSuccess ||
        hipEventCreateWithFlags(&w->in_ev[1], evf) != hipSuccess) { rc = CE_ERR_HIP; break; }
    for (int b = 0; b < 2 && rc == CE_OK; ++b) {
      if (hipEventCreateWithFlags(&w->out_ev[b], evf) != hipSuccess) rc = CE_ERR_HIP;
      else if (hipHostMalloc((void**)&w->rows_host[b], rows_bytes, hipHostMallocDefault) != hipSuccess) rc = CE_ERR_NOMEM;
      else if (hipHostMalloc((void**)&w->idx_host[b], idx_bytes, hipHostMallocDefault) != hipSuccess) rc = CE_ERR_NOMEM;
    }
    for (auto& ev : w->chunk_ev)
      if (rc == CE_OK && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) rc = CE_ERR_HIP;
    if (rc) break;
    // helper threads per direction: half of what the CPU budget leaves after the launch thread, the two (mostly
    // sleeping) workers and the runtime's own threads -- a cgroup that exceeds its CPU quota is frozen until the next
    // scheduler period, which shows up as millisecond stalls of the admission.  A helper is bound by the cache misses it can keep in flight (~45-75 ns per
    // 512-byte row), so the job time falls with the helper count until the quota is reached; beyond it the threads
    // only fight (16-CPU quota, spinning waits: 4 + 4 helpers 0.78 / 1.03 ms per job, 8 + 8 1.24 / 1.43 ms)
    static const int dflt = std::max(2, std::min(8, (cpu_budget() - 4) / 2));
    static const int out_threads = [] { const char* e = getenv("CE_WB_THREADS"); return e ? atoi(e) : dflt; }();
    static const int in_threads = [] { const char* e = getenv("CE_GATHER_THREADS"); return e ? atoi(e) : dflt; }();
    w->out_pool = new RowPool(std::max(1, std::min(out_threads, 64)));
    w->in_pool = new RowPool(std::max(1, std::min(in_threads, 64)));
    w->rowlen = h->rowlen;
    w->g_log2 = h->g_log2;
    w->vec = h->vec;
    w->out_thread = std::thread([w] { w->run_out(); });
    if (!w->chained) w->in_thread = std::thread([w] { w->run_in(); });
  } while (0);
  if (rc) {
    delete w;
    set_error("swap worker setup failed (pinned buffers / streams / events)");
    return rc;
  }
  w->deferred_rows = h->deferred_rows;
  h->wb = w;
  return CE_OK;
}

// rows moved per staged transfer: everything at once, or `buffer_rows` at a time (upstream's buffer_size /
// LimitBuffIndexCopyer: a bounded staging buffer walked in chunks)
static int64_t staged_chunk(const ce_cache* h, int64_t rows) {
  return (h->buffer_rows > 0 && h->buffer_rows < rows) ? h->buffer_rows : rows;
}

extern "C" int ce_cache_set_freq_bound(ce_cache_t* h, int64_t bound) {
  CE_REQUIRE(h && bound >= 0, CE_ERR_INVALID, "bad bound");
  h->freq_bound = std::max<uint64_t>(h->freq_bound, (uint64_t)bound);
  h->freq_bound_known = true;
  return CE_OK;
}

extern "C" int ce_cache_preload(ce_cache_t* h, const int32_t* rows, const int64_t* freq_vals, int64_t n,
                                ce_stream_t stream) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  CE_REQUIRE(n >= 0 && n <= h->cfg.cuda_row_num, CE_ERR_INVALID, "preload count out of range");
  CE_REQUIRE(!h->pend.active, CE_ERR_INVALID, "a cache op begun with ce_cache_prepare_ids_begin has not been finished");
  if (n == 0) return CE_OK;
  if (freq_vals) h->freq_bound_known = false;      // until ce_cache_set_freq_bound states their maximum
  int rc = before_call(h);
  if (rc) return rc;
  if (h->wb) {
    // rows are read zero-copy from the host table: every eviction queued on the worker transport lands first
    rc = h->wb->wait_out(h->wb->out_issued);
    if (rc) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  const ce_cache_config_t& c = h->cfg;
  rc = join_rows(h, s);
  if (rc) return rc;
  h->seq += 1;
  const int32_t epoch = (int32_t)(h->seq & 0x3fffffff);
  const int64_t groups_per_block = 256 >> h->g_log2;
  if (h->vec)
    hipLaunchKernelGGL((k_admit<f32x4>), dim3(grid_for(n, (int)groups_per_block * kSwapRows)), dim3(256), 0, s, rows,
                       (const int32_t*)nullptr, (const long long*)nullptr, (long long)n,
                       (const f32x4*)c.host_weight_dev, (f32x4*)c.cache_weight, h->rowlen, h->g_log2,
                       (const Ctl*)nullptr, 0ll);
  else
    hipLaunchKernelGGL((k_admit<float>), dim3(grid_for(n, (int)groups_per_block * kSwapRows)), dim3(256), 0, s, rows,
                       (const int32_t*)nullptr, (const long long*)nullptr, (long long)n,
                       (const float*)c.host_weight_dev, (float*)c.cache_weight, h->rowlen, h->g_log2,
                       (const Ctl*)nullptr, 0ll);
  // preloaded rows must not look "protected" to the first prepare_ids: stamp them as never used
  hipLaunchKernelGGL(k_admit_maps, dim3(grid_for(n, 256)), dim3(256), 0, s, rows, (const int32_t*)nullptr,
                     (const long long*)nullptr, (long long)n, c.cached_idx_map, c.inverted_cached_idx,
                     c.freq_cnter, freq_vals, h->slot_epoch, kEpochNever, (Ctl*)nullptr,
                     (ce_call_stats_t*)nullptr, 0ll, (const unsigned long long*)nullptr, 0ll);
  (void)epoch;
  hipLaunchKernelGGL(k_preload_end, dim3(1), dim3(1), 0, s, (long long)n, h->ctl, h->ring_dev + (h->seq % kRing),
                     h->seq);
  CE_LAUNCH_CHECK();
  CE_HIP_CHECK(hipEventRecord(h->ev, s));
  return CE_OK;
}

// staged transport: rows move through pinned staging with host worker threads touching the table
static int staged_swap(ce_cache* h, hipStream_t s) {
  const ce_cache_config_t& c = h->cfg;
  const int D = c.embedding_dim;
  const size_t rowbytes = (size_t)D * sizeof(float);
  // the host needs the counts and the lists: one sync point per call (the reference has one per phase)
  if (!h->ctl_host) CE_HIP_CHECK(hipHostMalloc((void**)&h->ctl_host, sizeof(Ctl), hipHostMallocDefault));
  CE_HIP_CHECK(hipMemcpyAsync(h->ctl_host, h->ctl, sizeof(Ctl), hipMemcpyDeviceToHost, s));
  CE_HIP_CHECK(hipStreamSynchronize(s));
  const Ctl ctl = *h->ctl_host;
  if (ctl.status != CE_OK) return CE_OK;
  const int64_t k = ctl.k_evict, m = ctl.n_miss;
  const int64_t big = std::max<int64_t>(std::max(k, m), 1);
  int rc = ensure_staging(h, staged_chunk(h, big), big);
  if (rc) return rc;
  const int gpb = 256 >> h->g_log2;
  if (k > 0) {
    // D2H: evicted rows packed on the device, copied, scattered into the table by worker threads
    int32_t* evicted_rows_dev = h->free_list;   // free_list is not live yet: reuse as the row list
    hipLaunchKernelGGL(k_evict_maps, dim3(grid_for(k, 256)), dim3(256), 0, s, h->victims, c.cached_idx_map,
                       c.inverted_cached_idx, evicted_rows_dev, h->ctl);
    CE_HIP_CHECK(hipMemcpyAsync(h->list_host, evicted_rows_dev, (size_t)k * 4, hipMemcpyDeviceToHost, s));
    const int64_t chunk = staged_chunk(h, k);
    for (int64_t off = 0; off < k; off += chunk) {
      const int64_t cnt = std::min(chunk, k - off);
      if (h->vec)
        hipLaunchKernelGGL((k_pack_rows<f32x4>), dim3(grid_for(cnt, gpb)), dim3(256), 0, s,
                           (const int32_t*)h->victims + off, (long long)cnt, (const f32x4*)c.cache_weight,
                           (f32x4*)h->stage_dev, h->rowlen, h->g_log2);
      else
        hipLaunchKernelGGL((k_pack_rows<float>), dim3(grid_for(cnt, gpb)), dim3(256), 0, s,
                           (const int32_t*)h->victims + off, (long long)cnt, (const float*)c.cache_weight,
                           (float*)h->stage_dev, h->rowlen, h->g_log2);
      CE_HIP_CHECK(hipMemcpyAsync(h->stage_host, h->stage_dev, (size_t)cnt * rowbytes, hipMemcpyDeviceToHost, s));
      CE_HIP_CHECK(hipStreamSynchronize(s));
      float* table = c.host_weight;
      const float* st = h->stage_host;
      const int32_t* rows = h->list_host + off;
      host_pool(h)->parallel(cnt, [=](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) memcpy(table + (size_t)rows[i] * D, st + (size_t)i * D, rowbytes);
      });
    }
  }
  return CE_OK;
}

// The worker transport parks the cache-op stream in hipStreamWaitValue64 and relies on its copy streams making
// progress meanwhile, i.e. on their not sharing a hardware queue with the parked stream (HIP multiplexes streams onto
// GPU_MAX_HW_QUEUES queues; the copy streams sit in the highest-priority pool for that reason -- ensure_writeback).
// That is a property of the runtime and of the process's environment, not of this library, so it is TESTED the
// first time a stream is used: park the stream on a private word, run one small copy on each copy stream, release.
// Copies that do not finish while the stream is parked mean the transport would hang here: CE_ERR_UNSUPPORTED, and
// prepare_ids falls back to the zero-copy kernels with a message.  Costs one stream synchronisation, once.
static int worker_selftest(ce_cache* h, hipStream_t s) {
  SwapEngine* w = h->wb;
  if (w->tested && w->tested_stream == s) return CE_OK;
  CE_HIP_CHECK(hipStreamSynchronize(s));
  unsigned long long* word = w->sig + 2;
  __atomic_store_n(word, 0ull, __ATOMIC_RELEASE);
  CE_HIP_CHECK(hipStreamWaitValue64(s, word, 1, hipStreamWaitValueGte, ~0ull));
  std::this_thread::sleep_for(std::chrono::milliseconds(2));          // let the wait reach the queue
  hipError_t e = hipMemcpyAsync(w->rows_host[0], w->stage_dev[0], 64, hipMemcpyDeviceToHost, w->out_stream);
  if (e == hipSuccess) e = hipMemcpyAsync(w->rows_host[1], w->stage_dev[1], 64, hipMemcpyDeviceToHost, w->out_stream2);
  if (e == hipSuccess) e = hipMemcpyAsync(w->in_stage_dev, w->rows_host[0], 64, hipMemcpyHostToDevice, w->in_stream);
  bool done = e == hipSuccess;
  if (done) {
    for (hipStream_t st : {w->out_stream, w->out_stream2, w->in_stream})
      if (stream_wait_polite(st, 0.25) != hipSuccess) done = false;
  }
  __atomic_store_n(word, 1ull, __ATOMIC_RELEASE);                      // always release
  CE_HIP_CHECK(hipStreamSynchronize(s));
  if (e != hipSuccess) {
    set_error("worker transport self-test: %s", hipGetErrorString(e));
    return CE_ERR_HIP;
  }
  if (!done) {
    set_error("worker transport self-test failed: copies on the library's copy streams do not run while the cache-op "
              "stream is parked in hipStreamWaitValue64 (they share a hardware queue with it: GPU_MAX_HW_QUEUES = %s)",
              getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "unset");
    return CE_ERR_UNSUPPORTED;
  }
  w->tested = true;
  w->tested_stream = s;
  return CE_OK;
}

// k_keys' arguments for what the per-lookup front left behind (all NULL behind the bitmap front)
struct FrontTail {
  const int32_t* miss_tmp;
  const FrontWords* fw;
  uint32_t* bitmap;
  int32_t *fine, *coarse;
};
static FrontTail take_front_tail(ce_cache* h) {
  if (!h->front_cleanup_pending) return FrontTail{nullptr, nullptr, nullptr, nullptr, nullptr};
  h->front_cleanup_pending = false;
  return FrontTail{h->victims, h->front + (h->front_calls & 1), h->bitmap, h->fine_cnt, h->coarse_cnt};
}

// Second part of a cache op's front: victim selection, staging of the victims (and the write-back job), free-slot list.
// Runs inline behind the front, or -- a call in two halves on the worker transport -- at the start of the SECOND half:
// the admission kernel behind the front reads the host table over PCIe for ~0.7 ms, and these kernels run 2-3x slower
// beside it (find_evict_ids 0.042 -> 0.072 ms, evict_stage 0.030 -> 0.065 ms in the bench's phase timers); behind the
// window's training steps the admission has long finished and they run alone, while the admission itself starts as
// early as before.
struct SelArgs {
  hipStream_t s;
  int64_t n;
  bool worker, capturing, steady;
  long long out_job, seq_arg;
  int wbuf, n_vblocks, pslot;
  const void* prof_id;
  int32_t* free_list;
};

// radix passes the victim select needs (the top one is taken by the key kernel itself)
static int select_top_pass(ce_cache* h, int64_t n, bool capturing) {
  const ce_cache_config_t& c = h->cfg;
  const int64_t N = c.num_embeddings;
  // DATASET keys are < N: the digits above the highest digit of N-1 are zero for every eligible slot, so the
  // radix select starts there (3 passes of 11 bits at N = 178 M).
  int top_pass = kLevels - 1;
  if (c.evict_strategy != CE_EVICT_LFU) {
    top_pass = 0;
    while (top_pass < kLevels - 1 && ((uint64_t)(N - 1) >> (kDigitBits * (top_pass + 1))) != 0) ++top_pass;
  } else {
    // LFU keys are freq << slot_bits | slot.  No counter can exceed the largest value ever preloaded plus the ids
    // seen so far (a call adds at most its own length to a counter), so the digits above that bound are zero in
    // every eligible key.  (The key kernel clamps a counter that a caller pushed beyond it -- freq_cnter is the
    // caller's tensor -- so a key never has bits above the top digit.)
    // (a captured call is replayed an unknown number of times: its pass count allows for kGraphFreqHeadroom more
    // ids; ce_cache_graph_replayed keeps the bound and asks for a new capture once it is used up)
    uint64_t bound = h->freq_bound + (uint64_t)n;
    if (capturing) {
      bound = h->freq_bound + kGraphFreqHeadroom;
      h->graph_freq_limit = bound;
    } else {
      h->freq_bound = bound;
    }
    const int bits = 64 - __builtin_clzll(bound | 1ull) + h->slot_bits;
    top_pass = h->freq_bound_known ? std::min(kLevels - 1, std::max(0, (bits + kDigitBits - 1) / kDigitBits - 1))
                                   : kLevels - 1;
  }
  return top_pass;
}

static int select_and_stage(ce_cache* h, const SelArgs& a, int* pmark_io) {
  const ce_cache_config_t& c = h->cfg;
  const Layout& L = h->L;
  const int64_t N = c.num_embeddings, C = c.cuda_row_num, n = a.n;
  hipStream_t s = a.s;
  const bool worker = a.worker, capturing = a.capturing, steady = a.steady;
  const long long out_job = a.out_job, seq_arg = a.seq_arg;
  const int wbuf = a.wbuf, n_vblocks = a.n_vblocks, pslot = a.pslot;
  const int lfu = c.evict_strategy == CE_EVICT_LFU;
  const int gpb = 256 >> h->g_log2;
  ce_call_stats_t* const ring = h->ring_dev;
  PhaseProf* const prof = (h->prof && (const void*)h->prof == a.prof_id) ? h->prof : nullptr;
  int pmark = *pmark_io;
  int rc = CE_OK;
#define CE_PHASE() do { if (prof) (void)hipEventRecord(prof->ev[pslot][pmark++], s); } while (0)
  // ---- victim selection (all kernels return at once when k == 0)
  const int cgrid = grid_for(C, 256 * 4);
  const int top_pass = select_top_pass(h, n, capturing);
  const FrontTail ft = take_front_tail(h);
  hipLaunchKernelGGL(k_keys, dim3(std::min(cgrid, 512)), dim3(256), 0, s, c.cached_idx_map, c.freq_cnter, h->slot_epoch, C, N,
                     seq_arg, c.protect_depth, h->slot_bits, lfu, top_pass, h->keys, h->hist, h->ctl, ft.miss_tmp, ft.fw,
                     ft.bitmap, ft.fine, ft.coarse, (int)L.n_chunks);
  const int hgrid = (int)std::min<int64_t>(kNumCU, std::max<int64_t>(1, cdiv(C, 1024 * 4)));
  // (all passes in ONE workgroup for small caches was tried for the B = 2048 shapes: a single CU keeps too few key
  // loads in flight -- 0.38 ms per call against 0.05 ms for the 5 launch pairs.  What works per launch: every
  // workgroup of pass p recomputes the digits of the level above it from that level's histogram in its prologue --
  // select_level -- so there is no pick kernel at all)
  for (int pass = top_pass - 1; pass >= 0; --pass)
    hipLaunchKernelGGL(k_hist, dim3(hgrid), dim3(1024), 0, s, h->keys, C, pass, top_pass, h->hist, h->ctl);
  hipLaunchKernelGGL(k_victims, dim3((unsigned)n_vblocks), dim3(256), 0, s, h->keys, C, h->victims, L.list_cap, h->ctl,
                     (const uint32_t*)h->hist, top_pass, ring, seq_arg, steady ? h->blk_free : (int32_t*)nullptr);
  CE_PHASE();
  float* const stage_cur = (worker && wbuf) ? h->stage2 : h->stage;
  int32_t* const stage_idx_cur = (worker && wbuf) ? h->stage_idx2 : h->stage_idx;
  if (c.transport == CE_TRANSPORT_ZEROCOPY || worker) {
    // ---- victims -> HBM staging (fast); rows beyond the staging capacity (rare) are written back directly;
    // map clear, free-slot list, admit stay on the caller's stream
    const long long scap = (long long)L.stage_rows;
    const int sgrid = (int)std::min<int64_t>(512, std::max<int64_t>(1, cdiv(L.stage_rows, gpb)));
    WbMail* const mail = worker ? h->wb->mail_dev + wbuf : nullptr;
    const dim3 sg(sgrid + (steady ? n_vblocks : 0));        // + the free-list workgroups of the steady-state form
    const EvTable evt = (worker && h->wb->chained) ? EvTable{h->evt_keys[wbuf], h->evt_pos[wbuf], h->evt_mask}
                                                     : EvTable{nullptr, nullptr, 0u};
    if (h->vec) {
      hipLaunchKernelGGL((k_evict_stage<f32x4>), sg, dim3(256), 0, s, h->victims, c.cached_idx_map,
                         c.inverted_cached_idx, (const f32x4*)c.cache_weight, (f32x4*)stage_cur, stage_idx_cur, scap, h->rowlen, h->g_log2,
                         h->ctl, mail, out_job, sgrid, (const unsigned long long*)h->keys, C, (const int32_t*)h->blk_free,
                         a.free_list, evt, L.list_cap > L.stage_rows ? (f32x4*)c.host_weight_dev : (f32x4*)nullptr);
    } else {
      hipLaunchKernelGGL((k_evict_stage<float>), sg, dim3(256), 0, s, h->victims, c.cached_idx_map,
                         c.inverted_cached_idx, (const float*)c.cache_weight, (float*)stage_cur, stage_idx_cur, scap, h->rowlen, h->g_log2,
                         h->ctl, mail, out_job, sgrid, (const unsigned long long*)h->keys, C, (const int32_t*)h->blk_free,
                         a.free_list, evt, L.list_cap > L.stage_rows ? (float*)c.host_weight_dev : (float*)nullptr);
    }
    if (worker) {
      // the write-back worker takes it from here: D2H of the packed block + scatter into the table
      CE_HIP_CHECK(hipEventRecord(h->wb->out_ev[wbuf], s));
      h->wb->push_out();
    }
  } else {
    rc = staged_swap(h, s);
    if (rc) return rc;
  }
  CE_PHASE();
  // (one workgroup walks 4096 slots per round: beyond a few rounds the pair of wide kernels is faster --
  // C = 94 k, Avazu at 1 %: 18.3 us against 13.6 us for the pair)
  if (steady) {
    // (the list was written by k_evict_stage's free-list workgroups)
  } else if (C <= 16384) {
    hipLaunchKernelGGL(k_free_single, dim3(1), dim3(1024), 0, s, c.cached_idx_map, C, a.free_list, h->ctl);
  } else {
    hipLaunchKernelGGL(k_free_count, dim3((unsigned)L.n_slot_blocks), dim3(256), 0, s, c.cached_idx_map, C,
                       h->blk_free, h->ctl);
    hipLaunchKernelGGL(k_free_emit, dim3((unsigned)L.n_slot_blocks), dim3(256), 0, s, c.cached_idx_map, C,
                       h->blk_free, a.free_list, h->ctl);
  }
  CE_PHASE();
#undef CE_PHASE
  *pmark_io = pmark;
  (void)rc;
  return CE_OK;
}

// the window's keys, written by the call's last kernel (ce_cache_prepare_ids_keys)
struct KeysTail {
  int64_t n_batches, nnz_per_batch;
  int32_t src_keys;
  const void* offsets;
  int32_t offsets_are_i64;
  int64_t offsets_batch_stride, num_bags;
  int32_t include_last_offset;
  int64_t hook_features;
  uint64_t* keys_out;
};

// how the ids are marked: rows in frequency order (idx_map present) -- the hot rows sit in the lowest bitmap words, an
// LDS window absorbs them, cold lookups issue their atomicOr directly; rows in id order -- hot rows are scattered,
// every wave would hammer their words (421 us per 3.4 M ids), so equal words of a wave are merged first.  (rocprofv3,
// 3.4 M ids per call; the grid / unroll sweeps of rounds 3-5 ended at these values: docs/history.md)
struct MarkCfg {
  bool merge;
  int u, threads, blocks, hot_words;
};
static MarkCfg mark_cfg(const ce_cache* h, int64_t n) {
  const bool ranked = h->cfg.idx_map != nullptr;
  MarkCfg m;
  m.merge = !ranked;
  m.threads = ranked ? 512 : 256;
  m.hot_words = (int)std::min<int64_t>(h->L.bitmap_words, ranked ? 8192 : 2048);
  m.u = n >= 65536 ? 4 : 1;
  m.blocks = std::min(grid_for(n, m.threads * m.u), 256);
  return m;
}

// calls k_rank_victims takes (ce_cache_fused.h): its look-back words hold 23-bit counts
constexpr int64_t kScanMaxIds = (1 << 23) - 1;

// front of a cache op: reset, ids -> bitmap, then the ascending list of the missing rows + the plan
static void launch_front_kernels(ce_cache* h, const int64_t* ids, int64_t n, int64_t* slots_out, hipStream_t s,
                                 int allow_pad, bool steady, long long seq_arg, int32_t* miss_list, WbMail* mail_in,
                                 long long in_job, int32_t* miss_host, long long* n_admit_out, bool single_pass_ok) {
  const ce_cache_config_t& c = h->cfg;
  const Layout& L = h->L;
  const int64_t N = c.num_embeddings, C = c.cuda_row_num;
  if (h->front_cleanup_pending) {
    // the call before took the per-lookup front and failed on the host before its k_keys was launched
    hipLaunchKernelGGL(k_front_cleanup, dim3(64), dim3(256), 0, s, (const int32_t*)h->victims,
                       (const FrontWords*)(h->front + (h->front_calls & 1)), h->bitmap, h->fine_cnt, h->coarse_cnt,
                       (int)L.n_chunks);
    h->front_cleanup_pending = false;
  }
  if (single_pass_ok && !mail_in && !miss_host && n < (int64_t)INT32_MAX && L.n_chunks <= kRankMaxChunks) {
    // the per-lookup front (ce_cache_fused.h): two launches, no reset kernel, nothing proportional to the table --
    // every launched call of the zero-copy, staged and chained-admission transports, whatever its size (a window of
    // 3.4 M ids: 0.108 against 0.140 ms for k_begin + k_mark + k_count + k_emit, profiles/r06_ab_front.txt).  The
    // bitmap front stays for the host-gather admission (it reads k_emit's mailbox), for captured calls (no call number
    // to stamp with) and for tables beyond 2^29 rows.
    const long long fc = ++h->front_calls;
    FrontWords* const fw = h->front + (fc & 1);
    FrontWords* const fw_next = h->front + ((fc + 1) & 1);
    if (n > 0) {
      const int U = n >= 65536 ? 2 : 1;
      const int threads = (int)std::min<int64_t>(1024, std::max<int64_t>(64, cdiv(cdiv(n, U), 64) * 64));
      const dim3 tg((unsigned)cdiv(n, (int64_t)threads * U)), tb(threads);
#define CE_TOUCH(U_)                                                                                                \
  hipLaunchKernelGGL((k_touch<U_>), tg, tb, 0, s, ids, n, c.idx_map, c.inverted_cached_idx, N, h->bitmap, h->fine_cnt, \
                     h->coarse_cnt, h->slot_epoch, seq_arg, fw, h->victims, (unsigned)L.list_cap, slots_out, allow_pad)
      if (U == 2) CE_TOUCH(2); else CE_TOUCH(1);
#undef CE_TOUCH
    }
    const int rgrid = (int)std::min<int64_t>(kNumCU, std::max<int64_t>(1, cdiv(std::max(n, C), 4096)));
    hipLaunchKernelGGL(k_miss_rank, dim3(rgrid), dim3(256), (size_t)L.n_chunks * 4, s, (const int32_t*)h->victims, fw, fw_next,
                       (const uint32_t*)h->bitmap, (const int32_t*)h->fine_cnt, (const int32_t*)h->coarse_cnt,
                       (int)L.n_chunks, (unsigned)L.list_cap, miss_list, (const int32_t*)h->slot_epoch, C, h->hist, seq_arg, h->ctl, n, h->ring_dev,
                       (long long)L.stage_rows, steady ? 1 : 0, n_admit_out);
    h->front_cleanup_pending = true;
    return;
  }
  hipLaunchKernelGGL(k_begin, dim3(16), dim3(256), 0, s, h->ctl, h->coarse,
                     (int)(((L.n_chunks >> kCoarseShift) + 1) * 2 * kCoarsePad), h->hist, seq_arg);
  if (n > 0) {
    const MarkCfg mc = mark_cfg(h, n);
    const dim3 mg(mc.blocks), mb(mc.threads);
#define CE_MARK(M, U_)                                                                                             \
  hipLaunchKernelGGL((k_mark<M, U_>), mg, mb, mc.hot_words * 4, s, ids, n, c.idx_map, c.inverted_cached_idx, N,    \
                     h->word_bits, mc.hot_words, h->bitmap, h->ctl, slots_out, allow_pad)
    if (mc.merge) { if (mc.u == 4) CE_MARK(true, 4); else CE_MARK(true, 1); }
    else { if (mc.u == 4) CE_MARK(false, 4); else CE_MARK(false, 1); }
#undef CE_MARK
  }
  hipLaunchKernelGGL(k_count, dim3((unsigned)L.n_chunks), dim3(256), 0, s, (const uint4*)h->bitmap,
                     c.inverted_cached_idx, N, h->blk_unique, h->blk_miss, h->coarse);
  hipLaunchKernelGGL(k_emit, dim3((unsigned)cdiv(L.n_chunks, kEmitSub)), dim3(256), 0, s, (uint4*)h->bitmap, c.inverted_cached_idx, N,
                     h->blk_miss, h->coarse, (int)L.n_chunks, miss_list, h->slot_epoch, seq_arg, h->ctl, C, n, h->ring_dev,
                     mail_in, in_job, (long long)L.stage_rows, miss_host, steady ? 1 : 0, n_admit_out);
}

static int prepare_ids_second_half(ce_cache* h);
static int chained_second_half(ce_cache* h);
static int launch_slots_keys(ce_cache* h);

// Chained admission, first half: the fused front on the call's stream, the admission kernel behind its event on the
// admission stream.  (SwapEngine: why no thread and no parked stream.)
static int chained_first_half(ce_cache* h, const int64_t* ids, int64_t n, int64_t* slots_out, hipStream_t s,
                              int allow_pad, const KeysTail* tail, int split) {
  SwapEngine* const w = h->wb;
  const Layout& L = h->L;
  const long long out_job = w->out_issued + 1;
  const int wbuf = (int)(out_job & 1);
  // write-back out_job - 2 has landed: its staging buffer is this call's, and everything older is in the host table
  // the admission kernel is about to read (what job out_job - 1 still carries it takes from that job's staging buffer)
  int rc = w->wait_out(out_job - 2);
  if (rc) return rc;
  const long long call = ++w->chain_calls;
  const int parity = (int)(call & 1);
  const bool steady = h->free_zero;
  const long long seq_arg = (long long)h->seq;
  PhaseProf* const prof = h->prof;
  const int pslot = (int)(h->seq % kProfDepth);
  int pmark = 0;
  if (prof) {
    prof->collect(pslot);
    prof->chained[pslot] = true;
    (void)hipEventRecord(prof->ev[pslot][pmark++], s);
  }
  launch_front_kernels(h, ids, n, slots_out, s, allow_pad, steady, seq_arg, h->miss_list_b[parity], (WbMail*)nullptr, 0ll,
                       (int32_t*)nullptr, &h->chain->n_admit[parity], true);
  CE_HIP_CHECK(hipEventRecord(w->ev_miss[parity], s));
  CE_HIP_CHECK(hipStreamWaitEvent(w->in_stream, w->ev_miss[parity], 0));
  if (prof) (void)hipEventRecord(prof->adm0[pslot], w->in_stream);
  {
    // 32 workgroups for the calls of prefetch_num 1-2 (the rows are waited for almost at once), 20 for window-sized
    // ones (the admission runs beside the window's training kernels: wider grids finish sooner and slow those --
    // rounds 2-5: 8 / 16 / 20 / 32 / 64 workgroups)
    const int blocks = n <= 600000 ? 32 : 20;
    const long long prev = out_job - 1;                      // the write-back job whose rows may still be on their way
    const int pb = (int)(prev & 1);
    // rows the previous call evicted come out of that job's staging buffer only while the job has not landed (the
    // look-up costs every row of the admission a dependent round trip before its PCIe read can start)
    bool landed;
    {
      std::lock_guard<std::mutex> g(w->m);
      landed = w->out_done >= prev;
      w->in_jobs += 1;
      if (!landed) w->in_probed += 1;
    }
    const bool has_prev = prev >= w->probe_floor && !landed;
    const unsigned long long* ek = has_prev ? h->evt_keys[pb] : nullptr;
    const long long* n_ptr = &h->chain->n_admit[parity];
    if (h->vec)
      hipLaunchKernelGGL((k_admit_probe<f32x4>), dim3(blocks), dim3(1024), 0, w->in_stream, h->miss_list_b[parity], n_ptr,
                         (const f32x4*)h->cfg.host_weight_dev, (f32x4*)h->in_stage, h->rowlen, h->g_log2, ek,
                         (const int32_t*)h->evt_pos[pb], h->evt_mask, (uint32_t)prev,
                         (const f32x4*)(pb ? h->stage2 : h->stage));
    else
      hipLaunchKernelGGL((k_admit_probe<float>), dim3(blocks), dim3(1024), 0, w->in_stream, h->miss_list_b[parity], n_ptr,
                         (const float*)h->cfg.host_weight_dev, (float*)h->in_stage, h->rowlen, h->g_log2, ek,
                         (const int32_t*)h->evt_pos[pb], h->evt_mask, (uint32_t)prev,
                         (const float*)(pb ? h->stage2 : h->stage));
  }
  if (prof) (void)hipEventRecord(prof->ev[pslot][pmark++], s);
  ce_cache::Pending& x = h->pend;
  x.n = n; x.slots_out = slots_out; x.s = s; x.worker = true; x.capturing = false; x.has_tail = tail != nullptr;
  x.in_job = 0; x.seq_arg = seq_arg; x.pslot = pslot; x.pmark = pmark; x.prof = prof;
  x.chained = true; x.call = call; x.parity = parity;
  x.sel_pending = true;
  x.sel_s = s; x.sel_n = n; x.sel_steady = steady; x.sel_out_job = out_job; x.sel_wbuf = wbuf;
  x.sel_n_vblocks = (int)cdiv(h->cfg.cuda_row_num, 4096);
  if (tail)
    x.tail = {tail->n_batches, tail->nnz_per_batch, tail->src_keys, tail->offsets, tail->offsets_are_i64,
              tail->offsets_batch_stride, tail->num_bags, tail->include_last_offset, tail->hook_features, tail->keys_out};
  (void)L;
  if (split) {
    x.active = true;
    CE_LAUNCH_CHECK();
    return CE_OK;
  }
  return chained_second_half(h);
}

// ... second half: selection + staging (+ the maps of the admitted rows: they need no payload) on the call's stream,
// the unpack kernel behind it on the admission stream, the slots / keys of the call's ids on the call's stream again.
static int chained_second_half(ce_cache* h) {
  ce_cache::Pending& x = h->pend;
  x.active = false;
  x.sel_pending = false;
  SwapEngine* const w = h->wb;
  const ce_cache_config_t& c = h->cfg;
  const Layout& L = h->L;
  hipStream_t s = x.s;
  const int parity = x.parity, wbuf = x.sel_wbuf, pslot = x.pslot;
  const long long call = x.call, out_job = x.sel_out_job;
  PhaseProf* const prof = (h->prof && (const void*)h->prof == x.prof) ? h->prof : nullptr;
  if (prof) {
    (void)hipEventRecord(prof->resume[pslot], s);
    prof->resumed[pslot] = true;
  }
  // The selection rewrites the staging buffer and the row table of write-back job out_job - 2 -- which the PREVIOUS
  // call's admission / unpack kernels read (rows that job carried) -- and the previous-but-one call's miss / free
  // lists come up for reuse with the next front: this call's selection goes behind the previous call's rows.
  if (call > 1) CE_HIP_CHECK(hipStreamWaitEvent(s, w->ev_rows[(call - 1) % SwapEngine::kRowsRing], 0));
  hipEvent_t sel_done = w->out_ev[wbuf];
  if (x.sel_steady && x.sel_n <= kScanMaxIds) {
    // a full cache: keys, the histogram passes, the victims ranked into the ascending free-slot list, then ONE kernel
    // for staging + both maps
    const int64_t N = c.num_embeddings, C = c.cuda_row_num;
    const int lfu = c.evict_strategy == CE_EVICT_LFU;
    const int top_pass = select_top_pass(h, x.sel_n, false);
    const FrontTail ft = take_front_tail(h);
    hipLaunchKernelGGL(k_keys, dim3(std::min(grid_for(C, 256 * 4), 512)), dim3(256), 0, s, c.cached_idx_map, c.freq_cnter,
                       h->slot_epoch, C, N, x.seq_arg, c.protect_depth, h->slot_bits, lfu, top_pass, h->keys, h->hist, h->ctl,
                       ft.miss_tmp, ft.fw, ft.bitmap, ft.fine, ft.coarse, (int)L.n_chunks);
    const int hgrid = (int)std::min<int64_t>(kNumCU, std::max<int64_t>(1, cdiv(C, 1024 * 4)));
    for (int pass = top_pass - 1; pass >= 0; --pass)
      hipLaunchKernelGGL(k_hist, dim3(hgrid), dim3(1024), 0, s, h->keys, C, pass, top_pass, h->hist, h->ctl);
    if (prof) (void)hipEventRecord(prof->ev[pslot][x.pmark++], s);
    StageArgs a;
    a.cached_idx_map = c.cached_idx_map;
    a.inverted = c.inverted_cached_idx;
    a.freq = c.freq_cnter;
    a.slot_epoch = h->slot_epoch;
    a.C = C;
    a.seq_arg = x.seq_arg;
    a.top_pass = top_pass;
    a.keys = h->keys;
    a.hist = h->hist;
    a.ctl = h->ctl;
    a.lb = h->lb_remap;
    h->lb_tag = (h->lb_tag + 1) & 0xffffu;
    a.tag = h->lb_tag;
    a.free_list = h->free_list_b[parity];
    a.miss_list = h->miss_list_b[parity];
    a.ring = h->ring_dev;
    a.n_unpack_out = &h->chain->n_unpack[parity];
    a.cache = c.cache_weight;
    a.stage = wbuf ? h->stage2 : h->stage;
    a.stage_rows_idx = wbuf ? h->stage_idx2 : h->stage_idx;
    a.scap = (long long)L.stage_rows;
    a.rowlen = h->rowlen;
    a.g_log2 = h->g_log2;
    a.vec = h->vec;
    a.mail = w->mail_dev + wbuf;
    a.job = out_job;
    a.evt = EvTable{h->evt_keys[wbuf], h->evt_pos[wbuf], h->evt_mask};
    a.host_overflow = L.list_cap > L.stage_rows ? (void*)c.host_weight_dev : nullptr;
    hipLaunchKernelGGL(k_rank_victims, dim3((unsigned)cdiv(C, kRemapSlots)), dim3(256), 0, s, a);
    {
      const int gpb = 256 >> h->g_log2;
      const int sgrid = (int)std::min<int64_t>(x.sel_n <= 600000 ? 256 : 512,
                                               std::max<int64_t>(1, cdiv(L.stage_rows, gpb * kStageRowsInFlight)));
      hipLaunchKernelGGL(k_stage_maps, dim3(sgrid), dim3(256), 0, s, a);
    }
    CE_HIP_CHECK(hipEventRecord(w->out_ev[wbuf], s));
    w->push_out();
    if (prof)
      for (int j = 0; j < 2; ++j) (void)hipEventRecord(prof->ev[pslot][x.pmark++], s);      // (free list: inside)
  } else {
    // the cache still has free slots (warm-up): the per-phase kernels, then the maps -- before the rows, which they
    // do not need -- with the count the unpack kernel moves
    SelArgs sel{s, x.sel_n, true, false, x.sel_steady, out_job, x.seq_arg, wbuf, x.sel_n_vblocks, pslot, x.prof,
                h->free_list_b[parity]};
    int rc0 = select_and_stage(h, sel, &x.pmark);
    if (rc0) return rc0;
    hipLaunchKernelGGL(k_admit_maps, dim3(grid_for(L.list_cap, 256)), dim3(256), 0, s, h->miss_list_b[parity],
                       h->free_list_b[parity], (const long long*)&h->ctl->n_miss, 0ll, c.cached_idx_map,
                       c.inverted_cached_idx, c.freq_cnter, (const int64_t*)nullptr, h->slot_epoch, 0, h->ctl,
                       h->ring_dev, x.seq_arg, (const unsigned long long*)nullptr, 0ll, &h->chain->n_unpack[parity]);
    CE_HIP_CHECK(hipEventRecord(w->ev_miss[parity], s));
    sel_done = w->ev_miss[parity];
  }
  CE_HIP_CHECK(hipStreamWaitEvent(w->in_stream, sel_done, 0));
  {
    const int gpb = 256 >> h->g_log2;
    const int ugrid = (int)std::min<int64_t>(x.sel_n <= 600000 ? 128 : 512, std::max<int64_t>(1, cdiv(L.stage_rows, gpb)));
    const long long prev = out_job - 1;
    const int pb = (int)(prev & 1);
    const unsigned long long* ek = prev >= w->probe_floor ? h->evt_keys[pb] : nullptr;
    const bool tail_possible = L.list_cap > L.stage_rows;
    if (h->vec)
      hipLaunchKernelGGL((k_unpack_chained<f32x4>), dim3(ugrid), dim3(256), 0, w->in_stream, h->free_list_b[parity],
                         (const long long*)&h->chain->n_unpack[parity], (long long)L.stage_rows,
                         (const f32x4*)h->in_stage, (f32x4*)c.cache_weight, h->rowlen, h->g_log2,
                         (const int32_t*)h->miss_list_b[parity],
                         tail_possible ? (const f32x4*)c.host_weight_dev : (const f32x4*)nullptr, ek,
                         (const int32_t*)h->evt_pos[pb], h->evt_mask, (uint32_t)prev,
                         (const f32x4*)(pb ? h->stage2 : h->stage));
    else
      hipLaunchKernelGGL((k_unpack_chained<float>), dim3(ugrid), dim3(256), 0, w->in_stream, h->free_list_b[parity],
                         (const long long*)&h->chain->n_unpack[parity], (long long)L.stage_rows,
                         (const float*)h->in_stage, (float*)c.cache_weight, h->rowlen, h->g_log2,
                         (const int32_t*)h->miss_list_b[parity],
                         tail_possible ? (const float*)c.host_weight_dev : (const float*)nullptr, ek,
                         (const int32_t*)h->evt_pos[pb], h->evt_mask, (uint32_t)prev,
                         (const float*)(pb ? h->stage2 : h->stage));
  }
  if (prof) (void)hipEventRecord(prof->adm1[pslot], w->in_stream);
  hipEvent_t rows = w->ev_rows[call % SwapEngine::kRowsRing];
  CE_HIP_CHECK(hipEventRecord(rows, w->in_stream));
  if (prof) (void)hipEventRecord(prof->ev[pslot][x.pmark++], s);      // ("admit_swap" is timed on the admission stream)
  int rc = launch_slots_keys(h);
  if (rc) return rc;
  if (prof) prof->pending[pslot] = true;
  CE_LAUNCH_CHECK();
  // the caller's stream order covers the rows unless it asked to wait for them itself (ce_cache_wait_rows)
  if (!w->deferred_rows) CE_HIP_CHECK(hipStreamWaitEvent(s, rows, 0));
  CE_HIP_CHECK(hipEventRecord(h->ev, s));
  return CE_OK;
}

// split != 0: only the first half is enqueued -- ce_cache_prepare_ids_finish enqueues the rest.  See ce_api.h.
static int prepare_ids_impl(ce_cache_t* h, const int64_t* ids, int64_t n, int64_t* slots_out, ce_stream_t stream,
                            int allow_pad, const KeysTail* tail = nullptr, int split = 0) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  CE_REQUIRE(!h->pend.active, CE_ERR_INVALID, "a cache op begun with ce_cache_prepare_ids_begin has not been finished");
  CE_REQUIRE(n >= 0 && n <= std::max<int64_t>(h->cfg.max_ids_per_call, 0), CE_ERR_INVALID,
             "n=%lld exceeds max_ids_per_call=%lld", (long long)n, (long long)h->cfg.max_ids_per_call);
  CE_REQUIRE(n == 0 || (ids && slots_out), CE_ERR_INVALID, "null ids/slots");
  hipStream_t s = (hipStream_t)stream;
  // A call issued while `stream` is being captured becomes part of the caller's hipGraph: it does not run now, so
  // nothing host-side may depend on it -- no call number (the device counts replays itself, Ctl::seq), no
  // event, no profiling, no worker hand-shake.  The caller reports every replay with ce_cache_graph_replayed.
  hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
  if (s != nullptr) CE_HIP_CHECK(hipStreamIsCapturing(s, &cap_st));
  const bool capturing = cap_st == hipStreamCaptureStatusActive;
  if (capturing) {
    CE_REQUIRE(h->cfg.transport == CE_TRANSPORT_ZEROCOPY, CE_ERR_UNSUPPORTED,
               "only the zero-copy transport can be captured in a hipGraph (the others hand rows to host threads)");
    CE_REQUIRE(!h->prof, CE_ERR_UNSUPPORTED, "switch the phase timers off before capturing a cache op");
    CE_REQUIRE(!split, CE_ERR_UNSUPPORTED, "a cache op in two halves cannot be captured in a hipGraph");
  }
  int rc = capturing ? CE_OK : before_call(h);
  if (rc) return rc;
  const ce_cache_config_t& c = h->cfg;
  const Layout& L = h->L;
  const int64_t N = c.num_embeddings, C = c.cuda_row_num;
  bool worker = c.transport == CE_TRANSPORT_WORKER;
  if (worker) {
    rc = ensure_writeback(h);
    if (rc) return rc;
    if (h->wb->chained) {
      rc = h->wb->check();
      if (rc) return rc;
      h->seq += 1;
      return chained_first_half(h, ids, n, slots_out, s, allow_pad, tail, split);
    }
  }
  if (!capturing) h->seq += 1;
  const long long seq_arg = capturing ? 0ll : (long long)h->seq;     // 0: the device's own count
  ce_call_stats_t* const ring = h->ring_dev;
  // swap kernels: small grid (default 2 workgroups per CU's worth of slots is left to training kernels)
  // protect_depth > 0 means the call overlaps with training kernels on another stream: stay small (32
  // workgroups measured best: 1.43 -> 1.82 G lookups/s); alone on the GPU a wider grid finishes sooner
  const int swap_blocks = c.protect_depth > 0 ? 32 : 512;
  const int swap_threads = 256;
  const int cap_groups = (int)std::min<int64_t>(swap_blocks, std::max<int64_t>(1, cdiv(L.list_cap, (swap_threads >> h->g_log2) * kSwapRows)));

  // worker transport with the host-gather admission: job numbers / staging buffer of this call; the launch thread only
  // waits here when a worker is two calls behind (its buffers and events are about to be reused)
  long long out_job = 0, in_job = 0;
  int wbuf = 0;
  if (worker) {
    rc = worker_selftest(h, s);
    if (rc == CE_ERR_UNSUPPORTED) {
      // loud, once: the environment cannot run this transport; the zero-copy swap kernel needs nothing from it
      fprintf(stderr, "[libce_hip] %s -- falling back to the zero-copy transport\n", ce_last_error());
      rc = h->wb->wait_out(h->wb->out_issued);
      if (rc) return rc;
      h->cfg.transport = CE_TRANSPORT_ZEROCOPY;
      worker = false;
    } else if (rc) {
      return rc;
    }
  }
  if (worker) {
    {
      // the stream about to be parked must not live in the copy streams' hardware-queue pool (see ensure_writeback)
      int prio = 0, prio_lo = 0, prio_hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
      if (s != nullptr && prio_hi != prio_lo && hipStreamGetPriority(s, &prio) == hipSuccess)
        CE_REQUIRE(prio != prio_hi, CE_ERR_UNSUPPORTED,
                   "the worker transport's host-gather admission cannot run its cache op on a highest-priority stream "
                   "(its copy streams use that priority to stay out of the parked stream's hardware queue)");
    }
    out_job = h->wb->out_issued + 1;
    in_job = h->wb->in_issued + 1;
    wbuf = (int)(out_job & 1);
    rc = h->wb->wait_out(out_job - 2);
    if (rc == CE_OK) rc = h->wb->wait_in(in_job - 2);
    if (rc) return rc;
  }
  // steady-state form (see ce_cache::free_zero): the slots to fill are this call's victims
  const bool steady = h->free_zero && !capturing && (c.transport == CE_TRANSPORT_ZEROCOPY || worker);
  const int n_vblocks = (int)cdiv(C, 4096);
  PhaseProf* const prof = h->prof;
  const int pslot = (int)(h->seq % kProfDepth);
  int pmark = 0;
  if (prof) prof->collect(pslot);
#define CE_PHASE() do { if (prof) (void)hipEventRecord(prof->ev[pslot][pmark++], s); } while (0)
  CE_PHASE();
  // (a captured call keeps the two-pass front: its look-back tag would be frozen into the graph)
  launch_front_kernels(h, ids, n, slots_out, s, allow_pad, steady, seq_arg, h->miss_list,
                       worker ? h->wb->mail_dev + 2 : (WbMail*)nullptr, in_job,
                       worker ? h->wb->miss_host_dev : (int32_t*)nullptr, (long long*)nullptr, !capturing);
  // the admission worker starts gathering the missed rows (host table -> pinned staging -> in_stage) right behind
  // k_emit; it first lets every earlier write-back land
  if (worker) {
    CE_HIP_CHECK(hipEventRecord(h->wb->in_ev[in_job & 1], s));
    h->wb->push_in(out_job - 1);
  }
  CE_PHASE();
  SelArgs sel{s, n, worker, capturing, steady, out_job, seq_arg, wbuf, n_vblocks, pslot, (const void*)prof, h->free_list};
  // a call in two halves on the worker transport: the selection / staging part moves into the second half
  const bool defer_sel = split && worker;
  if (!defer_sel) {
    rc = select_and_stage(h, sel, &pmark);
    if (rc) return rc;
  }
#undef CE_PHASE
  // ---- second half: from here on the call needs the missed rows
  ce_cache::Pending& x = h->pend;
  x.n = n; x.slots_out = slots_out; x.s = s; x.worker = worker; x.capturing = capturing; x.has_tail = tail != nullptr;
  x.in_job = in_job; x.seq_arg = seq_arg; x.cap_groups = cap_groups; x.swap_threads = swap_threads; x.pslot = pslot;
  x.pmark = pmark;
  x.prof = prof;
  x.chained = false;
  x.sel_pending = defer_sel;
  x.sel_s = s; x.sel_n = n; x.sel_steady = steady; x.sel_out_job = out_job; x.sel_wbuf = wbuf; x.sel_n_vblocks = n_vblocks;
  if (tail)
    x.tail = {tail->n_batches, tail->nnz_per_batch, tail->src_keys, tail->offsets, tail->offsets_are_i64,
              tail->offsets_batch_stride, tail->num_bags, tail->include_last_offset, tail->hook_features, tail->keys_out};
  if (split) {
    x.active = true;
    CE_LAUNCH_CHECK();
    return CE_OK;
  }
  return prepare_ids_second_half(h);
}

// the slots (and the window's keys) of the call's ids: the call's last launch
static int launch_slots_keys(ce_cache* h) {
  ce_cache::Pending& x = h->pend;
  const ce_cache_config_t& c = h->cfg;
  const int64_t C = c.cuda_row_num, n = x.n;
  int64_t* const slots_out = x.slots_out;
  hipStream_t s = x.s;
  const ce_stream_t stream = (ce_stream_t)s;
  const int lfu = c.evict_strategy == CE_EVICT_LFU;
  PhaseProf* const prof = (h->prof && (const void*)h->prof == x.prof) ? h->prof : nullptr;
  const decltype(x.tail)* const tail = x.has_tail ? &x.tail : nullptr;
  int rc = CE_OK;
  if (n > 0 && lfu) {
    // ~8 k lookups per workgroup keep the LDS hash table (8192 entries) below half full
    hipLaunchKernelGGL(k_slots_lfu, dim3(std::min(grid_for(n, 8192), kMaxBlocks)), dim3(1024), 0, s, slots_out, n,
                       c.inverted_cached_idx, c.freq_cnter, (const Ctl*)h->ctl);
    if (tail) {
      rc = tail->src_keys
               ? ce_bag_presort_window_src(slots_out, tail->nnz_per_batch, tail->n_batches, C, tail->offsets,
                                           tail->offsets_are_i64, tail->offsets_batch_stride, tail->num_bags,
                                           tail->include_last_offset, tail->hook_features, tail->keys_out, stream)
               : ce_bag_presort_window(slots_out, tail->nnz_per_batch, tail->n_batches, C, tail->keys_out, stream);
      if (rc) return rc;
    }
  } else if (n > 0 && tail) {
    // rows -> slots AND the window's keys in one pass (k_slots would write 8 bytes per id for the presort to read back)
    rc = presort_window_from_rows(slots_out, tail->nnz_per_batch, tail->n_batches, C, c.inverted_cached_idx,
                                  &h->ctl->status, tail->src_keys, tail->offsets, tail->offsets_are_i64,
                                  tail->offsets_batch_stride, tail->num_bags, tail->include_last_offset,
                                  tail->hook_features, tail->keys_out, s);
    if (rc) return rc;
  } else if (n > 0) {
    hipLaunchKernelGGL(k_slots, dim3(grid_for(n, 256 * 4)), dim3(256), 0, s, slots_out, n, c.inverted_cached_idx,
                       (const Ctl*)h->ctl);
  }
  if (prof) (void)hipEventRecord(prof->ev[x.pslot][x.pmark++], s);                   // end of "ids_to_slots"
  return CE_OK;
}

static int prepare_ids_second_half(ce_cache* h) {
  ce_cache::Pending& x = h->pend;
  if (x.chained) return chained_second_half(h);
  x.active = false;
  if (x.sel_pending) {
    x.sel_pending = false;
    SelArgs sel{x.sel_s, x.sel_n, x.worker, x.capturing, x.sel_steady, x.sel_out_job, x.seq_arg, x.sel_wbuf, x.sel_n_vblocks,
                x.pslot, x.prof, h->free_list};
    if (h->prof && (const void*)h->prof == x.prof) {
      (void)hipEventRecord(h->prof->resume[x.pslot], x.sel_s);
      h->prof->resumed[x.pslot] = true;
    }
    int rc0 = select_and_stage(h, sel, &x.pmark);
    if (rc0) return rc0;
  }
  const ce_cache_config_t& c = h->cfg;
  const Layout& L = h->L;
  const int64_t n = x.n;
  hipStream_t s = x.s;
  const bool worker = x.worker, capturing = x.capturing;
  const long long in_job = x.in_job;
  const int cap_groups = x.cap_groups, gpb = 256 >> h->g_log2;
  const dim3 swap_block(x.swap_threads);
  PhaseProf* const prof = (h->prof && (const void*)h->prof == x.prof) ? h->prof : nullptr;
  const int pslot = x.pslot;
  int rc = CE_OK;
  if (worker) {
    // host-gather admission: the missed rows arrive in in_stage through the admission worker; this stream parks in
    // the command processor until the worker's store to the pinned word it polls
    const long long scap = (long long)L.stage_rows;
    CE_HIP_CHECK(hipStreamWaitValue64(s, h->wb->sig, (uint64_t)in_job, hipStreamWaitValueGte, ~0ull));
    const int ugrid = (int)std::min<int64_t>(1024, std::max<int64_t>(1, cdiv(L.stage_rows, gpb)));
    if (h->vec) {
      hipLaunchKernelGGL((k_unpack_admitted<f32x4>), dim3(ugrid), dim3(256), 0, s, h->free_list,
                         (const long long*)&h->ctl->n_miss, scap, (const f32x4*)h->in_stage, (f32x4*)c.cache_weight,
                         h->rowlen, h->g_log2, h->ctl, (const unsigned long long*)(h->wb->sig_dev + 1),
                         in_job, (const int32_t*)h->miss_list,
                         L.list_cap > L.stage_rows ? (const f32x4*)c.host_weight_dev : (const f32x4*)nullptr);
    } else {
      hipLaunchKernelGGL((k_unpack_admitted<float>), dim3(ugrid), dim3(256), 0, s, h->free_list,
                         (const long long*)&h->ctl->n_miss, scap, (const float*)h->in_stage, (float*)c.cache_weight,
                         h->rowlen, h->g_log2, h->ctl, (const unsigned long long*)(h->wb->sig_dev + 1),
                         in_job, (const int32_t*)h->miss_list,
                         L.list_cap > L.stage_rows ? (const float*)c.host_weight_dev : (const float*)nullptr);
    }
  } else if (c.transport == CE_TRANSPORT_ZEROCOPY) {
    // write-back of the staged victims + admission of the missed rows, one launch, both PCIe directions busy
    const long long scap = (long long)L.stage_rows;
    // rows in flight per lane group: 16 keeps a window-sized swap (50 k rows) on a small grid; a call of a few thousand
    // rows (B = 2048 shapes) would then fill only a handful of groups -- 4 spreads it (38 -> 21 us per call)
    const int swap_rows = n <= 131072 ? 4 : kSwapRows;
    // workgroups of the write-back part: as many as admit when the call has the GPU to itself; half as many when it
    // overlaps with training (protect_depth > 0) -- PCIe writes are what slows the kernels next to them, and fewer
    // rows leave than enter (32 + 16 workgroups: 2.11 -> 2.22 G lookups/s; 32 + 8 makes the write-back the bottleneck)
    const int wb_groups = c.protect_depth > 0 ? std::max(1, cap_groups / 2) : cap_groups;
#define CE_SWAP(VT, R)                                                                                          \
  hipLaunchKernelGGL((k_swap<VT, R>), dim3(wb_groups + cap_groups), swap_block, 0, s, h->stage_idx,             \
                     (const VT*)h->stage, scap, wb_groups, h->miss_list, h->free_list,                           \
                     (const long long*)&h->ctl->n_miss,                                                          \
                     (VT*)c.host_weight_dev, (VT*)c.cache_weight, h->rowlen, h->g_log2, (const Ctl*)h->ctl)
    if (h->vec) {
      if (swap_rows == 4) CE_SWAP(f32x4, 4); else CE_SWAP(f32x4, 16);
    } else {
      CE_SWAP(float, 16);
    }
#undef CE_SWAP
  } else {
    // H2D: worker threads gather the missed rows out of the table into pinned staging
    const Ctl ctl = *h->ctl_host;   // filled by staged_swap
    const int64_t m = (ctl.status == CE_OK) ? ctl.n_miss : 0;
    if (m > 0) {
      const int D = c.embedding_dim;
      const size_t rowbytes = (size_t)D * sizeof(float);
      CE_HIP_CHECK(hipMemcpyAsync(h->list_host, h->miss_list, (size_t)m * 4, hipMemcpyDeviceToHost, s));
      CE_HIP_CHECK(hipStreamSynchronize(s));
      const float* table = c.host_weight;
      float* st = h->stage_host;
      const int64_t chunk = staged_chunk(h, m);
      for (int64_t off = 0; off < m; off += chunk) {
        const int64_t cnt = std::min(chunk, m - off);
        if (off > 0) CE_HIP_CHECK(hipStreamSynchronize(s));      // the staging buffer is reused
        const int32_t* rows = h->list_host + off;
        host_pool(h)->parallel(cnt, [=](int64_t lo, int64_t hi) {
          for (int64_t i = lo; i < hi; ++i) memcpy(st + (size_t)i * D, table + (size_t)rows[i] * D, rowbytes);
        });
        CE_HIP_CHECK(hipMemcpyAsync(h->stage_dev, h->stage_host, (size_t)cnt * rowbytes, hipMemcpyHostToDevice, s));
        if (h->vec)
          hipLaunchKernelGGL((k_unpack_rows<f32x4>), dim3(grid_for(cnt, gpb)), dim3(256), 0, s,
                             (const int32_t*)h->free_list + off, (long long)cnt, (const f32x4*)h->stage_dev,
                             (f32x4*)c.cache_weight, h->rowlen, h->g_log2);
        else
          hipLaunchKernelGGL((k_unpack_rows<float>), dim3(grid_for(cnt, gpb)), dim3(256), 0, s,
                             (const int32_t*)h->free_list + off, (long long)cnt, (const float*)h->stage_dev,
                             (float*)c.cache_weight, h->rowlen, h->g_log2);
      }
    }
  }
  // map updates + publication of the call's record (the last kernel that can amend it; it also turns a LOST admission
  // job of the host-gather worker into a failed call with nothing marked resident)
  hipLaunchKernelGGL(k_admit_maps, dim3(grid_for(L.list_cap, 256)), dim3(256), 0, s, h->miss_list, h->free_list,
                     (const long long*)&h->ctl->n_miss, 0ll, c.cached_idx_map, c.inverted_cached_idx,
                     c.freq_cnter, (const int64_t*)nullptr, h->slot_epoch, 0, h->ctl, h->ring_dev,
                     x.seq_arg, x.worker ? (const unsigned long long*)(h->wb->sig_dev + 1) : nullptr, x.in_job);
  if (prof) (void)hipEventRecord(prof->ev[pslot][x.pmark++], s);      // end of "admit_swap"
  rc = launch_slots_keys(h);
  if (rc) return rc;
  if (prof) prof->pending[pslot] = true;
  CE_LAUNCH_CHECK();
  if (!capturing) CE_HIP_CHECK(hipEventRecord(h->ev, s));
  return CE_OK;
}

extern "C" int ce_cache_prepare_ids(ce_cache_t* h, const int64_t* ids, int64_t n, int64_t* slots_out,
                                    ce_stream_t stream) {
  return prepare_ids_impl(h, ids, n, slots_out, stream, 0);
}

extern "C" int ce_cache_prepare_ids_keys(ce_cache_t* h, const int64_t* ids, int64_t n_batches, int64_t nnz_per_batch,
                                         int64_t* slots_out, int32_t src_keys, const void* offsets,
                                         int32_t offsets_are_i64, int64_t offsets_batch_stride, int64_t num_bags,
                                         int32_t include_last_offset, int64_t hook_features, uint64_t* keys_out,
                                         ce_stream_t stream) {
  CE_REQUIRE(n_batches > 0 && nnz_per_batch > 0 && keys_out, CE_ERR_INVALID, "bad window shape / null keys_out");
  KeysTail t{n_batches, nnz_per_batch, src_keys, offsets, offsets_are_i64, offsets_batch_stride, num_bags,
             include_last_offset, hook_features, keys_out};
  return prepare_ids_impl(h, ids, n_batches * nnz_per_batch, slots_out, stream, 0, &t);
}

extern "C" int ce_cache_prepare_ids_begin(ce_cache_t* h, const int64_t* ids, int64_t n_batches, int64_t nnz_per_batch,
                                          int64_t* slots_out, int32_t src_keys, const void* offsets,
                                          int32_t offsets_are_i64, int64_t offsets_batch_stride, int64_t num_bags,
                                          int32_t include_last_offset, int64_t hook_features, uint64_t* keys_out,
                                          ce_stream_t stream) {
  CE_REQUIRE(n_batches > 0 && nnz_per_batch > 0, CE_ERR_INVALID, "bad window shape");
  if (!keys_out) return prepare_ids_impl(h, ids, n_batches * nnz_per_batch, slots_out, stream, 0, nullptr, 1);
  KeysTail t{n_batches, nnz_per_batch, src_keys, offsets, offsets_are_i64, offsets_batch_stride, num_bags,
             include_last_offset, hook_features, keys_out};
  return prepare_ids_impl(h, ids, n_batches * nnz_per_batch, slots_out, stream, 0, &t, 1);
}

extern "C" int ce_cache_prepare_ids_finish(ce_cache_t* h, ce_stream_t stream) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  CE_REQUIRE(h->pend.active, CE_ERR_INVALID, "no cache op has been begun");
  CE_REQUIRE((hipStream_t)stream == h->pend.s, CE_ERR_INVALID, "finish the cache op on the stream it was begun on");
  return prepare_ids_second_half(h);
}

extern "C" int ce_cache_prepare_ids_padded(ce_cache_t* h, const int64_t* ids, int64_t n, int64_t* slots_out,
                                           ce_stream_t stream) {
  return prepare_ids_impl(h, ids, n, slots_out, stream, 1);
}

extern "C" int ce_cache_prepare_ids_begin_padded(ce_cache_t* h, const int64_t* ids, int64_t n, int64_t* slots_out,
                                                 ce_stream_t stream) {
  return prepare_ids_impl(h, ids, n, slots_out, stream, 1, nullptr, 1);
}

extern "C" int ce_cache_set_deferred_rows(ce_cache_t* h, int32_t on) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  CE_REQUIRE(!h->pend.active, CE_ERR_INVALID, "a cache op begun with ce_cache_prepare_ids_begin has not been finished");
  h->deferred_rows = on != 0;
  if (h->wb) h->wb->deferred_rows = h->deferred_rows;
  return CE_OK;
}

extern "C" int64_t ce_cache_rows_ticket(ce_cache_t* h) { return (h && h->wb && h->wb->chained) ? h->wb->chain_calls : 0; }

extern "C" int ce_cache_wait_rows(ce_cache_t* h, int64_t ticket, ce_stream_t stream) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  if (!h->wb || !h->wb->chained || h->wb->chain_calls == 0) return CE_OK;
  SwapEngine* const w = h->wb;
  long long t = (ticket <= 0 || ticket > w->chain_calls) ? w->chain_calls : ticket;
  // the ring holds the last kRowsRing calls' events; an older call's rows are implied by any younger call's (one
  // stream moves them all, in order)
  if (t <= w->chain_calls - SwapEngine::kRowsRing) t = w->chain_calls - SwapEngine::kRowsRing + 1;
  CE_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, w->ev_rows[t % SwapEngine::kRowsRing], 0));
  return CE_OK;
}

extern "C" int ce_cache_graph_replayed(ce_cache_t* h, int64_t n_calls, int64_t ids_per_call, ce_stream_t stream) {
  CE_REQUIRE(h && n_calls >= 0 && ids_per_call >= 0, CE_ERR_INVALID, "bad arguments");
  if (n_calls == 0) return CE_OK;
  CE_REQUIRE(n_calls <= kRing / 4, CE_ERR_INVALID, "report at most %d replayed calls at a time", kRing / 4);
  // the ring holds kRing records and the replays were launched already: fold what has arrived long before the
  // device can lap the host
  if (h->seq + n_calls - h->drained >= kRing / 2) {
    const int rc = sync_and_drain(h);
    if (rc) return rc;
  } else {
    drain(h);
  }
  h->seq += n_calls;
  CE_HIP_CHECK(hipEventRecord(h->ev, (hipStream_t)stream));
  if (h->cfg.evict_strategy == CE_EVICT_LFU) {
    h->freq_bound += (uint64_t)n_calls * (uint64_t)ids_per_call;
    CE_REQUIRE(h->freq_bound <= h->graph_freq_limit, CE_ERR_UNSUPPORTED,
               "the captured cache op's LFU key width is used up: capture it again");
  }
  return CE_OK;
}

extern "C" int ce_cache_last_stats(ce_cache_t* h, ce_call_stats_t* out) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  int rc = sync_and_drain(h);
  if (rc) return rc;
  if (h->history.empty()) {
    if (out) memset(out, 0, sizeof(*out));
    return CE_OK;
  }
  const ce_call_stats_t& r = h->history.back();
  if (out) *out = r;
  if (r.status == CE_ERR_CAPACITY)
    set_error("You move %lld embedding rows from CPU to CUDA. %lld rows are available on CUDA. "
              "Please increase cuda_row_num or decrease the training batch size.",
              (long long)r.n_unique, (long long)h->cfg.cuda_row_num);
  else if (r.status == CE_ERR_RANGE)
    set_error("an id is outside [0, %lld)", (long long)h->cfg.num_embeddings);
  return r.status;
}

extern "C" int ce_cache_totals(ce_cache_t* h, int64_t* cpu_to_cuda_numel, int64_t* cuda_to_cpu_numel,
                               int64_t* cache_miss, int64_t* total_cache, int64_t* n_calls) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  drain(h);
  if (cpu_to_cuda_numel) *cpu_to_cuda_numel = h->cpu_to_cuda_numel;
  if (cuda_to_cpu_numel) *cuda_to_cpu_numel = h->cuda_to_cpu_numel;
  if (cache_miss) *cache_miss = h->cache_miss;
  if (total_cache) *total_cache = h->total_cache;
  if (n_calls) *n_calls = h->drained;
  return CE_OK;
}

extern "C" int ce_cache_failures(ce_cache_t* h, int64_t* n_failed, int32_t* last_status, int64_t* last_seq) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  drain(h);
  if (n_failed) *n_failed = h->n_failed;
  if (last_status) *last_status = h->last_fail_status;
  if (last_seq) *last_seq = h->last_fail_seq;
  return CE_OK;
}

extern "C" int64_t ce_cache_history(ce_cache_t* h, int64_t first_seq, ce_call_stats_t* out, int64_t cap) {
  if (!h || !out || cap <= 0) return 0;
  drain(h);
  if (h->history.empty()) return 0;
  // records are stored in call order with consecutive seq: index directly (records older than the kept window
  // are gone)
  int64_t i = first_seq - h->hist_base;
  if (i < 0) i = 0;
  int64_t w = 0;
  for (; i < (int64_t)h->history.size() && w < cap; ++i) out[w++] = h->history[(size_t)i];
  return w;
}

extern "C" int ce_cache_lookup_slots(ce_cache_t* h, const int64_t* ids, int64_t n, int64_t* slots_out,
                                     ce_stream_t stream) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  if (n == 0) return CE_OK;
  CE_REQUIRE(ids && slots_out && n > 0, CE_ERR_INVALID, "null ids/slots");
  const ce_cache_config_t& c = h->cfg;
  hipLaunchKernelGGL(k_lookup, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, ids, n, c.idx_map,
                     c.inverted_cached_idx, c.num_embeddings, slots_out);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int ce_cache_flush(ce_cache_t* h, ce_stream_t stream) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  CE_REQUIRE(!h->pend.active, CE_ERR_INVALID, "a cache op begun with ce_cache_prepare_ids_begin has not been finished");
  int rc = before_call(h);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const ce_cache_config_t& c = h->cfg;
  const int64_t C = c.cuda_row_num;
  if (h->wb) {
    // queued write-backs land first (a row they carry may be resident again and is about to be written by
    // k_flush_rows with its newer value)
    rc = h->wb->wait_out(h->wb->out_issued);
    if (rc) return rc;
  }
  rc = join_rows(h, s);
  if (rc) return rc;
  h->seq += 1;
  h->free_zero = false;            // every slot is free again once this has run
  h->free_reset_seq = h->seq;
  const int gpb = 256 >> h->g_log2;
  hipLaunchKernelGGL(k_begin, dim3(1), dim3(256), 0, s, h->ctl, (int32_t*)nullptr, 0, (uint32_t*)nullptr,
                     (long long)h->seq);
  if (h->vec)
    hipLaunchKernelGGL((k_flush_rows<f32x4>), dim3(grid_for(C, gpb)), dim3(256), 0, s, c.cached_idx_map, C,
                       (const f32x4*)c.cache_weight, (f32x4*)c.host_weight_dev, h->rowlen, h->g_log2);
  else
    hipLaunchKernelGGL((k_flush_rows<float>), dim3(grid_for(C, gpb)), dim3(256), 0, s, c.cached_idx_map, C,
                       (const float*)c.cache_weight, (float*)c.host_weight_dev, h->rowlen, h->g_log2);
  hipLaunchKernelGGL(k_flush_maps, dim3(grid_for(C, 256)), dim3(256), 0, s, c.cached_idx_map, C,
                     c.inverted_cached_idx, c.freq_cnter, h->slot_epoch, h->ctl);
  hipLaunchKernelGGL(k_flush_end, dim3(1), dim3(1), 0, s, C, h->ctl, h->ring_dev + (h->seq % kRing),
                     (long long)h->seq);
  CE_LAUNCH_CHECK();
  CE_HIP_CHECK(hipEventRecord(h->ev, s));
  return CE_OK;
}

extern "C" int ce_cache_set_protect_depth(ce_cache_t* h, int32_t depth) {
  CE_REQUIRE(h && depth >= 0 && depth < 1024, CE_ERR_INVALID, "bad protect depth");
  h->cfg.protect_depth = depth;
  return CE_OK;
}

extern "C" int ce_cache_set_buffer_rows(ce_cache_t* h, int64_t rows) {
  CE_REQUIRE(h && rows >= 0, CE_ERR_INVALID, "bad buffer_rows");
  h->buffer_rows = rows;
  return CE_OK;
}

extern "C" int ce_cache_set_transport(ce_cache_t* h, int32_t transport) {
  CE_REQUIRE(h && (transport == CE_TRANSPORT_ZEROCOPY || transport == CE_TRANSPORT_STAGED ||
                   transport == CE_TRANSPORT_WORKER), CE_ERR_INVALID, "bad transport");
  if (h->cfg.transport == transport) return CE_OK;
  CE_REQUIRE(!h->pend.active, CE_ERR_INVALID, "a cache op begun with ce_cache_prepare_ids_begin has not been finished");
  if (h->wb) {
    // leaving (or re-entering) the worker transport: everything queued reaches the table first.  Blocks.
    int rc = h->wb->wait_out(h->wb->out_issued);
    if (rc == CE_OK) rc = h->wb->wait_in(h->wb->in_issued);
    if (rc) return rc;
    CE_HIP_CHECK(hipEventSynchronize(h->ev));
    if (hipEvent_t ev = last_rows_event(h)) CE_HIP_CHECK(hipEventSynchronize(ev));
    h->wb->probe_floor = h->wb->out_issued + 1;
  }
  h->cfg.transport = transport;
  if (transport == CE_TRANSPORT_WORKER) return ensure_writeback(h);
  return CE_OK;
}

extern "C" int32_t ce_cache_get_transport(ce_cache_t* h) { return h ? h->cfg.transport : -1; }

extern "C" int ce_cache_set_cache_weight(ce_cache_t* h, float* cache_weight) {
  CE_REQUIRE(h && cache_weight, CE_ERR_INVALID, "null handle / pointer");
  CE_REQUIRE(!h->vec || ((uintptr_t)cache_weight & 15) == 0, CE_ERR_INVALID, "the cache must keep its 16-byte alignment");
  CE_HIP_CHECK(hipDeviceSynchronize());       // nothing in flight may still address the old allocation
  h->cfg.cache_weight = cache_weight;
  return CE_OK;
}

extern "C" int ce_cache_set_profiling(ce_cache_t* h, int32_t on) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  if (on && !h->prof) h->prof = new PhaseProf();
  if (!on && h->prof) {
    for (int i = 0; i < kProfDepth; ++i) h->prof->collect(i);
    delete h->prof;
    h->prof = nullptr;
  }
  return CE_OK;
}

extern "C" int32_t ce_cache_phase_count(void) { return kPhases; }
extern "C" const char* ce_cache_phase_name(int32_t i) { return (i >= 0 && i < kPhases) ? kPhaseNames[i] : ""; }

extern "C" int ce_cache_phase_times(ce_cache_t* h, double* ms_out, int32_t cap, int64_t* calls, int32_t reset) {
  CE_REQUIRE(h && ms_out && cap >= kPhases, CE_ERR_INVALID, "bad arguments");
  for (int j = 0; j < kPhases; ++j) ms_out[j] = 0;
  if (calls) *calls = 0;
  if (!h->prof) return CE_OK;
  for (int i = 0; i < kProfDepth; ++i) h->prof->collect(i);
  for (int j = 0; j < kPhases; ++j) ms_out[j] = h->prof->ms[j];
  if (calls) *calls = h->prof->calls;
  if (reset) {
    for (int j = 0; j < kPhases; ++j) h->prof->ms[j] = 0;
    h->prof->calls = 0;
  }
  return CE_OK;
}

extern "C" int ce_cache_writeback_wait(ce_cache_t* h) {
  CE_REQUIRE(h, CE_ERR_INVALID, "null handle");
  if (!h->wb) return CE_OK;
  return h->wb->wait_out(h->wb->out_issued);
}

extern "C" int ce_cache_swap_stats(ce_cache_t* h, double* seconds6, int64_t* counts4) {
  CE_REQUIRE(h && seconds6 && counts4, CE_ERR_INVALID, "null argument");
  double* seconds4 = seconds6;
  for (int i = 0; i < 4; ++i) counts4[i] = 0;
  for (int i = 0; i < 6; ++i) seconds6[i] = 0;
  if (h->wb) {
    std::lock_guard<std::mutex> g(h->wb->m);
    seconds4[0] = h->wb->out_wait_s; seconds4[1] = h->wb->out_busy_s;
    seconds4[2] = h->wb->in_wait_s;  seconds4[3] = h->wb->in_busy_s;
    seconds6[4] = h->wb->in_gather_s;
    seconds6[5] = (double)h->wb->in_probed;
    counts4[0] = h->wb->out_rows; counts4[1] = h->wb->out_jobs;
    counts4[2] = h->wb->in_rows;  counts4[3] = h->wb->in_jobs;
  }
  return CE_OK;
}

extern "C" int ce_cache_free_rows(ce_cache_t* h, int64_t* out) {
  CE_REQUIRE(h && out, CE_ERR_INVALID, "null argument");
  int rc = sync_and_drain(h);
  if (rc) return rc;
  *out = h->history.empty() ? h->cfg.cuda_row_num : h->history.back().n_free_after;
  // a failed call leaves the count untouched but reports the pre-call value too
  return CE_OK;
}
