// Cache op, host side of the transports: the row pool of the staged transport, phase timers and the swap engine
// (SDMA write-back + host scatter threads, chained admission) behind CE_TRANSPORT_WORKER.
// Part of the one translation unit ce_cache.hip (included there, in this order: ce_cache_index.h, ce_cache_select.h,
// ce_cache_rows.h, ce_cache_fused.h, ce_cache_worker.h); not a stand-alone header.
#pragma once

namespace ce {

// persistent worker pool for the staged transport's table gather/scatter (spawning 64 std::threads per call cost
// more than the copies themselves)
class RowPool {
 public:
  explicit RowPool(int n) {
    // the helpers move rows between the host table and pinned staging: keep them on the GPU's NUMA node
    cpu_set_t set;
    const bool near = near_gpu_cpus(&set);
    for (int i = 0; i < n; ++i)
      workers_.emplace_back([this, i, near, set] {
        if (near) (void)sched_setaffinity(0, sizeof set, &set);
        run(i);
      });
  }
  ~RowPool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
      ++gen_;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  void parallel(int64_t n, const std::function<void(int64_t, int64_t)>& fn) {
    const int t = (int)std::min<int64_t>((int64_t)workers_.size(), std::max<int64_t>(1, cdiv(n, 1024)));
    if (t <= 1) {
      fn(0, n);
      return;
    }
    start(n, t, fn);
    wait();
  }
  // non-blocking form: worker i < parts runs fn(lo_i, hi_i) over its share of [0, n); `fn` must outlive wait()
  void start(int64_t n, int parts, const std::function<void(int64_t, int64_t)>& fn) {
    parts = std::max(1, std::min(parts, (int)workers_.size()));
    {
      std::lock_guard<std::mutex> g(m_);
      fn_ = &fn;
      n_ = n;
      parts_ = parts;
      pending_ = parts;
      ++gen_;
    }
    cv_.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> g(m_);
    done_.wait(g, [this] { return pending_ == 0; });
  }
  int size() const { return (int)workers_.size(); }

 private:
  void run(int id) {
    unsigned long long seen = 0;
    for (;;) {
      const std::function<void(int64_t, int64_t)>* fn;
      int64_t n;
      int parts;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        fn = fn_;
        n = n_;
        parts = parts_;
      }
      if (id < parts) {
        const int64_t per = cdiv(n, parts);
        const int64_t lo = id * per, hi = std::min<int64_t>(n, lo + per);
        if (lo < hi) (*fn)(lo, hi);
        std::lock_guard<std::mutex> g(m_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int64_t, int64_t)>* fn_ = nullptr;
  int64_t n_ = 0;
  int parts_ = 0, pending_ = 0;
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

// Per-phase timers of prepare_ids (the reference brackets the same phases with its Timer / record_function
// ranges: recsys/dlrm_main.py:258, upstream CachedParamMgr._elapsed_dict printed by print_comm_stats :294).
// hipEvents on the call's own stream, read back lazily: no host sync is added to the call.
constexpr int kPhases = 6;
constexpr int kProfDepth = 8;
static const char* const kPhaseNames[kPhases] = {"unique_and_miss", "find_evict_ids", "evict_stage",
                                                  "free_slots", "admit_swap", "ids_to_slots"};
struct PhaseProf {
  hipEvent_t ev[kProfDepth][kPhases + 1];
  bool pending[kProfDepth];
  // a call in two halves whose selection / staging part was deferred to the second half: the second phase starts at
  // `resume` (recorded when the second half begins), not at the mark behind the front -- the training steps between
  // the two halves are no phase of the cache op
  hipEvent_t resume[kProfDepth];
  bool resumed[kProfDepth];
  // chained admission: the rows move on the admission stream -- "admit_swap" is the span from the start of the
  // admission kernel to the end of the unpack kernel THERE (it overlaps with the phases around it on the call's stream)
  hipEvent_t adm0[kProfDepth], adm1[kProfDepth];
  bool chained[kProfDepth];
  double ms[kPhases];
  long long calls;
  PhaseProf() : calls(0) {
    for (int i = 0; i < kProfDepth; ++i) {
      pending[i] = false;
      resumed[i] = false;
      chained[i] = false;
      (void)hipEventCreate(&resume[i]);
      (void)hipEventCreate(&adm0[i]);
      (void)hipEventCreate(&adm1[i]);
      for (int j = 0; j <= kPhases; ++j) (void)hipEventCreate(&ev[i][j]);
    }
    for (int j = 0; j < kPhases; ++j) ms[j] = 0;
  }
  ~PhaseProf() {
    for (int i = 0; i < kProfDepth; ++i) {
      (void)hipEventDestroy(resume[i]);
      (void)hipEventDestroy(adm0[i]);
      (void)hipEventDestroy(adm1[i]);
      for (int j = 0; j <= kPhases; ++j) (void)hipEventDestroy(ev[i][j]);
    }
  }
  void collect(int i) {              // blocks until call slot i has finished
    if (!pending[i]) return;
    if (hipEventSynchronize(ev[i][kPhases]) == hipSuccess &&
        (!chained[i] || hipEventSynchronize(adm1[i]) == hipSuccess)) {
      for (int j = 0; j < kPhases; ++j) {
        float t = 0;
        hipError_t e;
        if (chained[i] && j == 4) e = hipEventElapsedTime(&t, adm0[i], adm1[i]);
        else e = hipEventElapsedTime(&t, (j == 1 && resumed[i]) ? resume[i] : ev[i][j], ev[i][j + 1]);
        if (e == hipSuccess) ms[j] += t;
      }
      calls += 1;
    }
    pending[i] = false;
    resumed[i] = false;
    chained[i] = false;
  }
};

// host-side row copies of the swap workers: csrc/ce_rowcopy.cpp (widest streaming store the CPU has)
void row_copy_stream(float* dst, const float* src, size_t floats);
void row_copy_fence();

// waiting for a copy stream without burning a CPU of a quota-limited host and without any packet in a hardware
// queue: poll hipStreamQuery with short sleeps
// seconds a swap worker (or the launch thread waiting for one) gives a copy / a job before it declares it lost:
// the parked cache-op stream is then released with the job flagged as failed instead of hanging the GPU for ever
// (what would happen if a copy stream ever shared a hardware queue with the parked stream -- ensure_writeback)
static double worker_timeout_s() {
  static const double v = [] { const char* e = getenv("CE_WORKER_TIMEOUT_S"); const double t = e ? atof(e) : 30.0; return t > 0 ? t : 30.0; }();
  return v;
}
// The workers wait with short sleeps (no queue packets, see run_in).  A thread's default timer slack is 50 us, so
// sleep_for(15 us) returns after ~65 us: at prefetch_num 1 a write-back job is a few hundred microseconds of which
// those overshoots were a third (Kaggle 5 % P = 1: 0.327 ms per job).  Worker threads ask for 1 us of slack.
static inline void tight_timer_slack() { (void)prctl(PR_SET_TIMERSLACK, 1000ul, 0ul, 0ul, 0ul); }

static inline hipError_t stream_wait_polite(hipStream_t st, double timeout_s = worker_timeout_s()) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int spins = 0;; ++spins) {
    const hipError_t e = hipStreamQuery(st);
    if (e != hipErrorNotReady) return e;
    if (spins < 50) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(15));
    if ((spins & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
      return hipErrorNotReady;      // timed out: the caller reports it
  }
}

static inline hipError_t event_wait_polite(hipEvent_t ev, double timeout_s = worker_timeout_s()) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int spins = 0;; ++spins) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    if (spins < 50) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(15));
    if ((spins & 1023) == 1023 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
      return hipErrorNotReady;      // timed out: the caller reports it
  }
}

static const bool g_trace = [] { const char* e = getenv("CE_WORKER_TRACE"); return e && atoi(e) != 0; }();
#define CE_TRACE(...)                                                                            \
  do {                                                                                           \
    if (ce::g_trace) {                                                                           \
      const double t_ = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); \
      fprintf(stderr, "[ce %.6f %p] ", t_, (void*)this);                                         \
      fprintf(stderr, __VA_ARGS__);                                                              \
      fputc('\n', stderr);                                                                       \
    }                                                                                            \
  } while (0)

// Worker transport (CE_TRANSPORT_WORKER): both directions of the row swap leave the CUs.
//
// Measured on the box (profiles/r02_probe_sdma.txt): a pinned hipMemcpyAsync runs on an SDMA engine at ~51 GB/s per
// direction and does NOT slow an HBM-bound kernel running beside it (x1.02), whereas rows moved by waves over the
// mapped host table (round 1's k_swap) held the training kernels back by 40-47 % for as long as they ran.  So:
//
//   out  k_evict_stage packs the victims of call w into an HBM staging buffer; the `out` worker waits for that
//        kernel's event ON ITS OWN THREAD, copies the block out in chunks (hipMemcpyAsync, private stream) and
//        scatters each chunk into the host table with helper threads while the next one is in flight.
//   in   k_emit leaves the ascending list of missed rows in pinned host memory; the `in` worker waits for that
//        kernel's event, lets helper threads gather the rows out of the table into pinned staging, each helper
//        the worker copying each gathered chunk to the device at once (hipMemcpyAsync), and finally releases the
//        cache-op stream, which has been parked in a hipStreamWaitValue64 (no CU involved), with a plain store to the
//        pinned word the stream polls, once the copies have completed.  k_unpack_admitted then moves the rows to
//        their slots.
//
// Ordering: a row evicted by call w-1 and missed by call w must be read back with the payload w-1 staged.  The host
// gather of call w therefore starts only after the write-back of call w-1 has reached the table; the admission
// KERNEL waits for the write-back of call w-2 only and, while that of w-1 is still on its way, looks every missed
// row up in the table of rows w-1 staged (EvTable) and takes a hit out of w-1's staging buffer -- intact until call
// w+1 stages its own victims, which happens behind call w's admission wait.  (Waiting for w-1 put a 1 ms
// write-back, the slower PCIe direction beside the admission's reads, on the cache-op stream's cycle: front ->
// write-back -> admission of the next call.)  Rows evicted by call w itself are never in its miss list.
// The launch thread blocks only when a worker is two calls behind.  A failing HIP call inside a worker still
// releases the stream (the error surfaces at the next call / wait) so the GPU is never left parked.
struct SwapEngine {
  int device = 0;
  int64_t D = 0, stage_rows = 0;
  float* table = nullptr;
  // ---- out (evictions)
  hipStream_t out_stream = nullptr, out_stream2 = nullptr;      // alternating D2H copy streams
  hipEvent_t out_ev[2] = {nullptr, nullptr};       // staging of the job complete (recorded on the cache-op stream)
  static constexpr int kOutChunks = 8;
  hipEvent_t chunk_ev[kOutChunks + 1] = {nullptr};      // behind every chunk copy of the job being written back
  const float* stage_dev[2] = {nullptr, nullptr};
  const int32_t* idx_dev[2] = {nullptr, nullptr};
  float* rows_host[2] = {nullptr, nullptr};        // pinned landing buffers
  int32_t* idx_host[2] = {nullptr, nullptr};
  // ---- in (admissions)
  hipStream_t in_stream = nullptr;
  hipEvent_t in_ev[2] = {nullptr, nullptr};        // miss list of the job complete (by job parity)
  float* in_stage_dev = nullptr;
  float* in_host = nullptr;                        // pinned gather buffer (host-gather admission only)
  // CHAINED admission (round 6; the default: the table has a device mapping): the launch thread itself enqueues, on
  // in_stream, the admission kernel behind the front's event and the unpack kernel behind the selection's event.  No
  // library thread takes part and the cache-op stream never parks: what used to be "front -> [event wake-up of a
  // worker thread, its kernel launch, its polling of the stream, its store to a pinned word the parked stream polls]
  // -> unpack" is two stream-to-stream event edges.  Ordering: the admission of call w reads the host table for rows
  // that no write-back in flight carries -- the launch thread has waited for write-back w - 2 before it enqueues
  // call w (its staging buffer is about to be reused anyway), and rows of write-back w - 1 come out of that job's
  // staging buffer (EvTable), intact until call w + 1's selection, which waits for call w's rows (ev_rows).
  // The HOST-GATHER admission (CE_WORKER_ADMIT=sdma, or a table without device mapping) keeps the worker thread
  // below: helper threads gather the rows into pinned staging, SDMA copies bring them in, the cache-op stream parks in
  // hipStreamWaitValue64 until the thread releases it.
  bool chained = false;
  hipEvent_t ev_miss[2] = {nullptr, nullptr};      // front of the call of either parity complete (cache-op stream)
  static constexpr int kRowsRing = 4;
  hipEvent_t ev_rows[kRowsRing] = {nullptr};       // rows of call c in their slots (in_stream), c % kRowsRing
  long long chain_calls = 0;                       // chained calls issued (1-based ticket of the latest)
  // write-back jobs below this number are never looked up in their staging buffer: calls of ANOTHER transport ran
  // since (ce_cache_set_transport), which may have re-admitted and evicted the same rows past that buffer -- the host
  // table, where every one of those jobs has landed by then, is the up-to-date copy
  long long probe_floor = 1;
  bool deferred_rows = false;                      // prepare_ids does not make its stream wait for the rows
  const unsigned long long* evt_keys[2] = {nullptr, nullptr};
  const int32_t* evt_pos[2] = {nullptr, nullptr};
  uint32_t evt_mask = 0;
  long long in_probed = 0;         // admissions enqueued while the previous write-back was still on its way
#ifdef CE_TEST_HOOKS
  // fault / delay injection for tests/test_gpu_worker.py: only in libce_hip_testhooks.so (build.py, -DCE_TEST_HOOKS);
  // the product library has neither the fields nor the strings (tests/test_abi.py)
  int out_delay_us = 0;            // CE_WORKER_OUT_DELAY_US: every write-back job starts this much late
  long long fail_in_job = 0;       // CE_WORKER_FAIL_IN_JOB: this host-gather admission job reports a failed HIP call
#endif
  int rowlen = 0, g_log2 = 0, vec = 0;
  int32_t* miss_host = nullptr;                    // pinned + mapped: written by k_emit
  int32_t* miss_host_dev = nullptr;
  unsigned long long* sig = nullptr;               // pinned + mapped: [0] the value the cache-op stream waits for,
                                                   // [1] the last admission job that was LOST (k_admit_maps reads it)
  unsigned long long* sig_dev = nullptr;
  // ---- mailboxes (pinned + mapped): [0], [1] = out staging buffers, [2] = in
  WbMail* mail = nullptr;
  WbMail* mail_dev = nullptr;
  std::thread out_thread, in_thread;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  long long out_issued = 0, out_done = 0, in_issued = 0, in_done = 0;
  bool stop = false;
  int err = 0;
  char errmsg[256] = {0};
  RowPool* out_pool = nullptr;
  RowPool* in_pool = nullptr;
  // statistics (what upstream's swap_in_bandwidth / swap_out_bandwidth report)
  double out_wait_s = 0, out_busy_s = 0, in_wait_s = 0, in_busy_s = 0, in_gather_s = 0;
  double out_wait0_s = 0;
  double out_copy_wait_s = 0, out_scatter_s = 0;       // parts of out_busy_s (CE_WORKER_PROFILE=1 prints them at exit)
  long long out_rows = 0, out_jobs = 0, in_rows = 0, in_jobs = 0;

  void fail(const char* what, hipError_t e) {
    std::lock_guard<std::mutex> g(m);
    if (!err) {
      err = CE_ERR_HIP;
      snprintf(errmsg, sizeof errmsg, "swap worker: %s failed: %s", what, hipGetErrorString(e));
    }
  }
  bool failed() {
    std::lock_guard<std::mutex> g(m);
    return err != 0;
  }

  void run_out() {
    (void)hipSetDevice(device);
    bind_thread_near_gpu();
    tight_timer_slack();
    for (;;) {
      long long job;
      {
        std::unique_lock<std::mutex> g(m);
        cv_job.wait(g, [&] { return stop || out_done < out_issued; });
        if (out_done >= out_issued) return;
        job = out_done + 1;
      }
      const int b = (int)(job & 1);
      const auto t0 = std::chrono::steady_clock::now();
      CE_TRACE("out job %lld: waiting for its staging event", job);
      hipError_t e = hipEventSynchronize(out_ev[b]);
      if (e != hipSuccess) fail("hipEventSynchronize(out)", e);
      const auto t1 = std::chrono::steady_clock::now();
      long long k = mail[b].count;
      double copy_wait = 0, scatter = 0, wait0 = 0;
#ifdef CE_TEST_HOOKS
      if (out_delay_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(out_delay_us));
#endif
      CE_TRACE("out job %lld: event done (%s), mail job %lld count %lld", job, hipGetErrorString(e), mail[b].job, k);
      if (mail[b].job != job || k < 0 || k > stage_rows) k = 0;     // a failed / foreign record moves nothing
      if (k > 0 && !failed()) {
        // The packed block leaves in kOutChunks copies, ALL issued at once on the two copy streams in turn, each with
        // an event behind it; chunk c is scattered into the table while the later ones are on the wire.  (Issuing
        // chunk c + 1 only when chunk c was being waited for left the copies and the scatters back to back:
        // 0.6 + 0.55 ms per 54 k-row job instead of overlapped.)  The host polls the events (no queue packets).
        e = hipMemcpyAsync(idx_host[b], idx_dev[b], (size_t)k * 4, hipMemcpyDeviceToHost, out_stream);
        const int64_t per = std::max<int64_t>(4096, cdiv(k, kOutChunks));
        int nch = 0;
        for (int64_t off = 0; off < k && e == hipSuccess; off += per, ++nch) {
          const int64_t cnt = std::min<int64_t>(per, k - off);
          hipStream_t cs = (nch & 1) ? out_stream2 : out_stream;
          e = hipMemcpyAsync(rows_host[b] + off * D, stage_dev[b] + off * D, (size_t)cnt * D * 4, hipMemcpyDeviceToHost, cs);
          if (e == hipSuccess) e = hipEventRecord(chunk_ev[nch], cs);
        }
        int c = 0;
        for (int64_t off = 0; off < k && e == hipSuccess; off += per, ++c) {
          const int64_t cnt = std::min<int64_t>(per, k - off);
          const auto tc0 = std::chrono::steady_clock::now();
          // (chunk 0 also needs the row numbers, which went first on out_stream)
          hipError_t e2 = event_wait_polite(chunk_ev[c]);
          const auto tc1 = std::chrono::steady_clock::now();
          copy_wait += std::chrono::duration<double>(tc1 - tc0).count();
          if (c == 0) wait0 = std::chrono::duration<double>(tc1 - tc0).count();
          if (e == hipSuccess) e = e2;
          if (e != hipSuccess) break;
          float* tb = table;
          const float* st = rows_host[b] + off * D;
          const int32_t* ri = idx_host[b] + off;
          const int64_t d = D;
          out_pool->parallel(cnt, [=](int64_t lo, int64_t hi) {
            // every row lands on a page of its own: a read prefetch of ANOTHER line of the page of the row 8 ahead
            // starts its page walk early (the row's own lines are streamed past the cache and must not be pulled
            // in): 71 -> 48 ns per row on 4 KB pages, 18.5 -> 13 on 2 MB pages (profiles/probes/probe_scatter2.cpp)
            for (int64_t i = lo; i < hi; ++i) {
              if (i + 8 < hi) __builtin_prefetch((const void*)((uintptr_t)(tb + (size_t)ri[i + 8] * d) ^ 2048u), 0, 0);
              row_copy_stream(tb + (size_t)ri[i] * d, st + (size_t)i * d, (size_t)d);
            }
            row_copy_fence();
          });
          scatter += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc1).count();
        }
        if (e == hipSuccess) {         // both streams idle again before the staging buffer is reused
          e = stream_wait_polite(out_stream);
          if (e == hipSuccess) e = stream_wait_polite(out_stream2);
        }
        if (e != hipSuccess) fail(e == hipErrorNotReady ? "write-back copy timed out (CE_WORKER_TIMEOUT_S)" : "D2H copy", e);
      }
      const auto t2 = std::chrono::steady_clock::now();
      CE_TRACE("out job %lld: done", job);
      {
        std::lock_guard<std::mutex> g(m);
        out_done = job;
        out_wait_s += std::chrono::duration<double>(t1 - t0).count();
        out_busy_s += std::chrono::duration<double>(t2 - t1).count();
        out_copy_wait_s += copy_wait;
        out_wait0_s += wait0;
        out_scatter_s += scatter;
        out_rows += k;
        out_jobs += 1;
      }
      cv_done.notify_all();
    }
  }

  void run_in() {
    (void)hipSetDevice(device);
    bind_thread_near_gpu();
    tight_timer_slack();
    for (;;) {
      long long job, need_out;
      {
        std::unique_lock<std::mutex> g(m);
        cv_job.wait(g, [&] { return stop || in_done < in_issued; });
        if (in_done >= in_issued) return;
        job = in_done + 1;
        need_out = out_issued_at[job & 7];
      }
      const auto t0 = std::chrono::steady_clock::now();
      CE_TRACE("in job %lld: waiting for its miss-list event (needs out job %lld)", job, need_out);
      hipError_t e = hipEventSynchronize(in_ev[job & 1]);
      if (e != hipSuccess) fail("hipEventSynchronize(in)", e);
      CE_TRACE("in job %lld: event done (%s)", job, hipGetErrorString(e));
      {
        // rows the earlier calls evicted must be in the table before the host threads read it
        std::unique_lock<std::mutex> g(m);
        cv_done.wait(g, [&] { return out_done >= need_out || err != 0; });
      }
      const auto t1 = std::chrono::steady_clock::now();
      long long n = mail[2].count;
      CE_TRACE("in job %lld: earlier write-backs landed; mail job %lld count %lld", job, mail[2].job, n);
      if (mail[2].job != job || n < 0 || n > stage_rows) n = 0;
#ifdef CE_TEST_HOOKS
      if (fail_in_job > 0 && job == fail_in_job) fail("admission (injected: CE_WORKER_FAIL_IN_JOB)", hipErrorUnknown);
#endif
      if (n > 0 && !failed()) {
        const float* tb = table;
        float* st = in_host;
        float* dv = in_stage_dev;
        const int32_t* rows = miss_host;
        const int64_t d = D;
        hipStream_t cs = in_stream;
        // ONE wake-up of the helpers per job (a condition-variable round trip per chunk cost more than the chunk):
        // they pull 2048-row pieces off a shared counter and flag each finished piece; this thread -- the only one
        // that talks to the runtime -- copies every run of finished pieces to the device as soon as it is 8192 rows
        // long, so the copies trail the gather by one chunk.
        constexpr int kAhead = 8;
        constexpr int64_t kPiece = 2048, kCopyPieces = 4;
        const int64_t npieces = cdiv(n, kPiece);
        std::atomic<int64_t> next{0};
        std::vector<std::atomic<unsigned char>> ready((size_t)npieces);
        for (auto& r : ready) r.store(0, std::memory_order_relaxed);
        const std::function<void(int64_t, int64_t)> work = [&](int64_t, int64_t) {
          for (;;) {
            const int64_t pc = next.fetch_add(1, std::memory_order_relaxed);
            if (pc >= npieces) break;
            const int64_t lo = pc * kPiece, hi = std::min<int64_t>(n, lo + kPiece);
            for (int64_t i = lo; i < hi; ++i) {
              if (i + kAhead < hi) {
                const char* q = (const char*)(tb + (size_t)rows[i + kAhead] * d);
                for (int64_t l = 0; l < d * 4; l += 64) __builtin_prefetch(q + l);
              }
              row_copy_stream(st + (size_t)i * d, tb + (size_t)rows[i] * d, (size_t)d);
            }
            row_copy_fence();
            ready[(size_t)pc].store(1, std::memory_order_release);
          }
        };
        const int helpers = (int)std::min<int64_t>(in_pool->size(), npieces);
        in_pool->start(helpers, helpers, work);
        int64_t cursor = 0;
        int idle = 0;
        while (cursor < npieces) {
          int64_t k = 0;
          while (cursor + k < npieces && ready[(size_t)(cursor + k)].load(std::memory_order_acquire)) ++k;
          if (k >= kCopyPieces || (k > 0 && cursor + k == npieces)) {
            const int64_t lo = cursor * kPiece, hi = std::min<int64_t>(n, (cursor + k) * kPiece);
            e = hipMemcpyAsync(dv + (size_t)lo * d, st + (size_t)lo * d, (size_t)(hi - lo) * d * 4,
                               hipMemcpyHostToDevice, cs);
            if (e != hipSuccess) { fail("hipMemcpyAsync(H2D)", e); break; }
            cursor += k;
            idle = 0;
          } else if (++idle < 64) {
            std::this_thread::yield();
          } else {
            std::this_thread::sleep_for(std::chrono::microseconds(10));
          }
        }
        in_pool->wait();
      }
      const auto tg = std::chrono::steady_clock::now();
      // Wait for the copies on the host (their completion signals: no packet goes through a hardware queue), then
      // release the cache-op stream with a plain store to the pinned word it polls.  Nothing here may depend on a
      // GPU queue making progress: HIP multiplexes streams onto a few hardware queues (4 by default), so the copy
      // stream can share one with the parked stream -- a hipStreamWriteValue64 / event marker queued behind the
      // parked wait would never execute (seen as a hang of the full test suite).
      CE_TRACE("in job %lld: rows gathered, copies enqueued", job);
      e = stream_wait_polite(in_stream);
      if (e != hipSuccess) fail(e == hipErrorNotReady ? "admission timed out (CE_WORKER_TIMEOUT_S)" : "waiting for the H2D copies", e);
      // a job that did not bring its rows in is flagged BEFORE the stream is released: k_unpack_admitted /
      // k_admit_maps then admit nothing and the call's record says CE_ERR_HIP
      if (n > 0 && failed()) __atomic_store_n(sig + 1, (unsigned long long)job, __ATOMIC_RELEASE);
      __atomic_store_n(sig, (unsigned long long)job, __ATOMIC_RELEASE);
      CE_TRACE("in job %lld: released the stream (%s)", job, hipGetErrorString(e));
      const auto t2 = std::chrono::steady_clock::now();
      {
        std::lock_guard<std::mutex> g(m);
        in_done = job;
        in_wait_s += std::chrono::duration<double>(t1 - t0).count();
        in_busy_s += std::chrono::duration<double>(t2 - t1).count();
        in_gather_s += std::chrono::duration<double>(tg - t1).count();
        in_rows += n;
        in_jobs += 1;
      }
      cv_done.notify_all();
    }
  }
  long long out_issued_at[8] = {0};      // write-back jobs that must have landed before in-job j gathers
  hipStream_t tested_stream = nullptr;   // cache-op stream the self-test below has passed on
  bool tested = false;

  int check() {
    std::lock_guard<std::mutex> g(m);
    if (err) {
      set_error("%s", errmsg);
      return err;
    }
    return CE_OK;
  }
  // The launch thread never waits for a worker without a deadline: a worker stuck in the runtime (GPU hang, a copy
  // that never starts) would otherwise block the caller for ever with the cache-op stream parked.  On a timeout the
  // pending admission is flagged lost and the stream released from here.
  int give_up(const char* what) {
    {
      std::lock_guard<std::mutex> g(m);
      if (!err) {
        err = CE_ERR_HIP;
        snprintf(errmsg, sizeof errmsg, "swap worker: %s did not finish within %.0f s (CE_WORKER_TIMEOUT_S)", what,
                 2 * worker_timeout_s());
      }
    }
    if (sig) {
      __atomic_store_n(sig + 1, (unsigned long long)in_issued, __ATOMIC_RELEASE);
      __atomic_store_n(sig, ~0ull >> 1, __ATOMIC_RELEASE);
    }
    return check();
  }
  int wait_out(long long upto) {       // blocks until write-back job `upto` has reached the host table
    bool ok;
    {
      std::unique_lock<std::mutex> g(m);
      ok = cv_done.wait_for(g, std::chrono::duration<double>(2 * worker_timeout_s()),
                            [&] { return out_done >= upto || out_done >= out_issued; });
    }
    return ok ? check() : give_up("a write-back job");
  }
  int wait_in(long long upto) {
    bool ok;
    {
      std::unique_lock<std::mutex> g(m);
      ok = cv_done.wait_for(g, std::chrono::duration<double>(2 * worker_timeout_s()),
                            [&] { return in_done >= upto || in_done >= in_issued; });
    }
    return ok ? check() : give_up("an admission job");
  }
  void push_out() {
    {
      std::lock_guard<std::mutex> g(m);
      ++out_issued;
      CE_TRACE("push out job %lld", out_issued);
    }
    cv_job.notify_all();
  }
  void push_in(long long need_out) {
    {
      std::lock_guard<std::mutex> g(m);
      ++in_issued;
      out_issued_at[in_issued & 7] = need_out;
      CE_TRACE("push in job %lld (needs out %lld)", in_issued, need_out);
    }
    cv_job.notify_all();
  }

  ~SwapEngine() {
    {
      std::lock_guard<std::mutex> g(m);
      stop = true;
    }
    if (const char* e = getenv("CE_WORKER_PROFILE"))
      if (atoi(e) != 0 && in_jobs > 0)
        fprintf(stderr, "[libce_hip] admission worker: %lld jobs, %lld of them with the previous write-back still on its "
                "way (rows it evicted taken from its staging buffer)\n", in_jobs, in_probed);
    if (const char* e = getenv("CE_WORKER_PROFILE"))
      if (atoi(e) != 0 && out_jobs > 0)
        fprintf(stderr, "[libce_hip] write-back worker: %lld jobs, %.3f ms busy per job = %.3f waiting for copies + %.3f "
                "scattering + %.3f other; %.0f rows per job; first chunk's wait %.3f\n", out_jobs, out_busy_s / out_jobs * 1e3,
                out_copy_wait_s / out_jobs * 1e3, out_scatter_s / out_jobs * 1e3,
                (out_busy_s - out_copy_wait_s - out_scatter_s) / out_jobs * 1e3, (double)out_rows / out_jobs,
                out_wait0_s / out_jobs * 1e3);
    cv_job.notify_all();
    if (in_thread.joinable()) in_thread.join();
    if (out_thread.joinable()) out_thread.join();
    delete out_pool;
    delete in_pool;
    for (int b = 0; b < 2; ++b) {
      if (out_ev[b]) (void)hipEventDestroy(out_ev[b]);
      if (rows_host[b]) (void)hipHostFree(rows_host[b]);
      if (idx_host[b]) (void)hipHostFree(idx_host[b]);
    }
    for (int b = 0; b < 2; ++b)
      if (in_ev[b]) (void)hipEventDestroy(in_ev[b]);
    for (auto& ev : ev_miss)
      if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : ev_rows)
      if (ev) (void)hipEventDestroy(ev);

    for (auto& ev : chunk_ev)
      if (ev) (void)hipEventDestroy(ev);
    if (in_host) (void)hipHostFree(in_host);
    if (miss_host) (void)hipHostFree(miss_host);
    if (sig) (void)hipHostFree(sig);
    if (mail) (void)hipHostFree(mail);
    if (out_stream) (void)hipStreamDestroy(out_stream);
    if (out_stream2) (void)hipStreamDestroy(out_stream2);
    if (in_stream) (void)hipStreamDestroy(in_stream);
  }
};

}  // namespace ce
