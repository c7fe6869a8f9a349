// PCIe duplex probe for the row swap (round 5).  The worker transport moves both directions of a
// window's swap at once: the WRITE-BACK of ~55 k evicted rows as pinned hipMemcpyAsync device-to-host copies (SDMA, 8
// chunks on two streams, then a host scatter) and the ADMISSION of ~55 k missed rows by a small kernel that reads 512-byte
// rows at random places of the mapped pinned table (20 workgroups x 1024 threads, R rows in flight per 32-lane group).
// What can the link do when both run together?  Measured here, each alone and both at once, at the job sizes of the
// bench (28 MB each way); plus the opposite engine assignment (a contiguous SDMA host-to-device copy beside a kernel that
// SCATTERS rows into the host table), and the admission kernel at several depths / widths (rows in flight are what
// occupies the L2's miss queues for a PCIe round trip each).
//
//   hipcc --offload-arch=gfx950 -O3 probe_pcie_duplex.hip -o probe_pcie_duplex && ./probe_pcie_duplex [table_GB]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int D4 = 32;   // 128 floats per row = 32 x 16 bytes: one row per 32-lane group

// rows[i] of the host table -> dst[i] (the admission kernel's shape: admit_rows in csrc/ce_cache.hip)
template <int R>
__global__ __launch_bounds__(1024) void k_read_rows(const int32_t* __restrict__ rows, int64_t n, const f32x4* __restrict__ host,
                                                    f32x4* __restrict__ dst) {
  const int gl = threadIdx.x & 31;
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> 5;
  for (int64_t i = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5) * R; i < n; i += gstride * R) {
    f32x4 v[R];
#pragma unroll
    for (int t = 0; t < R; ++t)
      if (i + t < n) v[t] = host[(int64_t)rows[i + t] * D4 + gl];
#pragma unroll
    for (int t = 0; t < R; ++t)
      if (i + t < n) dst[(i + t) * D4 + gl] = v[t];
  }
}

// src[i] -> rows[i] of the host table (the zero-copy write-back's shape)
template <int R>
__global__ __launch_bounds__(1024) void k_write_rows(const int32_t* __restrict__ rows, int64_t n, const f32x4* __restrict__ src,
                                                     f32x4* __restrict__ host) {
  const int gl = threadIdx.x & 31;
  const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) >> 5;
  for (int64_t i = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5) * R; i < n; i += gstride * R) {
    f32x4 v[R];
#pragma unroll
    for (int t = 0; t < R; ++t)
      if (i + t < n) v[t] = src[(i + t) * D4 + gl];
#pragma unroll
    for (int t = 0; t < R; ++t)
      if (i + t < n) host[(int64_t)rows[i + t] * D4 + gl] = v[t];
  }
}

// an HBM-bound neighbour: what the bag kernels are to the swap
__global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ a, f32x4* __restrict__ b, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(a[i], b + i);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const double table_gb = argc > 1 ? atof(argv[1]) : 8.0;
  const int64_t N = (int64_t)(table_gb * 1e9 / 512), n = 55000;
  const size_t row_b = 512, job = n * row_b;
  float* table;
  CK(hipHostMalloc((void**)&table, (size_t)N * row_b, hipHostMallocMapped | hipHostMallocPortable));
  for (int64_t i = 0; i < N * 128; i += 1024) table[i] = 1.0f;          // touch the pages
  float* table_dev;
  CK(hipHostGetDevicePointer((void**)&table_dev, table, 0));
  std::mt19937_64 rng(7);
  std::vector<int32_t> rows_h(n);
  for (auto& r : rows_h) r = (int32_t)(rng() % N);
  std::sort(rows_h.begin(), rows_h.end());                              // the miss list is ascending
  int32_t* rows;
  CK(hipMalloc(&rows, n * 4));
  CK(hipMemcpy(rows, rows_h.data(), n * 4, hipMemcpyHostToDevice));
  f32x4 *stage_in, *stage_out;
  CK(hipMalloc(&stage_in, job));
  CK(hipMalloc(&stage_out, job));
  CK(hipMemset(stage_out, 0, job));
  float* land;                                                          // pinned landing buffer of the write-back
  CK(hipHostMalloc((void**)&land, job, hipHostMallocPortable));
  const int64_t big4 = (int64_t)(1 << 30) / 16;
  f32x4 *ba, *bb;
  CK(hipMalloc(&ba, (size_t)big4 * 16));
  CK(hipMalloc(&bb, (size_t)big4 * 16));
  CK(hipMemset(ba, 0, (size_t)big4 * 16));
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s_in, s_out[2], s_main;
  CK(hipStreamCreateWithPriority(&s_in, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&s_out[0], hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&s_out[1], hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));

  auto d2h = [&](int chunks) {              // the write-back's copies: `chunks` pieces on two streams in turn
    const int64_t per = (n + chunks - 1) / chunks;
    int c = 0;
    for (int64_t off = 0; off < n; off += per, ++c) {
      const int64_t cnt = std::min<int64_t>(per, n - off);
      CK(hipMemcpyAsync((char*)land + off * row_b, (char*)stage_out + off * row_b, cnt * row_b, hipMemcpyDeviceToHost, s_out[c & 1]));
    }
  };
  auto h2d_sdma = [&]() { CK(hipMemcpyAsync(stage_in, land, job, hipMemcpyHostToDevice, s_in)); };
  auto admit = [&](int blocks, int R) {
    if (R == 16) hipLaunchKernelGGL((k_read_rows<16>), dim3(blocks), dim3(1024), 0, s_in, rows, n, (const f32x4*)table_dev, stage_in);
    else if (R == 8) hipLaunchKernelGGL((k_read_rows<8>), dim3(blocks), dim3(1024), 0, s_in, rows, n, (const f32x4*)table_dev, stage_in);
    else if (R == 4) hipLaunchKernelGGL((k_read_rows<4>), dim3(blocks), dim3(1024), 0, s_in, rows, n, (const f32x4*)table_dev, stage_in);
    else if (R == 2) hipLaunchKernelGGL((k_read_rows<2>), dim3(blocks), dim3(1024), 0, s_in, rows, n, (const f32x4*)table_dev, stage_in);
    else hipLaunchKernelGGL((k_read_rows<1>), dim3(blocks), dim3(1024), 0, s_in, rows, n, (const f32x4*)table_dev, stage_in);
  };
  auto wb_kernel = [&](int blocks) {
    hipLaunchKernelGGL((k_write_rows<16>), dim3(blocks), dim3(1024), 0, s_out[0], rows, n, (const f32x4*)stage_out, (f32x4*)table_dev);
  };
  auto stream_kernel = [&](int reps) {
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s_main, (const f32x4*)ba, bb, big4);
  };
  auto sync_all = [&]() { CK(hipDeviceSynchronize()); };
  auto timeit = [&](const char* what, int reps, auto fn, double bytes_in, double bytes_out) {
    fn();
    sync_all();
    const double t0 = now();
    for (int r = 0; r < reps; ++r) { fn(); sync_all(); }
    const double dt = (now() - t0) / reps;
    printf("%-74s %8.3f ms", what, dt * 1e3);
    if (bytes_in > 0) printf("   in %6.1f GB/s", bytes_in / dt / 1e9);
    if (bytes_out > 0) printf("   out %6.1f GB/s", bytes_out / dt / 1e9);
    if (bytes_in > 0 && bytes_out > 0) printf("   both %6.1f GB/s", (bytes_in + bytes_out) / dt / 1e9);
    printf("\n");
    return dt;
  };
  printf("# PCIe duplex probe: %lld rows of 512 B per job and direction (%.1f MB), host table %.1f GB pinned + mapped\n",
         (long long)n, job / 1e6, table_gb);
  const int R_ = 20;
  timeit("write-back: SDMA D2H, 8 chunks on two streams, alone", R_, [&] { d2h(8); }, 0, job);
  timeit("write-back: SDMA D2H, 1 chunk, alone", R_, [&] { d2h(1); }, 0, job);
  timeit("admission: kernel 20 x 1024, 16 rows in flight per group, alone", R_, [&] { admit(20, 16); }, job, 0);
  for (int R : {1, 2, 4, 8})
    for (int blocks : {20, 40}) {
      char buf[128];
      snprintf(buf, sizeof buf, "admission: kernel %d x 1024, %d rows in flight per group, alone", blocks, R);
      timeit(buf, R_, [&] { admit(blocks, R); }, job, 0);
    }
  timeit("BOTH (the worker transport): SDMA D2H 8 chunks + admission kernel 20 x 1024 x 16", R_, [&] { d2h(8); admit(20, 16); }, job, job);
  timeit("BOTH: SDMA D2H 8 chunks + admission kernel 20 x 1024 x 4", R_, [&] { d2h(8); admit(20, 4); }, job, job);
  timeit("BOTH: SDMA D2H 8 chunks + admission kernel 40 x 1024 x 2", R_, [&] { d2h(8); admit(40, 2); }, job, job);
  timeit("opposite engines: SDMA H2D (contiguous 28 MB), alone", R_, [&] { h2d_sdma(); }, job, 0);
  timeit("opposite engines: write-back kernel 16 x 1024 scattering rows into the host table, alone", R_, [&] { wb_kernel(16); }, 0, job);
  timeit("BOTH, opposite engines: SDMA H2D contiguous + write-back kernel 16 x 1024", R_, [&] { h2d_sdma(); wb_kernel(16); }, job, job);
  timeit("BOTH, two SDMA copies: D2H 8 chunks + H2D contiguous", R_, [&] { d2h(8); h2d_sdma(); }, job, job);
  // ---- where the admission kernel's workgroups sit.  Its reads hold miss-queue entries of the L2 they pass through for a
  // PCIe round trip each; every XCD has its own L2, and a 20-workgroup grid on an unmasked stream lands on all eight.
  // hipExtStreamCreateWithCUMask: bit i of the mask = CU i; which XCD a bit belongs to is what the two patterns probe
  // (every 8th bit = one XCD if the driver deals the bits round-robin to the XCCs; the low 32 bits = one XCD if not).
  auto masked_stream = [&](int pattern) {
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pattern == 0) for (int i = 0; i < 256; i += 8) m[i / 32] |= 1u << (i % 32);        // every 8th CU: 32 CUs
    else if (pattern == 1) m[0] = 0xffffffffu;                                             // CUs 0..31
    else if (pattern == 2) for (int i = 0; i < 256; i += 4) m[i / 32] |= 1u << (i % 32);   // every 4th: 64 CUs
    else { m[0] = 0xffffffffu; m[1] = 0xffffffffu; }                                       // CUs 0..63
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, 8, m));
    return st;
  };
  const char* pat_name[4] = {"every 8th CU (32)", "CUs 0..31", "every 4th CU (64)", "CUs 0..63"};
  hipStream_t s_saved = s_in;
  for (int pat = 0; pat < 4; ++pat) {
    s_in = masked_stream(pat);
    char buf[160];
    snprintf(buf, sizeof buf, "admission 20 x 1024 x 16 on a stream masked to %s, alone", pat_name[pat]);
    timeit(buf, R_, [&] { admit(20, 16); }, job, 0);
    const double b0 = timeit("   HBM-bound neighbour alone (8 x copy of 1 GiB)", 5, [&] { stream_kernel(8); }, 0, 0);
    snprintf(buf, sizeof buf, "   neighbour beside admission kernels on the stream masked to %s", pat_name[pat]);
    const double d1 = timeit(buf, 5, [&] { for (int k = 0; k < 6; ++k) admit(20, 16); stream_kernel(8); }, 0, 0);
    printf("       -> x%.3f\n", d1 / b0);
    snprintf(buf, sizeof buf, "   neighbour beside 32 x 1024 x 16 admission kernels, masked to %s", pat_name[pat]);
    const double d2 = timeit(buf, 5, [&] { for (int k = 0; k < 6; ++k) admit(32, 16); stream_kernel(8); }, 0, 0);
    printf("       -> x%.3f\n", d2 / b0);
    CK(hipStreamDestroy(s_in));
  }
  s_in = s_saved;
  // both directions as KERNELS (no SDMA, no host scatter): write-back kernel + admission kernel, unmasked and masked
  timeit("BOTH as kernels: write-back 16 x 1024 + admission 20 x 1024 x 16, unmasked streams", R_, [&] { wb_kernel(16); admit(20, 16); }, job, job);
  {
    hipStream_t keep_in = s_in, keep_out = s_out[0];
    s_in = masked_stream(0);
    s_out[0] = masked_stream(0);
    timeit("BOTH as kernels, both streams masked to every 8th CU", R_, [&] { wb_kernel(16); admit(20, 16); }, job, job);
    const double b0 = timeit("   HBM-bound neighbour alone (8 x copy of 1 GiB)", 5, [&] { stream_kernel(8); }, 0, 0);
    const double d1 = timeit("   neighbour beside both kernels (masked), back to back for its duration", 5,
                             [&] { for (int k = 0; k < 5; ++k) { wb_kernel(16); admit(20, 16); } stream_kernel(8); }, 0, 0);
    printf("       -> x%.3f\n", d1 / b0);
    CK(hipStreamDestroy(s_in));
    CK(hipStreamDestroy(s_out[0]));
    s_in = keep_in;
    s_out[0] = keep_out;
  }
  // what the admission does to an HBM-bound neighbour (1 GiB copy, 8 launches), by rows in flight
  const double base = timeit("HBM-bound neighbour alone (8 x copy of 1 GiB)", 5, [&] { stream_kernel(8); }, 0, 0);
  for (int R : {16, 4, 2}) {
    char buf[160];
    snprintf(buf, sizeof buf, "neighbour beside admission kernels 20 x 1024 x %d (back to back for its duration)", R);
    const double dt = timeit(buf, 5, [&] { for (int k = 0; k < 6; ++k) admit(20, R); stream_kernel(8); }, 0, 0);
    printf("    -> x%.3f\n", dt / base);
  }
  return 0;
}
