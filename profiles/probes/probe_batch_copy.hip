// Can the copy engines gather the missed rows?  hipMemcpyBatchAsync with one 512-byte copy per row (55 k rows out of a
// pinned host table into contiguous HBM staging), alone and beside an HBM-bound kernel; for comparison the same rows as
// ONE contiguous hipMemcpyAsync.  DESIGN.md section 8 item 2 named this as unmeasured.
// hipcc --offload-arch=gfx950 -O2 -o probe_batch_copy profiles/probes/probe_batch_copy.hip && ./probe_batch_copy
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_stream(const float4* a, float4* b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main() {
  const size_t row = 512, table_rows = 8u << 20, n = 55000;     // 4 GiB pinned table
  char* table;
  CK(hipHostMalloc((void**)&table, table_rows * row, hipHostMallocDefault));
  char* stage;
  CK(hipMalloc((void**)&stage, n * row));
  std::vector<void*> dst(n), src(n);
  std::vector<size_t> sz(n, row);
  srand(1);
  for (size_t i = 0; i < n; ++i) {
    dst[i] = stage + i * row;
    src[i] = table + ((((size_t)rand() << 15) ^ (size_t)rand()) % table_rows) * row;
  }
  hipStream_t s, s2;
  CK(hipStreamCreate(&s));
  CK(hipStreamCreate(&s2));
  float4 *a, *b;
  const size_t big = (size_t)1 << 28;       // 256 MiB each
  CK(hipMalloc((void**)&a, big));
  CK(hipMalloc((void**)&b, big));
  hipEvent_t e0, e1, k0, k1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1));
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, s));
    CK(hipMemcpyAsync(stage, table, n * row, hipMemcpyHostToDevice, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("one contiguous copy of %zu x %zu B: %.3f ms = %.1f GB/s\n", n, row, ms, n * row / ms / 1e6);
  }
  for (size_t cnt : {(size_t)1000, (size_t)10000, n}) {
    for (int rep = 0; rep < 2; ++rep) {
      size_t fail = 0;
      const auto t0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(e0, s));
      hipError_t e = hipMemcpyBatchAsync(dst.data(), src.data(), sz.data(), cnt, nullptr, nullptr, 0, &fail, s);
      if (e != hipSuccess) { printf("hipMemcpyBatchAsync: %s (fail index %zu)\n", hipGetErrorString(e), fail); return 0; }
      CK(hipEventRecord(e1, s));
      const auto t1 = std::chrono::steady_clock::now();
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("batch of %zu row copies: enqueue %.3f ms on the host, %.3f ms on the device = %.2f GB/s\n", cnt,
             std::chrono::duration<double, std::milli>(t1 - t0).count(), ms, cnt * row / ms / 1e6);
    }
  }
  // beside an HBM-bound kernel
  for (int with = 0; with < 2; ++with) {
    size_t fail = 0;
    CK(hipDeviceSynchronize());
    if (with) CK(hipMemcpyBatchAsync(dst.data(), src.data(), sz.data(), n, nullptr, nullptr, 0, &fail, s));
    CK(hipEventRecord(k0, s2));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s2, a, b, big / 16);
    CK(hipEventRecord(k1, s2));
    CK(hipEventSynchronize(k1));
    CK(hipEventElapsedTime(&ms, k0, k1));
    printf("20 x 256 MiB stream copies %s the batch: %.3f ms\n", with ? "beside" : "without", ms);
  }
  return 0;
}
