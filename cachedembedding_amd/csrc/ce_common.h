// Shared helpers for the HIP sources of libce_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "ce_api.h"

namespace ce {

void set_error(const char* fmt, ...);

#define CE_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      ce::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return CE_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

#define CE_REQUIRE(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      ce::set_error(__VA_ARGS__);    \
      return (code);                 \
    }                                \
  } while (0)

#define CE_LAUNCH_CHECK() CE_HIP_CHECK(hipGetLastError())

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <typename VT> __device__ __forceinline__ VT vzero();
template <> __device__ __forceinline__ float vzero<float>() { return 0.f; }
template <> __device__ __forceinline__ f32x4 vzero<f32x4>() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// MI355X: 256 CUs; memory-bound grid-stride kernels are capped at 8 blocks of 256 per CU.
constexpr int kNumCU = 256;
constexpr int kMaxBlocks = kNumCU * 8;

static inline int grid_for(int64_t work_items, int per_block) {
  int64_t b = cdiv(work_items, per_block);
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// ce_bag.hip, for the cache manager (ce_cache.hip): the window presort with the cache op's last step folded in.
// slots_io holds the ROW of every id (what k_mark left there; -1 = no lookup); the kernel turns it into the slot in
// place (inverted[row]; -1 everywhere when *status != CE_OK) and writes the window's keys in the same pass -- instead
// of k_slots writing 8 bytes per id that the presort reads straight back.  lay_* as ce_bag_presort_window_src;
// src_keys == 0: keys = row << 32 | lookup in segment (ce_bag_presort_window).
int presort_window_from_rows(int64_t* slots_io, int64_t nnz_per_batch, int64_t n_batches, int64_t num_rows,
                             const int32_t* inverted, const int* status, int32_t src_keys, const void* offsets,
                             int32_t offsets_are_i64, int64_t offsets_batch_stride, int64_t num_bags,
                             int32_t include_last_offset, int64_t hook_features, uint64_t* keys_out, hipStream_t stream);

// ce_host.hip: pins the CALLING thread to the CPUs of the current GPU's NUMA node (sysfs local_cpulist of its PCI
// function, intersected with the CPUs the process may use); no-op when that cannot be read or CE_NUMA_BIND=0.  For
// the library's own threads only -- the swap workers, their helpers and the first touch of ce_host_alloc -- so that
// the host table and the staging buffers sit behind the GPU's own root complex.
void bind_thread_near_gpu();
// the same CPU set for threads that have not selected the device themselves (taken by their creator); false = none
bool near_gpu_cpus(cpu_set_t* out);

// lanes cooperating on one embedding row: 16 B per lane, power of two, at most one wave
static inline int group_lanes_for_dim(int dim) {
  int v = (dim + 3) / 4;
  int g = 1;
  while (g < v && g < 64) g <<= 1;
  return g;
}

}  // namespace ce
