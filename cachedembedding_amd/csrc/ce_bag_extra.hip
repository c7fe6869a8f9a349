// The F.embedding_bag arguments the reference forwards (recsys/models/dlrm.py:99-110 -> upstream A.7) but none of its
// scripts sets: mode='max', the gradient w.r.t. per_sample_weights, max_norm / norm_type.  They are off the benchmarked
// path; the kernels here are plain (any dim, fp32, one wave per bag / lookup / row) and bandwidth-bound like the rest.
#include <algorithm>

#include "ce_common.h"

namespace ce {
namespace {

struct XParams {
  const int64_t* indices;
  const void* offsets;
  int64_t nnz;
  int32_t num_bags, dim, off64, include_last, hookF, hookB;
  uint32_t num_rows;
};

__device__ __forceinline__ int x_off(const XParams& p, int i) {
  return p.off64 ? (int)((const int64_t*)p.offsets)[i] : ((const int32_t*)p.offsets)[i];
}
__device__ __forceinline__ int x_end(const XParams& p, int b) {
  return (p.include_last || b + 1 < p.num_bags) ? x_off(p, b + 1) : (int)p.nnz;
}
// row of the [B, F, D] output that feature-major bag g = f * B + b lands in (hook_features > 0)
__device__ __forceinline__ int64_t x_out_row(const XParams& p, int g) {
  if (p.hookF == 0) return g;
  const int f = g / p.hookB;
  return (int64_t)(g - f * p.hookB) * p.hookF + f;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  return v;
}

// ---- mode = 'max': out[bag][d] = max over the bag's rows; max_pos[bag][d] = the lookup that supplied it (-1: none).
// One wave per bag, lanes over d.  The first of equal maxima wins, as in torch's CPU kernel.
__global__ __launch_bounds__(256) void k_bag_fwd_max(XParams p, const float* __restrict__ W, float* __restrict__ out,
                                                     int32_t* __restrict__ max_pos) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t b = wave; b < p.num_bags; b += nwaves) {
    const int lo = x_off(p, (int)b), hi = x_end(p, (int)b);
    const int64_t orow = x_out_row(p, (int)b);
    for (int d0 = 0; d0 < p.dim; d0 += 64) {
      const int d = d0 + lane;
      float best = 0.f;
      int pos = -1;
      for (int j = lo; j < hi; ++j) {
        const int64_t ri = p.indices[j];
        if ((uint64_t)ri >= (uint64_t)p.num_rows || d >= p.dim) continue;
        const float v = W[ri * p.dim + d];
        // NaN: exactly torch's CPU kernel -- the bag's first row is taken as it is, a later row only if `v > best`;
        // so a NaN in the first row stays, a NaN further on never wins (tests/test_gpu_bag.py pins this against torch)
        if (pos < 0 || v > best) { best = v; pos = j; }
      }
      if (d < p.dim) {
        out[orow * p.dim + d] = best;
        if (max_pos) max_pos[b * p.dim + d] = pos;
      }
    }
  }
}

// dst[indices[max_pos[bag][d]]][d] += alpha * grad_out[bag][d]   (grad_weight: alpha 1; fused SGD: alpha -lr)
__global__ __launch_bounds__(256) void k_bag_bwd_max(XParams p, float* dst, const float* __restrict__ grad_out,
                                                     const int32_t* __restrict__ max_pos, float alpha) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t b = wave; b < p.num_bags; b += nwaves) {
    const int64_t orow = x_out_row(p, (int)b);
    for (int d = lane; d < p.dim; d += 64) {
      const int pos = max_pos[b * p.dim + d];
      if (pos < 0) continue;
      const int64_t ri = p.indices[pos];
      if ((uint64_t)ri >= (uint64_t)p.num_rows) continue;
      atomicAdd(dst + ri * p.dim + d, alpha * grad_out[orow * p.dim + d]);
    }
  }
}

// ---- d loss / d per_sample_weights[j] = < grad_out[bag of j], weight[indices[j]] >   (mode 'sum').  One wave per bag
// walks its lookups (the bag's gradient row stays in registers).
__global__ __launch_bounds__(256) void k_bag_bwd_psw(XParams p, const float* __restrict__ W,
                                                     const float* __restrict__ grad_out, float* __restrict__ grad_psw) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t b = wave; b < p.num_bags; b += nwaves) {
    const int lo = x_off(p, (int)b), hi = x_end(p, (int)b);
    const int64_t orow = x_out_row(p, (int)b);
    for (int j = lo; j < hi; ++j) {
      const int64_t ri = p.indices[j];
      float acc = 0.f;
      if ((uint64_t)ri < (uint64_t)p.num_rows)
        for (int d = lane; d < p.dim; d += 64) acc += grad_out[orow * p.dim + d] * W[ri * p.dim + d];
      acc = wave_sum_f(acc);
      if (lane == 0) grad_psw[j] = acc;
    }
  }
}

// ---- max_norm: every row an index names is scaled to norm max_norm if its p-norm exceeds it, ONCE however often it
// is named (torch.embedding_renorm_ works on the unique indices): the named rows are marked in a bitmap, then one wave
// per marked row does the norm and the scaling.
__global__ __launch_bounds__(256) void k_renorm_mark(const int64_t* __restrict__ indices, int64_t n, uint32_t num_rows,
                                                     uint32_t* bitmap) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t r = indices[i];
    if ((uint64_t)r < (uint64_t)num_rows) atomicOr(bitmap + (r >> 5), 1u << (r & 31));
  }
}

__global__ __launch_bounds__(256) void k_renorm_rows(float* W, uint32_t num_rows, int dim, const uint32_t* bitmap,
                                                     float max_norm, float norm_type) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  const int64_t nwords = ((int64_t)num_rows + 31) >> 5;
  const bool p2 = norm_type == 2.f, p1 = norm_type == 1.f, pinf = isinf(norm_type);
  for (int64_t w0 = wave * 64; w0 < nwords; w0 += nwaves * 64) {
    const uint32_t mine = w0 + lane < nwords ? bitmap[w0 + lane] : 0u;
    unsigned long long any = __ballot(mine != 0);
    while (any) {
      const int src = __ffsll((long long)any) - 1;
      any &= any - 1;
      uint32_t bits = __shfl(mine, src);
      while (bits) {
        const int bit = __ffs(bits) - 1;
        bits &= bits - 1;
        const int64_t r = (w0 + src) * 32 + bit;
        float* row = W + r * dim;
        float acc = 0.f;
        for (int d = lane; d < dim; d += 64) {
          const float a = fabsf(row[d]);
          acc = pinf ? fmaxf(acc, a) : acc + (p2 ? a * a : (p1 ? a : powf(a, norm_type)));
        }
        acc = pinf ? wave_max_f(acc) : wave_sum_f(acc);
        const float norm = pinf ? acc : (p2 ? sqrtf(acc) : (p1 ? acc : powf(acc, 1.f / norm_type)));
        if (norm > max_norm) {
          const float scale = max_norm / (norm + 1e-7f);
          for (int d = lane; d < dim; d += 64) row[d] *= scale;
        }
      }
    }
  }
}

int fill_x(XParams& p, int64_t num_rows, int32_t dim, const int64_t* indices, int64_t nnz, const void* offsets,
           int32_t off64, int64_t num_bags, int32_t include_last, int64_t hookF) {
  CE_REQUIRE(dim > 0 && num_rows > 0 && num_rows < (int64_t)INT32_MAX, CE_ERR_INVALID, "bad table shape");
  CE_REQUIRE(num_bags >= 0 && nnz >= 0 && num_bags < (int64_t)INT32_MAX - 64 && nnz < (int64_t)INT32_MAX,
             CE_ERR_UNSUPPORTED, "more than 2^31 bags / lookups in one launch");
  CE_REQUIRE(hookF >= 0 && (hookF == 0 || num_bags % hookF == 0), CE_ERR_INVALID, "hook_features must divide num_bags");
  p.indices = indices;
  p.offsets = offsets;
  p.nnz = nnz;
  p.num_bags = (int32_t)num_bags;
  p.dim = dim;
  p.off64 = off64;
  p.include_last = include_last;
  p.hookF = (int32_t)hookF;
  p.hookB = hookF ? (int32_t)(num_bags / hookF) : 0;
  p.num_rows = (uint32_t)num_rows;
  return CE_OK;
}

int wave_grid(int64_t items) { return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(items, 4), 16384)); }

}  // namespace
}  // namespace ce

using namespace ce;

extern "C" int ce_bag_forward_max(const float* weight, int64_t num_rows, int32_t dim, const int64_t* indices,
                                  int64_t nnz, const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                                  int32_t include_last_offset, int64_t hook_features, float* out, int32_t* max_pos,
                                  ce_stream_t stream) {
  if (num_bags == 0) return CE_OK;
  CE_REQUIRE(weight && out && offsets && (indices || nnz == 0), CE_ERR_INVALID, "null pointer");
  XParams p{};
  const int rc = fill_x(p, num_rows, dim, indices, nnz, offsets, offsets_are_i64, num_bags, include_last_offset,
                        hook_features);
  if (rc) return rc;
  hipLaunchKernelGGL(k_bag_fwd_max, dim3(wave_grid(num_bags)), dim3(256), 0, (hipStream_t)stream, p, weight, out,
                     max_pos);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int ce_bag_backward_max(float* dst, int64_t num_rows, int32_t dim, const int64_t* indices, int64_t nnz,
                                   int64_t num_bags, int64_t hook_features, const float* grad_out,
                                   const int32_t* max_pos, float alpha, ce_stream_t stream) {
  if (num_bags == 0 || nnz == 0) return CE_OK;
  CE_REQUIRE(dst && indices && grad_out && max_pos, CE_ERR_INVALID, "null pointer");
  XParams p{};
  const int rc = fill_x(p, num_rows, dim, indices, nnz, nullptr, 0, num_bags, 1, hook_features);
  if (rc) return rc;
  hipLaunchKernelGGL(k_bag_bwd_max, dim3(wave_grid(num_bags)), dim3(256), 0, (hipStream_t)stream, p, dst, grad_out,
                     max_pos, alpha);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" int ce_bag_backward_psw(const float* weight, int64_t num_rows, int32_t dim, const int64_t* indices,
                                   int64_t nnz, const void* offsets, int32_t offsets_are_i64, int64_t num_bags,
                                   int32_t include_last_offset, int64_t hook_features, const float* grad_out,
                                   float* grad_psw, ce_stream_t stream) {
  if (num_bags == 0 || nnz == 0) return CE_OK;
  CE_REQUIRE(weight && indices && offsets && grad_out && grad_psw, CE_ERR_INVALID, "null pointer");
  XParams p{};
  const int rc = fill_x(p, num_rows, dim, indices, nnz, offsets, offsets_are_i64, num_bags, include_last_offset,
                        hook_features);
  if (rc) return rc;
  hipLaunchKernelGGL(k_bag_bwd_psw, dim3(wave_grid(num_bags)), dim3(256), 0, (hipStream_t)stream, p, weight, grad_out,
                     grad_psw);
  CE_LAUNCH_CHECK();
  return CE_OK;
}

extern "C" size_t ce_rows_renorm_workspace(int64_t num_rows) {
  return num_rows > 0 ? (size_t)((num_rows + 31) / 32) * 4 : 0;
}

extern "C" int ce_rows_renorm(float* weight, int64_t num_rows, int32_t dim, const int64_t* indices, int64_t n,
                              float max_norm, float norm_type, void* workspace, size_t workspace_bytes,
                              ce_stream_t stream) {
  if (n == 0) return CE_OK;
  CE_REQUIRE(weight && indices && workspace, CE_ERR_INVALID, "null pointer");
  CE_REQUIRE(dim > 0 && num_rows > 0 && num_rows < (int64_t)INT32_MAX && n > 0, CE_ERR_INVALID, "bad sizes");
  CE_REQUIRE(max_norm >= 0.f && norm_type > 0.f, CE_ERR_INVALID, "max_norm must be >= 0 and norm_type > 0");
  CE_REQUIRE(workspace_bytes >= ce_rows_renorm_workspace(num_rows), CE_ERR_INVALID, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  CE_HIP_CHECK(hipMemsetAsync(workspace, 0, ce_rows_renorm_workspace(num_rows), s));
  hipLaunchKernelGGL(k_renorm_mark, dim3(grid_for(n, 256)), dim3(256), 0, s, indices, n, (uint32_t)num_rows,
                     (uint32_t*)workspace);
  const int64_t nwords = (num_rows + 31) / 32;
  hipLaunchKernelGGL(k_renorm_rows, dim3(wave_grid(cdiv(nwords, 64))), dim3(256), 0, s, weight, (uint32_t)num_rows, dim,
                     (const uint32_t*)workspace, max_norm, norm_type);
  CE_LAUNCH_CHECK();
  return CE_OK;
}
