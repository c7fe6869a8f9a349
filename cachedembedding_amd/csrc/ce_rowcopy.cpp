// Host-side row copy for the swap workers of libce_hip (plain C++: x86 intrinsics stay out of the HIP passes).
//
// A worker helper moves 512-byte rows between the host table (random row addresses) and contiguous pinned staging.
// Measured on the bench box's EPYC 9575F (profiles/probes/probe_scatter.cpp: 54 k rows, 6 threads): into the TABLE a plain
// memcpy pays a read-for-ownership miss per destination line (0.81 ms) and 16-byte non-temporal stores are no better
// (0.84 ms); 64-byte (AVX-512) / 32-byte (AVX2) non-temporal stores write whole lines without reading them (0.61 ms).
// Into the STAGING buffer (sequential) memcpy and 64-byte non-temporal stores tie (0.42 ms), 16-byte ones lose (0.54).
// So: the widest streaming store the CPU has, chosen once at load time; memcpy when there is none or the pointers are
// not 64-byte aligned.
#include <immintrin.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

namespace ce {

typedef void (*row_copy_fn)(float*, const float*, size_t);

__attribute__((target("avx512f"))) static void copy_nt64(float* d, const float* s, size_t n) {
  size_t i = 0;
  for (; i + 16 <= n; i += 16) _mm512_stream_ps(d + i, _mm512_loadu_ps(s + i));
  if (i < n) memcpy(d + i, s + i, (n - i) * sizeof(float));
}

__attribute__((target("avx2"))) static void copy_nt32(float* d, const float* s, size_t n) {
  size_t i = 0;
  for (; i + 8 <= n; i += 8) _mm256_stream_ps(d + i, _mm256_loadu_ps(s + i));
  if (i < n) memcpy(d + i, s + i, (n - i) * sizeof(float));
}

static void copy_plain(float* d, const float* s, size_t n) { memcpy(d, s, n * sizeof(float)); }

static row_copy_fn pick() {
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx512f")) return copy_nt64;
  if (__builtin_cpu_supports("avx2")) return copy_nt32;
  return copy_plain;
}

static const row_copy_fn g_stream = pick();

// dst is written whole and not read again by this core: stream it past the cache when it is line-aligned
void row_copy_stream(float* dst, const float* src, size_t floats) {
  if ((((uintptr_t)dst) & 63) == 0) g_stream(dst, src, floats);
  else copy_plain(dst, src, floats);
}

void row_copy_fence() { _mm_sfence(); }      // streaming stores are weakly ordered: fence before publishing

const char* row_copy_kind() { return g_stream == copy_nt64 ? "avx512 nt" : (g_stream == copy_nt32 ? "avx2 nt" : "memcpy"); }

}  // namespace ce
