// Probe helper: device memory through HIP's virtual-memory API, so that the ALIGNMENT of the virtual range and the
// size of the physical pieces behind it are the caller's choice (hipMalloc decides both itself).
// hipcc -shared -fPIC -O2 -o profiles/probes/libvmm_alloc.so profiles/probes/vmm_alloc.cpp
#include <hip/hip_runtime.h>
#include <cstdio>

extern "C" int vmm_granularity(size_t* out) {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  return (int)hipMemGetAllocationGranularity(out, &prop, hipMemAllocationGranularityRecommended);
}

// size bytes at a virtual address aligned to `align`, backed by physical pieces of `piece` bytes each (size % piece == 0)
extern "C" int vmm_alloc(size_t size, size_t align, size_t piece, void** out) {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  void* va = nullptr;
  hipError_t e = hipMemAddressReserve(&va, size, align, nullptr, 0);
  if (e != hipSuccess) { fprintf(stderr, "reserve: %s\n", hipGetErrorString(e)); return (int)e; }
  for (size_t off = 0; off < size; off += piece) {
    hipMemGenericAllocationHandle_t h;
    e = hipMemCreate(&h, piece, &prop, 0);
    if (e != hipSuccess) { fprintf(stderr, "create: %s\n", hipGetErrorString(e)); return (int)e; }
    e = hipMemMap((char*)va + off, piece, 0, h, 0);
    if (e != hipSuccess) { fprintf(stderr, "map: %s\n", hipGetErrorString(e)); return (int)e; }
    (void)hipMemRelease(h);          // (the mapping keeps the memory)
  }
  hipMemAccessDesc acc{};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = 0;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(va, size, &acc, 1);
  if (e != hipSuccess) { fprintf(stderr, "access: %s\n", hipGetErrorString(e)); return (int)e; }
  *out = va;
  return 0;
}

// the same physical pieces mapped at TWO virtual ranges (out[0], out[1]); the second one `shift` bytes into a larger
// reservation, so that the two ranges differ in more than their high bits
extern "C" int vmm_alloc_twice(size_t size, size_t piece, size_t shift, void** out) {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  void *va1 = nullptr, *va2 = nullptr;
  hipError_t e = hipMemAddressReserve(&va1, size, 0, nullptr, 0);
  if (e == hipSuccess) e = hipMemAddressReserve(&va2, size + shift, 0, nullptr, 0);
  if (e != hipSuccess) { fprintf(stderr, "reserve: %s\n", hipGetErrorString(e)); return (int)e; }
  va2 = (char*)va2 + shift;
  for (size_t off = 0; off < size; off += piece) {
    hipMemGenericAllocationHandle_t h;
    e = hipMemCreate(&h, piece, &prop, 0);
    if (e == hipSuccess) e = hipMemMap((char*)va1 + off, piece, 0, h, 0);
    if (e == hipSuccess) e = hipMemMap((char*)va2 + off, piece, 0, h, 0);
    if (e != hipSuccess) { fprintf(stderr, "create/map: %s\n", hipGetErrorString(e)); return (int)e; }
    (void)hipMemRelease(h);
  }
  hipMemAccessDesc acc{};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = 0;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(va1, size, &acc, 1);
  if (e == hipSuccess) e = hipMemSetAccess(va2, size, &acc, 1);
  if (e != hipSuccess) { fprintf(stderr, "access: %s\n", hipGetErrorString(e)); return (int)e; }
  out[0] = va1;
  out[1] = va2;
  return 0;
}
