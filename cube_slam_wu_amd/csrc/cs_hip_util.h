// cs_hip_util.h -- small host-side HIP helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

namespace cs {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's instance of a kernel: a handle on a second GPU of the
// same process (cs_ba_create(device = 1), sharded ranks as threads on several devices) needs its own call.  One table per kernel, a
// slot per device: 0 = not tried, 1 = set, 2 = refused.  Two threads racing on a slot both make the (idempotent) call.
struct DynLdsOnce {
  enum { MAX_DEV = 64 };
  std::atomic<unsigned char> slot[MAX_DEV];
  // true when `fn` may be launched with `bytes` of dynamic LDS on the current device
  bool set(const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    unsigned char s = slot[dev].load(std::memory_order_acquire);
    if (s == 0) {
      s = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 1 : 2;
      if (s == 2) (void)hipGetLastError();     // (the refusal is reported through the return value, not left as the thread's sticky error)
      slot[dev].store(s, std::memory_order_release);
    }
    return s == 1;
  }
};

}  // namespace cs
