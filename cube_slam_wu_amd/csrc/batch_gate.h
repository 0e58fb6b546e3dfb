// batch_gate.h -- the images of a batch reach the host stage chunk by chunk: the per-pixel kernels and the copy back of chunk c + 1 run
// while the worker pool is already walking the images of chunk c (the segment producers' host halves, lines_host.cpp / lsd_host.cpp).
// Task 0 of the pool run watches the chunks' events (blocking-sync events: the watcher sleeps) and opens the gate; an image's task
// sleeps on the gate until its chunk has arrived.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <vector>

namespace cs {

enum { BATCH_CHUNK = 4 };      // images per chunk (a chunk's copy back is ~1 ms at KITTI size: the pool starts that long after the call)

struct ChunkGate {
  std::mutex m;
  std::condition_variable cv;
  int ready = 0;               // images [0, ready) are in host memory
  bool failed = false;

  void watch(int device, const hipEvent_t* done, int n_chunks, int chunk, int n_images) {
    (void)hipSetDevice(device);
    for (int c = 0; c < n_chunks; c++) {
      const hipError_t e = hipEventSynchronize(done[c]);
      std::lock_guard<std::mutex> lk(m);
      if (e != hipSuccess) { failed = true; ready = n_images; cv.notify_all(); return; }   // (nobody may wait for ever)
      ready = std::min(n_images, (c + 1) * chunk);
      cv.notify_all();
    }
  }
  bool wait_for(int i) {       // false: a chunk's event reported an error
    std::unique_lock<std::mutex> lk(m);
    cv.wait(lk, [&] { return ready > i; });
    return !failed;
  }
};

// grow-only set of events of a producer's scratch: `done` (blocking sync, no timing) and a timed pair per chunk around its kernels
struct ChunkEvents {
  std::vector<hipEvent_t> done, k0, k1;
  hipError_t reserve(int n) {
    while ((int)done.size() < n) {
      hipEvent_t a = nullptr, b = nullptr, c = nullptr;
      hipError_t e = hipEventCreateWithFlags(&a, hipEventBlockingSync | hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreate(&b);
      if (e == hipSuccess) e = hipEventCreate(&c);
      if (e != hipSuccess) { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); if (c) (void)hipEventDestroy(c); return e; }
      done.push_back(a); k0.push_back(b); k1.push_back(c);
    }
    return hipSuccess;
  }
  void release() {
    for (hipEvent_t e : done) (void)hipEventDestroy(e);
    for (hipEvent_t e : k0) (void)hipEventDestroy(e);
    for (hipEvent_t e : k1) (void)hipEventDestroy(e);
    done.clear(); k0.clear(); k1.clear();
  }
};

}  // namespace cs
