// LD_PRELOAD shim for HOST-SIDE PROFILING in a container without a GPU: device memory is host memory, copies are memcpy, kernel
// launches, events and streams do nothing.  NOTHING is computed -- it exists so that the structure phase of the bundle adjustment
// (index maps, orderings, schedules: ba_host.cpp, pure host work ending in uploads) can be timed and its uploaded arrays compared
// between two builds (tools/hostonly/structure_time.py).  Not part of the product, not loaded by any test of results.
#include <hip/hip_runtime_api.h>
#include <cstdlib>
#include <cstring>
extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  strcpy(p->name, "no-gpu shim"); strcpy(p->gcnArchName, "gfx950");
  p->multiProcessorCount = 256; p->warpSize = 64; p->sharedMemPerBlock = 65536; p->maxSharedMemoryPerMultiProcessor = 163840;
  p->totalGlobalMem = 288ull << 30; p->maxThreadsPerBlock = 1024; p->maxThreadsPerMultiProcessor = 2048; p->sharedMemPerBlockOptin = 163840;
  return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 8); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 8); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 2; return hipSuccess; }
int rocblas_create_handle(void** h) { *h = malloc(8); return 0; }
int rocblas_destroy_handle(void* h) { free(h); return 0; }
int rocblas_set_stream(void*, void*) { return 0; }
}
