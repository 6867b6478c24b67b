// A SYNCHRONOUS stand-in for the HIP runtime, for the CPU suite only: "device" memory is host memory, every asynchronous call
// completes before it returns, streams and events are tokens.  With it the HOST side of the C ABI (flowgger_amd/csrc/fg_capi.cpp:
// contexts, staging buffers, the sliced host paths with their entry-range bookkeeping, retries, error paths) compiles with g++ and
// runs in `pytest -m "not gpu"` against fake kernel launchers (tests/native/host_pipeline_fake.cpp).  It says nothing about
// concurrency -- every interleaving it executes is the sequential one.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 } hipError_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
struct fakehip_stream { int id; };
struct fakehip_event { int recorded; };
typedef fakehip_stream* hipStream_t;
typedef fakehip_event* hipEvent_t;
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; char gcnArchName[64]; };
struct uint4 { uint32_t x, y, z, w; };
struct dim3 { uint32_t x, y, z; };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };

namespace fakehip {
// fault injection + accounting for the tests
inline long long& fail_malloc_after() { static long long v = -1; return v; }  // >= 0: that many more hipMalloc calls succeed
inline unsigned long long& copies() { static unsigned long long v = 0; return v; }
inline unsigned long long& bytes_h2d() { static unsigned long long v = 0; return v; }
inline unsigned long long& bytes_d2h() { static unsigned long long v = 0; return v; }
}  // namespace fakehip

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "ok" : "fake hip error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p);
    p->multiProcessorCount = 4;
    strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    strcpy(p->name, "fake MI355X");
    return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) {
    if (fakehip::fail_malloc_after() == 0) { *p = nullptr; return hipErrorOutOfMemory; }
    if (fakehip::fail_malloc_after() > 0) --fakehip::fail_malloc_after();
    *p = malloc(n ? n : 1);
    if (*p) memset(*p, 0xCD, n);  // (device memory is not zeroed)
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
// pinned allocations are remembered: hipPointerGetAttributes tells pinned host memory (the zero-copy host path) from pageable
namespace fakehip {
struct Range { const char* p; size_t n; };
inline std::vector<Range>& pinned() { static std::vector<Range> v; return v; }
}  // namespace fakehip
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) {
    *p = malloc(n ? n : 1);
    if (*p) {
        memset(*p, 0xAB, n);
        fakehip::pinned().push_back({(const char*)*p, n ? n : 1});
    }
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipHostFree(void* p) {
    auto& v = fakehip::pinned();
    for (size_t i = 0; i < v.size(); ++i)
        if (v[i].p == (const char*)p) {
            v.erase(v.begin() + (long)i);
            break;
        }
    free(p);
    return hipSuccess;
}
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    for (const auto& r : fakehip::pinned())
        if ((const char*)p >= r.p && (const char*)p < r.p + r.n) {
            a->type = hipMemoryTypeHost;
            a->device = 0;
            a->devicePointer = const_cast<void*>(p);
            a->hostPointer = const_cast<void*>(p);
            return hipSuccess;
        }
    a->type = hipMemoryTypeUnregistered;
    a->devicePointer = nullptr;
    return hipErrorInvalidValue;
}
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) {
    memmove(d, s, n);
    ++fakehip::copies();
    if (k == hipMemcpyHostToDevice) fakehip::bytes_h2d() += n;
    if (k == hipMemcpyDeviceToHost) fakehip::bytes_d2h() += n;
    return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new fakehip_stream{0}; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new fakehip_event{0}; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->recorded = 1; return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }

// the few device builtins that non-template inline code of the shared headers names
static inline uint32_t fakehip_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8u * (s & 3u))); }
#define __builtin_amdgcn_alignbyte(hi, lo, s) fakehip_alignbyte((hi), (lo), (s))
#define __builtin_amdgcn_ubfe(v, off, w) (((uint32_t)(v) >> (off)) & ((1u << (w)) - 1u))

// (wave-level code of the shared headers is compiled but never run here)
struct fakehip_idx { uint32_t x, y, z; };
static const fakehip_idx threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0};
static inline uint32_t __lane_id() { return 0; }
template <class T> static inline T __shfl(T v, int, int = 64) { return v; }
template <class T> static inline T __shfl_up(T v, unsigned, int = 64) { return v; }
