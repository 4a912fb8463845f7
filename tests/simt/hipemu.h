// TEST INFRASTRUCTURE ONLY -- a minimal single-threaded SIMT emulator that lets the HIP
// sources under nrsc5_amd/csrc be compiled with g++ and executed on a CPU so that kernel
// LOGIC can be debugged in this GPU-less container (`-m "not gpu"` tests).  It is never a
// fallback: the shipped library (libnrsc5hip.so) is built by hipcc from the same sources and
// the Python binding refuses to run without it.  The emulated build lives in tests/simt/ and
// is only ever loaded explicitly by tests.
//
// Model: one workgroup at a time; every work-item is a fiber (own stack, hand-rolled x86-64
// context switch); __syncthreads() and wave64 collectives (__shfl*, __ballot, ...) are
// rendezvous points.  Divergent collectives/barriers are reported as deadlocks.
#pragma once
#ifndef __x86_64__
#error "hipemu needs x86-64"
#endif
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <functional>
#include <vector>
#include <algorithm>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define HIP_KERNEL_NAME(...) __VA_ARGS__

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct uint3_emu { unsigned x, y, z; };
extern uint3_emu threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 64;

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct short2 { short x, y; };
struct short4 { short x, y, z, w; };
struct ushort2 { unsigned short x, y; };
struct char2 { signed char x, y; };
struct char4 { signed char x, y, z, w; };
struct uchar2 { unsigned char x, y; };
struct uchar4 { unsigned char x, y, z, w; };
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline int2 make_int2(int a, int b) { return int2{a, b}; }
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline short2 make_short2(short a, short b) { return short2{a, b}; }
static inline uchar4 make_uchar4(unsigned char a, unsigned char b, unsigned char c, unsigned char d) { return uchar4{a, b, c, d}; }

namespace simt {
void launch(dim3 grid, dim3 block, size_t shmem, std::function<void()> body);
void sync_block();
// wave collective: deposits `v` (8 bytes max) for this lane, waits for the wave, returns slots.
const uint64_t *wave_exchange(uint64_t v, uint64_t *active_mask);
int lane_id();
void *dynamic_shared();
}
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)simt::dynamic_shared();

static inline void __syncthreads() { simt::sync_block(); }

template <typename T> static inline uint64_t simt_pack(T v) { uint64_t u = 0; static_assert(sizeof(T) <= 8, "shfl type"); memcpy(&u, &v, sizeof(T)); return u; }
template <typename T> static inline T simt_unpack(uint64_t u) { T v; memcpy(&v, &u, sizeof(T)); return v; }

template <typename T> static inline T __shfl(T v, int src, int width = 64)
{
    uint64_t act; const uint64_t *s = simt::wave_exchange(simt_pack(v), &act);
    int lane = simt::lane_id();
    int idx = (lane & ~(width - 1)) | (src & (width - 1));
    return ((act >> idx) & 1) ? simt_unpack<T>(s[idx]) : v;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64)
{
    uint64_t act; const uint64_t *s = simt::wave_exchange(simt_pack(v), &act);
    int lane = simt::lane_id();
    int idx = lane ^ mask;
    if ((idx & ~(width - 1)) != (lane & ~(width - 1))) idx = lane;
    return ((act >> idx) & 1) ? simt_unpack<T>(s[idx]) : v;
}
template <typename T> static inline T __shfl_down(T v, unsigned delta, int width = 64)
{
    uint64_t act; const uint64_t *s = simt::wave_exchange(simt_pack(v), &act);
    int lane = simt::lane_id();
    int idx = lane + (int)delta;
    if ((idx & ~(width - 1)) != (lane & ~(width - 1))) idx = lane;
    return ((act >> idx) & 1) ? simt_unpack<T>(s[idx]) : v;
}
template <typename T> static inline T __shfl_up(T v, unsigned delta, int width = 64)
{
    uint64_t act; const uint64_t *s = simt::wave_exchange(simt_pack(v), &act);
    int lane = simt::lane_id();
    int idx = lane - (int)delta;
    if (idx < 0 || (idx & ~(width - 1)) != (lane & ~(width - 1))) idx = lane;
    return ((act >> idx) & 1) ? simt_unpack<T>(s[idx]) : v;
}
static inline unsigned long long __ballot(int pred)
{
    uint64_t act; const uint64_t *s = simt::wave_exchange(pred ? 1 : 0, &act);
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) if (((act >> i) & 1) && s[i]) m |= 1ull << i;
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0; }
static inline int __all(int pred)
{
    uint64_t act; const uint64_t *s = simt::wave_exchange(pred ? 1 : 0, &act);
    for (int i = 0; i < 64; i++) if (((act >> i) & 1) && !s[i]) return 0;
    return 1;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }

template <typename T> static inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMax(T *p, T v) { T o = *p; *p = std::max(o, v); return o; }
template <typename T> static inline T atomicCAS(T *p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
template <typename T> static inline T atomicMin(T *p, T v) { T o = *p; *p = std::min(o, v); return o; }
template <typename T> static inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> static inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
static inline long long clock64() { return 0; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __threadfence_system() {}

static inline void sincosf_emu(float x, float *s, float *c) { *s = sinf(x); *c = cosf(x); }
#define __sincosf(x, s, c) sincosf_emu((x), (s), (c))
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __saturatef(float x) { return x < 0 ? 0 : x > 1 ? 1 : x; }

// ---- runtime API subset ------------------------------------------------------------------
typedef int hipError_t;
typedef struct simt_stream *hipStream_t;
typedef struct simt_event *hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };
static inline hipError_t hipMalloc(void **p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned f = 0) { return hipMalloc((void **)p, n); }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = 0) { for (size_t r = 0; r < h; r++) memmove((char *)d + r * dp, (const char *)s + r * sp, w); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
#define hipHostMallocMapped 0u
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = 0; return hipSuccess; }
#define hipStreamDefault 0u
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = 0; return hipSuccess; }
#define hipEventDisableTiming 2u
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = 0; return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = 0) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template <typename F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "hipemu"); strcpy(p->gcnArchName, "emu"); p->multiProcessorCount = 1; return hipSuccess; }
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0

#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    simt::launch(dim3(grid), dim3(block), (shmem), [=]() { kern(__VA_ARGS__); })
using std::min;
using std::max;
