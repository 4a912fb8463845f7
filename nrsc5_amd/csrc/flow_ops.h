// Hand-off primitives for work items of ONE launch that depend on each other (k_flow.hip): gfx950 has eight XCDs whose L2s are not coherent with each other and a
// vector L1 per CU that no other CU's store ever refreshes, so a value crosses from one workgroup to another only through
//   * a store that goes THROUGH the caches (sc1: "write-through", the line is dropped from the writer's L2) and a load that goes past the reader's L1 (sc1), or
//   * plain stores -> agent-scope RELEASE (buffer_wbl2 sc1: the XCD L2's dirty lines written back) -> flag -> agent-scope ACQUIRE on the reader (buffer_inv sc1) -> plain loads.
// (/opt/skills/guides/cdna_hip_programming.md, Guideline 16.)  The CPU twin (HIPEMU) runs the workgroups of a launch one after the other in dispatch order: plain accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrsc5 {

#ifdef HIPEMU
__device__ inline void flow_store_u64(unsigned long long *p, unsigned long long v) { *p = v; }
__device__ inline unsigned long long flow_load_u64(const unsigned long long *p) { return *p; }
__device__ inline unsigned flow_load_u32(const unsigned *p) { return *p; }
__device__ inline void flow_store_u32(unsigned *p, unsigned v) { *p = v; }
__device__ inline unsigned flow_add_u32(unsigned *p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
__device__ inline void flow_release() {}
__device__ inline void flow_acquire() {}
__device__ inline void flow_drain_stores() {}
__device__ inline void flow_sleep() {}
__device__ inline int flow_xcc_id() { return (int)(blockIdx.x & 7); }
#else
// relaxed, agent scope: global_store / global_load ... sc1
__device__ __forceinline__ void flow_store_u64(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long flow_load_u64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned flow_load_u32(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flow_store_u32(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned flow_add_u32(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane, after a workgroup barrier that every storing wave reached with its stores drained.  The asm wait restates the one the compiler may drop behind buffer_wbl2
// (ROCm 7.2: whenever its scoreboard says the wave has nothing outstanding at the fence -- the flag could then overtake the write-back)
__device__ __forceinline__ void flow_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void flow_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
// every wave that stored write-through, before the barrier in front of the counter
__device__ __forceinline__ void flow_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void flow_sleep() { __builtin_amdgcn_s_sleep(8); }
__device__ __forceinline__ int flow_xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return (int)(v & 7u); }
#endif

// an 8-byte value through the caches (a complex bin; a {tag, value} granule)
template <typename T> __device__ __forceinline__ void flow_store_through(T *p, T v)
{
    static_assert(sizeof(T) == 8, "one aligned 8-byte store");
    unsigned long long bits; __builtin_memcpy(&bits, &v, 8);
    flow_store_u64(reinterpret_cast<unsigned long long *>(p), bits);
}

}  // namespace nrsc5
