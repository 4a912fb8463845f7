// Wave64 helpers for gfx950.  Everything here is wave-synchronous: all 64 lanes execute it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrsc5 {

#ifdef HIPEMU   // CPU logic-test build (tests/simt): same semantics through the emulator's collectives
__device__ inline int wave_readlane(int v, int lane) { return __shfl(v, lane); }
#else
// v_readlane_b32: lane index must be wave-uniform
__device__ __forceinline__ int wave_readlane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
#endif
// Order this wave's own LDS traffic without a workgroup barrier: the LDS unit serves one wave's requests in issue order,
// so a ds_read issued after a ds_write of the same wave sees it; all that is needed is that the compiler keeps the
// order.  (__syncthreads() would also wait for outstanding GLOBAL stores -- hundreds of cycles per trellis step.)
// Only valid in single-wave workgroups.
#ifdef HIPEMU
#define WAVE_LDS_SYNC() __syncthreads()
#else
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
// the same ordering without the wave barrier's HIPEMU stand-in (a workgroup barrier): for code that only one wave of a larger
// workgroup executes
#ifdef HIPEMU
#define WAVE_LDS_FENCE() do { } while (0)
#else
#define WAVE_LDS_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
// Raise this wave's issue priority (s_setprio): the per-block kernels are a serial chain of 224 steps per pass, the
// trellis passes that share their SIMDs are long-running background work.
#ifdef HIPEMU
__device__ inline void wave_set_priority(int) {}
#else
// s_setprio takes an immediate: 0 (default) .. 3
__device__ __forceinline__ void wave_set_priority(int p)
{
    if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 3) __builtin_amdgcn_s_setprio(3);
}
#endif
#ifdef HIPEMU
__device__ inline void wave_set_priority_high() {}
#else
__device__ __forceinline__ void wave_set_priority_high() { __builtin_amdgcn_s_setprio(3); }
#endif

#ifdef HIPEMU
__device__ inline int wave_uniform(int v) { return __shfl(v, 0); }
#else
// tell the compiler a value is wave-uniform (moves it to an SGPR: scalar ALU from here on)
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif
// place a wave-uniform value into one lane of a VGPR (v_cmp_eq + v_cndmask; v_writelane_b32 would need
// the lane select as an inline constant next to an SGPR value -- constant-bus limit on gfx9-class VOP3)
__device__ __forceinline__ int wave_writelane(int old, int sval, int lane) { return ((int)(threadIdx.x & 63) == lane) ? sval : old; }

// write a wave-uniform value into lane LANE (compile-time) of a VGPR: one v_writelane_b32 (SGPR value +
// inline-constant lane select fits the constant bus)
template <int LANE> __device__ __forceinline__ int wave_writelane_c(int old, int sval)
{
#ifdef HIPEMU
    return ((int)(threadIdx.x & 63) == LANE) ? sval : old;
#else
    asm("v_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(sval), "n"(LANE));
    return old;
#endif
}

// Fused ACS select for the rotating-layout Viterbi:
//   ballot = (xc > y) per lane;  pm = ballot ? x : y;  and, in the shadow of the compare, the PREVIOUS
//   step's ballot (plo, phi) is parked in lane LANE of (lo, hi).
// v_writelane_b32 reads its SGPR data operand early in the pipe: it must not follow the VALU write of that
// SGPR by fewer than ~4 wait states (observed on gfx950: a v_writelane right behind v_cmp stores the OLD
// mask).  Writing the previous step's mask here guarantees >= 8 instructions of distance without s_nops.
template <int LANE>
__device__ __forceinline__ unsigned long long acs_select_park(int &pm, int x, int xc, int y, unsigned long long prev, int &lo, int &hi)
{
#ifdef HIPEMU
    const bool own = xc > y;
    pm = own ? x : y;
    if ((int)(threadIdx.x & 63) == LANE) { lo = (int)(uint32_t)prev; hi = (int)(uint32_t)(prev >> 32); }
    return __ballot(own);
#else
    unsigned long long b;
    int npm;
    const int plo = (int)(uint32_t)prev, phi = (int)(uint32_t)(prev >> 32);
    asm("v_cmp_gt_i32_e64 %0, %5, %6\n\t"
        "v_writelane_b32 %2, %7, %9\n\t"
        "v_writelane_b32 %3, %8, %9\n\t"
        "v_cndmask_b32_e64 %1, %6, %4, %0"
        : "=&s"(b), "=v"(npm), "+v"(lo), "+v"(hi)
        : "v"(x), "v"(xc), "v"(y), "s"(plo), "s"(phi), "n"(LANE));
    pm = npm;
    return b;
#endif
}

// plain park with explicit wait states (chunk epilogue only)
template <int LANE>
__device__ __forceinline__ void park_ballot(unsigned long long v, int &lo, int &hi)
{
#ifdef HIPEMU
    if ((int)(threadIdx.x & 63) == LANE) { lo = (int)(uint32_t)v; hi = (int)(uint32_t)(v >> 32); }
#else
    const int vlo = (int)(uint32_t)v, vhi = (int)(uint32_t)(v >> 32);
    asm("s_nop 4\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4" : "+v"(lo), "+v"(hi) : "s"(vlo), "s"(vhi), "n"(LANE));
#endif
}

// lane_xor<M>(v): value of lane (id ^ M), M a power of two.  On gfx950 these are register-file moves
// (DPP quad_perm / row_ror, v_permlane16_swap, v_permlane32_swap) -- no LDS crossbar round trip as with
// ds_bpermute.  Encodings are verified against __shfl_xor on the device by nrsc5hip_stage_selftest.
template <int M> __device__ __forceinline__ int lane_xor(int v)
{
#ifdef HIPEMU
    return __shfl_xor(v, M);
#else
    if constexpr (M == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    else if constexpr (M == 4) {
        const int t = __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xa, false);                  // row_ror:4 -> lanes 4-7, 12-15
        return __builtin_amdgcn_update_dpp(t, v, 0x12C, 0xf, 0x5, false);                         // row_ror:12 -> lanes 0-3, 8-11
    }
    else if constexpr (M == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);  // row_ror:8
    else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);   // {vdst', vsrc'}: odd rows of vdst <-> even rows of vsrc
        return (threadIdx.x & 16) ? (int)r[0] : (int)r[1];
    }
    else {
        static_assert(M == 32, "lane_xor: M must be 1,2,4,8,16,32");
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // upper half of vdst <-> lower half of vsrc
        return (threadIdx.x & 32) ? (int)r[0] : (int)r[1];
    }
#endif
}

// Wave reductions over lane_xor's register-file moves (the generic wave_max_i32 / wave_min_i32 / wave_sum_f64 below go through __shfl_xor = ds_bpermute: twelve LDS
// crossbar round trips for a max + arg-min, ~100 shader cycles each on a chain one lone wave waits for).  Same butterfly, same pairing order: identical results.
template <int M> __device__ __forceinline__ double lane_xor_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)lane_xor<M>((int)(unsigned)b), hi = (unsigned)lane_xor<M>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int wave_max_i32_rf(int v)
{
    int o;
    o = lane_xor<32>(v); v = o > v ? o : v; o = lane_xor<16>(v); v = o > v ? o : v; o = lane_xor<8>(v); v = o > v ? o : v;
    o = lane_xor<4>(v); v = o > v ? o : v; o = lane_xor<2>(v); v = o > v ? o : v; o = lane_xor<1>(v); v = o > v ? o : v;
    return v;
}
__device__ __forceinline__ int wave_min_i32_rf(int v)
{
    int o;
    o = lane_xor<32>(v); v = o < v ? o : v; o = lane_xor<16>(v); v = o < v ? o : v; o = lane_xor<8>(v); v = o < v ? o : v;
    o = lane_xor<4>(v); v = o < v ? o : v; o = lane_xor<2>(v); v = o < v ? o : v; o = lane_xor<1>(v); v = o < v ? o : v;
    return v;
}
__device__ __forceinline__ double wave_sum_f64_rf(double v)
{
    v += lane_xor_f64<32>(v); v += lane_xor_f64<16>(v); v += lane_xor_f64<8>(v); v += lane_xor_f64<4>(v); v += lane_xor_f64<2>(v); v += lane_xor_f64<1>(v);
    return v;
}

// signed 4 x int8 dot product + accumulator (v_dot4_i32_i8)
__device__ __forceinline__ int dot4_i8(int a, int b, int acc)
{
#ifdef HIPEMU
    int s = acc;
    for (int k = 0; k < 4; k++) s += (int)(int8_t)(a >> (8 * k)) * (int)(int8_t)(b >> (8 * k));
    return s;
#else
    return __builtin_amdgcn_sdot4(a, b, acc, false);
#endif
}

__device__ inline int wave_min_i32(int v)
{
    for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(v, m); v = o < v ? o : v; }
    return v;
}
__device__ inline int wave_max_i32(int v)
{
    for (int m = 32; m >= 1; m >>= 1) { int o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
}
__device__ inline int wave_sum_i32(int v)
{
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ inline float wave_sum_f32(float v)
{
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ inline double wave_sum_f64(double v)
{
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

}  // namespace nrsc5
