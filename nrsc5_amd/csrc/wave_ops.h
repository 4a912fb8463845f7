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
// place a wave-uniform value into one lane of a VGPR (v_cmp_eq + v_cndmask; v_writelane_b32 would need
// the lane select as an inline constant next to an SGPR value -- constant-bus limit on gfx9-class VOP3)
__device__ __forceinline__ int wave_writelane(int old, int sval, int lane) { return ((int)(threadIdx.x & 63) == lane) ? sval : old; }

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
