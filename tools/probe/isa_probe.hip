// gfx950 instruction-cost probe for the trellis step: time long unrolled runs of single instruction patterns on one
// wave per workgroup (hipcc tools/probe/isa_probe.hip -o gpurun_out/isa_probe --offload-arch=gfx950; run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))
#define KERNEL(NAME, BODY, NINSTR)                                                                     \
    __global__ __launch_bounds__(64) void NAME(int *out, int iters) {                                  \
        int v0 = threadIdx.x, v1 = threadIdx.x * 3 + 1, v2 = 7, v3 = 9, v4 = 11, v5 = 13, s = iters;   \
        int sg = __builtin_amdgcn_readfirstlane(iters * 77);                                          \
        for (int i = 0; i < iters; i++) {                                                              \
            asm volatile(REP32(BODY) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5) : "s"(sg));  \
        }                                                                                              \
        out[blockIdx.x * 64 + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + s;                          \
    }                                                                                                  \
    static const int NAME##_n = NINSTR;
KERNEL(k_add_dep, "v_add_u32 %0, %0, %1\n\t", 1)
KERNEL(k_add_ind, "v_add_u32 %0, %1, %2\n\tv_add_u32 %3, %1, %2\n\tv_add_u32 %4, %1, %2\n\tv_add_u32 %5, %1, %2\n\t", 4)
KERNEL(k_dot4_ind, "v_dot4_i32_i8 %0, %6, %1, 0\n\tv_dot4_i32_i8 %3, %6, %2, 0\n\tv_dot4_i32_i8 %4, %6, %1, 0\n\tv_dot4_i32_i8 %5, %6, %2, 0\n\t", 4)
KERNEL(k_dot4_acc, "v_dot4_i32_i8 %0, %6, %1, %0\n\t", 1)
KERNEL(k_max_dep, "v_max_i32 %0, %0, %1\n\t", 1)
KERNEL(k_andor_dep, "v_and_or_b32 %0, %0, -2, %1\n\t", 1)
KERNEL(k_alignbit_dep, "v_alignbit_b32 %0, %1, %0, 1\n\t", 1)
KERNEL(k_subdpp_ind, "v_sub_u32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_sub_u32_dpp %3, %1, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_sub_u32_dpp %4, %1, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_sub_u32_dpp %5, %1, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t", 4)
KERNEL(k_subdpp_dep2, "v_sub_u32_dpp %0, %0, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_u32 %3, %1, %2\n\tv_add_u32 %4, %1, %2\n\t", 3)
KERNEL(k_swap32, "v_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\t", 1)
KERNEL(k_swap16_fill, "v_permlane16_swap_b32 %0, %1\n\tv_add_u32 %3, %4, %2\n\tv_add_u32 %5, %4, %2\n\t", 3)
KERNEL(k_snop0, "s_nop 0\n\t", 1)
KERNEL(k_snop1, "s_nop 1\n\t", 1)
KERNEL(k_snop2, "s_nop 2\n\t", 1)
KERNEL(k_dot_then_add, "v_dot4_i32_i8 %3, %6, %1, 0\n\ts_nop 2\n\tv_add_u32 %0, %0, %3\n\t", 2)
KERNEL(k_step_dpp, "v_add_u32 %3, %0, %1\n\tv_dot4_i32_i8 %4, %6, %2, 0\n\tv_sub_u32_dpp %1, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_alignbit_b32 %5, %3, %5, 1\n\tv_max_i32 %3, %3, %1\n\tv_and_or_b32 %0, %3, -2, %2\n\t", 6)

template <typename K> static double run(K kern, int n_per_body, int nblocks, const char *name, int *out)
{
    const int iters = 4096;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64), 0, 0, out, 16);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64), 0, 0, out, iters);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double ns_per = ms * 1e6 / ((double)iters * 32 * n_per_body);
    printf("%-16s blocks %4d: %7.3f ns per instruction (%6.2f cycles @2.4 GHz)\n", name, nblocks, ns_per, ns_per * 2.4);
    return ns_per;
}
#define RUN(NAME) for (int nb : {1, 1024}) run(NAME, NAME##_n, nb, #NAME, out);
int main()
{
    int *out; hipMalloc(&out, 1024 * 64 * sizeof(int));
    RUN(k_add_dep) RUN(k_add_ind) RUN(k_dot4_ind) RUN(k_dot4_acc) RUN(k_max_dep) RUN(k_andor_dep) RUN(k_alignbit_dep)
    RUN(k_subdpp_ind) RUN(k_subdpp_dep2) RUN(k_swap32) RUN(k_swap16_fill) RUN(k_snop0) RUN(k_snop1) RUN(k_snop2) RUN(k_dot_then_add) RUN(k_step_dpp)
    return 0;
}
