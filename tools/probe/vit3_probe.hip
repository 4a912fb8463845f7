// pure-compute cost of the generated 8-step trellis blocks (no memory): ns per trellis step
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "viterbi_v3_asm.h"
#define OPERANDS                                                                                                       \
    : [u] "+v"(u), [h] "+v"(hist), [ns] "+v"(ns), [x] "=&v"(x), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)   \
    : [a0] "s"(a0), [a1] "s"(a1), [a2] "s"(a2), [a3] "s"(a3), [a4] "s"(a4), [a5] "s"(a5), [a6] "s"(a6), [a7] "s"(a7),   \
      [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [p4] "v"(p4), [q4] "v"(q4), [p5] "v"(p5), [q5] "v"(q5),   \
      [z0] "v"(z0), [z1] "v"(z1), [z2] "v"(z2), [z3] "v"(z3), [z4] "v"(z4), [z5] "v"(z5)
__global__ __launch_bounds__(64) void k(int *out, int iters, int mode)
{
    const int lane = threadIdx.x;
    int u = lane & 1, hist = 0, ns = 0, x, d0, d1, d2, d3;
    const int w0 = 0x020202 ^ (lane * 0x40404 & 0xfcfcfc), w1 = 0xfe02fe, w2 = 0x02fefe, w3 = 0xfefe02, p4 = 0x0202fe, q4 = 0xfefe02, p5 = 0x02fe02, q5 = 0xfe02fe;
    const int z0 = lane & 1, z1 = (lane >> 1) & 1, z2 = (lane >> 2) & 1, z3 = (lane >> 3) & 1, z4 = (lane >> 4) & 1, z5 = (lane >> 5) & 1;
    int a0 = __builtin_amdgcn_readfirstlane(iters * 0x010203), a1 = a0 + 5, a2 = a0 ^ 0x70102, a3 = a0 + 9, a4 = a0 * 3, a5 = a0 - 1, a6 = a0 + 77, a7 = a0 ^ 0x1111;
    for (int i = 0; i < iters; i++) {
        if (mode == 0) {          // the three chunk variants' 24 blocks (192 steps), as the kernel runs them
            asm volatile(VIT3_ASM_PH0_OPEN OPERANDS); asm volatile(VIT3_ASM_PH2_CONT OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS);
            asm volatile(VIT3_ASM_PH2_OPEN OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS); asm volatile(VIT3_ASM_PH2_CONT OPERANDS);
            asm volatile(VIT3_ASM_PH4_OPEN OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS); asm volatile(VIT3_ASM_PH2_CONT OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS);
        } else if (mode == 1) { asm volatile(VIT3_ASM_PH0_CONT OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS);
                                asm volatile(VIT3_ASM_PH0_CONT OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS);
                                asm volatile(VIT3_ASM_PH0_CONT OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS); asm volatile(VIT3_ASM_PH0_CONT OPERANDS);
        } else { asm volatile(VIT3_ASM_PH4_CONT OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS);
                 asm volatile(VIT3_ASM_PH4_CONT OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS);
                 asm volatile(VIT3_ASM_PH4_CONT OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS); asm volatile(VIT3_ASM_PH4_CONT OPERANDS); }
    }
    out[blockIdx.x * 64 + lane] = u + hist + ns;
}
int main()
{
    int *out; (void)hipMalloc(&out, 1024 * 64 * sizeof(int));
    const char *names[3] = {"mixed (as run)", "PH0_CONT x12 (2 swap steps / 8)", "PH4_CONT x12 (4 swap steps / 8)"};
    for (int mode = 0; mode < 3; mode++) for (int nb : {1, 256, 1024}) {
        const int iters = 4096;
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, out, 16, mode);
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(k, dim3(nb), dim3(64), 0, 0, out, iters, mode);
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-34s blocks %4d: %6.2f ns per trellis step\n", names[mode], nb, ms * 1e6 / ((double)iters * 96));
    }
    return 0;
}
