// gfx950 latency probe for the exact-oscillator chain (k_nco_exact): one float complex multiplication per step, every step dependent on the last.
//   hipcc tools/probe/nco_probe.hip -o gpurun_out/nco_probe --offload-arch=gfx950 -O3 -ffp-contract=off && gpurun_out/nco_probe
// Variants: packed (2 v_pk_mul_f32 + 1 v_pk_add_f32), scalar (4 v_mul_f32 + v_sub_f32 + v_add_f32), two lanes per chain with the partner's
// component through DPP (2 v_mul_f32_dpp + 1 v_add_f32 + the wait states a DPP read of a fresh VALU result needs).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float cf __attribute__((ext_vector_type(2)));
constexpr int STEPS = 69120;

__global__ __launch_bounds__(64) void k_packed(float2 *out, float c, float d)
{
    if (threadIdx.x) return;
    cf P = {1.0f, 0.0f}; const cf K = {c, d};
#pragma unroll 8
    for (int j = 0; j < STEPS; j++) {
        cf t1, t2;
        asm("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[0,1]\n\tv_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\tv_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "+v"(P), "=&v"(t1), "=&v"(t2) : "v"(K));
    }
    out[blockIdx.x] = make_float2(P.x, P.y);
}
__global__ __launch_bounds__(64) void k_scalar(float2 *out, float c, float d)
{
    if (threadIdx.x) return;
    float a = 1.0f, b = 0.0f;
#pragma unroll 8
    for (int j = 0; j < STEPS; j++) {
        float ac, bd, ad, bc;
        asm("v_mul_f32 %2, %0, %6\n\tv_mul_f32 %3, %1, %7\n\tv_mul_f32 %4, %0, %7\n\tv_mul_f32 %5, %1, %6\n\tv_sub_f32 %0, %2, %3\n\tv_add_f32 %1, %4, %5"
            : "+v"(a), "+v"(b), "=&v"(ac), "=&v"(bd), "=&v"(ad), "=&v"(bc) : "v"(c), "v"(d));
    }
    out[blockIdx.x] = make_float2(a, b);
}
// lanes 0 / 1 of a quad: r = re in lane 0, im in lane 1.  k1 = (c, d), k2 = (-d, c) per lane: lane 0: a c + b (-d), lane 1: a d + b c
template <int NOP>
__global__ __launch_bounds__(64) void k_dpp(float2 *out, float c, float d)
{
    if (threadIdx.x > 1) return;
    float r = threadIdx.x == 0 ? 1.0f : 0.0f;
    const float k1 = threadIdx.x == 0 ? c : d, k2 = threadIdx.x == 0 ? -d : c;
#pragma unroll 8
    for (int j = 0; j < STEPS; j++) {
        float t1, t2;
        if (NOP == 1)
            asm("s_nop 1\n\tv_mul_f32_dpp %1, %0, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %2, %0, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32 %0, %1, %2"
                : "+v"(r), "=&v"(t1), "=&v"(t2) : "v"(k1), "v"(k2));
        else
            asm("s_nop 0\n\tv_mul_f32_dpp %1, %0, %3 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\tv_mul_f32_dpp %2, %0, %4 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_add_f32 %0, %1, %2"
                : "+v"(r), "=&v"(t1), "=&v"(t2) : "v"(k1), "v"(k2));
    }
    if (threadIdx.x == 0) out[blockIdx.x].x = r; else out[blockIdx.x].y = r;
}

template <typename K> static void run(K kern, const char *name, float2 *out, int nblocks)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const float c = 0.99999976f, d = 6.9e-4f;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64), 0, 0, out, c, d);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64), 0, 0, out, c, d);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    float2 h; hipMemcpy(&h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-10s blocks %4d: %8.3f ms per launch = %6.2f ns per step   result (%.9g, %.9g)\n", name, nblocks, ms / 5, ms / 5 * 1e6 / STEPS, h.x, h.y);
}

int main()
{
    float2 *out; hipMalloc(&out, 4096 * sizeof(float2));
    for (int nb : {1, 256, 2048}) {
        run(k_packed, "packed", out, nb);
        run(k_scalar, "scalar", out, nb);
        run(k_dpp<1>, "dpp nop1", out, nb);
        run(k_dpp<0>, "dpp nop0", out, nb);
    }
    return 0;
}
