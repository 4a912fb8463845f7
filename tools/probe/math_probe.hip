// accuracy of nrsc5_amd/csrc/fastmath.h on the device against double precision
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include "fastmath.h"
using namespace nrsc5;
__global__ void k(double *err, float range)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
    double es = 0, ec = 0, ea = 0, el = 0, er = 0;
    for (int r = 0; r < 64; r++) {
        unsigned h = (unsigned)(i * 64 + r) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float x = ((float)(h >> 8) / 8388608.0f - 1.0f) * range;
        float s, c; fast_sincos(x, s, c);
        es = fmax(es, fabs((double)s - sin((double)x))); ec = fmax(ec, fabs((double)c - cos((double)x)));
        float s2, c2; sincosf(x, &s2, &c2);
        el = fmax(el, fmax(fabs((double)s2 - sin((double)x)), fabs((double)c2 - cos((double)x))));
        if (fabsf(x) <= 3.2f) { float s3, c3; fast_sincos_reduced(x, s3, c3); er = fmax(er, fmax(fabs((double)s3 - sin((double)x)), fabs((double)c3 - cos((double)x)))); }
        unsigned g = h * 3266489917u; g ^= g >> 16;
        const float yy = ((float)(g >> 8) / 8388608.0f - 1.0f) * ((r & 7) == 0 ? 1e-3f : 4.0f), xx = ((float)(h & 0xffffff) / 8388608.0f - 1.0f) * ((r & 3) == 0 ? 1e-3f : 4.0f);
        ea = fmax(ea, fabs((double)fast_atan2(yy, xx) - atan2((double)yy, (double)xx)));
    }
    err[i] = es; err[n + i] = ec; err[2 * n + i] = ea; err[3 * n + i] = el; err[4 * n + i] = er;
}
int main()
{
    const int nb = 1024, nt = 256, n = nb * nt;
    double *d; hipMalloc(&d, 5 * n * sizeof(double));
    double *h = new double[5 * n];
    for (float range : {3.1415927f, 100.0f, 2000.0f, 1e5f}) {
        hipLaunchKernelGGL(k, dim3(nb), dim3(nt), 0, 0, d, range);
        hipMemcpy(h, d, 5 * n * sizeof(double), hipMemcpyDeviceToHost);
        double m[5] = {0, 0, 0, 0, 0};
        for (int j = 0; j < 5; j++) for (int i = 0; i < n; i++) m[j] = fmax(m[j], h[j * n + i]);
        printf("range +-%g: max |err| fast sin %.3g cos %.3g | atan2 %.3g rad | ocml sincosf %.3g | fast reduced (|x|<=3.2) %.3g\n", range, m[0], m[1], m[2], m[3], m[4]);
    }
    return 0;
}
