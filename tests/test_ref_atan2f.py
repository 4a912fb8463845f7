"""CPU-only: `ref_atan2f` (nrsc5_amd/csrc/fastmath.h) -- the float arc tangent the device uses wherever the reference calls cargf once per block (the coarse carrier
angle of an acquisition block, acquire.c:153; the AM carrier / equaliser phases) -- returns glibc's atan2f BIT FOR BIT: fdlibm's float algorithm restated, float
operations only.  Compiled here with g++ from the very header the device build includes (no contraction, as the device build) and compared with this container's libm
on 2e7 arguments of four distributions (unit square, tall / flat ratios, raw bit patterns incl. NaN / inf / denormals, the right half-plane) plus the special cases."""
import os
import subprocess
import tempfile

from tests import common

SRC = r'''
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fastmath.h"
using namespace nrsc5;
static long check(float y, float x) { float r = ref_atan2f(y, x), g = atan2f(y, x); return (memcmp(&r, &g, 4) != 0 && !(r != r && g != g)) ? 1 : 0; }
int main() {
    srand48(7); long n = 20000000, bad = 0;
    for (long i = 0; i < n; i++) {
        float y, x; const int mode = i % 4;
        if (mode == 0) { y = (float)(drand48() * 2 - 1); x = (float)(drand48() * 2 - 1); }
        else if (mode == 1) { y = (float)((drand48() * 2 - 1) * 1e3); x = (float)((drand48() * 2 - 1) * 1e-3); }
        else if (mode == 2) { unsigned a = (unsigned)mrand48(), b = (unsigned)mrand48(); memcpy(&y, &a, 4); memcpy(&x, &b, 4); }
        else { y = (float)((drand48() * 2 - 1) * 50); x = (float)(drand48() * 100); }
        bad += check(y, x);
    }
    const float sp[] = { 0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, 1e-40f, -1e-40f, 3.4e38f, 0.4375f, 0.6875f, 1.1875f, 2.4375f, 33554432.0f };
    for (float y : sp) for (float x : sp) bad += check(y, x);
    printf("%ld\n", bad);
    return 0;
}
'''


def test_ref_atan2f_equals_glibc_bit_for_bit():
    simt = os.path.join(common.ROOT, "tests", "simt")
    csrc = os.path.join(common.ROOT, "nrsc5_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.cpp"), os.path.join(d, "t")
        open(src, "w").write(SRC)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-w", "-DHIPEMU", "-I" + simt, "-I" + csrc, "-o", exe, src, "-lm"])
        out = subprocess.check_output([exe]).decode().strip()
    assert out == "0", f"{out} mismatches against glibc's atan2f"
