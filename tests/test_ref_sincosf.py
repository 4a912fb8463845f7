"""CPU-only: `ref_sincosf` (nrsc5_amd/csrc/fastmath.h) -- what the device uses wherever the reference calls cexpf(I y) (the Costas loops, sync.c:103-135, the equaliser
phases, sync.c:271-272, the NCO set-up, acquire.c:153-168) -- returns glibc's sincosf BIT FOR BIT, in the build glibc's ifunc selects on a host with FMA + AVX2
(`__sincosf_fma`): the published algorithm (double-precision reduction + polynomial pair) restated with the same fused / unfused operations.  Compiled here with g++ from
the very header the device build includes (no contraction beyond the explicit fma calls, as the device build) and compared with this container's libm on 1e8 arguments of
five distributions: the Costas loops' steady range (+-2 pi), the CFO search's range (+-2000 rad), |y| < pi/4, raw bit patterns (incl. NaN / inf / denormals / huge), and the
neighbourhood of multiples of pi/2.  Skipped (not failed) on a host whose CPU lacks FMA: its libm runs the unfused build, which is a different function in the last bit."""
import os
import subprocess
import tempfile

import pytest

from tests import common

SRC = r'''
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "fastmath.h"
using namespace nrsc5;
static long check(float y) {
    float rs, rc, gs, gc; ref_sincosf(y, rs, rc); sincosf(y, &gs, &gc);
    const bool bs = memcmp(&rs, &gs, 4) != 0 && !(rs != rs && gs != gs), bc = memcmp(&rc, &gc, 4) != 0 && !(rc != rc && gc != gc);
    return (bs || bc) ? 1 : 0;
}
int main(int argc, char **argv) {
    srand48(11); long n = atol(argv[1]), bad = 0;
    for (long i = 0; i < n; i++) {
        float y; const int mode = i % 5;
        if (mode == 0) y = (float)((drand48() * 2 - 1) * 6.5);
        else if (mode == 1) y = (float)((drand48() * 2 - 1) * 2000.0);
        else if (mode == 2) y = (float)((drand48() * 2 - 1) * 0.79);
        else if (mode == 3) { unsigned a = (unsigned)mrand48(); memcpy(&y, &a, 4); }
        else y = (float)((double)(lrand48() % 2001 - 1000) * 1.5707963267948966 + (drand48() * 2 - 1) * 1e-3);
        bad += check(y);
    }
    const float sp[] = { 0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, 1e-40f, -1e-40f, 3.4e38f, -3.4e38f, 0.78539816f, 0.78539822f, 120.0f, 119.99999f, 2.4414062e-4f, 2.4414059e-4f,
                         3.14159274f, -3.14159274f, 6.28318548f, 1.57079637f, 1e9f, 16777216.0f };
    for (float y : sp) bad += check(y);
    printf("%ld\n", bad);
    return 0;
}
'''


def _host_has_fma() -> bool:
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
    except (OSError, StopIteration):
        return False
    return "fma" in flags and "avx2" in flags


def test_ref_sincosf_equals_glibc_bit_for_bit():
    if not _host_has_fma():
        pytest.skip("host CPU without FMA + AVX2: its glibc dispatches to the unfused sincosf")
    simt = os.path.join(common.ROOT, "tests", "simt")
    csrc = os.path.join(common.ROOT, "nrsc5_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.cpp"), os.path.join(d, "t")
        open(src, "w").write(SRC)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-w", "-DHIPEMU", "-I" + simt, "-I" + csrc, "-o", exe, src, "-lm"])
        out = subprocess.check_output([exe, os.environ.get("NRSC5_SINCOSF_ARGS", "100000000")]).decode().strip()
    assert out == "0", f"{out} mismatches against glibc's sincosf"
