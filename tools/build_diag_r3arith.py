"""Diagnostic twin of the library with the symbol kernel's FLOAT SEQUENCE of round 3 / the first half of round 4 (three-instruction complex product, every sample divided
by 32767 as the reference does) -- to tell whether a parity finding comes from this round's arithmetic change (fused products, scale carried by the phasor) or not.
Built from a patched COPY of nrsc5_amd/csrc in a temporary directory: the tree, and with it the source fingerprint the records are stamped with, stays as it is.
    python tools/build_diag_r3arith.py   ->   nrsc5_amd/libnrsc5hip_r3arith.so"""
import os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nrsc5_amd import build

PATCHES = [
    # complex product: the three-instruction form
    ('''    cf p;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(p) : "v"(a), "v"(b));
    return p;''', '''    return cmul3(a, b);'''),
    ('''    cf p;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "s"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(p) : "v"(a), "s"(b));
    return p;''', '''    return cmul3(a, b);'''),
    # Q15 -> float per sample (cq15_to_cf's correctly rounded quotient), phasor unscaled
    ('''// forward 4-point DFT in place, natural order out''', '''__device__ __forceinline__ cf q15_to_cf(cf x)
{
    const cf r = {1.0f / 32767.0f, 1.0f / 32767.0f}, d = {32767.0f, 32767.0f};
    const cf q0 = x * r;
    const cf e = __builtin_elementwise_fma(-q0, d, x);
    return __builtin_elementwise_fma(e, r, q0);
}

// forward 4-point DFT in place, natural order out'''),
    ('''        ph = emul(unit_phasor((float)a0p), cf_make(1.0f / 32767.0f, 1.0f / 32767.0f));''', '''        ph = unit_phasor((float)a0p);'''),
    ('''            if (RAW) return lds[j];                                // the tile holds Q15 integers, conjugated
            const c16 s16 = win[j];
            return cf_make((float)s16.r, -(float)s16.i);''', '''            if (RAW) return q15_to_cf(lds[j]);
            const c16 s16 = win[j];
            return q15_to_cf(cf_make((float)s16.r, -(float)s16.i));'''),
]


def main():
    with tempfile.TemporaryDirectory() as d:
        csrc = os.path.join(d, "csrc")
        shutil.copytree(build.CSRC, csrc)
        p = os.path.join(csrc, "k_mixfft.hip")
        s = open(p).read()
        for old, new in PATCHES:
            assert s.count(old) == 1, old[:60]
            s = s.replace(old, new)
        open(p, "w").write(s)
        out = os.path.join(ROOT, "nrsc5_amd", "libnrsc5hip_r3arith.so")
        srcs = [os.path.join(csrc, f) for f in build.HIP_SOURCES]
        cmd = [build.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", '-DNRSC5HIP_SOURCE_SHA="%s"' % build.source_sha(),
               "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-o", out] + srcs
        subprocess.check_call(cmd)
        print(out)


if __name__ == "__main__":
    main()
