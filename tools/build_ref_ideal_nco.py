"""EXPERIMENT for DESIGN.md (c) limit 2 -- test infrastructure, build container only (/root/reference must be present; nothing of it is stored here).
Builds oracle/_ref/libnrsc5_ref_sse_nco.so: the reference's translation units as oracle/Makefile compiles them, EXCEPT that acquire.c goes through four textual
substitutions on its way into the compiler (stdin; no patched file is written): inside each OFDM symbol the oscillator is advanced in DOUBLE precision from the float
state and written back to the float state at the symbol's end -- an ideal NCO with the same rounded increment, the same renormalisation, the same everything else.
The reference advances its oscillator by a float complex recurrence (acquire.c:242,250: 2160 dependent multiplications per symbol, ~3e-6 rad of rounding drift);
this design evaluates the phase in closed form (k_mixfft.hip).  tools/cpu_cfo_lock_sweep.py --nco compares the UNMODIFIED reference with this variant: if the
locks after a CFO search deviate as often as they do against the library, the oscillator's drift -- not the FFT -- is what the chaotic part of the search amplifies."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("REF", "/root/reference")
ORC = os.path.join(ROOT, "oracle")
SUBST = [
    ("        for (j = 0; j < st->fftcp; ++j)\n        {\n            float complex sample = st->phase * st->buffer",
     "        double complex nco_d = st->phase, inc_d = phase_increment;\n        for (j = 0; j < st->fftcp; ++j)\n        {\n            float complex sample = (float complex)nco_d * st->buffer"),
    ("            st->phase *= phase_increment;\n", "            nco_d *= inc_d;\n"),
    ("        st->phase /= cabsf(st->phase);\n", "        st->phase = (float complex)nco_d;\n        st->phase /= cabsf(st->phase);\n"),
]
TUS = "acquire decode here_images input nrsc5 output pids rtltcp sync firdecim_q15 conv_dec rs_init rs_decode unicode".split()


def main():
    src = open(os.path.join(REF, "src", "acquire.c")).read()
    for old, new in SUBST:
        assert src.count(old) == 1, old
        src = src.replace(old, new)
    cflags = ["-O3", "-std=gnu11", "-fPIC", "-D_GNU_SOURCE", "-msse2", "-msse3", "-mssse3", "-DHAVE_SSE2", "-DHAVE_SSE3", "-DGIT_COMMIT_HASH=\"oracle\"",
              "-Iref_shim", "-I../integration/shim", "-I" + os.path.join(REF, "include"), "-I" + os.path.join(REF, "src")]
    os.makedirs(os.path.join(ORC, "_ref", "obj_nco"), exist_ok=True)
    obj = os.path.join(ORC, "_ref", "obj_nco", "acquire.o")
    subprocess.run(["cc"] + cflags + ["-x", "c", "-", "-c", "-o", obj], input=src.encode(), cwd=ORC, check=True)
    subprocess.check_call(["make", "-C", ORC, "_ref/libnrsc5_ref_sse.so"])           # the other objects
    objs = [obj] + [os.path.join("_ref", "obj_sse", t + ".o") for t in TUS if t != "acquire"]
    mk = open(os.path.join(ORC, "Makefile")).read().replace("\\\n", " ")
    wraps = [l for l in mk.split("\n") if l.startswith("WRAPS")][0].split(":=", 1)[1].split()
    wrapflags = ["-Wl,--wrap=" + w for w in wraps]
    out = os.path.join("_ref", "libnrsc5_ref_sse_nco.so")
    subprocess.check_call(["cc"] + cflags + ["-shared", "-o", out] + objs + ["ref_shim/stubs.c", "../integration/shim/rtlsdr_stubs.c", "ref_shim/ref_harness.c", "ref_shim/frame_indexed.c", "cpu_fft.c"]
                          + wrapflags + ["-lm", "-lpthread"], cwd=ORC)
    print(os.path.join(ORC, out))


if __name__ == "__main__":
    main()
