"""EXPERIMENT (DESIGN.md (c) limit 2, (h) item 0) -- a CPU-emulated twin of the library whose symbol kernel gives its NCO phasor the AMPLITUDE the reference's oscillator has:
the reference multiplies its float phase by the rounded pair phase_increment = (cosf, sinf)(dtheta) once per sample (acquire.c:168,250) and renormalises at the symbol's end
(acquire.c:252); |phase_increment| is not 1 but 1 + g with |g| up to 6e-8, so inside a symbol its oscillator's amplitude runs as (1 + g)^n -- up to 1.3e-4 by sample 2159,
deterministic, and 100 x larger than the recurrence's rounding noise.  The library's closed-form phasor has amplitude 1.  This script builds the twin from a patched COPY of
nrsc5_amd/csrc (the tree and its fingerprint stay as they are): prepare_values also returns g, the symbol kernel scales the work-item's start phasor by 1 + tid g and the
128-sample step by 1 + 128 g (three instructions per work-item).  Result (tools/cpu_cfo_lock_sweep.py --emu-lib /tmp/nrsc5_emu_growth/libemu_growth.so):
profiles/r04_cfo_lock_transients.txt -- 5 of 900 CFO-search locks deviate instead of 18.
    python tools/build_emu_nco_growth.py   ->   /tmp/nrsc5_emu_growth/libemu_growth.so"""
import os, shutil, subprocess, sys
sys.path.insert(0, '/root/repo')
from nrsc5_amd import build
ROOT='/root/repo'
os.makedirs('/tmp/nrsc5_emu_growth', exist_ok=True)
d='/tmp/nrsc5_emu_growth/csrc'
shutil.rmtree(d, ignore_errors=True); shutil.copytree(build.CSRC, d)
def patch(f, subs):
    p=os.path.join(d,f); s=open(p).read()
    for old,new in subs:
        assert s.count(old)==1, (f, old[:70], s.count(old))
        s=s.replace(old,new)
    open(p,'w').write(s)
patch('prepare_block.h', [
 ("    double dtheta, theta;       // NCO step and start phase of the block\n", "    double dtheta, theta;       // NCO step and start phase of the block\n    double growth;              // |phase_increment| - 1: the reference's oscillator grows / shrinks by this much per sample until it is renormalised at the symbol's end\n"),
 ("    p.samperr = 0; p.prev_angle = st.prev_angle; p.to_coarse = 0; p.dtheta = st.dtheta; p.theta = st.theta;\n", "    p.samperr = 0; p.prev_angle = st.prev_angle; p.to_coarse = 0; p.dtheta = st.dtheta; p.theta = st.theta; p.growth = 0.0;\n"),
 ("    // phase *= e^{-i (1080 - samperr) angle / 2048}            (acquire.c:166)\n", "    p.growth = sqrt((double)inc_c * (double)inc_c + (double)inc_s * (double)inc_s) - 1.0;\n    // phase *= e^{-i (1080 - samperr) angle / 2048}            (acquire.c:166)\n"),
 ("    st.dtheta = p.dtheta;\n", "    st.dtheta = p.dtheta;\n    st.growth = p.growth;\n"),
])
patch('nrsc5_dev.h', [
 ("    int px_nch, px_slot, px_record;\n};", "    int px_nch, px_slot, px_record;\n    double growth;\n};"),
])
patch('k_mixfft.hip', [
 ("struct SymParams { long long a00; double dtheta, theta; int active; };", "struct SymParams { long long a00; double dtheta, theta; int active; double growth; };"),
 ("            sh_sp.active = p.active; sh_sp.a00 = (st.rd - st.base) + p.samperr; sh_sp.dtheta = p.dtheta; sh_sp.theta = p.theta;\n        }\n        __syncthreads();\n        sp = sh_sp;\n    } else {\n        sp.active = st.active; sp.a00 = (st.rd - st.base) + st.samperr_cur; sp.dtheta = st.dtheta; sp.theta = st.theta;\n    }\n    sp.active = wave_uniform(sp.active); sp.a00 = uniform64(sp.a00); sp.dtheta = uniform64(sp.dtheta); sp.theta = uniform64(sp.theta);   // scalar registers",
  "            sh_sp.active = p.active; sh_sp.a00 = (st.rd - st.base) + p.samperr; sh_sp.dtheta = p.dtheta; sh_sp.theta = p.theta; sh_sp.growth = p.growth;\n        }\n        __syncthreads();\n        sp = sh_sp;\n    } else {\n        sp.active = st.active; sp.a00 = (st.rd - st.base) + st.samperr_cur; sp.dtheta = st.dtheta; sp.theta = st.theta; sp.growth = st.growth;\n    }\n    sp.active = wave_uniform(sp.active); sp.a00 = uniform64(sp.a00); sp.dtheta = uniform64(sp.dtheta); sp.theta = uniform64(sp.theta);   // scalar registers"),
 ("        ph = emul(unit_phasor((float)a0p), cf_make(1.0f / 32767.0f, 1.0f / 32767.0f));\n", "        ph = emul(unit_phasor((float)a0p), cf_make(1.0f / 32767.0f, 1.0f / 32767.0f));\n        if (GROWTH_ON) { const float g0 = (float)(1.0 + (double)tid * sp.growth); ph = emul(ph, cf_make(g0, g0)); }\n"),
 ("#pragma unroll 1\n    for (int i = 0; i < SPW; i++) {\n        const int sym = sym0 + i;", "    if (GROWTH_ON) { const float g1 = (float)(1.0 + 128.0 * sp.growth); stp = emul(stp, cf_make(g1, g1)); }\n#pragma unroll 1\n    for (int i = 0; i < SPW; i++) {\n        const int sym = sym0 + i;"),
 ("namespace nrsc5 {\n\n__device__ inline int stream_of(const int *ids, int idx) { return ids ? ids[idx] : idx; }", "namespace nrsc5 {\n#ifndef GROWTH_ON\n#define GROWTH_ON 1\n#endif\n\n__device__ inline int stream_of(const int *ids, int idx) { return ids ? ids[idx] : idx; }"),
])
simt=os.path.join(ROOT,'tests','simt')
out='/tmp/nrsc5_emu_growth/libemu_growth.so'
srcs=[os.path.join(d,f) for f in build.HIP_SOURCES]
cmd=["g++","-O2","-g","-std=c++17","-fPIC","-shared","-ffp-contract=off","-w",'-DNRSC5HIP_SOURCE_SHA="%s"'%build.source_sha(),"-I"+simt,"-I"+os.path.join(ROOT,"include"),"-I"+d,"-o",out,os.path.join(simt,"hipemu.cpp")]
for s in srcs: cmd += ["-x","c++",s]
subprocess.check_call(cmd); print(out)
