"""Build recipe for libnrsc5hip.so (hipcc, gfx950 only).  `python -m nrsc5_amd.build`.

The CPU-emulated twin used by the `-m "not gpu"` logic tests is built by `build_emu()` into
tests/simt/ -- test infrastructure, never loaded by the package itself."""
from __future__ import annotations

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nrsc5_amd", "csrc")
HIP_SOURCES = ["engine.hip", "k_decimate.hip", "k_acquire.hip", "k_mixfft.hip", "k_sync.hip", "k_decode.hip", "k_replay.hip", "k_am.hip", "k_l2.hip", "hdc_consumer.hip"]
LIB = os.path.join(ROOT, "nrsc5_amd", "libnrsc5hip.so")
EMU_LIB = os.path.join(ROOT, "tests", "simt", "libnrsc5hip_emu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    d.append(os.path.join(ROOT, "include", "nrsc5hip.h"))
    return d


def source_sha() -> str:
    """Fingerprint of the device sources: measurements stored under profiles/ carry it, so that bench.py can tell whether a PMC
    figure was collected from THIS tree."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(_deps()):
        if f.endswith((".hip", ".h")):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """Cross-compiles on a GPU-less host too (hipcc --offload-arch=gfx950)."""
    if force or _stale(LIB, _deps()):
        srcs = [os.path.join(CSRC, f) for f in HIP_SOURCES if os.path.exists(os.path.join(CSRC, f))]
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
               '-DNRSC5HIP_SOURCE_SHA="%s"' % source_sha(),        # nrsc5hip_source_sha(): callers can tell a stale build from the tree they run in
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", LIB] + srcs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_hip_diag(flags, out_name: str) -> str:
    """A diagnostic twin of the library with extra -D flags (e.g. -DNRSC5HIP_MIXFFT_PHASES: k_mixfft's phase timers), same sources and
    fingerprint, written next to the release library under its own name; tools/ scripts load it by path."""
    out = os.path.join(ROOT, "nrsc5_amd", out_name)
    srcs = [os.path.join(CSRC, f) for f in HIP_SOURCES if os.path.exists(os.path.join(CSRC, f))]
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", '-DNRSC5HIP_SOURCE_SHA="%s"' % source_sha(),
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", out] + list(flags) + srcs
    subprocess.check_call(cmd)
    return out


def _host_has_fma() -> bool:
    try:
        flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
    except (OSError, StopIteration):
        return False
    return "fma" in flags


def build_emu(force: bool = False) -> str:
    simt = os.path.join(ROOT, "tests", "simt")
    deps = _deps() + [os.path.join(simt, "hipemu.h"), os.path.join(simt, "hipemu.cpp")]
    if force or _stale(EMU_LIB, deps):
        srcs = [os.path.join(CSRC, f) for f in HIP_SOURCES if os.path.exists(os.path.join(CSRC, f))]
        # -mfma (where the host has it): __builtin_fma of ref_sincosf becomes the instruction instead of a libm call; -ffp-contract=off keeps every other a * b + c unfused
        fma = ["-mfma"] if _host_has_fma() else []
        cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-w"] + fma + ['-DNRSC5HIP_SOURCE_SHA="%s"' % source_sha(),
               "-I" + simt, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", EMU_LIB,
               os.path.join(simt, "hipemu.cpp")]
        for s in srcs:
            cmd += ["-x", "c++", s]
        subprocess.check_call(cmd)
    return EMU_LIB


if __name__ == "__main__":
    if "--emu" in sys.argv:
        print(build_emu(force=True))
    elif "--accurate-trig" in sys.argv:
        print(build_hip_diag(["-DNRSC5HIP_ACCURATE_TRIG"], "libnrsc5hip_acctrig.so"))
    elif "--cmul-unfused" in sys.argv:
        print(build_hip_diag(["-DNRSC5HIP_CMUL_UNFUSED"], "libnrsc5hip_unfused.so"))
    elif "--mixfft-noload" in sys.argv:
        print(build_hip_diag(["-DNRSC5HIP_MIXFFT_NOLOAD"], "libnrsc5hip_noload.so"))
    elif "--mixfft-phases" in sys.argv:
        print(build_hip_diag(["-DNRSC5HIP_MIXFFT_PHASES"], "libnrsc5hip_mixphases.so"))
    else:
        print(build_hip(force=True, verbose=True))
