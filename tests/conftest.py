import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) is ~20 minutes of single-threaded work -- the SIMT emulator runs every kernel's logic on the
    host -- and embarrassingly parallel: spread it over the cores with pytest-xdist unless the caller chose a worker count.  The GPU
    suite (`-m gpu`) is left alone: its tests share one device.  (Round 4 did this with an early `-p tests.xdist_auto` plugin in
    pytest.ini, which made a bare `pytest` fail to start: the package was not importable that early.  A conftest hook needs nothing.)"""
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER"):
        return
    try:
        import xdist  # noqa: F401
    except Exception:
        return
    opt = config.option
    if not hasattr(opt, "numprocesses") or opt.numprocesses is not None or getattr(opt, "dist", "no") != "no":
        return
    if (getattr(opt, "markexpr", "") or "").replace(" ", "") != "notgpu":
        return
    if getattr(opt, "usepdb", False) or getattr(opt, "collectonly", False):
        return
    n = min(8, os.cpu_count() or 1)
    if n > 1:
        opt.numprocesses = n
        opt.dist = "load"
        opt.tx = ["popen"] * n


def pytest_sessionstart(session):
    """GPU runs: initialise torch's HIP runtime BEFORE anything loads libnrsc5hip.so.  torch ships its own libamdhip64 (ROCm 7.0 here); the library links /opt/rocm's (7.2).
    Whichever is loaded first serves both; with the system runtime first, torch's device enumeration fails ("No HIP GPUs are available") in any later test that uses torch on
    the device -- seen when the GPU test files ran in another order than the alphabetical one (tests/test_gpu_batch256.py first).  bench.py initialises torch first for the same reason."""
    expr = (getattr(session.config.option, "markexpr", "") or "").replace(" ", "")
    if "gpu" in expr and "notgpu" not in expr:
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement (oracle/liboracle.so), compiled on demand with gcc."""
    from oracle import port
    return port.Oracle()


@pytest.fixture(scope="session")
def reflib():
    """The unmodified reference built from /root/reference (only in the build container, or when
    oracle/_ref/ travelled with the snapshot)."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref/libnrsc5_ref.so not built (needs /root/reference)")
    return ref.RefLib()


@pytest.fixture(scope="session")
def emu_lib():
    """CPU SIMT-emulated build of the HIP sources: logic tests only, never a fallback."""
    from nrsc5_amd import build
    return build.build_emu()


@pytest.fixture(scope="session")
def hip_lib():
    """The real gfx950 library; GPU tests must run THIS, so a missing build is an error."""
    from nrsc5_amd import build, engine
    build.build_hip()                                          # rebuilds only when a source is newer than the library
    try:
        engine.check_fresh()
    except engine.Nrsc5HipError:                               # same mtimes, other content (a checkout): rebuild
        build.build_hip(force=True)
        engine.check_fresh()
    return engine.DEFAULT_LIB


@pytest.fixture(scope="session")
def captures():
    """Golden capture inputs, regenerated deterministically (float64 numpy) and cached per session."""
    from nrsc5_amd import synth
    from tests import common
    cache = {}

    def get(name):
        if name not in cache:
            if name in common.GOLDEN_AM_CASES:
                from nrsc5_amd import synth_am
                cache[name] = synth_am.am_ma1_capture(**common.GOLDEN_AM_CASES[name])
            else:
                cache[name] = synth.fm_mp1_capture(**common.GOLDEN_CASES[name])
        return cache[name]
    return get
