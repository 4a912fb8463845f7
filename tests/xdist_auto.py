"""Early plugin (pytest.ini: `-p tests.xdist_auto`): the CPU suite (`-m "not gpu"`) is ~20 minutes of single-threaded work -- the
SIMT emulator runs every kernel's logic on the host -- and embarrassingly parallel, so it is spread over the cores with pytest-xdist
unless the caller chose a worker count.  The GPU suite (`-m gpu`) is left alone: its tests share one device."""
import os


def pytest_load_initial_conftests(early_config, parser, args):
    try:
        import xdist  # noqa: F401
    except Exception:
        return
    if any(a == "-n" or a.startswith("-n") or a.startswith("--numprocesses") or a == "--dist" for a in args):
        return
    expr = ""
    for i, a in enumerate(args):
        if a == "-m" and i + 1 < len(args):
            expr = args[i + 1]
        elif a.startswith("-m") and len(a) > 2:
            expr = a[2:]
    if expr.replace(" ", "") != "notgpu":
        return
    n = min(8, os.cpu_count() or 1)
    if n > 1:
        args[:] = list(args) + ["-n", str(n)]
