"""diagnostic: which decoded frames differ from the transmitted truth, and how (timing-dependent failure hunt)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
if os.environ.get("DIAG_LIB"):
    from nrsc5_amd import engine as _e
    _e.DEFAULT_LIB = os.path.join(ROOT, os.environ["DIAG_LIB"])
    _orig = _e.load_library
    def _ll(path=None):
        import ctypes
        lib = ctypes.CDLL(path or _e.DEFAULT_LIB)
        try:
            return _orig(path)
        except AttributeError:
            pass
        # older library: bind what exists
        import types
        src = open(_e.__file__).read()
        return _orig_partial(path or _e.DEFAULT_LIB)
    def _orig_partial(path):
        import ctypes, re
        lib = ctypes.CDLL(path)
        class Shim:
            def __getattr__(self, name):
                return getattr(lib, name)
        code = open(_e.__file__).read()
        body = code[code.index("    vp, ci = ctypes.c_void_p, ctypes.c_int"):code.index("    return lib\n\n\nEXPORTED_SYMBOLS")]
        ns = {"ctypes": ctypes, "lib": lib, "_Config": _e._Config, "HDC_CB": _e.HDC_CB}
        for line in body.splitlines():
            try:
                exec(line.strip(), ns)
            except AttributeError:
                pass
        return lib
    _e.load_library = _ll
if not os.environ.get("DIAG_LIB"):
    from nrsc5_amd import engine as _e2
    _e2.check_fresh()
args = bench.parse(["--no-cpu-baseline"] + sys.argv[1:])
dev = torch.device("cuda", 0)
S = args.streams or 256
W = bench.Fm(args, dev, 0, list(range(S)))
eng = W.eng
RECREATE = os.environ.get("DIAG_RECREATE") == "1"
if os.environ.get("DIAG_PROFILE"):
    W.E.profile(int(os.environ["DIAG_PROFILE"], 0))
for p in range(3):
    if RECREATE and p:
        W.E.close(); W.E = W.make_engine(S, 0, in_order=False)
    steps, (recs, counts, frames) = W.one_pass()
    bad_streams = 0; kinds = {}
    for k in range(S):
        r = recs[k, :counts[k]]
        truth = W.pool[k % args.payloads][0]
        p1r = r[(r["flags"] & eng.REC_P1) != 0]
        lost = int(((r["flags"] & eng.REC_LOST_SYNC) != 0).sum())
        desc = []
        for j, rr in enumerate(p1r):
            b = frames[k, int(rr["p1_slot"])].view(np.uint8)
            m = np.nonzero((truth == b[None, :]).all(axis=1))[0]
            if m.size:
                desc.append(str(int(m[0])))
            else:
                nd = int(np.unpackbits(truth ^ b[None, :], axis=1).sum(axis=1).min())
                desc.append(f"X({nd},ber{float(rr['ber']):.3f})")
        seq = ",".join(desc)
        ok = all(not d.startswith("X") for d in desc[1:])
        if lost or not ok:
            bad_streams += 1
            if bad_streams <= int(os.environ.get("DIAG_SHOW", "4")):
                print(f"pass {p} stream {k}: lost_sync {lost} blocks {len(r)} frames [{seq}]")
    print(f"pass {p}: steps {steps}, streams with lost sync or non-truth frames after the first: {bad_streams}", flush=True)
try:
    print("fwd stats", W.E.fwd_stats())
except Exception as ex:
    print("no fwd stats in this library")
