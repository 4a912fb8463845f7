"""CPU only: the UNMODIFIED reference against ITSELF on another FFT (oracle/_ref/libnrsc5_ref_sse.so: float Stockham stand-in; libnrsc5_ref_sse_dp.so: a double-precision FFT rounded
once -- outputs ~2e-7 of the largest bin apart), on the captures of the GPU fuzz batch (tests/test_gpu_batch256.py's generator: CFO uniform in +-3 kHz, every 4th stream through an
impaired channel; stream ids from the command line), compared under the very rule the device is held to (bench.compare_with_reference: strict rule of tests/common.py + the counted
classes).  The reference links fftw3f, whose results depend on the plan and the host's SIMD level: what this measures is how far the reference is from itself when ONLY its FFT's last
bits change.        python tools/cpu_ref_self_fft_batch.py [base=100256] [streams=256] [processes=8]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def work(gs):
    import torch
    torch.set_num_threads(1)
    from nrsc5_amd import synth_torch as stt
    from tests import test_gpu_batch256 as t
    from oracle import ref
    import bench
    dev = torch.device("cpu")
    p1, pids, m = stt.payload_stream(t.N_FRAMES, seed=900 + gs % 8)
    out = stt.receive_cu8(stt.modulate(m, dev), t.batch_params(gs), tail=8640)
    iq = out.cpu().numpy(); iq = iq[:iq.shape[0] - iq.shape[0] % 4]
    A = ref.RefLib(path=os.path.join(ROOT, "oracle", "_ref", "libnrsc5_ref_sse.so"))
    B = ref.RefLib(path=os.path.join(ROOT, "oracle", "_ref", "libnrsc5_ref_sse_dp.so"))
    la, lb = A.run(iq, mode=0)[0], B.run(iq, mode=0)[0]
    del bench.TRANSIENT_DETAILS[:]
    diffs, nex, mb, ntr = bench.compare_with_reference(la, lb, False)
    locks = sum(1 for k, v in la if k == "sync")
    return gs, diffs[:4], ntr, list(bench.TRANSIENT_DETAILS[:3]), sorted({bench._diff_class(d) for d in diffs}), locks


if __name__ == "__main__":
    from multiprocessing import get_context
    base = int(sys.argv[1]) if len(sys.argv) > 1 else 100256
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    nproc = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    t0 = time.time()
    with get_context("spawn").Pool(nproc) as p:
        res = p.map(work, list(range(base, base + n)), chunksize=2)
    failing = [r for r in res if r[1]]
    transient = [r for r in res if not r[1] and r[2]]
    classes = {}
    for r in failing:
        for c in r[4]:
            classes[c] = classes.get(c, 0) + 1
    print(json.dumps({"streams": n, "base": base, "seconds": round(time.time() - t0, 1), "sync_events": sum(r[5] for r in res),
                      "strict": n - len(failing) - len(transient), "counted_transient": len(transient), "failing": len(failing), "failing_by_class": classes,
                      "failing_streams": [{"stream": r[0], "diffs": r[1]} for r in failing[:8]],
                      "transient_streams": [{"stream": r[0], "details": r[3]} for r in transient[:8]]}, indent=1))
