"""GPU: BASELINE's 2048 streams end to end -- one batch pass of the engine (zero-copy, window pipeline, replay), the records of
every stream through ONE slim HDC consumer, and for every stream the packets == the NRSC5_EVENT_HDC sequence the unmodified
reference produces for that stream's capture; host memory stays bounded (the reference: one 22.9 MB nrsc5_t per stream)."""
import resource

import numpy as np
import pytest

from nrsc5_amd import engine as eng, synth

pytestmark = pytest.mark.gpu

S = 2048


def _rss_mb():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmRSS:"):
                return int(line.split()[1]) / 1024.0
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


def test_gpu_2048_streams_hdc_equals_reference_with_bounded_host_memory(hip_lib, reflib):
    import torch
    from oracle import ref
    prm = [(25.0, 400, 22.0), (-310.0, 3100, 18.0), (140.0, 1777, 25.0), (-60.0, 64, 20.0)]
    caps = [synth.fm_mp1_capture(3, seed=81 + k, cfo_hz=c, offset=o, snr_db=snr) for k, (c, o, snr) in enumerate(prm)]
    exp = []
    for cap in caps:
        log, _, _ = reflib.run(cap.iq, taps=ref.TAP_HDC)
        exp.append([(v["program"], v["count"], v["flags"], bytes(v["data"])) for kk, v in log if kk == "hdc"])
        assert len(exp[-1]) >= 32
    stride = max(c.iq.size for c in caps); stride += (-stride) % 256
    base = np.zeros((len(caps), stride), dtype=np.uint8)
    for k, c in enumerate(caps):
        base[k, :c.iq.size] = c.iq
    dev = torch.device("cuda", 0)
    which = torch.arange(S, device=dev) % len(caps)
    iq = torch.from_numpy(base).to(dev)[which].contiguous()            # [2048, stride] resident in HBM: 27 GB
    nbytes = np.array([caps[k % len(caps)].iq.size - caps[k % len(caps)].iq.size % 4 for k in range(S)], dtype=np.uint32)

    E = eng.Engine(max_streams=S, q15_capacity=2 * 71280, record_capacity=512, p1_slots=8, p1_async=True, l2_feedback=True,
                   batch_zero_copy=True, lib_path=hip_lib)
    E.batch_append_cu8(iq.data_ptr(), stride, nbytes)
    steps = E.batch_process(S)
    recs, counts, frames = E.batch_fetch_view(S, with_frames=False)
    assert steps >= 48

    rss0 = _rss_mb()
    H = eng.HdcConsumer(S, lib=E.lib)
    packets = 0
    for k in range(S):
        H.events.clear()
        eng.feed_hdc(E, H, k, recs[k, :counts[k]])
        got = [(p, c, f, d) for (s, p, c, f, d) in H.events if s == k]
        want = exp[k % len(caps)]
        assert len(got) == len(want) and all(a == b for a, b in zip(got, want)), (k, len(got), len(want))
        packets += len(got)
    H.events.clear()
    per_stream = H.host_bytes() / S
    growth = _rss_mb() - rss0
    print(f"2048 streams: {steps} block steps, {packets} HDC packets == reference, consumer state {per_stream / 1024:.1f} KB per stream, "
          f"process RSS grew {growth:.0f} MB while consuming (reference: 2048 x 22.9 MB = 46.9 GB)")
    assert packets >= S * 32
    assert per_stream < 128 * 1024, per_stream
    assert growth < S * 0.25, growth                                   # < 512 MB for 2048 streams
    H.close()
    E.close()
