"""Regenerates tests/golden/l2_reference_taps.json from the UNMODIFIED reference (oracle/_ref): every L2 test frame of
nrsc5_amd/synth_l2.py is handed to the reference's own frame_push and the output_align / output_push calls it makes
(tapped with --wrap) are recorded, payloads as CRC-32.  Run in the build container only:

    python tests/golden/make_golden_l2.py

The GPU box has no /root/reference: there the device index is checked against this file directly."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import common  # noqa: E402
from nrsc5_amd import synth_l2  # noqa: E402
from oracle import ref  # noqa: E402


def main():
    R = ref.RefLib()
    out = {}
    for nbits in sorted(synth_l2.LAYOUT):
        cases = []
        for name, bits, safe in synth_l2.test_frames(nbits):
            if not safe or name.startswith("fixed"):
                continue                          # reference undefined on it / fixed-data walk is not indexed
            log = R.l2_frames([bits])[0]
            cases.append({"name": name, "bits_sha1": hashlib.sha1(bits.tobytes()).hexdigest(),
                          "taps": common.l2_taps_digest(common.l2_reference_taps(log))})
        out[str(nbits)] = cases
        print(nbits, len(cases), "cases,", sum(len(c["taps"]) for c in cases), "taps")
    json.dump(out, open(os.path.join(HERE, "l2_reference_taps.json"), "w"), separators=(",", ":"))


if __name__ == "__main__":
    main()
