"""Aggregates one rocprofv3 --pmc pass of SQ counters over ONE pass of `python bench.py` into per-kernel issue statistics and
the VALU roofline input of the bench line (profiles/sq_latest.json, stamped with the fingerprint of the device sources).

MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* tick in QUAD-cycles; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY
~= WAVE_CYCLES (disjoint), so the fractions say where a wave's life goes: parked on s_waitcnt / barriers, stalled at issue, or
issuing; ACTIVE_INST_VALU / WAVE_CYCLES is the VALU share of it.  VALU roofline of the pass: SIMD cycles in which a VALU
instruction issued = 4 x sum(SQ_ACTIVE_INST_VALU) over every kernel of the pass, against 1024 SIMDs x clock x pass time (the
bench divides by ITS pass time: counter collection serialises the kernels, so the profiled wall time means nothing).
Clock: GRBM_GUI_ACTIVE / kernel duration of the long single-wave trellis kernel when the kernel trace is present, else 2.4 GHz."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
COUNTERS = ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
from profiles.collect_pmc import CLASSES, kernel_base                                  # kernel-name fragments -> the classes of nrsc5hip_profile


def main(dirname, out, workload="fm"):
    from nrsc5_amd import build
    tot, calls, gui_by_dispatch = {}, {}, {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "nrsc5::" not in row["Kernel_Name"].split("(")[0]:
                continue
            name = "nrsc5::" + kernel_base(row["Kernel_Name"])
            c = row["Counter_Name"]
            d = tot.setdefault(name, {})
            d[c] = d.get(c, 0.0) + float(row["Counter_Value"])
            if c == "SQ_WAVES":
                calls[name] = calls.get(name, 0) + 1
            if c == "GRBM_GUI_ACTIVE":
                gui_by_dispatch[row.get("Dispatch_Id")] = (name, float(row["Counter_Value"]))
    # effective clock from the kernel trace of the same run, if there is one
    clock, clock_note = 2.4, "nominal 2.4 GHz (no kernel trace / GRBM_GUI_ACTIVE in this collection)"
    dur = {}
    for f in glob.glob(os.path.join(dirname, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row.get("Dispatch_Id")] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
    pairs = [(g, dur[d]) for d, (name, g) in gui_by_dispatch.items() if d in dur and dur[d] > 200e-6 and g > 0]
    if pairs:
        # the counter is summed over the chip's 8 XCDs (one GRBM each): 8 x the shader clock while the kernel has all of them busy
        clock = sum(g for g, _ in pairs) / sum(t for _, t in pairs) / 1e9 / 8.0
        clock_note = f"GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / kernel duration over the {len(pairs)} dispatches longer than 200 us of the profiled (serialised) pass"
        if not 1.0 < clock < 3.0:
            clock, clock_note = 2.4, f"nominal 2.4 GHz (GRBM_GUI_ACTIVE / duration gave {clock:.2f} GHz: not plausible)"
    res = {}
    for name, d in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
        wc = d.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        res[name] = {"launches": calls.get(name, 0), "waves": d.get("SQ_WAVES", 0.0),
                     "wave_quad_cycles": d.get("SQ_WAVE_CYCLES", 0.0),
                     "frac_issuing": round(d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
                     "frac_issuing_valu": round(d.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4),
                     "frac_issue_stall": round(d.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                     "valu_insts_per_wave": round(d.get("SQ_INSTS_VALU", 0.0) / max(d.get("SQ_WAVES", 0.0), 1.0), 1),
                     "valu_simd_cycles": 4.0 * d.get("SQ_ACTIVE_INST_VALU", 0.0),
                     "sq_busy_cycles": d.get("SQ_BUSY_CYCLES", 0.0), "grbm_gui_active": d.get("GRBM_GUI_ACTIVE", 0.0)}
    per_class, per_class_res = {}, {}
    for cls, frags in CLASSES.items():
        ks = [k for k in res if any(k.replace("nrsc5::", "").startswith(x) for x in frags)]
        v = sum(res[k]["valu_simd_cycles"] for k in ks)
        if v:
            per_class[cls] = v
            wq = sum(res[k]["wave_quad_cycles"] for k in ks)
            per_class_res[cls] = round(v / (4.0 * wq), 4) if wq else None
    out_d = {"source_sha": build.source_sha(), "workload": workload, "passes": 1, "counters": COUNTERS, "clock_ghz": round(clock, 3), "clock_note": clock_note,
             "valu_simd_cycles_per_pass": sum(v["valu_simd_cycles"] for v in res.values()),
             "per_class_valu_simd_cycles": per_class, "per_class_valu_busy_while_resident": per_class_res, "kernels": res}
    json.dump(out_d, open(out, "w"), indent=1)
    print(json.dumps({k: out_d[k] for k in ("source_sha", "clock_ghz", "valu_simd_cycles_per_pass", "per_class_valu_simd_cycles", "per_class_valu_busy_while_resident")}))
    for k, v in list(res.items())[:14]:
        print(k, v)


if __name__ == "__main__":
    main(*sys.argv[1:4])
