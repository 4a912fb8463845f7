"""After `gpurun -- 'bash tools/gpu_r6_final.sh TAG'`: copy the run's summaries from gpurun_out/ into profiles/ (the tracked copies the bench line and DESIGN.md (f) cite)
and print the figures DESIGN.md / README.md quote.   python tools/r6_install_records.py TAG"""
import csv, json, shutil, sys
tag = sys.argv[1]
for src, dst in ((f"gpurun_out/{tag}_kernel_stats_fm.csv", "profiles/r06_kernel_stats_fm_256x20s.csv"), (f"gpurun_out/{tag}_kernel_stats_am-cs16.csv", "profiles/r06_kernel_stats_am-cs16_256x61s.csv"),
                 ("gpurun_out/kernel_stats_latest.json", "profiles/kernel_stats_latest.json"), ("gpurun_out/traffic_fm.json", "profiles/traffic_latest.json"), ("gpurun_out/sq_fm.json", "profiles/sq_latest.json"),
                 (f"gpurun_out/{tag}_bench.json", "profiles/r06_bench_fm.json"), (f"gpurun_out/{tag}_trace_summary.txt", "profiles/r06_trace_final.txt")):
    shutil.copyfile(src, dst)
out = {}
for base, f in ((0, f"gpurun_out/{tag}_bench.json"), (256, f"gpurun_out/{tag}_parity_base256.json"), (512, f"gpurun_out/{tag}_parity_base512.json")):
    d = json.load(open(f)); out[f"stream_base_{base}"] = {"ms_per_step": d["ms_per_step"], "parity": d["parity"]["reference_equality_rank0"], "parity_failures": d["parity_failures"]}
json.dump(out, open("profiles/r06_parity_all_256_streams.json", "w"), indent=1)
for k in ("kernel_stats_latest", "traffic_latest", "sq_latest"):
    print(k, json.load(open(f"profiles/{k}.json")).get("source_sha"))
b = json.load(open("profiles/r06_bench_fm.json")); r = b["roofline"]
print("ms", b["ms_per_step"], b["ms_per_step_median"], b["ms_per_step_min_max"], "x", b["x_realtime"], "value", b["value"], "failures", b["parity_failures"])
print("frac", r["frac"], r["avg_launch_ms"], "rocprof", r["rocprof"]["avg_launch_us"], r["rocprof"]["calls"], r["rocprof"]["achieved_GBps"], r["rocprof"]["frac"])
print("whole", r["whole_pass"], "valu", r["frac_valu"])
k = r["kernel_by_device_time"]; print("by_dev", k["kernel"], k["device_ms_per_pass"], k["launches_per_pass"], k["avg_launch_ms"], k["achieved"], k["frac"])
w = r["whole_path_traffic"]; print("traffic", r["traffic"], w["hbm_bytes_per_pass"], w["over_algorithmic"], w["counter_GBps"], w["counter_frac_of_peak"])
print(r["valu"]["per_class_frac_of_pass"]); print(r["valu"]["per_class_valu_busy_while_resident"])
print("single", b["single_stream"]["x_realtime"], "inorder", b["in_order"]["ms_per_step"])
ds = b["dropin"]["dropin_strict_delivery"]; print("dropin", b["dropin"]["dropin"]["x_realtime"], b["dropin"]["dropin"]["x_realtime_min_max"], ds["x_realtime"], ds["x_realtime_min_max"], ds["breakdown_us_per_block"])
print("cpu", b["cpu_baseline"]["value"], b["cpu_baseline"]["x_realtime"], b["cpu_baseline"]["all_cores"]["x_realtime"])
print("am", b["config4"]["am_cs16"]["ms_per_step"], b["config4"]["am_cs16"]["x_realtime"], "mixed", b["config4"]["mixed"]["ms_per_step"], b["config4"]["mixed"]["x_realtime"])
print(r["device_ms_per_pass"])
for f in ("profiles/r06_kernel_stats_fm_256x20s.csv", "profiles/r06_kernel_stats_am-cs16_256x61s.csv"):
    for row in csv.reader(open(f)):
        if row and "nrsc5::" in row[0] and any(x in row[0] for x in ("k_mixfft", "k_sync<", "k_p1_forward", "k_p1_tbwalk", "k_p1_traceback(", "k_nco_exact", "k_am_decode_fwd", "k_am_block", "k_am_decode_tb", "k_am_decode_finish")):
            print(row[0][:44], row[1], row[3][:8])
for p in ("256", "512"):
    q = out[f"stream_base_{p}"]["parity"]; print("base", p, out[f"stream_base_{p}"]["ms_per_step"], q["streams_equal_under_the_strict_rule"], q["streams_with_transient_loop_state_deviation"], q["streams_failing_by_class"])
