"""Reads the counter files of tools/time_hist_clock.py runs (directories given as mode=dir) and prints, per mode and kernel, the
dispatches' mean time and the clock GRBM_GUI_ACTIVE / 8 / time (MI355X_MICROARCH.md, DVFS give-back), first and second half of the run."""
import csv, glob, os, sys
for arg in sys.argv[1:]:
    mode, d = arg.split("=", 1)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(mode, "no counter file"); continue
    rows = {}
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        rows.setdefault(k, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    for k in ("hist_lanes_kernel", "tree_wave_kernel", "pack_kernel", "decode_sub_kernel", "decode_fast_kernel"):
        v = sorted(rows.get(k, []))
        if len(v) < 8: continue
        v = v[4:]                                               # (the first calls: cold)
        def stat(part):
            t = sum(x[2] for x in part) / len(part); clk = sum(x[1] for x in part) / 8.0 / sum(x[2] for x in part)
            return "%.1f us at %.2f GHz" % (t / 1e3, clk)
        ts = sorted(x[2] for x in v)
        print("%-9s %-20s n=%d  all: %s   fastest quarter: %.1f us  slowest quarter: %.1f us" % (
            mode, k, len(v), stat(v), sum(ts[: len(ts) // 4]) / (len(ts) // 4) / 1e3, sum(ts[-(len(ts) // 4):]) / (len(ts) // 4) / 1e3))
