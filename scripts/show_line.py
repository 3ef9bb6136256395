"""Helper: the judged fields of the bench lines in a directory, and the top kernels of the rocprofv3 stats beside them.  Usage: show_line.py DIR"""
import csv, glob, json, os, sys
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable:", e); continue
    r = j.get("roofline") or {}
    print(os.path.basename(f), "value %.1f %s, ms_per_step %.4f, launch mean %.4f median %.4f, roofline.frac %s, fp32 frac %s, cpu_baseline %s, forward %.4f ms, training step %.4f ms"
          % (j["value"], j["unit"], j["ms_per_step"], j["launch"]["mean_ms"], j["launch"]["median_ms"], r.get("frac"), r.get("frac_fp32_peak"),
             (j.get("cpu_baseline") or {}).get("value"), j["forward"]["launch_ms"], j["training_step"]["ms_per_step"]))
    print("   kernel:", (r.get("kernel") or "")[:600])
for f in sorted(glob.glob(os.path.join(d, "*kernel_stats.csv"))):
    for row in list(csv.DictReader(open(f)))[:4]:
        print(os.path.basename(f), row["Name"].split("::")[-1][:60], row["Calls"], "calls, avg %.1f us" % (float(row["AverageNs"]) / 1e3))
