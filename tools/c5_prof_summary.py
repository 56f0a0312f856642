#!/usr/bin/env python3
"""Summarise the config-5 passes of tools/profile_round.sh (the closed MPC loop as ONE launch: B = 8192, T = 100, N = 30) into
profiles/<tag>_kernel_stats_config5.csv and profiles/<tag>_pmc_config5.csv.

  python tools/c5_prof_summary.py gpurun_out/prof_stats_c5 gpurun_out/prof_c5_sq r05
"""
import csv, glob, os, shutil, sys
from collections import defaultdict

stats_dir, pmc_dir, tag = sys.argv[1:4]
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
st = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)
if not st:
    sys.exit(f"no kernel stats under {stats_dir}")
rows = list(csv.DictReader(open(st[0])))
out = os.path.join(REPO, "profiles", f"{tag}_kernel_stats_config5.csv")
with open(out, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:120]
        w.writerow(r)
for r in rows:
    if "k_rollout_mpc" in r["Name"]:
        print("k_rollout_mpc: calls %s, average %.3f ms, max %.3f ms" % (r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MaxNs"]) / 1e6))
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_rollout_mpc" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
if not acc:
    sys.exit(f"no k_rollout_mpc counter rows under {pmc_dir}: the counter file is left as it is")
outp = os.path.join(REPO, "profiles", f"{tag}_pmc_config5.csv")
with open(outp, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "sum_over_dispatches", "dispatch_rows"])
    for c, (s, n) in sorted(acc.items()):
        w.writerow(["k_rollout_mpc", c, "%.0f" % s, n])
print(open(outp).read())
g = lambda c: acc[c][0]
if g("SQ_WAVE_CYCLES"):
    print("VALU active / wave cycles %.3f | waiting / wave cycles %.3f | VALU wave-instructions %.4e (sum of both dispatches: warm-up 2 steps + 100 steps)"
          % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_INSTS_VALU")))
