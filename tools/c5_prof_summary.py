#!/usr/bin/env python3
"""Summarise the config-5 passes of tools/profile_round.sh (the closed MPC loop as ONE launch: B = 8192, T = 100, N = 30) into
profiles/<tag>_kernel_stats_config5.csv and profiles/<tag>_pmc_config5.csv.

  python tools/c5_prof_summary.py gpurun_out/prof_stats_c5 gpurun_out/prof_c5_sq [gpurun_out/prof_c5_fp] r05

Also writes profiles/config5_counters.json (issued fp64 FLOPs and VALU wave-instructions per aircraft-step of the ONE-launch loop, from
the launch of 100 steps alone) for bench.py's roofline.config5.
"""
import csv, glob, os, shutil, sys
from collections import defaultdict

stats_dir, tag = sys.argv[1], sys.argv[-1]
pmc_dirs = sys.argv[2:-1]
pmc_dir = pmc_dirs[0]
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
big = defaultdict(float)          # the 100-step launch alone: the dispatch with the larger counter sum
for d in pmc_dirs:
    per_disp = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_rollout_mpc" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
                per_disp[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    if per_disp:
        last = max(per_disp, key=lambda k: sum(per_disp[k].values()))
        big.update(per_disp[last])
if not acc:
    sys.exit(f"no k_rollout_mpc counter rows under {pmc_dir}: the counter file is left as it is")
outp = os.path.join(REPO, "profiles", f"{tag}_pmc_config5.csv")
with open(outp, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "sum_over_dispatches", "dispatch_rows"])
    for c, (s, n) in sorted(acc.items()):
        w.writerow(["k_rollout_mpc", c, "%.0f" % s, n])
print(open(outp).read())
if big.get("SQ_INSTS_VALU_FMA_F64"):
    import json
    B, T = 8192, 100
    ms = max(float(r["MaxNs"]) for r in rows if "k_rollout_mpc" in r["Name"]) / 1e6
    vec = 64.0 * (2 * big["SQ_INSTS_VALU_FMA_F64"] + big["SQ_INSTS_VALU_ADD_F64"] + big["SQ_INSTS_VALU_MUL_F64"])
    mfma = 2048.0 * big.get("SQ_INSTS_VALU_MFMA_F64", 0.0)            # v_mfma_f64_16x16x4_f64: 16 x 16 x 4 x 2 FLOP
    rec = {"batch": B, "steps": T, "hzn": 30, "kernel": "k_rollout_mpc", "launch_ms_under_rocprofv3": ms,
           "vector_f64_flop_per_launch": vec, "mfma_flop_per_launch": mfma, "issued_flop_per_launch": vec + mfma,
           "issued_tflops": (vec + mfma) / (ms * 1e-3) / 1e12, "frac_of_fp64_peak_78p6": (vec + mfma) / (ms * 1e-3) / 78.6e12,
           "valu_wave_instructions_per_launch": big.get("SQ_INSTS_VALU"), "lds_bank_conflict_fraction": big["SQ_LDS_BANK_CONFLICT"] / big["SQ_LDS_IDX_ACTIVE"] if big.get("SQ_LDS_IDX_ACTIVE") else None,
           "source": f"profiles/{tag}_pmc_config5.csv + profiles/{tag}_kernel_stats_config5.csv (the 100-step dispatch alone)"}
    json.dump(rec, open(os.path.join(REPO, "profiles", "config5_counters.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))
g = lambda c: acc[c][0]
if g("SQ_WAVE_CYCLES"):
    print("VALU active / wave cycles %.3f | waiting / wave cycles %.3f | VALU wave-instructions %.4e (sum of both dispatches: warm-up 2 steps + 100 steps)"
          % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_INSTS_VALU")))
