#!/usr/bin/env python3
"""Summarise the dynamics SQ-counter pass of tools/profile_round.sh into profiles/<tag>_pmc_dyn.csv.

  python tools/dyn_pmc_summary.py gpurun_out/prof_dyn r01

One row per (kernel, counter): mean per dispatch, plus derived VALU issue utilisation =
SQ_ACTIVE_INST_VALU / (4 x SQ_BUSY_CYCLES-equivalent SIMD cycles) printed to stdout."""
import csv, glob, os, sys
from collections import defaultdict

dirs, tag = sys.argv[1:-1], sys.argv[-1]
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
acc = defaultdict(lambda: [0.0, 0])
for f in [g for d in dirs for g in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)]:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "k_rollout" not in name:
            continue
        short = name.split("(")[0].replace("void f16::", "")
        a = acc[(short, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
out = os.path.join(REPO, "profiles", f"{tag}_pmc_dyn.csv")
if not acc:
    # (round 4: a second run after gpurun_out/ had been cleaned replaced 42 rows with a header line -- and bench.py went on citing it)
    sys.exit(f"no k_rollout rows under {dirs}: {out} is left as it is")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "dispatch_rows", "mean_per_dispatch"])
    for (k, c), (s, n) in sorted(acc.items()):
        w.writerow([k, c, n, "%.1f" % (s / n)])
kern = sorted({k for k, _ in acc})
for k in kern:
    g = lambda c: acc[(k, c)][0] / max(acc[(k, c)][1], 1)
    print("%-28s VALU wave-instr %.3e | active VALU cycles / wave cycles %.3f | wait / wave cycles %.3f | LDS instr %.3e, conflict/active %.3f"
          % (k, g("SQ_INSTS_VALU"), g("SQ_ACTIVE_INST_VALU") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
             g("SQ_INSTS_LDS"), g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_ACTIVE_INST_LDS"), 1)))
    if acc.get((k, "SQ_LDS_IDX_ACTIVE")):
        fp = g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64")
        print("%-28s LDS bank-conflict cycles / LDS index-active cycles %.3f | address conflicts / index-active %.3f | fp64 FMA+ADD+MUL wave-instr %.3e = %.1f %% of VALU"
              % ("", g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1), g("SQ_LDS_ADDR_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1), fp,
                 100 * fp / max(g("SQ_INSTS_VALU"), 1)))
# VALU wave-instructions per launch of the two benchmark kernels -> profiles/issue_valu.json (bench.py: roofline.issue)
import json
ip = os.path.join(REPO, "profiles", "issue_valu.json")
try:
    rec = json.load(open(ip))
except Exception:
    rec = {}
for key, pat, meta in (("k_rollout_q", "k_rollout_q<1, false>", dict(batch=4096, euler_steps=1000)),
                       ("k_rollout_i", "k_rollout_i<512, false>", dict(batch=262144, euler_steps=200))):
    ks = [k for k in kern if pat in k and (k, "SQ_INSTS_VALU") in acc]
    if ks:
        a = acc[(ks[0], "SQ_INSTS_VALU")]
        rec[key] = dict(meta, kernel=ks[0], insts_valu_per_launch=a[0] / a[1], source=f"profiles/{tag}_pmc_dyn.csv")
json.dump(rec, open(ip, "w"), indent=1)
print("wrote", out, ip)
