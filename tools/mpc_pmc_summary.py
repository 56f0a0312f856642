#!/usr/bin/env python3
"""Summarise the MPC counter passes of tools/profile_round.sh into profiles/<tag>_pmc_mpc.csv.

  python tools/mpc_pmc_summary.py gpurun_out/prof_mfma gpurun_out/prof_lds r01

One row per (kernel, counter): sum over all dispatches and the per-dispatch mean.  Kernels: the MPC build kernel
(k_mpc<true>: DARE on the matrix cores), the ADMM kernel (k_mpc_fast: KKT inverse on the matrix cores)."""
import csv, glob, os, sys
from collections import defaultdict

dirs, tag = sys.argv[1:-1], sys.argv[-1]
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
acc = defaultdict(lambda: [0.0, 0])
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if "k_mpc" not in name and "k_lqr" not in name:
                continue
            short = "k_mpc_fast" if "k_mpc_fast" in name else ("k_mpc<true> (build)" if ("ILb1" in name or "k_mpc<true" in name) else name[:40])
            a = acc[(short, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
out = os.path.join(REPO, "profiles", f"{tag}_pmc_mpc.csv")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "sum_over_dispatches", "dispatch_rows", "mean_per_row"])
    for (k, c), (s, n) in sorted(acc.items()):
        w.writerow([k, c, f"{s:.0f}", n, f"{s / n:.1f}"])
print(open(out).read())
# compact record for bench.py (mpc.mfma): per launch of 4096 solves, N = 30
import json
rec = {"source": f"profiles/{tag}_pmc_mpc.csv", "batch": 4096, "hzn": 30, "unit_note": "SQ_INSTS_VALU_MFMA_MOPS_F64 counts 512-FLOP units"}
for short in ("k_mpc_fast", "k_mpc<true> (build)"):
    g = lambda c: acc[(short, c)][0] / max(acc[(short, c)][1], 1)
    rec[short] = {"mfma_f64_instructions_per_launch": g("SQ_INSTS_VALU_MFMA_F64"),
                  "mfma_flop_per_launch": g("SQ_INSTS_VALU_MFMA_MOPS_F64") * 512,
                  "mfma_busy_cycles_per_launch": g("SQ_VALU_MFMA_BUSY_CYCLES"),
                  "valu_instructions_per_launch": g("SQ_INSTS_VALU"),
                  "wait_fraction_of_wave_cycles": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
                  "lds_bank_conflict_fraction": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")}
json.dump(rec, open(os.path.join(REPO, "profiles", "mfma_mpc.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
