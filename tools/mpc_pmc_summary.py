#!/usr/bin/env python3
"""Summarise the MPC passes of tools/profile_round.sh (headline settings only: B = 4096, N = 30, osqp defaults) into
profiles/<tag>_pmc_mpc.csv, profiles/<tag>_kernel_stats_mpc.csv and the compact record profiles/mfma_mpc.json that bench.py
attaches to its line (mpc.roofline: issued FLOPs, MFMA-busy fraction; mpc.mfma).

  python tools/mpc_pmc_summary.py gpurun_out/prof_stats_mpc gpurun_out/prof_mfma gpurun_out/prof_lds r03

Kernels: the build kernel (k_mpc<true>: DARE on the matrix cores), the solver (k_mpc_wave: one wavefront per aircraft; or
k_mpc_fast: one 512-lane workgroup per aircraft when F16_MPC_WAVE=0)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

stats_dir, dirs, tag = sys.argv[1], sys.argv[2:-1], sys.argv[-1]
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CLOCK_GHZ = 2.4      # MI355X_MICROARCH.md: max clock; GRBM_GUI_ACTIVE / 8 is used instead where it was collected


def short(name):
    if "k_mpc_wave" in name:
        return "k_mpc_wave"
    if "k_mpc_fast" in name:
        return "k_mpc_fast"
    if "k_mpc<true" in name or "ILb1" in name:
        return "k_mpc<true> (build)"
    return None


acc = defaultdict(lambda: [0.0, 0])
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            a = acc[(k, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
out = os.path.join(REPO, "profiles", f"{tag}_pmc_mpc.csv")
if not acc:
    sys.exit(f"no MPC kernel rows found: {out} is left as it is")
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "sum_over_dispatches", "dispatch_rows", "mean_per_launch"])
    for (k, c), (s, n) in sorted(acc.items()):
        w.writerow([k, c, f"{s:.0f}", n, f"{s / n:.1f}"])
print(open(out).read())
# kernel durations of the same command (headline settings only)
dur = {}
for f in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(REPO, "profiles", f"{tag}_kernel_stats_mpc.csv"))
    for r in csv.DictReader(open(f)):
        k = short(r["Name"])
        if k:
            dur[k] = dict(calls=int(r["Calls"]), avg_ms=float(r["AverageNs"]) / 1e6, min_ms=float(r["MinNs"]) / 1e6, max_ms=float(r["MaxNs"]) / 1e6)
rec = {"source": f"profiles/{tag}_pmc_mpc.csv + profiles/{tag}_kernel_stats_mpc.csv", "batch": 4096, "hzn": 30,
       "settings": "osqp_defaults (headline)", "unit_note": "SQ_INSTS_VALU_MFMA_MOPS_F64 counts 512-FLOP units; SQ_WAVE_CYCLES / SQ_WAIT_ANY in quad-cycles"}
for k in ("k_mpc_wave", "k_mpc_fast", "k_mpc<true> (build)"):
    g = lambda c: acc[(k, c)][0] / max(acc[(k, c)][1], 1)
    if not acc[(k, "SQ_INSTS_VALU")][1]:
        continue
    vec = 64.0 * (2 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64"))
    mfma = 512.0 * g("SQ_INSTS_VALU_MFMA_MOPS_F64")
    ms = dur.get(k, {}).get("avg_ms")
    cyc = g("GRBM_GUI_ACTIVE") / 8 if acc[(k, "GRBM_GUI_ACTIVE")][1] else (ms * 1e-3 * CLOCK_GHZ * 1e9 if ms else None)
    r = {"name": k, "avg_ms": ms, "calls": dur.get(k, {}).get("calls"),
         "mfma_f64_instructions_per_launch": g("SQ_INSTS_VALU_MFMA_F64"), "mfma_flop_per_launch": mfma,
         "vector_f64_flop_per_launch": vec, "issued_flop_per_launch": vec + mfma,
         "mfma_busy_cycles_per_launch": g("SQ_VALU_MFMA_BUSY_CYCLES"), "kernel_cycles": cyc,
         "mfma_busy_frac": g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc) if cyc else None,
         "mfma_tflops": mfma / (ms * 1e-3) / 1e12 if ms else None,
         "valu_instructions_per_launch": g("SQ_INSTS_VALU"),
         "wait_fraction_of_wave_cycles": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
         "valu_active_fraction_of_wave_cycles": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else None,
         "lds_bank_conflict_fraction": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None}
    rec[k] = r
    if k in ("k_mpc_wave", "k_mpc_fast") and "solve_kernel" not in rec:
        rec["solve_kernel"] = r
json.dump(rec, open(os.path.join(REPO, "profiles", "mfma_mpc.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
# VALU wave-instructions per launch of the solver (build kernel added: bench.py times the two together) -> profiles/issue_valu.json
ip = os.path.join(REPO, "profiles", "issue_valu.json")
try:
    iv = json.load(open(ip))
except Exception:
    iv = {}
if "k_mpc_wave" in rec:
    n = rec["k_mpc_wave"]["valu_instructions_per_launch"] + rec.get("k_mpc<true> (build)", {}).get("valu_instructions_per_launch", 0.0)
    iv["k_mpc_wave"] = dict(batch=4096, hzn=30, kernel="k_mpc<true> (build) + k_mpc_wave", insts_valu_per_launch=n,
                            per_aircraft_iteration=None, source=f"profiles/{tag}_pmc_mpc.csv")
    json.dump(iv, open(ip, "w"), indent=1)
