#!/usr/bin/env python3
"""Summarise the horizon-sweep passes of tools/profile_round.sh into profiles/<tag>_kernel_stats_sweep.csv and profiles/<tag>_sweep.json.

  python tools/sweep_prof_summary.py gpurun_out r03"""
import collections, csv, json, shutil, sys
O, tag = sys.argv[1], sys.argv[2]
shutil.copy(f"{O}/prof_stats_sweep/sw_kernel_stats.csv", f"profiles/{tag}_kernel_stats_sweep.csv")
out = {}
for f in (f"{O}/prof_sweep_fetch/f_counter_collection.csv", f"{O}/prof_sweep_write/w_counter_collection.csv", f"{O}/prof_sweep_sq/s_counter_collection.csv"):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_mpc_big" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 512 * 256:      # the sweep's solve launches (first + repeated call)
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    out.update(agg)
rows = list(csv.DictReader(open(f"{O}/prof_stats_sweep/sw_kernel_trace.csv")))
big = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "k_mpc_big" in r["Kernel_Name"] and int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) >= 512 * 256)
builds = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "k_mpc<true" in r["Kernel_Name"]]
log = [l.strip() for l in open(f"{O}/prof_stats_sweep.log") if "calc_constr" in l or l.startswith("again")]
rec = {"command": "rocprofv3 --kernel-trace --stats / --pmc ... -- python3 tools/gpu_sweep_only.py 64 150 (tools/profile_round.sh): the sweep twice",
       "solve_launches_ms": [d / 1e6 for d in big], "build_launches": len(builds), "build_launches_total_ms": sum(builds) / 1e6,
       "wall_clock_under_the_profiler": log,
       "counters_of_both_solve_launches": out,
       "derived": {"fetch_TB_counter_x_1KiB": out.get("FETCH_SIZE", 0) * 1024 / 1e12, "fetch_TB_doubled_upper_bound": 2 * out.get("FETCH_SIZE", 0) * 1024 / 1e12,
                   "write_TB": out.get("WRITE_SIZE", 0) * 1024 / 1e12,
                   "wait_fraction_of_wave_cycles": out.get("SQ_WAIT_ANY", 0) / max(out.get("SQ_WAVE_CYCLES", 1), 1),
                   "valu_active_fraction_of_wave_cycles": out.get("SQ_ACTIVE_INST_VALU", 0) / max(out.get("SQ_WAVE_CYCLES", 1), 1)},
       "note": "each solve launch: k_mpc_big over the 7,552 (horizon, aircraft) pairs N = 33..150 x 64 aircraft, taken from a work queue (second "
               "launch: costliest first by the first one's iteration counts); FETCH_SIZE on gfx950 under-reports wide reads by up to 2 x "
               "(MI355X_MICROARCH.md HBM): both readings given"}
json.dump(rec, open(f"profiles/{tag}_sweep.json", "w"), indent=1)
print(json.dumps(rec, indent=1)[:1200])
