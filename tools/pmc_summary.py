#!/usr/bin/env python3
"""Summarise rocprofv3 output of `bench.py` into profiles/ (run after copying gpurun_out/prof_* back).

  python tools/pmc_summary.py gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write r01

writes profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats summary), profiles/<tag>_pmc_k_rollout.csv (the
two separate --pmc passes, FETCH_SIZE and WRITE_SIZE rows of the dominant kernel) and profiles/traffic_k_rollout.json
(HBM bytes per k_rollout launch).  Unit/correction per /opt/skills/guides/MI355X_MICROARCH.md "HBM": counters are in
KiB; WRITE_SIZE is exact for streaming stores; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, so the
read side is doubled (an upper bound for our 8-byte-per-lane state loads)."""
import csv
import glob
import json
import os
import sys

stats_dir, fetch_dir, write_dir, tag = sys.argv[1:5]
batch, euler = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (4096, 1000)
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
out = os.path.join(REPO, "profiles")
os.makedirs(out, exist_ok=True)


def one(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    assert f, (d, pat)
    return f[0]


rows = list(csv.DictReader(open(one(stats_dir, "*kernel_stats.csv"))))
with open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys())
    w.writeheader()
    for r in rows:
        r["Name"] = r["Name"][:120]
        w.writerow(r)


def counter(d, name, kern="k_rollout_q"):
    vals, meta = [], None
    for r in csv.DictReader(open(one(d, "*counter_collection.csv"))):
        if kern in r["Kernel_Name"] and ", true>" not in r["Kernel_Name"] and r["Counter_Name"] == name:      # (", true>": the LQR closed-loop instantiation)
            vals.append(float(r["Counter_Value"]))
            meta = r
    return vals, meta


fetch, meta = counter(fetch_dir, "FETCH_SIZE")
kname = meta["Kernel_Name"].replace("void ", "").split("(")[0].replace(",", ";")
write, _ = counter(write_dir, "WRITE_SIZE")
kern = [r for r in rows if kname.split("::")[-1].split("<")[0] in r["Name"] and ("4w" in kname) == ("4w" in r["Name"]) and ", true>" not in r["Name"]][0]
with open(os.path.join(out, f"{tag}_pmc_k_rollout.csv"), "w") as f:
    f.write("kernel,counter,launches,mean_KiB,min_KiB,max_KiB,VGPR,LDS_bytes,workgroup\n")
    for nm, v in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        f.write(f"{kname},{nm},{len(v)},{sum(v)/len(v):.3f},{min(v):.3f},{max(v):.3f},{meta['VGPR_Count']},"
                f"{meta['LDS_Block_Size']},{meta['Workgroup_Size']}\n")
fk, wk = sum(fetch) / len(fetch), sum(write) / len(write)
rec = {"kernel": kname, "batch": batch, "euler_steps": euler,
       "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "fetch_correction": 2.0,
       "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024),
       "algorithmic_bytes_per_launch": batch * euler * 144,
       "rocprof_avg_kernel_ns": float(kern["AverageNs"]), "source": f"profiles/{tag}_pmc_k_rollout.csv"}
json.dump(rec, open(os.path.join(out, "traffic_k_rollout.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))

# the large-batch leg of the same command (bench.py: roofline_large_batch): k_rollout_i<512>, B = 262,144, 200 steps
LB, LT = 262144, 200
fl, ml = counter(fetch_dir, "FETCH_SIZE", "k_rollout_i")
if not fl:
    fl, ml = counter(fetch_dir, "FETCH_SIZE", "k_rollout<512")
wl, _ = counter(write_dir, "WRITE_SIZE", "k_rollout_i")
if not wl:
    wl, _ = counter(write_dir, "WRITE_SIZE", "k_rollout<512")
if fl and wl:
    fk, wk = sum(fl) / len(fl), sum(wl) / len(wl)
    kl = [r for r in rows if "k_rollout_i" in r["Name"] or "k_rollout<512" in r["Name"] or "k_rolloutILi512" in r["Name"]]
    rec = {"kernel": "f16::k_rollout_i<512>", "batch": LB, "euler_steps": LT, "launches": len(fl),
           "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "fetch_correction": 2.0,
           "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024), "algorithmic_bytes_per_launch": LB * LT * 144,
           "ratio_to_algorithmic": (2.0 * fk + wk) * 1024 / (LB * LT * 144),
           "rocprof_avg_kernel_ns": float(kl[0]["AverageNs"]) if kl else None,
           "scratch_bytes_per_lane": ml.get("Scratch_Size", ml.get("Private_Segment_Size")),
           "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py ({tag})"}
    with open(os.path.join(out, f"{tag}_pmc_k_rollout.csv"), "a") as f:
        for nm, v in (("FETCH_SIZE", fl), ("WRITE_SIZE", wl)):
            f.write(f"f16::k_rollout_i<512>,{nm},{len(v)},{sum(v)/len(v):.3f},{min(v):.3f},{max(v):.3f},{ml['VGPR_Count']},"
                    f"{ml['LDS_Block_Size']},{ml['Workgroup_Size']}\n")
    json.dump(rec, open(os.path.join(out, "traffic_k_rollout_large.json"), "w"), indent=1)
    print(json.dumps(rec, indent=1))
