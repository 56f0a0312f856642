"""Print the headline figures of a bench.py JSON line:  python tools/show_bench.py gpurun_out/bench.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("dyn", d["value"], d["ms_per_step"], d["roofline"]["frac"])
if "roofline_large_batch" in d: print("large", d["roofline_large_batch"]["steps_per_s"], d["roofline_large_batch"]["frac"])
if "mpc" in d: print("mpc first call", d["mpc"]["first_call_value"], "repeated", d["mpc"].get("repeated_call_value"), d["mpc"]["ms_per_batch"], d["mpc"]["admm_iters"])
c = d.get("config5_closed_loop")
if c:
    for k in ("headline", "host_loop", "fused_hold_command", "fused", "prepared_plan", "one_shot", "prepared_plan_warm_start"):
        if k in c: print(k, c[k])
