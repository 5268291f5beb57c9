"""Summary of a rocprofv3 --kernel-trace --stats run for bench.py: the step kernel's name, calls and average
duration, tagged with the workload, as profiles/rNN_<tag>_kernel_trace.json.

    python tools/kernel_stats_summary.py <kernel_stats.csv> <envs_per_gpu> <agents> <streams> <out.json> [<bench log of the traced run>]

With the traced run's own bench line the summary also holds its step time: per-dispatch tracing slows the launch pipeline,
so the traced durations belong to that step time (launches in flight = streams x avg_us / step time), not to an untraced run's.
"""
import csv
import json
import re
import sys

path, E, n_a, S, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
best = None
with open(path) as f:
    for row in csv.DictReader(f):
        if "k_step" in row["Name"] and (best is None or float(row["TotalDurationNs"]) > float(best["TotalDurationNs"])):
            best = row
assert best is not None, "no step kernel in " + path
name = best["Name"].split("(")[0].replace("void ", "")
summary = {"workload": {"envs_per_gpu": E, "agents": n_a, "streams": S}, "kernel": name, "calls": int(best["Calls"]),
           "avg_us": float(best["AverageNs"]) / 1e3, "min_us": float(best["MinNs"]) / 1e3, "max_us": float(best["MaxNs"]) / 1e3,
           "stats_csv": path.split("/")[-1]}
if len(sys.argv) > 6:
    m = re.search(r'"ms_per_step": ([0-9.eE+-]+)', open(sys.argv[6]).read())
    if m:
        summary["bench_ms_per_step_under_tracing"] = float(m.group(1))
json.dump(summary, open(out, "w"), indent=1)
print(open(out).read())
