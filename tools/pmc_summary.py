"""Summarise rocprofv3 --pmc CSVs written by tools/pmc.sh: per-launch averages of every
counter for the kernels whose name contains a pattern (default k_step), as text and,
with --json FILE, as the small record bench.py reads (HBM bytes per launch, VALU busy).

    python tools/pmc_summary.py <dir> [pattern] [--json out.json --envs E --agents N]
"""
import argparse
import collections
import csv
import glob
import json

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("pattern", nargs="?", default="k_step")
ap.add_argument("--json")
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--agents", type=int, default=30)
ap.add_argument("--streams", type=int, default=2)
a = ap.parse_args()

acc = collections.defaultdict(list)
names = set()
for f in sorted(glob.glob(f"{a.dir}/pass*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if a.pattern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            names.add(r["Kernel_Name"].split("(")[0])
mean = {k: sum(v) / len(v) for k, v in acc.items()}
for k, v in acc.items():
    print(f"{k:32s} launches={len(v):4d} mean={mean[k]:16.1f}")

if a.json:
    rec = {"workload": {"envs_per_gpu": a.envs, "agents": a.agents, "streams": a.streams}, "kernel": sorted(names),
           "source": "rocprofv3 --pmc, one pass per counter group (tools/pmc.sh), bench.py --steps 50; mean over the launches",
           "counters": {k: round(v, 1) for k, v in sorted(mean.items())}}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        # FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide
        # coalesced read, so reads are doubled (upper bound of the true read traffic)
        rec["FETCH_SIZE_KB_raw"] = mean["FETCH_SIZE"]
        rec["WRITE_SIZE_KB"] = mean["WRITE_SIZE"]
        rec["hbm_bytes_per_launch"] = int(mean["FETCH_SIZE"] * 1024 * 2 + mean["WRITE_SIZE"] * 1024)
    if "SQ_ACTIVE_INST_VALU" in mean and "GRBM_GUI_ACTIVE" in mean:
        # rocprof's VALUBusy (gfx94x formula): SQ_ACTIVE_INST_VALU * 4 / SIMDs / GRBM_GUI_ACTIVE, with the
        # GRBM count summed over the 8 XCDs and 1024 SIMDs on the chip
        rec["valu_busy"] = mean["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (mean["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_WAVE_CYCLES" in mean:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in mean:
                rec[k.lower() + "_frac_of_wave_cycles"] = mean[k] / mean["SQ_WAVE_CYCLES"]
    with open(a.json, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
