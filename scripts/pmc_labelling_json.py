#!/usr/bin/env python3
"""Sums rocprofv3's FETCH_SIZE / WRITE_SIZE over every dispatch of the min-cut kernels of ONE expansion (scripts/profile_labelling_pmc.sh)
and writes the constants bench.py quotes next to `roofline_labelling` (profiles/pmc_labelling.json).  HBM bytes per expansion = 2 x
FETCH + WRITE in KiB x 1024 (gfx950 reports half of a read: MI355X_MICROARCH.md HBM section, calibrated in
profiles/round5_fetch_calibration.txt); `atomic_write` is the WRITE_SIZE of the kernels that write through atomics only, which the
calibration showed to be L2 traffic tallied at 32 B per atomic - reported separately, not subtracted (the labelling kernels mix both).
usage: pmc_labelling_json.py <tag> <prefix of the rocprofv3 output directories>"""
import glob
import json
import os
import sqlite3
import sys

tag, prefix = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINCUT = ("mf_k_", "t_move_kernel", "t_mini_kernel", "t_region_mini_kernel", "r_init_mark_kernel", "r_promote_kernel", "r_build_kernel", "energy_kernel")


def db_of(d):
    hits = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    return sqlite3.connect(hits[0]) if hits else None


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("pgx::", "")


out = {}
for cfg in ("C3", "C5", "C4"):
    rec = {}
    stats = db_of(f"{prefix}{cfg}_stats")
    print(f"==== {cfg}: one expansion from zeros (scripts/ab_expansion.py {cfg} --reps 1)")
    if stats is not None:
        print(f"{'kernel':48s} {'calls':>7s} {'total_us':>12s} {'avg_us':>9s}")
        tot = 0.0
        for name, calls, total, avg in stats.execute("select name,total_calls,total_duration,average from top_kernels"):
            if any(m in name for m in MINCUT):
                print(f"{short(name)[:48]:48s} {calls:7d} {total:12.1f} {avg:9.2f}")
                tot += total
        rec["kernel_us"] = tot
        print(f"{'all min-cut kernels':48s} {'':7s} {tot:12.1f}")
    for counter, key in (("FETCH_SIZE", "fetch_kib"), ("WRITE_SIZE", "write_kib")):
        db = db_of(f"{prefix}{cfg}_{'fetch' if key == 'fetch_kib' else 'write'}")
        if db is None:
            continue
        rows = db.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name = ? group by kernel_name",
                          (counter,)).fetchall()
        total = 0.0
        print(f"-- {counter} (KiB, summed over the dispatches)")
        for name, v, nd in sorted(rows, key=lambda r: -r[1]):
            if any(m in name for m in MINCUT):
                total += v
                if v > 0.005 * max(1.0, sum(r[1] for r in rows)):
                    print(f"   {short(name)[:48]:48s} {v:14.1f} over {nd} dispatches")
        rec[key] = total
        print(f"   {'all min-cut kernels':48s} {total:14.1f}")
    if "fetch_kib" in rec and "write_kib" in rec:
        rec["bytes"] = int((2 * rec["fetch_kib"] + rec["write_kib"]) * 1024)
        rec["source"] = f"profiles/round6_labelling_pmc.txt ({tag}): 2 x FETCH_SIZE + WRITE_SIZE over every min-cut kernel of one expansion"
        out[cfg.lower()] = rec
    print()
with open(os.path.join(ROOT, "gpurun_out", f"pmc_labelling_{tag}.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out))
