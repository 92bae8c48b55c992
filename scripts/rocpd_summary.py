#!/usr/bin/env python3
"""Text summary of rocprofv3 (ROCm 7.2, rocpd sqlite output): per-kernel time stats and per-kernel PMC averages.

usage: rocpd_summary.py <results.db> [<results.db> ...] > profiles/<name>.txt
Counter values are summed over the dimension instances (SEs / XCCs / channels) of one dispatch, then averaged over
dispatches of the same kernel.  FETCH_SIZE / WRITE_SIZE are in KiB as rocprofv3 reports them; the gfx950 x2 correction
for wide coalesced reads (MI355X_MICROARCH.md §HBM) is NOT applied here, it is applied where the numbers are quoted.
"""
import sqlite3
import sys


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    print(f"==== {path}")
    print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{short(name)[:70]:70s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}")
    rows = db.execute("""select kernel_name, counter_name, dispatch_id, sum(value), max(vgpr_count), max(sgpr_count),
                                max(lds_block_size), max(grid_size), max(workgroup_size), max(duration)
                         from counters_collection group by kernel_name, counter_name, dispatch_id""").fetchall()
    if rows:
        agg = {}
        for k, c, d, v, vg, sg, lds, grid, wg, dur in rows:
            a = agg.setdefault((k, c), [0.0, 0, vg, sg, lds, grid, wg, 0.0])
            a[0] += v
            a[1] += 1
            a[7] += dur
        print(f"\n{'kernel':50s} {'counter':24s} {'avg/dispatch':>16s} {'disp':>5s} {'vgpr':>5s} {'sgpr':>5s} {'lds':>6s} {'grid':>10s} {'avg_ns':>10s}")
        for (k, c), a in sorted(agg.items()):
            print(f"{short(k)[:50]:50s} {c:24s} {a[0] / a[1]:16.1f} {a[1]:5d} {a[2]:5d} {a[3]:5d} {a[4]:6d} {a[5]:10d} {a[7] / a[1]:10.0f}")
    print()
