#!/usr/bin/env python3
"""Duration distribution of one kernel from a rocprofv3 rocpd database: min / percentiles / max (us).
usage: rocpd_hist.py <results.db> <kernel name substring> [...]"""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t] or [t for t in tabs if "kernel" in t]
print("tables:", kd[:6])
for pat in sys.argv[2:]:
    try:
        rows = db.execute("select (end - start) from kernels where name like ?", (f"%{pat}%",)).fetchall()
    except sqlite3.Error as e:
        print("query failed:", e); rows = []
    d = np.array([r[0] for r in rows], dtype=np.float64) / 1e3
    if d.size:
        q = np.percentile(d, [0, 5, 25, 50, 75, 95, 100])
        print(f"{pat}: n={d.size} min {q[0]:.1f} p5 {q[1]:.1f} p25 {q[2]:.1f} p50 {q[3]:.1f} p75 {q[4]:.1f} p95 {q[5]:.1f} max {q[6]:.1f} us")
