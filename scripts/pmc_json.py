#!/usr/bin/env python3
"""profile_<tag>.txt (scripts/rocpd_summary.py) -> the PMC constants bench.py quotes (profiles/pmc_current.json):
FETCH_SIZE / WRITE_SIZE in KiB summed over the kernels of one scoring launch, the VALU-busy fraction of the dominant
kernel from the SQ pass (SQ_ACTIVE_INST_VALU counts quad-cycles: x4 SIMD-cycles; 1024 SIMDs x kernel cycles at 2.4 GHz)."""
import json
import sys

src, committed_as = sys.argv[1], sys.argv[2]
vals = {}
for line in open(src):
    name, parts = line[:50].strip(), line[50:].split()      # fixed-width columns: kernel names contain blanks
    if len(parts) >= 8 and name.startswith("pgx::score_") and parts[0] in ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU"):
        if "group_kernel" in name and (", true, " in name or name.rstrip().endswith(", true>")):
            continue                                          # the counting variant of pgx_score_stats (never timed)
        vals[(name.split("<")[0], parts[0])] = (float(parts[1]), float(parts[-1]))
fetch = sum(v[0] for (k, c), v in vals.items() if c == "FETCH_SIZE")
write = sum(v[0] for (k, c), v in vals.items() if c == "WRITE_SIZE")
# the group kernel writes nothing but its accumulator atomics (an L2-resident 48 KB table): WRITE_SIZE tallies each atomic as a 32-byte
# request although no HBM byte moves (profiles/round5_fetch_calibration.txt: 2^24 atomics -> 524 288 KiB), so this part is not traffic
atomic_write = vals.get(("pgx::score_group_kernel", "WRITE_SIZE"), (0.0, 0.0))[0]
busy = None
key = ("pgx::score_group_kernel", "SQ_ACTIVE_INST_VALU")
if key in vals:
    quad_cycles, avg_ns = vals[key]
    busy = quad_cycles * 4.0 / (1024 * avg_ns * 2.4)
lds_busy = None
key = ("pgx::score_group_kernel", "SQ_ACTIVE_INST_LDS")    # quad-cycles of the CU's LDS pipe: x4 / (256 CUs x kernel cycles)
if key in vals:
    quad_cycles, avg_ns = vals[key]
    lds_busy = quad_cycles * 4.0 / (256 * avg_ns * 2.4)
insts = vals.get(("pgx::score_group_kernel", "SQ_INSTS_VALU"), (None, None))[0]
print(json.dumps({"source": committed_as, "fetch_kib": fetch, "write_kib": write, "atomic_write_kib": atomic_write, "valu_busy_frac": busy, "lds_busy_frac": lds_busy,
                  "valu_wave_instructions": insts}))
