#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, SQ_INSTS_MFMA,
SQ_INSTS_VALU) + kernel trace.  usage: mfma_util.py <dir> [min_us]
util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32)   (SQ_BUSY_CYCLES is summed over 32 shader engines' SQs)"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
vals = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        vals[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
print("| kernel | launches | avg µs | MFMA instr / launch | vector instr / launch (incl. MFMA) | matrix pipe busy |")
print("|---|---|---|---|---|---|")
for name, xs in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    avg = sum(xs) / len(xs)
    if avg < min_us:
        continue
    c = {k: sum(v) / len(v) for k, v in vals.get(name, {}).items()}
    if not c.get("SQ_BUSY_CYCLES"):
        continue
    util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * c["SQ_BUSY_CYCLES"] / 32)
    print(f"| `{name[:80]}` | {len(xs)} | {avg:.1f} | {c.get('SQ_INSTS_MFMA', 0):.3g} | {c.get('SQ_INSTS_VALU', 0):.3g} | {100 * util:.0f} % |")
