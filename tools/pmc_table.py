#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 --pmc output directories (+ kernel-trace durations).
usage: pmc_table.py <dir> [<dir> ...] [--min-us 20] [--match substr]"""
import csv, glob, os, sys
from collections import defaultdict
args = sys.argv[1:]
min_us, match = 20.0, ""
dirs = []
i = 0
while i < len(args):
    if args[i] == "--min-us": min_us = float(args[i + 1]); i += 2
    elif args[i] == "--match": match = args[i + 1]; i += 2
    else: dirs.append(args[i]); i += 1
vals = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            vals[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for name, xs in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    avg = sum(xs) / len(xs)
    if avg < min_us or match not in name:
        continue
    c = {k: sum(v) / len(v) for k, v in vals.get(name, {}).items()}
    print(f"## `{name[:110]}`\n\nlaunches {len(xs)}, avg {avg:.1f} us (under the counter passes)\n")
    print("| counter | per launch |")
    print("|---|---|")
    for k in sorted(c):
        print(f"| {k} | {c[k]:.4g} |")
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        print(f"\nmatrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x SQ_BUSY_CYCLES / 32) = {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * c['SQ_BUSY_CYCLES'] / 32):.1f} %")
    if c.get("SQ_INSTS_MFMA") and c.get("SQ_INSTS_VALU"):
        print(f"vector instructions (non-MFMA) per MFMA = {(c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']) / c['SQ_INSTS_MFMA']:.2f}")
    print()
