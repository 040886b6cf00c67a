#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/profile.sh into a small markdown report
(kernel durations from --kernel-trace --stats; HBM bytes from the FETCH_SIZE / WRITE_SIZE PMC passes,
corrected as MI355X_MICROARCH.md prescribes: on gfx950 FETCH_SIZE counts 64 B per 128-B request for
wide streaming reads, so the read side is doubled; both counters are in KiB)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, suffix):
    return sorted(glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True))


def kernel_durations(d):
    out = defaultdict(list)
    for f in find(d, "kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "?")
            out[name].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)   # us
    return out


def counters(d, counter):
    out = defaultdict(list)
    for f in find(d, "counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                out[row.get("Kernel_Name", "?")].append(float(row["Counter_Value"]))
    return out


def short(n):
    return n if len(n) < 90 else n[:87] + "..."


def main():
    d = sys.argv[1]
    dur = kernel_durations(os.path.join(d, "stats"))
    fetch = counters(os.path.join(d, "fetch"), "FETCH_SIZE")
    write = counters(os.path.join(d, "write"), "WRITE_SIZE")
    print(f"# rocprofv3 summary: {os.path.basename(d)}\n")
    print("| kernel | calls | avg us | min us | max us | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch | HBM bytes/launch (fetch x2 + write) |")
    print("|---|---|---|---|---|---|---|---|")
    summary = {}
    for name, xs in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        fs, ws = fetch.get(name), write.get(name)
        f_avg = sum(fs) / len(fs) if fs else None
        w_avg = sum(ws) / len(ws) if ws else None
        hbm = (2 * f_avg + w_avg) * 1024 if (f_avg is not None and w_avg is not None) else None
        print(f"| `{short(name)}` | {len(xs)} | {sum(xs)/len(xs):.1f} | {min(xs):.1f} | {max(xs):.1f} | "
              f"{f_avg if f_avg is None else round(f_avg,1)} | {w_avg if w_avg is None else round(w_avg,1)} | "
              f"{hbm if hbm is None else int(hbm)} |")
        summary[name] = dict(calls=len(xs), avg_us=sum(xs) / len(xs), fetch_kib_raw=f_avg, write_kib=w_avg,
                             hbm_bytes_per_launch=hbm)
    # SQ counter passes (any counter found is reported per launch, summed over the chip)
    extra = defaultdict(dict)
    for sub in ("sq", "sq2"):
        for f in find(os.path.join(d, sub), "counter_collection.csv"):
            acc = defaultdict(lambda: defaultdict(list))
            for row in csv.DictReader(open(f)):
                acc[row.get("Kernel_Name", "?")][row.get("Counter_Name")].append(float(row["Counter_Value"]))
            for kn, cs in acc.items():
                for cn, vals in cs.items():
                    extra[kn][cn] = sum(vals) / len(vals)
    for kn, cs in extra.items():
        if kn in summary:
            summary[kn]["sq_counters_per_launch"] = cs
            print(f"\nSQ counters per launch, `{short(kn)}`:\n")
            for cn, val in sorted(cs.items()):
                print(f"- {cn}: {val:.4g}")
    json.dump(summary, open(os.path.join(d, "summary.json"), "w"), indent=1)
    for n in ("stats", "fetch", "write"):
        lg = os.path.join(d, n + ".log")
        if os.path.exists(lg):
            lines = [l for l in open(lg) if l.startswith("{")]
            if lines:
                print(f"\nbench line under the `{n}` pass:\n```\n{lines[-1].strip()}\n```")


if __name__ == "__main__":
    main()
