#!/bin/bash
# rocprofv3 kernel stats of the dp_step workload; usage: tools/dpstep_stats.sh <tag> [bench args...]
TAG="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/dpstats_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --workload dp_step --steps 6 --warmup 2 --precondition-ms 0 "$@" > "$OUT/run.log" 2>&1 || { echo failed; tail -5 "$OUT/run.log"; exit 1; }
cd "$ROOT"
python3 - <<PY
import csv,glob
f=sorted(glob.glob("$OUT/**/*kernel_stats.csv",recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (8 steps incl warm-up):", tot/1e6/8)
for r in rows[:28]:
    print("%-100s calls=%5s avg_us=%8.1f pct=%5s"%(r["Name"][:100],r["Calls"],float(r["AverageNs"])/1e3,r["Percentage"]))
PY
