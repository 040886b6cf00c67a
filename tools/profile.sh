#!/bin/bash
# rocprofv3 passes for the headline bench on the GPU box.  usage: tools/profile.sh <tag> [bench args...]
# Writes CSVs under gpurun_out/prof_<tag>/{stats,fetch,write}; summarise with tools/summarize_profile.py
TAG="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1 || { echo "stats pass failed"; tail -5 "$OUT/stats.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/write.log"; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/sq" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/sq.log" 2>&1 || { echo "sq pass failed"; tail -5 "$OUT/sq.log"; }
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d "$OUT/sq2" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/sq2.log" 2>&1 || { echo "sq2 pass failed"; tail -5 "$OUT/sq2.log"; }
cd "$ROOT"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.md" 2>&1
cat "$OUT/summary.md"
