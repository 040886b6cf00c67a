#!/bin/bash
# three rocprofv3 counter passes (+ one kernel-trace --stats pass) over tools/trace_case.py <case args>
# usage: tools/pmc_case.sh <tag> <trace_case args...>     -> gpurun_out/pmc_<tag>/{stats,a,b,c}, summary.md
TAG="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/tools/trace_case.py" "$@" > "$OUT/stats.log" 2>&1 || { echo "stats pass failed"; tail -5 "$OUT/stats.log"; exit 1; }
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/a" -- python3 "$ROOT/tools/trace_case.py" "$@" > "$OUT/a.log" 2>&1 || { echo "pass a failed"; tail -5 "$OUT/a.log"; }
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d "$OUT/b" -- python3 "$ROOT/tools/trace_case.py" "$@" > "$OUT/b.log" 2>&1 || { echo "pass b failed"; tail -5 "$OUT/b.log"; }
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/c" -- python3 "$ROOT/tools/trace_case.py" "$@" > "$OUT/c.log" 2>&1 || { echo "pass c failed"; tail -5 "$OUT/c.log"; }
if [ -n "$PMC_HBM" ]; then   # HBM traffic: FETCH_SIZE and WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md: TCC slots)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/d" -- python3 "$ROOT/tools/trace_case.py" "$@" > "$OUT/d.log" 2>&1 || { echo "pass d failed"; tail -5 "$OUT/d.log"; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/e" -- python3 "$ROOT/tools/trace_case.py" "$@" > "$OUT/e.log" 2>&1 || { echo "pass e failed"; tail -5 "$OUT/e.log"; }
fi
cd "$ROOT"
python3 tools/pmc_table.py "$OUT/a" "$OUT/b" "$OUT/c" "$OUT/d" "$OUT/e" > "$OUT/summary.md" 2>&1
cat "$OUT/summary.md"
