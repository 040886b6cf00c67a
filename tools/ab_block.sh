#!/bin/bash
# rocprofv3 kernel stats of tools/trace_block.py under a list of environment settings.
# usage: tools/ab_block.sh "<grep pattern>" "ENV1=a ENV2=b" "ENV1=c" ...   (each argument = one run's environment)
PAT="$1"; shift
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  out="$ROOT/gpurun_out/ab_block_$i"
  rm -rf "$out"
  ( cd /tmp && export $cfg && rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$ROOT/tools/trace_block.py" tinyllama linearmax > "$out.log" 2>&1 ) || { echo "run failed: $cfg"; tail -5 "$out.log"; exit 1; }
  echo "== $cfg"
  python3 "$ROOT/tools/stats_table.py" "$out" 8 60 | grep -E "$PAT|^total"
done
