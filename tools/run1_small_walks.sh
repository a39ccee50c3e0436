#!/bin/bash
# The walk of a rank-sized block (OHXBoosterRun1 host form, tools/run1_block_ticks.py): which tile kernel and how many runs
# of trees per tile.  ohx_kernel auto at this size = predict_fields_kernel<2,2,tops> with the trees in 5 runs.
# usage (GPU box): tools/run1_small_walks.sh [block] [ticks]
cd "$(dirname "$0")/.."
block=${1:-48,24,72}; ticks=${2:-300}
run() { printf '%-44s ' "$*"; python3 tools/run1_block_ticks.py --block "$block" --ticks "$ticks" "$@" 2>&1 | tail -1; }
run
for k in super2 super4 super1 super3; do
  run --param ohx_kernel=$k
  run --param ohx_kernel=$k --param ohx_tree_split=8
  run --param ohx_kernel=$k --param ohx_tree_split=3
done
run --param ohx_tree_tops=off
run --param ohx_tree_split=off
run
