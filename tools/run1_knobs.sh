#!/bin/bash
# A rank's OH tick (tools/run1_block_ticks.py: OHXBoosterRun1's host form, registered arrays) under the library's experiment
# knobs, one line each, in ONE session (boxes differ by a few per cent):  OHX_RUN1_GATE (launch a kernel when the host has
# seen its inputs' event | enqueue it behind a wait on the event), OHX_RUN1_SLAB_IN_PLACE (the slab count reads PLE and TROPP
# over PCIe | after their copy), OHX_RUN1_STREAMS (copies beside the kernels | in one stream with them).
# usage (GPU box): tools/run1_knobs.sh [block] [ticks]
cd "$(dirname "$0")/.."
block=${1:-48,24,72}; ticks=${2:-300}
run() { printf '%-58s ' "$*"; env "$@" python3 tools/run1_block_ticks.py --block "$block" --ticks "$ticks" 2>&1 | tail -1; }
run OHX_RUN1_GATE=1 OHX_RUN1_SLAB_IN_PLACE=1 OHX_RUN1_STREAMS=2
run OHX_RUN1_GATE=0 OHX_RUN1_SLAB_IN_PLACE=1 OHX_RUN1_STREAMS=2
run OHX_RUN1_GATE=1 OHX_RUN1_SLAB_IN_PLACE=0 OHX_RUN1_STREAMS=2
run OHX_RUN1_GATE=0 OHX_RUN1_SLAB_IN_PLACE=0 OHX_RUN1_STREAMS=2
run OHX_RUN1_GATE=0 OHX_RUN1_SLAB_IN_PLACE=1 OHX_RUN1_STREAMS=1
run OHX_RUN1_GATE=1 OHX_RUN1_SLAB_IN_PLACE=1 OHX_RUN1_STREAMS=2
# the walk of a rank's block: its trees cut into N runs walked by different waves (ohx_tree_split; auto = 5 at this size)
for n in 5 6 8 10; do
  printf '%-58s ' "ohx_tree_split=$n"; python3 tools/run1_block_ticks.py --block "$block" --ticks "$ticks" --param ohx_tree_split=$n 2>&1 | tail -1
done
