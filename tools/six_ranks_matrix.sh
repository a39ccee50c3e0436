#!/bin/bash
# Six ranks on one GPU (tools/ranks_per_gpu.py --ranks ${RANKS:-6} --block 48,24) under the two knobs a rank's environment has for the
# number of queues the tick uses: GPU_MAX_HW_QUEUES (HIP: hardware queues per process, default 4) and OHX_RUN1_STREAMS
# (this library: streams of a Run1 tick, default 3).  One line per combination: the tick's deciles, p99, max, mean and the
# aggregate, for the five calls and for Run1 registered.   usage (GPU box): tools/six_ranks_matrix.sh [outdir]
set -e
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/six_ranks}
mkdir -p "$out"
COMBOS=${COMBOS:--:3 3:3 3:1 2:3 2:2 2:1 1:1}
for combo in $COMBOS; do
  q=${combo%%:*}; s=${combo##*:}
  ( [ "$q" = "-" ] || export GPU_MAX_HW_QUEUES=$q
    export OHX_RUN1_STREAMS=$s
    python3 tools/ranks_per_gpu.py --ranks ${RANKS:-6} --block 48,24 --prep-s ${PREP_S:-20} --ticks ${TICKS:-200} ${EXTRA_ARGS} \
        > "$out/q${q}_s${s}.json" 2> "$out/q${q}_s${s}.err" )
  python3 - "$out/q${q}_s${s}.json" "$q" "$s" <<'PY'
import json, sys
for P, e in json.load(open(sys.argv[1]))["by_ranks"].items():
    for mode in ("reference", "run1_registered"):
        t = e[mode]["tick_ms"]
        agg = e[mode]["aggregate_gridcells_per_s"]
        print("queues %s streams %s  P=%s %-16s deciles %s  p99 %.2f max %.1f mean %.3f  aggregate %.0f M" % (
            sys.argv[2], sys.argv[3], P, mode, " ".join("%.2f" % d for d in t["deciles"]), t["p99"], t["max"], t["mean"], (agg or 0) / 1e6), flush=True)
PY
done | tee "$out/summary.txt"
