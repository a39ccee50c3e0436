#!/bin/bash
# usage (GPU box): tools/round_table.sh <outfile>   - the round's table: the final kernels over the workloads and the
# switches that matter, one bench.py run each (5 steps after 2, --no-pcie, no CPU leg), through tools/sweep.sh
out=${1:-gpurun_out/round_table.txt}
mkdir -p "$(dirname "$out")"
bash tools/sweep.sh "$out" --steps 5 --warmup 2 --no-pcie --cpu-seconds 0 -- \
  "" "--no-grid" "--consecutive" "--shuffle" "--shuffle --param ohx_cluster=off" \
  "--missing-ppm 100" "--missing-ppm 100 --param ohx_defer_missing=off" "--missing-ppm 1000" "--missing-ppm 10000" \
  "--shuffle --missing-ppm 100" \
  "--path fields" "--path fields --missing-ppm 100" "--path run1" \
  "--grid C720L137 --steps 2 --warmup 1" "--grid C720L137 --shuffle --steps 2 --warmup 1" \
  "--grid C48" "--grid C48 --param ohx_tree_split=off" "--grid C90" "--grid C180" "--grid C180 --shuffle" \
  "--depth 6" "--depth 10" "--depth 14" "--trees 200" \
  "--kernel super3" "--param ohx_coop_rows=0" "--param ohx_tree_tops=off" "--param ohx_brick_k_fastest=0" \
  "--param ohx_launches_per_residency=3" "--trees 1 --depth 0"
