#!/bin/bash
# usage (here, after gpurun has merged gpurun_out/ back): tools/install_profiles.sh <round tag>
# Copies what tools/profile_round.sh and tools/profile_paths.sh left under gpurun_out/ into profiles/ under the round's names.
tag=$1; R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/profile_$tag; P=$R/gpurun_out/paths_$tag; D=$R/profiles
cp $O/kernel_stats.csv $D/${tag}_rocprofv3_kernel_stats.csv
cp $O/kernel_stats_timed_only.csv $D/${tag}_rocprofv3_kernel_stats_timed_region_only.csv
cp $O/bench.json $D/${tag}_bench_under_rocprofv3.json
cp $O/bench_timed_only.json $D/${tag}_bench_under_rocprofv3_timed_region_only.json
cp $O/pmc_summary.txt $D/${tag}_pmc_predict_kernel.txt
cp $O/calib_fetch_write.txt $D/${tag}_pmc_fetch_calibration.txt
cp $O/traffic.json $D/${tag}_traffic.json
cp $O/bench_plain.json $D/${tag}_bench_c360_n1.json
if [ -d $P ]; then
  cp $P/run1_kernel_stats.csv $D/${tag}_rocprofv3_run1_kernel_stats.csv
  cp $P/fields_kernel_stats.csv $D/${tag}_rocprofv3_fields_kernel_stats.csv
  cp $P/pmc_fields.txt $D/${tag}_pmc_fields_kernel.txt
  cp $P/bench_fields.json $D/${tag}_bench_fields.json
  cp $P/bench_run1.json $D/${tag}_bench_run1.json
fi
grep kernel_source_hash $D/${tag}_traffic.json
python3 -c "import sys; sys.path.insert(0, '$R'); import bench; print('sources now:', bench.kernel_source_hash())"
