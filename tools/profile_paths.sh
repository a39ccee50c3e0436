#!/bin/bash
# usage: tools/profile_paths.sh <round tag>   (on the GPU box through gpurun)
# The two paths the GridComp shell calls, profiled like the headline path:
#   bench.py --path fields   OHXBoosterPredictFieldsDevice   (fused SoA call)
#   bench.py --path run1     OHXBoosterRun1Device            (imports -> INTERNAL OH)
# For each: rocprofv3 --kernel-trace --stats summary, the bench line under the profiler, and the PMC set
# (separate --pmc passes, tools/pmc.sh) per kernel.  Results under gpurun_out/paths_<tag>/.
tag=$1
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/paths_$tag && mkdir -p $O && cd $R
for path in fields run1; do
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$path -- python3 bench.py --path $path --steps 5 --warmup 2 > $O/bench_${path}_under_rocprof.log 2>&1 || exit 1
  grep "^{\"metric\"" $O/bench_${path}_under_rocprof.log | tail -1 > $O/bench_${path}_under_rocprof.json
  cp $O/trace_$path/*/*_kernel_stats.csv $O/${path}_kernel_stats.csv
  rm -rf $O/trace_$path
  timeout -k 10 600 python3 bench.py --path $path --steps 10 --warmup 3 > $O/bench_${path}.log 2>&1 || exit 1
  grep "^{\"metric\"" $O/bench_${path}.log | tail -1 > $O/bench_${path}.json
  PMC_SETS=base tools/pmc.sh ${tag}_$path --path $path > /dev/null 2>&1
  cp gpurun_out/pmc_${tag}_$path/summary.txt $O/pmc_${path}.txt
  rm -rf gpurun_out/pmc_${tag}_$path/*/
  echo "$path done"
done
ls -la $O
