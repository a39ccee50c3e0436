#!/bin/bash
# usage: tools/profile_round.sh <round tag>   (run on the GPU box through gpurun)
# Produces, under gpurun_out/profile_<tag>/, what profiles/ keeps for the round:
#   kernel_stats.csv        rocprofv3 --kernel-trace --stats of the default `python3 bench.py`
#   bench.json              the JSON line of that same command;  bench_plain.json the same without the profiler
#   pmc_summary.txt         per-step means of the predict kernel's counters (separate --pmc passes, tools/pmc.sh)
#   calib_fetch_write.txt   FETCH_SIZE calibration on an almost pure streaming run (1 tree of depth 0)
#   traffic.json            roofline.traffic for bench.py, tagged with the hash of the kernel sources it was measured on
# PHASE=1 (the rocprofv3 --stats runs and the plain line) or PHASE=2 (counters, calibration, traffic, the line again)
# splits it over two gpurun calls - round 6's default bench run carries a CPU thread sweep, a whole-batch oracle
# comparison and the rank ticks, and all of it no longer fits one call's 20 minutes; unset = everything.
tag=$1
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/profile_$tag && mkdir -p $O && cd $R
if [ "${PHASE:-1}" = "1" ]; then
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof.log | tail -1 > $O/bench.json
cp $O/trace/*/*_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
timeout -k 10 600 python3 bench.py > $O/bench_plain.log 2>&1
grep "^{\"metric\"" $O/bench_plain.log | tail -1 > $O/bench_plain.json
# the same with nothing but the timed region launching the walk kernel (no pcie_inclusive leg, no CPU leg)
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace2 -- python3 bench.py --no-pcie --cpu-seconds 0 > $O/bench_under_rocprof_timed_only.log 2>&1
grep "^{\"metric\"" $O/bench_under_rocprof_timed_only.log | tail -1 > $O/bench_timed_only.json
cp $O/trace2/*/*_kernel_stats.csv $O/kernel_stats_timed_only.csv 2>/dev/null
fi
if [ "${PHASE:-2}" = "2" ]; then
# (phase 2 of a split round starts on a fresh box: the plain line the traffic file is made from, in its short form)
[ -s $O/bench_plain.json ] || { timeout -k 10 600 python3 bench.py --no-pcie --cpu-seconds 0 > $O/bench_plain.log 2>&1; grep "^{\"metric\"" $O/bench_plain.log | tail -1 > $O/bench_plain.json; }
PMC_KERNEL="${WALK_KERNEL:-predict_rows_ring_kernel}" tools/pmc.sh $tag > /dev/null 2>&1
cp gpurun_out/pmc_$tag/summary.txt $O/pmc_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/calib_$c -- python3 bench.py --cpu-seconds 0 --no-verify --steps 2 --warmup 1 --trees 1 --depth 0 --kernel ring > $O/calib_$c.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
with open(O + '/calib_fetch_write.txt', 'w') as out:
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        tot = 0.0
        for f in glob.glob(f'{O}/calib_{c}/*/*_counter_collection.csv'):
            for r in csv.DictReader(open(f)):
                if 'predict_rows_ring_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c:
                    tot += float(r['Counter_Value'])
        line = f"{c} per_step={tot/3:.6g} (1 tree of depth 0: rows 6 046 617 600 B read, 223 948 800 B written per step)"
        print(line); out.write(line + "\n")
PY
python3 tools/make_traffic_json.py $O/pmc_summary.txt $O/calib_fetch_write.txt $O/bench_plain.json > $O/traffic.json
# the default line once more, now that the measurement of these very kernels exists: it carries roofline.traffic
cp $O/traffic.json profiles/${tag}_traffic.json
timeout -k 10 600 python3 bench.py > $O/bench_plain.log 2>&1
grep "^{\"metric\"" $O/bench_plain.log | tail -1 > $O/bench_plain.json
fi
rm -rf $O/trace $O/trace2 $O/calib_FETCH_SIZE $O/calib_WRITE_SIZE
ls -la $O
