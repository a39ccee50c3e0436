#!/bin/bash
# usage (GPU box): tools/run1_timeline.sh <tag> [bench args...]  - rocprofv3 --kernel-trace of `bench.py --path run1`
# (2 steps after 1), then the LAST step's kernels in start order: start and end in microseconds from the step's first
# kernel, queue, name - which kernels of an OH Run1 tick really run beside which.  Output: gpurun_out/run1_timeline_<tag>.txt
#   tools/run1_timeline.sh <tag> --block 48,24,72 [run1_block_ticks args]   the same for a rank-sized block's HOST tick
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/run1_tl_$tag && mkdir -p $O && cd $R
if [ "$1" = "--block" ]; then
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 tools/run1_block_ticks.py --ticks 20 "$@" > $O/bench.log 2>&1
else
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 bench.py --path run1 --steps 2 --warmup 1 --no-pcie --cpu-seconds 0 --no-verify "$@" > $O/bench.log 2>&1
fi
python3 - $O > $R/gpurun_out/run1_timeline_$tag.txt <<'PY'
import csv, glob, re, sys
rows = []
for f in glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']))
for f in glob.glob(sys.argv[1] + '/*/*_memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'dma', 'COPY ' + r.get('Direction', '') + ' ' + r.get('Name', '')))
rows.sort()
def short(n):
    n = re.sub(r'^void ', '', n).replace('(anonymous namespace)::', '').replace('ohx::', '')
    return re.sub(r'\(.*$', '', n)[:70]
# the last step: from the last k_slab_kernel on
last = max(i for i, r in enumerate(rows) if 'k_slab_kernel' in r[3])
start = min(i for i in range(last, -1, -1) if rows[last][0] - rows[i][0] < 3_000_000)     # what was enqueued just before it
t0 = rows[start][0]
for s, e, q, n in rows[start:]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} us  q{q}  {short(n)}")
PY
rm -rf $O/*/
cat $R/gpurun_out/run1_timeline_$tag.txt
