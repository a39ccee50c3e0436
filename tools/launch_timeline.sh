#!/bin/bash
# usage: tools/launch_timeline.sh <tag> <bench args...>  (GPU box) - per-launch start/duration/gap of the predict kernels
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/timeline_$tag && mkdir -p $O && cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --cpu-seconds 0 --no-verify "$@" > $O/bench.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
O = sys.argv[1]
rows = []
for f in glob.glob(O + '/trace/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60]))
rows.sort()
prev_end = None
with open(O + '/timeline.txt', 'w') as out:
    for s, e, name in rows:
        if 'predict_' not in name and 'cluster' not in name and 'scan' not in name and 'period' not in name:
            prev_end = e
            continue
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        out.write(f"{(s - rows[0][0]) / 1e3:12.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:8.1f} us  {name}\n")
        prev_end = e
print(open(O + '/timeline.txt').read()[-6000:])
PY
rm -rf $O/trace
