#!/bin/bash
# usage: tools/valu_calib.sh <tag>   (on the GPU box) - tools/bin/valu_issue_microbench plain, then the same under
# rocprofv3 --pmc with the SQ "who is busy" counters of tools/pmc.sh, so that what those counters read for a KNOWN
# stream of vector instructions (1 / 4 waves per SIMD, plain adds and the walk's step) stands beside what they read for
# predict_rows_ring_kernel.  Output: gpurun_out/valu_<tag>/{microbench.txt,calib.txt}
tag=$1
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/valu_$tag && mkdir -p $O && cd $R
timeout -k 10 120 tools/bin/valu_issue_microbench > $O/microbench.txt 2>&1 || exit 1
C="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVES"
: > $O/calib.txt
for spec in "0 1" "0 4" "6 1" "6 4" "6 8"; do
  set -- $spec
  d=$O/k$1_w$2
  timeout -k 10 120 rocprofv3 --pmc $C --output-format csv -d $d -- tools/bin/valu_issue_microbench --kind $1 --waves $2 > $d.log 2>&1
  python3 - $d $1 $2 >> $O/calib.txt <<'PY'
import csv, glob, sys, collections
d, kind, w = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(d + '/*/*_counter_collection.csv'):
    rows = list(csv.DictReader(open(f)))
    last = max(int(r['Dispatch_Id']) for r in rows)          # the timed launch (the first is the warm-up)
    for r in rows:
        if int(r['Dispatch_Id']) == last:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
per_trip = {0: 32, 6: 56}[kind]
print(f"kind {kind} waves/SIMD {w}: " + "  ".join(f"{k}={v:.6g}" for k, v in sorted(tot.items())))
if tot.get('SQ_INSTS_VALU'):
    print(f"    SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU = {4 * tot['SQ_ACTIVE_INST_VALU'] / tot['SQ_INSTS_VALU']:.3f} cycles per wave64 vector instruction;"
          f"  SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CU_CYCLES x 4 SIMDs) = {tot['SQ_ACTIVE_INST_VALU'] / max(tot['SQ_BUSY_CU_CYCLES'], 1):.3f}")
PY
  rm -rf $d
done
cat $O/microbench.txt $O/calib.txt
