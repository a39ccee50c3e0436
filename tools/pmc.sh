#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   — separate rocprofv3 --pmc passes over bench.py (short run),
# then a one-screen summary of the predict kernel's counters (mean per dispatch).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/pmc_$tag && mkdir -p $O && cd $R
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$n -- python3 bench.py --cpu-seconds 0 --steps 2 --warmup 1 "$@" > $O/$n.log 2>&1
done
python3 - $O <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
tot = collections.defaultdict(float); cnt = collections.defaultdict(int); steps = 3
for f in glob.glob(O + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'predict_' in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); cnt[r['Counter_Name']] += 1
with open(O + '/summary.txt', 'w') as out:
    for k in sorted(tot):
        line = f"{k:32s} per_step={tot[k]/steps:.6g}  dispatches={cnt[k]}"
        print(line); out.write(line + "\n")
PY
