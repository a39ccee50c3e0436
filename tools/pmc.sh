#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   — separate rocprofv3 --pmc passes over bench.py (short run),
# then a one-screen summary of the kernels' counters, per bench step and per kernel.
#   PMC_KERNEL=<substring>  only kernels whose name contains it (default: every kernel whose name contains "_kernel")
#   PMC_STEPS / PMC_WARMUP  steps of the short run (default 2 / 1); per_step divides by their sum
# The run never includes bench.py's whole-batch cross-check (--no-verify) or the CPU leg (--cpu-seconds 0).
tag=$1; shift
steps=${PMC_STEPS:-2}; warm=${PMC_WARMUP:-1}
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/pmc_$tag && mkdir -p $O && cd $R
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$n -- python3 bench.py --cpu-seconds 0 --no-verify --steps $steps --warmup $warm "$@" > $O/$n.log 2>&1
done
python3 - $O $((steps + warm)) "${PMC_KERNEL:-_kernel}" <<'PY'
import csv, glob, collections, re, sys
O, steps, want = sys.argv[1], int(sys.argv[2]), sys.argv[3]
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)
for f in glob.glob(O + '/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if want in r['Kernel_Name']:
            k = (short(r['Kernel_Name']), r['Counter_Name'])
            tot[k] += float(r['Counter_Value']); cnt[k] += 1
with open(O + '/summary.txt', 'w') as out:
    for kern in sorted({k[0] for k in tot}):
        head = f"# {kern}   ({steps} bench steps incl. warm-up)"
        print(head); out.write(head + "\n")
        for k in sorted(tot):
            if k[0] != kern:
                continue
            line = f"{k[1]:32s} per_step={tot[k]/steps:.6g}  dispatches={cnt[k]}"
            print(line); out.write(line + "\n")
PY
