#!/bin/bash
# usage: tools/pmc.sh <tag> <bench args...>   — separate rocprofv3 --pmc passes over bench.py (short run),
# then a one-screen summary of the kernels' counters, per bench step and per kernel.
#   PMC_KERNEL=<substring>  only kernels whose name contains it (default: every kernel whose name contains "_kernel")
#   PMC_STEPS / PMC_WARMUP  steps of the short run (default 2 / 1); per_step divides by their sum
# The run never includes bench.py's whole-batch cross-check (--no-verify) or the CPU leg (--cpu-seconds 0).
tag=$1; shift
steps=${PMC_STEPS:-2}; warm=${PMC_WARMUP:-1}
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/pmc_$tag && mkdir -p $O && cd $R
# counter groups, one rocprofv3 pass each (<= 8 SQ counters per pass).  PMC_SETS=base|busy|all (default all):
#   base = rounds 1-4's set;  busy = (r5) what says WHICH unit is busy: cycles a wave spends issuing VALU / LDS / VMEM /
#   scalar instructions (quad-cycles), CU-busy cycles, the addresser's and the data path's stall reasons.
base=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
      "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
      "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
      "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum")
busy=("SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS"
      "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_CYCLES"
      "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
      "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum"
      "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum")
case "${PMC_SETS:-all}" in base) sets=("${base[@]}");; busy) sets=("${busy[@]}");; *) sets=("${base[@]}" "${busy[@]}");; esac
for c in "${sets[@]}"; do
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
