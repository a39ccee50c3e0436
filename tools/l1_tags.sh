#!/bin/bash
# usage: tools/l1_tags.sh <tag> [table KiB]   (on the GPU box) - tools/bin/l1_tag_microbench plain, then under rocprofv3
# --pmc with the L1's tag-conflict counter; one table: pattern, cycles per gather, tag-conflict stall cycles and tag
# look-ups per gather (per CU).  Output: gpurun_out/l1_tags_<tag>/summary.txt
tag=$1; kib=${2:-1024}
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/l1_tags_$tag && mkdir -p $O && cd $R
timeout -k 10 120 tools/bin/l1_tag_microbench $kib > $O/plain.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/pmc -- tools/bin/l1_tag_microbench $kib > $O/pmc.log 2>&1 || exit 1
python3 - $O <<'PY'
import csv, glob, collections, sys
O = sys.argv[1]
rows = [ln.rstrip("\n") for ln in open(O + "/plain.txt") if not ln.startswith("#") and not ln.startswith("pattern")]
per = collections.defaultdict(dict)
for f in glob.glob(O + "/pmc/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "tag_kernel" in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(per)
timed = ids[1::2]                      # every row: a warm-up dispatch, then the timed one
gathers = 256 * 16 * 2048              # CUs x waves per CU x gathers per wave (the tool's constants)
with open(O + "/summary.txt", "w") as out:
    head = open(O + "/plain.txt").readline().rstrip()
    print(head); out.write(head + "\n")
    hdr = "%-34s %10s %10s %14s %14s %12s" % ("pattern", "ms", "cyc/gather", "tagconfl/gath", "lookups/gath", "L2 req/gath")
    print(hdr); out.write(hdr + "\n")
    for ln, d in zip(rows, timed):
        c = per[d]
        line = "%s %14.2f %14.2f %12.2f" % (ln, c.get("TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", 0) / gathers,
                                             c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / gathers, c.get("TCP_TCC_READ_REQ_sum", 0) / gathers)
        print(line); out.write(line + "\n")
PY
