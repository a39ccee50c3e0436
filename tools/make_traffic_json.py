#!/usr/bin/env python3
"""profiles/rNN_traffic.json from a round's PMC passes (tools/profile_round.sh).
usage: make_traffic_json.py pmc_summary.txt calib_fetch_write.txt bench_plain.json > traffic.json

HBM-side bytes of the predict kernel per bench step, as MI355X_MICROARCH.md prescribes for FETCH_SIZE / WRITE_SIZE
from separate rocprofv3 --pmc passes: both counters are in KiB-ish units of 1 KB = 1000 B on this pool's
rocprofv3 (checked against a pure stream in round 1), FETCH_SIZE under-reports an access pattern that is not the
16-B contiguous stream and is calibrated on this kernel's own row stream (1 tree of depth 0: a step reads exactly
6 046 617 600 B of rows), WRITE_SIZE is taken as read.  Infinity-Cache hits are counted, so this is L2-miss
traffic: an upper bound on HBM bytes.  The kernel source hash ties the number to the code it was measured on;
bench.py emits it as roofline.traffic only on a match."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def value(path, name):
    for line in open(path):
        m = re.match(rf"\s*{name}\s+per_step=([0-9.eE+-]+)", line)
        if m:
            return float(m.group(1))
    raise SystemExit(f"{name} not in {path}")


def opt(path, name):
    try:
        return value(path, name)
    except SystemExit:
        return None


def main():
    pmc, calib, bench = sys.argv[1:4]
    import bench as bench_py
    line = json.loads(open(bench).read().strip().splitlines()[-1])
    fetch_kb, write_kb = value(pmc, "FETCH_SIZE"), value(pmc, "WRITE_SIZE")
    rows_bytes = 112 * 0 + 27 * 4 * line["config"]["rows_total"]
    scale = rows_bytes / (value(calib, "FETCH_SIZE") * 1000.0)
    traffic = fetch_kb * 1000.0 * scale + write_kb * 1000.0
    print(json.dumps({
        "comment": "Memory-side traffic of the predict kernel per bench step, separate rocprofv3 --pmc passes; "
                   "FETCH_SIZE calibrated on this kernel's own row stream (see tools/make_traffic_json.py), WRITE_SIZE as read; "
                   "Infinity-Cache hits included (upper bound on HBM bytes).",
        "workload": line["config"]["grid"] and "C360" if line["config"]["grid"] == [360, 2160, 72] else str(line["config"]["grid"]),
        "kernel": "ring", "model_nodes": line["config"]["booster"]["nodes"],
        "fetch_size_kb": fetch_kb, "write_size_kb": write_kb, "fetch_calibration": round(scale, 4),
        "traffic_bytes_per_step": round(traffic),
        # (r5) the counters bench.py's roofline.valu_issue / roofline.ta_busy are made of, same passes, per step
        "counters_per_step": {name: opt(pmc, name) for name in
                              ("SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "TA_TA_BUSY_sum", "SQ_BUSY_CU_CYCLES",
                               "TCP_TOTAL_CACHE_ACCESSES_sum")},
        "kernel_source_hash": bench_py.kernel_source_hash(),
        "kernel_ms_when_measured": line["roofline"]["kernel_ms"]}, indent=1))


if __name__ == "__main__":
    main()
