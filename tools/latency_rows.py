#!/usr/bin/env python3
"""Latency of one predict on N device-resident rows, N = 1e4 ... 1e6: what a rank-sized block costs
when nothing crosses PCIe.  Rows are the first N of the C360 batch (so a whole number of levels only from 777 600 on);
with the grid hint, with the level size the library finds by itself, and with nothing known (64 consecutive rows).
Median and p95 of 50 calls, each call = OHXBoosterPredictDevice + stream synchronise.  Prints one JSON object.
usage (GPU box): python3 tools/latency_rows.py"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quickchem_amd import capi, synth  # noqa: E402


def main():
    grid = synth.GRIDS["C360"]
    model = synth.make_model()
    b = capi.Booster(model_buffer=model.image)
    nmax = 2_000_000
    rows = torch.empty((nmax, synth.NFEAT), dtype=torch.float32, device="cuda")
    synth.rows_device(grid, 0, nmax, rows)
    out = torch.empty(nmax, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    res = {"rows": {}, "booster": {"trees": model.num_trees, "nodes": model.num_nodes}, "calls": 50}
    for n in (10_000, 30_000, 100_000, 300_000, 777_600, 1_000_000, 1_555_200):
        entry = {}
        for mode in ("hint", "consecutive"):
            d = capi.DMatrix(device_ptr=rows.data_ptr(), nrow=n, ncol=synth.NFEAT, missing=synth.XX_MISS)
            if mode == "hint":
                d.set_grid(grid[0], grid[1], 0)
            else:
                d.set_grid(0, 0, 0)
            ts = []
            for it in range(55):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                b.predict_device(d, out.data_ptr(), stream=stream)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ts = sorted(ts[5:])
            entry[mode] = {"p50_us": ts[len(ts) // 2] * 1e6, "p95_us": ts[int(0.95 * (len(ts) - 1))] * 1e6,
                           "gridcells_per_s_at_p50": n / ts[len(ts) // 2]}
            d.free()
        res["rows"][str(n)] = entry
    b.check()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
