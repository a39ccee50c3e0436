#!/usr/bin/env python3
"""Random grids, shards, brick shapes and lane orders through OHXBoosterPredictDevice on device buffers of exactly
nrow x 27 floats, against the CPU oracle (bit for bit).  Aimed at the geometry of the rows kernel: bricks that
overhang the grid or the shard, runs of rows that start before or end after the matrix, tiles fetched by the wave
together or lane by lane.  usage (GPU box): python3 tools/fuzz_tiles.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quickchem_amd import capi, synth  # noqa: E402
from tests import helpers  # noqa: E402


def run(cases, seed):
    """Returns the descriptions of the cases whose margins differ from the oracle's."""
    rng = np.random.default_rng(seed)
    torch.cuda.set_device(0)
    model = synth.make_model(num_trees=12, max_depth=9, sample_log2=13, min_leaf=2, grid=synth.GRIDS["C12"])
    deep = synth.make_model(num_trees=6, max_depth=16, sample_log2=14, min_leaf=1, grid=synth.GRIDS["C12"])
    bricks = ["auto", "4,4,4", "8,4,2", "8,8,1", "2,2,16", "64,1,1", "16,1,4", "4,16,1", "1,8,8"]
    bad = []
    for c in range(cases):
        im, jm, nk = int(rng.integers(1, 70)), int(rng.integers(1, 40)), int(rng.integers(1, 12))
        n = im * jm * nk
        r0 = int(rng.integers(0, n))
        m = int(rng.integers(1, n - r0 + 1))
        if rng.random() < 0.3:
            r0, m = 0, n
        rows = synth.rows_cpu((im, jm, max(nk, 2)), 0, n) if im * jm >= 4 else synth.rows_cpu((4, 4, 72), 0, n)
        holes = rng.random(rows.shape) < 0.002
        rows[holes] = synth.XX_MISS
        image = (deep if c % 3 == 0 else model).image
        want = helpers.oracle_predict(image, rows[r0:r0 + m], synth.XX_MISS)
        d_rows = torch.from_numpy(rows[r0:r0 + m].copy()).to("cuda:0")
        out = torch.full((m,), 7.0, dtype=torch.float32, device="cuda:0")
        params = {"ohx_brick": bricks[int(rng.integers(0, len(bricks)))], "ohx_brick_k_fastest": str(int(rng.integers(0, 2))),
                  "ohx_coop_rows": "1" if rng.random() < 0.8 else "0", "ohx_tree_tops": ["auto", "on", "off"][int(rng.integers(0, 3))],
                  "ohx_launches_per_residency": str(int(rng.integers(0, 4))),
                  # round 3: small batches with their trees split over waves, rows with missing values left to a second launch
                  "ohx_tree_split": ["auto", "off", "2", "3", "5"][int(rng.integers(0, 5))],
                  "ohx_defer_missing": ["auto", "on", "off"][int(rng.integers(0, 3))]}
        b = capi.Booster(model_buffer=image)
        for k, v in params.items():
            if not (k == "ohx_brick" and v == "auto"):
                b.set_param(k, v)
        d = capi.DMatrix(device_ptr=d_rows.data_ptr(), nrow=m, ncol=27, missing=synth.XX_MISS)
        mode = int(rng.integers(0, 3))
        if mode == 0:
            d.set_grid(im, jm, r0)
        elif mode == 1:
            d.set_grid(0, 0, 0)
        b.predict_device(d, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        b.check()
        if not np.array_equal(helpers.bits(out.cpu().numpy()), helpers.bits(want)):
            bad.append(((im, jm, nk), (r0, m), mode, params))
        d.free()
        b.free()
    return bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    bad = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    for b in bad:
        print("MISMATCH", *b)
    print(f"{cases} cases, {len(bad)} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
