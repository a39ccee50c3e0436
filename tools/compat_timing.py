"""PCIe-inclusive timing of the host-pointer entry points on one C360/8 row shard
(6 998 400 gridcells): the reference's own call sequence (XGDMatrixCreateFromMat ->
XGBoosterPredict -> XGDMatrixFree) and the fused OHXBoosterPredictFields, from pageable host
arrays, as a Fortran caller would hand them over.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quickchem_amd import capi, synth  # noqa: E402


def main():
    grid = synth.GRIDS["C360"]
    n = grid[0] * grid[1] * grid[2] // 8
    model = synth.make_model()
    b = capi.Booster(model_buffer=model.image)
    rows = synth.rows_cpu(grid, 3 * n, n)
    out = {}
    # "compat": with the one-line layout hint the Fortran mirror adds; "reference": the reference's own five calls and
    # nothing else (the first predict on the matrix looks for the level size: one more wait).  The first repetition
    # pays hipMalloc of the matrix; the later ones take the parked buffer, as the second tick of a run does.
    for name, hint in (("compat", True), ("reference", False)):
        reps = []
        for rep in range(4):
            t0 = time.perf_counter()
            d = capi.DMatrix(rows, missing=synth.XX_MISS)
            if hint:
                d.set_grid(grid[0], grid[1], 3 * n)
            t1 = time.perf_counter()
            p = b.predict(d)
            t2 = time.perf_counter()
            d.free()
            t3 = time.perf_counter()
            reps.append({"create_ms": (t1 - t0) * 1e3, "predict_ms": (t2 - t1) * 1e3, "free_ms": (t3 - t2) * 1e3,
                         "gridcells_per_s": n / (t3 - t0)})
        out[name] = {"first_tick": reps[0], "later_tick": reps[-1]}
    # fused: a k-slab of a (im, jm/8, km) sub-domain with all levels
    im, jm, km = grid[0], grid[1] // 8, grid[2]
    sub = (im, jm, km)
    fields = [np.ascontiguousarray(synth.field_cpu(sub, f).T) for f in range(27)]
    oh = np.zeros(im * jm * km, dtype=np.float32)
    for rep in range(3):
        t0 = time.perf_counter()
        b.predict_fields(fields, synth.IS2D, synth.PL_FEATURE, im, jm, km, 1, km, synth.XX_MISS, oh, ohscale=0.85)
        t1 = time.perf_counter()
        out["fused"] = {"call_ms": (t1 - t0) * 1e3, "gridcells_per_s": im * jm * km / (t1 - t0)}
    out["rows"] = n
    out["bytes_h2d"] = rows.nbytes
    print(json.dumps(out))


if __name__ == "__main__":
    main()
