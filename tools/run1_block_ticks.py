#!/usr/bin/env python3
"""A GEOS rank's OH tick, alone on the GPU, for a look at its kernels (measurement aid): OHXBoosterRun1's HOST form on one
rank-sized block (default 48 x 24 x 72) whose arrays are registered (ohx_register_host), --ticks times; prints the median
tick.  Under `rocprofv3 --kernel-trace --stats -- python3 tools/run1_block_ticks.py` the per-kernel averages say where
the device side of a tick goes.  usage (GPU box): python3 tools/run1_block_ticks.py [--block 48,24,72] [--ticks 200]
[--param name=value ...]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--block", default="48,24,72")
    ap.add_argument("--ticks", type=int, default=200)
    ap.add_argument("--trees", type=int, default=100)
    ap.add_argument("--depth", type=int, default=18)
    ap.add_argument("--param", action="append", default=[])
    ap.add_argument("--call", default="run1", choices=["run1", "fused"], help="OHXBoosterRun1 (default) or OHXBoosterPredictFields, host forms")
    ap.add_argument("--ab", default="", help="name=a,b: alternate two values of a parameter from tick to tick and report both")
    args = ap.parse_args()
    from quickchem_amd import capi, synth
    block = tuple(int(x) for x in args.block.split(","))
    model = synth.make_model(num_trees=args.trees, max_depth=args.depth, sample_log2=20)
    booster = capi.Booster(model_buffer=model.image)
    booster.set_param("ohx_register_host", "1")
    for kv in args.param:
        name, _, val = kv.partition("=")
        booster.set_param(name, val)
    if args.call == "fused":
        # the fused predict-only call from host arrays (OHXBoosterPredictFields): 27 fields in, OH_ML out, all levels
        im, jm, km = block
        fields = [np.ascontiguousarray(synth.field_cpu(block, f).T) for f in range(synth.NFEAT)]
        oh = np.zeros((km, jm, im), dtype=np.float32)

        class Fused:
            def run1_call(self, _):
                booster.predict_fields(fields, synth.IS2D, synth.PL_FEATURE, im, jm, km, 1, km, synth.XX_MISS, oh, ohscale=0.85)
                return {"k1": 1, "k2": km}
        runner, call = Fused(), None
    else:
        st = synth.run1_state(block, seed=5)
        call = booster.run1_prepare(st, dynamic_k_range=True)
        runner = booster
    if args.ab:
        # two settings of one parameter, tick about: the drift of a box over a minute cancels
        name, _, vals = args.ab.partition("=")
        a, b2 = vals.split(",")
        t = {a: [], b2: []}
        for i in range(2 * args.ticks):
            v = (a, b2)[i & 1]
            booster.set_param(name, v)
            t0 = time.perf_counter()
            r = runner.run1_call(call)
            t[v].append(time.perf_counter() - t0)
        for v in (a, b2):
            print(f"block {block}: {name}={v}: tick median {np.median(t[v][5:]) * 1e3:.3f} ms, p95 {np.percentile(t[v][5:], 95) * 1e3:.3f} ms")
        return
    ticks = []
    for _ in range(args.ticks):
        t0 = time.perf_counter()
        r = runner.run1_call(call)
        ticks.append(time.perf_counter() - t0)
    print(f"block {block}: levels {int(r['k1'])}..{int(r['k2'])}, tick median {np.median(ticks[5:]) * 1e3:.3f} ms, "
          f"p95 {np.percentile(ticks[5:], 95) * 1e3:.3f} ms over {args.ticks - 5} ticks")


if __name__ == "__main__":
    main()
