#!/usr/bin/env python3
"""Soak of the ring kernels' barrier-free protocol (measurement aid): the same C360 batch predicted --iters times, the
launch geometry changed from call to call (rounds per launch, CUs left free, XCD remap, a tree limit now and then), every
output compared bit for bit with the `wide` kernel's.  A race between the waves of a block would show as a difference
or as a time-out (since round 5 a re-run by the tile kernel, counted: OHXBoosterRingReruns); prints the count of each.  usage (GPU box): python3 tools/ring_soak.py [--iters 600]"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=600)
    ap.add_argument("--grid", default="C360")
    ap.add_argument("--missing-ppm", type=int, default=0)
    ap.add_argument("--fields", action="store_true", help="the fused call (predict_fields_ring_kernel) instead of the rows call")
    args = ap.parse_args()
    from quickchem_amd import capi, synth
    dev = torch.device("cuda:0")
    grid = synth.GRIDS[args.grid]
    n = grid[0] * grid[1] * grid[2]
    if args.fields:
        return soak_fields(args, grid, n, dev)
    rows = torch.empty((n, synth.NFEAT), dtype=torch.float32, device=dev)
    synth.rows_device(grid, 0, n, rows)
    if args.missing_ppm:
        synth.inject_missing_device(rows, args.missing_ppm)
    model = synth.make_model(num_trees=100, max_depth=18, sample_log2=20)
    stream = torch.cuda.current_stream()
    dm = capi.DMatrix(device_ptr=rows.data_ptr(), nrow=n, ncol=synth.NFEAT, missing=synth.XX_MISS)
    dm.set_grid(grid[0], grid[1], 0)
    plain = capi.DMatrix(device_ptr=rows.data_ptr(), nrow=n, ncol=synth.NFEAT, missing=synth.XX_MISS)
    plain.set_grid(0, 0, 0)
    wide = capi.Booster(model_buffer=model.image)
    wide.set_param("ohx_kernel", "wide")
    refs = {}
    for limit in (0, 37, 64):
        r = torch.empty(n, dtype=torch.float32, device=dev)
        wide.predict_device(plain, r.data_ptr(), stream=stream.cuda_stream, ntree_limit=limit)
        torch.cuda.synchronize()
        wide.check()
        refs[limit] = r
    booster = capi.Booster(model_buffer=model.image)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    rng = random.Random(4)
    wrong = errors = 0
    for it in range(args.iters):
        booster.set_param("ohx_ring_rounds", rng.choice([0, 1, 3, 16, 64, 64, 64]))
        booster.set_param("ohx_reserve_cus", rng.choice([0, 0, 0, 8, 24]))
        booster.set_param("ohx_xcd_remap", rng.choice([1, 1, 0]))
        limit = rng.choice([0, 0, 0, 37, 64])
        out.fill_(float("nan"))
        try:
            booster.predict_device(dm, out.data_ptr(), stream=stream.cuda_stream, ntree_limit=limit)
            torch.cuda.synchronize()
            booster.check()
        except capi.OhxError as e:
            errors += 1
            print(f"iteration {it}: {e}", flush=True)
            continue
        if not torch.equal(out.view(torch.int32), refs[limit].view(torch.int32)):
            wrong += 1
            bad = int((out.view(torch.int32) != refs[limit].view(torch.int32)).sum())
            print(f"iteration {it}: {bad} rows differ", flush=True)
        if it % 100 == 99:
            print(f"{it + 1} predicts, {wrong} wrong, {errors} errors", flush=True)
    reruns = booster.ring_reruns()       # (r5) a time-out is no error any more: the rows are predicted again, and counted
    print(f"ring soak: {args.iters} predicts of {n} rows ({booster.kernel_symbol(27)}), {wrong} with a wrong row, {errors} errors, "
          f"{reruns} time-outs (re-runs)")
    sys.exit(1 if wrong or errors else 0)


def soak_fields(args, grid, n, dev):
    from quickchem_amd import capi, synth
    im, jm, km = grid
    plane = im * jm
    fields = []
    for f in range(synth.NFEAT):
        t = torch.empty(plane * (1 if synth.IS2D[f] else km), dtype=torch.float32, device=dev)
        synth.field_device(grid, f, t)
        if args.missing_ppm:
            synth.inject_missing_device(t, args.missing_ppm)
        fields.append(t)
    ptrs = [t.data_ptr() for t in fields]
    model = synth.make_model(num_trees=100, max_depth=18, sample_log2=20)
    stream = torch.cuda.current_stream()

    def run(b, oh, k1):
        b.predict_fields_device(ptrs, synth.IS2D, synth.PL_FEATURE, im, jm, km, k1, km, synth.XX_MISS, oh.data_ptr(),
                                apply_pow10=True, ohscale=0.85, stream=stream.cuda_stream)
        torch.cuda.synchronize()
        b.check()
    wide = capi.Booster(model_buffer=model.image)
    wide.set_param("ohx_kernel", "wide")
    refs = {}
    for k1 in (1, 22):
        refs[k1] = torch.zeros(plane * km, dtype=torch.float32, device=dev)
        run(wide, refs[k1], k1)
    booster = capi.Booster(model_buffer=model.image)
    out = torch.empty(plane * km, dtype=torch.float32, device=dev)
    rng = random.Random(5)
    wrong = errors = 0
    for it in range(args.iters):
        booster.set_param("ohx_ring_rounds", rng.choice([0, 1, 3, 16, 64, 64, 64]))
        booster.set_param("ohx_xcd_remap", rng.choice([1, 1, 0]))
        k1 = rng.choice([1, 1, 22])
        out.zero_()
        try:
            run(booster, out, k1)
        except capi.OhxError as e:
            errors += 1
            print(f"iteration {it}: {e}", flush=True)
            continue
        if not torch.equal(out.view(torch.int32), refs[k1].view(torch.int32)):
            wrong += 1
            print(f"iteration {it}: {int((out.view(torch.int32) != refs[k1].view(torch.int32)).sum())} gridcells differ", flush=True)
        if it % 100 == 99:
            print(f"{it + 1} fused calls, {wrong} wrong, {errors} errors", flush=True)
    reruns = booster.ring_reruns()
    print(f"ring soak: {args.iters} fused calls on {n} gridcells ({booster.fields_kernel_symbol(n)}), {wrong} with a wrong gridcell, "
          f"{errors} errors, {reruns} time-outs (re-runs)")
    sys.exit(1 if wrong or errors else 0)


if __name__ == "__main__":
    main()
