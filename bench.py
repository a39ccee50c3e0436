#!/usr/bin/env python3
"""bench.py — OH gridcells/s of the XGBoost-predict hot path on MI355X.

A step = one pass of the hot path over the synthetic C360 L72 batch
(55 987 200 gridcells x 27 float32 features, already resident in HBM as the
row-major xx_carr the reference builds): OHXBoosterPredictDevice -> raw OH margins
in HBM.  With N > 1 ranks (one process per GPU, launched by torch.distributed.run)
the batch is cut into N contiguous row shards, every rank predicts its shard with a
replicated booster, and one RCCL all-gather over xGMI reassembles the OH field on
every GPU — that is part of the step.

Prints ONE JSON line (rank 0).  `roofline` prices the predict kernel against the
HBM roofline with the ALGORITHMIC bytes of SURVEY.md §8(d): 112 B per gridcell plus
8 B per node slot once per launch; `cpu_baseline` is the CPU oracle (own restatement
of xgboost 1.6.0 semantics; "port") timed on this node's host cores on a bounded
sample of the same rows.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
BYTES_PER_CELL = 112           # 27 x 4 B read + 4 B write


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", default="C360", help="BASELINE.json workload grid (default: the headline C360 L72)")
    ap.add_argument("--kernel", default="auto", help="auto | wide | packed1 | packed2 | packed4")
    ap.add_argument("--trees", type=int, default=100)
    ap.add_argument("--depth", type=int, default=18)
    ap.add_argument("--sample-log2", type=int, default=20)
    ap.add_argument("--top-levels", type=int, default=None)
    ap.add_argument("--line-slots", type=int, default=None)
    ap.add_argument("--param", action="append", default=[], help="extra XGBoosterSetParam name=value (repeatable)")
    ap.add_argument("--missing-ppm", type=int, default=0, help="inject -999.0/NaN at this rate per million entries")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--shuffle", action="store_true", help="permute the rows (destroys spatial coherence)")
    ap.add_argument("--infer-grid", action="store_true",
                    help="rows path: no hint either, but let the library look for the level size in the rows "
                         "(OHXDMatrixInferGrid), as XGDMatrixCreateFromMat does unasked for host matrices")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed steps rank 0 predicts the whole batch in one piece, untiled, and compares "
                         "the gathered field with it bit for bit (small grids: it generates all rows on rank 0)")
    ap.add_argument("--no-grid", action="store_true",
                    help="rows path: do not tell the library which grid the rows come from (OHXDMatrixSetGrid); "
                         "waves then take 64 consecutive rows instead of bricks of neighbouring gridcells")
    ap.add_argument("--gather-chunks", type=int, default=4,
                    help="N > 1: cut the shard into this many pieces so that a piece's all-gather overlaps the next "
                         "piece's prediction (1 = predict everything, then one all-gather)")
    ap.add_argument("--path", default="rows", choices=["rows", "fields", "run1"],
                    help="rows: AoS xx_carr -> margins (the headline); fields: the fused SoA call, 27 MAPL fields -> "
                         "10**pred*OHscale; run1: OHXBoosterRun1Device, imports -> INTERNAL OH (both 1 GPU only)")
    return ap.parse_args()


def cpu_baseline(model_image, rows_dev, budget_s):
    """The oracle, all host cores (OpenMP), on the first S rows of the batch; S sized to the budget."""
    from quickchem_amd import capi, synth
    lib = capi.declare_xgb_api(C.CDLL(os.path.join(ROOT, "oracle", "lib", "liboracle_xgb.so")))
    lib.oracle_num_threads.restype = C.c_int
    cores = int(lib.oracle_num_threads())
    booster = capi.Booster(model_buffer=model_image, lib=lib)

    def run(n):
        host = rows_dev[:n].cpu().numpy()
        t0 = time.perf_counter()
        d = capi.DMatrix(host, missing=synth.XX_MISS, lib=lib)     # XGDMatrixCreateFromMat is on the path
        out = booster.predict(d)
        d.free()
        return time.perf_counter() - t0, out

    probe = min(524288, rows_dev.shape[0])
    run(min(4096, probe))                       # thread pool and page faults out of the way
    t_probe, _ = run(probe)
    rate = probe / max(t_probe, 1e-6)
    n = int(min(rows_dev.shape[0], max(probe, 0.7 * rate * budget_s)))
    n = max(64, n // 64 * 64)
    t, out = run(n)
    # how one GEOS rank runs it (one thread), on a slice sized to about two seconds
    lib.oracle_set_num_threads.argtypes = [C.c_int]
    lib.oracle_set_num_threads(1)
    tp, _ = run(8192)
    n1 = max(8192, int(min(n, 2.0 * 8192 / max(tp, 1e-6))) // 64 * 64)
    t1, _ = run(n1)
    lib.oracle_set_num_threads(cores)
    return {"value": n / t, "unit": "gridcells/s", "cores": cores, "kind": "port",
            "sample": f"first {n} rows of the batch, oracle/xgb_oracle.c (OpenMP, {cores} threads), "
                      f"XGDMatrixCreateFromMat + XGBoosterPredict, {t:.2f} s",
            "one_thread": {"value": n1 / t1, "unit": "gridcells/s", "cores": 1,
                           "sample": f"first {n1} rows, {t1:.2f} s"}}, out, n


def bench_run1(args, grid, n_total, model, booster, dev, t_model):
    """OH Run1 from the imports to INTERNAL OH, everything resident in HBM."""
    import ctypes as C
    from quickchem_amd import capi, synth
    im, jm, km = grid
    plane, vol, edge = im * jm, im * jm * km, im * jm * (km + 1)
    g = torch.Generator(device=dev).manual_seed(23)

    def u(n, lo, hi):
        return (lo + (hi - lo) * torch.rand(n, device=dev, generator=g, dtype=torch.float32)).contiguous()
    keep = {}
    a = capi.OHXRun1Args()
    a.im, a.jm, a.km = im, jm, km
    a.dynamic_k_range, a.tropp_min, a.ohscale, a.missing = 1, 4000.0, 0.85, synth.XX_MISS
    a.avogad, a.runiv, a.epsilon = 6.023e26, 8314.47, 18.015 / 28.965
    sig = (torch.arange(km + 1, device=dev, dtype=torch.float32) / km) ** 2
    ps = u(plane, 6.0e4, 1.04e5)
    keep["ple_mod"] = (1.0 + (ps[None, :] - 1.0) * sig[:, None]).contiguous().reshape(-1)        # (im,jm,0:km) Fortran order
    keep["ple_bst"] = keep["ple_mod"].clone()
    keep["zle_bst"] = (8.0e4 * (1.0 - sig[:, None]) * u(plane, 0.9, 1.1)[None, :]).contiguous().reshape(-1)
    for name, lo, hi, n in (("t_mod", 190, 310, vol), ("q_mod", 1e-7, 2e-2, vol), ("tropp_mod", 9e3, 3e4, plane),
                            ("tauclw", 0, 4, vol), ("taucli", 0, 2, vol), ("gmito3", 250, 450, plane),
                            ("gmitto3", 20, 60, plane), ("lat_deg", -90, 90, plane), ("albuv", 0.02, 0.9, plane),
                            ("sza", 0, 113, plane), ("default_oh", 1e-15, 5e-13, vol)):
        keep[name] = u(n, lo, hi)
    # the features that are used as they are come from the seeded generator (spatially coherent)
    direct = {"t_bst": 2, "no2": 3, "o3": 4, "ch4": 5, "co": 6, "isop": 7, "acet": 8, "c2h6": 9, "c3h8": 10, "prpe": 11,
              "alk4": 12, "mp": 13, "h2o2": 14, "cloud": 19, "qv": 20, "ch2o": 25}
    for name, f in direct.items():
        t = torch.empty(vol, dtype=torch.float32, device=dev)
        synth.field_device(grid, f, t)
        keep[name] = t
    sca = [u(vol, 0, 5e-6) for _ in range(7)]
    a.scacoef = (C.c_void_p * 7)(*[t.data_ptr() for t in sca])
    for name, t in keep.items():
        setattr(a, name, t.data_ptr())
    oh = torch.empty(vol, dtype=torch.float32, device=dev)
    boost = torch.empty(vol, dtype=torch.float32, device=dev)
    k1, k2 = C.c_int32(), C.c_int32()
    a.oh, a.oh_boost, a.ndwet = oh.data_ptr(), boost.data_ptr(), None
    a.k1, a.k2 = C.cast(C.pointer(k1), C.c_void_p), C.cast(C.pointer(k2), C.c_void_p)
    lib = booster.lib
    stream = torch.cuda.current_stream()

    def step():
        capi.check(lib, lib.OHXBoosterRun1Device(booster.handle, C.byref(a), stream.cuda_stream))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    per_step = (time.perf_counter() - t0) / args.steps
    booster.check()
    nslab = plane * (k2.value - k1.value + 1)
    print(json.dumps({
        "metric": "OH gridcells/sec (XGBoost predict), C360 L72 batch", "value": nslab / per_step,
        "unit": "gridcells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": per_step * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.grid} L{km}: OH Run1 imports -> INTERNAL OH in HBM (feature engineering, "
                               f"k-slab {k1.value}..{k2.value}, predict, tropopause mask, unit conversion)",
                   "grid": list(grid), "rows_predicted": nslab, "kernel": args.kernel, "params": args.param},
        "roofline": None, "cpu_baseline": None}), flush=True)


def measured_traffic(grid_name, kernel, model_nodes):
    """HBM-side bytes per step from the committed rocprofv3 --pmc passes (profiles/), if they
    were taken on this very workload; None otherwise (PMC cannot be collected from in here)."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None
    for name in sorted(os.listdir(pdir)):
        if not name.endswith("_traffic.json"):
            continue
        try:
            t = json.load(open(os.path.join(pdir, name)))
        except Exception:
            continue
        if t.get("workload") == grid_name and t.get("model_nodes") == model_nodes and \
                kernel in ("auto", t.get("kernel")):
            best = t.get("traffic_bytes_per_step")
    return best


def gather_issue(info, nrows, kernel_s, dev):
    import torch
    props = torch.cuda.get_device_properties(dev)
    tiles = -(-nrows // 64)
    gathers = tiles * (info.get("gathers_per_wave", 0) + 7)          # + the 7 loads of a tile's 27-float rows
    clock_hz = 2.4e9                                                 # MI355X max clock (MI355X_MICROARCH.md)
    peak = props.multi_processor_count * clock_hz / 14.0
    return {"unit": "wave64 gathers/s", "achieved": gathers / kernel_s, "peak": peak,
            "frac": gathers / kernel_s / peak,
            "gathers_per_wave_and_tree": info.get("gathers_per_wave", 0) / max(info["num_trees"], 1),
            "min_cycles_per_gather": 14, "cus": props.multi_processor_count, "clock_mhz": clock_hz / 1e6}


def bench_fields(args, grid, n_total, model, booster, dev, t_model):
    """The fused SoA variant (23 3-D reads + 4 2-D reads + 1 write per gridcell)."""
    from quickchem_amd import synth
    im, jm, km = grid
    plane = im * jm
    fields = []
    for f in range(synth.NFEAT):
        t = torch.empty(plane * (1 if synth.IS2D[f] else km), dtype=torch.float32, device=dev)
        synth.field_device(grid, f, t)
        fields.append(t)
    oh = torch.zeros(plane * km, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()
    ptrs = [t.data_ptr() for t in fields]

    def step():
        booster.predict_fields_device(ptrs, synth.IS2D, synth.PL_FEATURE, im, jm, km, 1, km, synth.XX_MISS,
                                      oh.data_ptr(), apply_pow10=True, ohscale=0.85, stream=stream.cuda_stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    booster.check()
    info = booster.info()
    algo = 4 * (23 * n_total + 4 * plane + n_total) + info["node_bytes"]
    per_step = elapsed / args.steps
    achieved = algo / per_step / 1e9
    print(json.dumps({
        "metric": "OH gridcells/sec (XGBoost predict), C360 L72 batch", "value": n_total / per_step,
        "unit": "gridcells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": per_step * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.grid} L{km}: fused SoA call, 27 MAPL fields in HBM -> 10**pred * OHscale",
                   "grid": list(grid), "rows_total": n_total, "kernel": args.kernel, "params": args.param,
                   "booster": {"trees": info["num_trees"], "nodes": info["num_nodes"], "node_bytes": info["node_bytes"],
                               "build_s": round(t_model, 2)}},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "predict_fields_kernel<2,2>",
                     "kernel_ms": per_step * 1e3, "algorithmic_bytes": algo},
        "cpu_baseline": None}), flush=True)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    # rehearsal of the multi-rank path on a one-GPU box: OHX_BENCH_SHARE_GPU=1 puts every rank on device 0
    # (RCCL refuses two ranks on one device, so OHX_BENCH_BACKEND=gloo goes with it); never for numbers
    dev_index = 0 if os.environ.get("OHX_BENCH_SHARE_GPU") == "1" else local_rank
    backend = os.environ.get("OHX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    force_dist = os.environ.get("OHX_BENCH_FORCE_DIST") == "1"     # exercise the RCCL calls with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if force_dist and world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend=backend)

    from quickchem_amd import capi, shard, synth

    grid = synth.GRIDS[args.grid]
    n_total = grid[0] * grid[1] * grid[2]
    row0, n_local = shard.row_shard(n_total, world, rank)

    # ---- the booster: seeded synthetic OH model, identical on every rank ----
    synth.set_threads(max(1, (os.cpu_count() or 8) // max(world, 1)))
    t0 = time.perf_counter()
    model = synth.make_model(num_trees=args.trees, max_depth=args.depth, sample_log2=args.sample_log2)
    booster = capi.Booster(model_buffer=model.image)
    booster.set_param("ohx_kernel", args.kernel)
    if args.top_levels is not None:
        booster.set_param("ohx_top_levels", args.top_levels)
    if args.line_slots is not None:
        booster.set_param("ohx_line_slots", args.line_slots)
    for kv in args.param:
        name, _, val = kv.partition("=")
        booster.set_param(name, val)
    t_model = time.perf_counter() - t0

    if args.path in ("fields", "run1"):
        if world != 1:
            raise SystemExit(f"--path {args.path} is a single-GPU measurement")
        fn = bench_fields if args.path == "fields" else bench_run1
        return fn(args, grid, n_total, model, booster, dev, t_model)

    # ---- the batch: this rank's contiguous row shard, generated in HBM ----
    rows = torch.empty((n_local, synth.NFEAT), dtype=torch.float32, device=dev)
    synth.rows_device(grid, row0, n_local, rows)
    if args.missing_ppm:
        synth.inject_missing_device(rows, args.missing_ppm)
    if args.shuffle:
        perm = torch.randperm(n_local, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
        rows = rows[perm].contiguous()
        del perm
    torch.cuda.synchronize()
    out_local = torch.empty(n_local, dtype=torch.float32, device=dev)
    even = (n_total % world == 0)
    # pieces of the shard: one DMatrix view each (device pointers into `rows`)
    gather = world > 1 or force_dist
    use_grid = not (args.no_grid or args.shuffle or args.infer_grid)
    plane = grid[0] * grid[1]
    # pieces are whole levels when the shard is (bricks then have no idle lanes), else whole launches
    granule = plane if (use_grid and row0 % plane == 0 and n_local % plane == 0) else 64 * 256 * 20 * 2
    round_rows = torch.cuda.get_device_properties(dev).multi_processor_count * 20 * 64   # one residency of the chip
    pieces = shard.plan_pieces(n_local, args.gather_chunks if (gather and even) else 1, granule, round_rows)
    dmats = [capi.DMatrix(device_ptr=rows.data_ptr() + lo * synth.NFEAT * 4, nrow=hi - lo, ncol=synth.NFEAT,
                          missing=synth.XX_MISS) for lo, hi in pieces]
    if use_grid:
        for (lo, hi), dm in zip(pieces, dmats):
            dm.set_grid(grid[0], grid[1], row0 + lo)
    elif args.infer_grid:
        for dm in dmats:
            dm.infer_grid()
    out_full = torch.empty(n_total, dtype=torch.float32, device=dev) if gather else out_local
    stream = torch.cuda.current_stream()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None):
        if i is not None:
            ev0[i].record(stream)
        works = []
        for (lo, hi), dm in zip(pieces, dmats):
            booster.predict_device(dm, out_local.data_ptr() + lo * 4, stream=stream.cuda_stream)
            if gather and len(pieces) > 1:
                works.append(shard.all_gather_chunk_async(out_full, out_local, lo, hi, n_local, world))
        if i is not None:
            ev1[i].record(stream)      # with N > 1 this spans the predict launches of all pieces
        if gather and len(pieces) == 1:
            shard.all_gather_rows(out_full, out_local, n_total, world, even)
        for w in works:
            w.wait()

    def fence():
        torch.cuda.synchronize()
        if gather:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    booster.check()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    verified = None
    if args.verify:
        ok = 1
        if rank == 0:
            all_rows = torch.empty((n_total, synth.NFEAT), dtype=torch.float32, device=dev)
            synth.rows_device(grid, 0, n_total, all_rows)
            if args.missing_ppm:
                raise SystemExit("--verify regenerates the rows: not with --missing-ppm")
            ref = torch.empty(n_total, dtype=torch.float32, device=dev)
            dm_all = capi.DMatrix(device_ptr=all_rows.data_ptr(), nrow=n_total, ncol=synth.NFEAT, missing=synth.XX_MISS)
            booster.predict_device(dm_all, ref.data_ptr(), stream=stream.cuda_stream)
            torch.cuda.synchronize()
            booster.check()
            ok = int(torch.equal(ref.view(torch.int32), out_full.view(torch.int32)))
            del all_rows, ref
        if world > 1:
            t = torch.tensor([ok], dtype=torch.int64, device=dev)
            dist.broadcast(t, src=0)
            ok = int(t.item())
        verified = bool(ok)
        if not verified:
            raise SystemExit("bench --verify: the gathered field differs from the one-piece prediction")
    kernel_ms = [a.elapsed_time(b) for a, b in zip(ev0, ev1)]
    kernel_s = float(np.mean(kernel_ms)) * 1e-3
    if world > 1:
        t = torch.tensor([kernel_s], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kernel_s = float(t.item())

    info = booster.info()
    launches_per_step = sum(-(-((hi - lo + 63) // 64) // (256 * 20 * 2)) for lo, hi in pieces)  # per 2 residencies
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total / (elapsed / args.steps)
    algo_bytes = BYTES_PER_CELL * n_local + info["node_bytes"]
    achieved = algo_bytes / kernel_s / 1e9

    # ---- self-check + CPU baseline (rank 0, N = 1): the oracle on a bounded sample of the same rows ----
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        cpu, want, n_chk = cpu_baseline(model.image, rows, args.cpu_seconds)
        got = out_local[:n_chk].cpu().numpy()
        if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
            raise SystemExit("bench: GPU margins differ from the oracle on the cpu_baseline sample")
    if gather:
        # every rank must hold the whole field, and the shards in row order
        lo = out_full[row0:row0 + n_local]
        if not torch.equal(lo.view(torch.int32), out_local.view(torch.int32)):
            raise SystemExit("bench: all-gather did not put this rank's shard at its rows")

    if rank == 0:
        line = {
            "metric": "OH gridcells/sec (XGBoost predict), C360 L72 batch",
            "value": value, "unit": "gridcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{args.grid} L{grid[2]}: {n_total} gridcells x 27 float32 features (AoS rows in HBM), "
                            f"{world} contiguous row shard(s)" + (", all-gather of the OH field" if world > 1 else ""),
                "grid": list(grid), "rows_total": n_total, "rows_per_gpu": n_local,
                "booster": {"trees": info["num_trees"], "max_depth": info["max_depth"], "nodes": info["num_nodes"],
                          "node_slots": info["num_slots"], "node_bytes": info["node_bytes"],
                          "mean_path": round(model.mean_path, 3), "seed": synth.MODEL_SEED,
                          "build_s": round(t_model, 2)},
                "kernel": args.kernel, "params": args.param, "missing_ppm": args.missing_ppm, "shuffled": bool(args.shuffle), "grid_hint": bool(use_grid), "grid_inferred": bool(args.infer_grid), "verified": verified,
                "parallelism": f"rows{world}", "gather_pieces": len(pieces) if gather else 0,
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic(args.grid, args.kernel, info["num_nodes"])
                         if (world == 1 and use_grid and not args.shuffle and not args.missing_ppm and not args.param) else None,
                         "kernel": "predict_rows_tile_kernel<2,2>", "kernel_ms": kernel_s * 1e3,
                         "per": "step = the train of launches of one pass over the batch",
                         "launches_per_step": launches_per_step,
                         "avg_launch_us": kernel_s * 1e6 / launches_per_step,
                         "algorithmic_bytes": algo_bytes,
                         # what actually bounds the walk: gather instructions through the
                         # texture addresser, priced at the 14 cycles a wave64 gather costs at the very least
                         "gather_issue": gather_issue(info, n_local, kernel_s, dev)},
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if gather:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
