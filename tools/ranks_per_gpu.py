#!/usr/bin/env python3
"""How the reference really calls the path on a node: many MPI ranks, each with a small block in pageable host
arrays, all sharing one GPU (OH_GridCompMod.F90:1199-1202: a rank predicts its own im x jm x km block; :305-383: the
five calls per OH tick; NOTES.wiki:14,33: one rank per core).

Starts P fresh processes on ONE GPU, P in --ranks.  Each is a "rank" that owns C360 / (8 P) gridcells (a
(360, 2160 / (8 P), 72) sub-domain), loads the model FROM A FILE as a GEOS rank does (XGBoosterLoadModel), and runs
--ticks ticks of
   reference:  XGDMatrixCreateFromMat -> XGBoosterPredict -> XGDMatrixFree   (the booster made and loaded at the
               first tick, as predict_OH_with_XGB does: :242-271), and
   fused:      OHXBoosterPredictFields (gather, PL/100, walk, 10**, *OHscale in one call)
   run1:       OHXBoosterRun1, the HOST form - the one call quickchem_amd/fortran/oh_gridcomp.F90 makes on a Boost tick
               (oh_run1_boost: 37 import arrays in, INTERNAL OH, OH_boost and NDWET out; OH_GridCompMod.F90:1444-1595)
   run1_registered:  the same with XGBoosterSetParam("ohx_register_host", "1"): the arrays are registered with the
               driver at their first tick and read / written by the GPU in place from then on - by copy kernels
               (run1_registered: ohx_copy_engine = kernel, the default), by the DMA engines (run1_registered_dma), or
               whichever of the two the library's own trial finds faster on this card at this moment
               (run1_registered_auto); run1_registered_oh_only is the default engine with INTERNAL OH as the only output
               (no OH_boost, no NDWET: what the shell asks for when HISTORY wants neither)
from pageable numpy arrays.  The ranks start their ticks together (a start time handed to all of them).  Prints ONE JSON
object: per P the aggregate gridcells/s over the common window, per-tick latency p50 / p95 / max, the first tick, and
the HBM the processes hold together.

The GPU boxes of this pool admit at most 6 processes on a card at once (gpurun's process guard), so P is taken from
{1, 2, 3, 6}; the parent never touches the GPU.  usage (GPU box): python3 tools/ranks_per_gpu.py [--ranks 1,2,3,6]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


RUN1_MODES = ("run1", "run1_registered", "run1_registered_dma", "run1_registered_auto", "run1_registered_oh_only")


def hbm_free_bytes():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    if hip.hipMemGetInfo(C.byref(free), C.byref(total)) != 0:
        return None, None
    return free.value, total.value


def child(args):
    from quickchem_amd import capi, synth
    grid = synth.GRIDS[args.grid]
    im, jm_all, km = grid
    parts = 8 * args.nranks
    jm = jm_all // parts
    if args.block:                                        # a GEOS-sized rank block instead: im x jm columns, all levels
        im, jm = args.block
    sub = (im, jm, km)
    n = im * jm * km
    # the rank's block, pageable host memory: the reference's xx_carr rows, and the same block as 27 SoA fields
    # (a (360, jm, 72) sub-domain generated as a grid of its own, seeded by the rank)
    seed = synth.FEATURE_SEED + args.rank
    fields = [np.ascontiguousarray(synth.field_cpu(sub, f, seed=seed).T) for f in range(synth.NFEAT)]
    rows = np.empty((n, synth.NFEAT), dtype=np.float32)            # the gather of :308-345, PL / 100 in float32
    for f in range(synth.NFEAT):
        col = np.tile(fields[f].reshape(-1), km) if synth.IS2D[f] else fields[f].reshape(-1)
        rows[:, f] = col / np.float32(100.0) if f == synth.PL_FEATURE else col
    oh = np.zeros(n, dtype=np.float32)
    free0, total = None, None
    out = {"rank": args.rank, "rows": n, "pid": os.getpid()}
    while time.time() < args.start_at:
        time.sleep(0.001)
    t_begin = time.time()
    free_before, total = hbm_free_bytes()                  # first HIP call of the process: context creation
    out["hip_init_s"] = time.time() - t_begin
    ticks = {"reference": [], "fused": []}
    ends = {"reference": [], "fused": []}
    b = None

    def meet(name):
        """Every mode starts on all ranks together (files beside the model file): ranks that ran their modes back to back
        drifted apart, and by the fourth mode the first rank had finished before the last had begun - a common window of
        negative length."""
        # (the file names carry the rank count: the runs of a --ranks list share the directory, and the files of the run
        # before let the ranks of this one through before their slowest peer had arrived - P = 3 windows of negative length)
        d = os.path.dirname(args.model)
        open(os.path.join(d, f"at_{args.nranks}_{name}_{args.rank}"), "w").close()
        while sum(os.path.exists(os.path.join(d, f"at_{args.nranks}_{name}_{r}")) for r in range(args.nranks)) < args.nranks:
            time.sleep(0.0005)

    t_ref0 = time.time()
    for tick in range(args.ticks):
        t0 = time.perf_counter()
        if b is None:                                      # first_time (:242-271): create + load, from the file
            b = capi.Booster(args.model)
            for kv in args.param:
                b.set_param(*kv.split("=", 1))
        d = capi.DMatrix(rows, missing=synth.XX_MISS)
        p = b.predict(d, copy=False)                       # the caller reads the booster's buffer in place (:362-374)
        d.free()
        ticks["reference"].append(time.perf_counter() - t0)
        ends["reference"].append(time.time())
        if tick == 0:
            meet("reference")                              # the first tick loads the model file; the others run together
    t_ref1 = time.time()
    p = p.copy()
    ref_sum = float(np.float64(p).sum())
    meet("fused")
    t_ref1 = time.time()
    for tick in range(args.ticks):
        t0 = time.perf_counter()
        b.predict_fields(fields, synth.IS2D, synth.PL_FEATURE, im, jm, km, 1, km, synth.XX_MISS, oh, ohscale=1.0,
                         apply_pow10=False)
        ticks["fused"].append(time.perf_counter() - t0)
        ends["fused"].append(time.time())
    t_fused1 = time.time()
    # OH Run1 through its host form, on the rank's block: a synthetic import state (quickchem_amd.synth.run1_state: plausible
    # magnitudes), the arrays at fixed addresses from tick to tick as MAPL's state pointers are
    st = synth.run1_state(sub, seed=17 + args.rank)
    call = b.run1_prepare(st, dynamic_k_range=True, want_boost=True, want_ndwet=True)
    windows = {"reference": [t_ref0, t_ref1], "fused": [t_ref1, t_fused1]}
    first_oh = None
    # (r6) registered arrays three ways: ohx_copy_engine = kernel (the default), dma, and auto, whose first eighteen ticks
    # are its trial of the other two and are among the timed ones, as a rank sees them; then the default once more with
    # only INTERNAL OH coming back
    for mode in RUN1_MODES:
        ticks[mode], ends[mode] = [], []
        b.set_param("ohx_register_host", "0" if mode == "run1" else "1")
        if mode != "run1":
            b.set_param("ohx_copy_engine", {"run1_registered_dma": "dma", "run1_registered_auto": "auto"}.get(mode, "kernel"))
            if mode == "run1_registered":
                b.run1_call(call)              # the tick that registers the arrays is not one of the timed ones
            if mode == "run1_registered_oh_only":
                # what the shell asks for when HISTORY wants neither OH_boost nor DIAG_NDWET and Boost runs at every
                # alarm (oh_gridcomp.F90: want_boost / want_ndwet): INTERNAL OH alone comes back
                call = b.run1_prepare(st, dynamic_k_range=True, want_boost=False, want_ndwet=False)
                b.run1_call(call)
        meet(mode)
        w0 = time.time()
        for tick in range(args.ticks):
            t0 = time.perf_counter()
            r1 = b.run1_call(call)
            ticks[mode].append(time.perf_counter() - t0)
            ends[mode].append(time.time())
        windows[mode] = [w0, time.time()]
        if mode == "run1_registered_auto":
            out["copy_engine_choice"] = dict(zip(("choice", "trials", "picked_dma"), b.copy_engine_choice()))
        if first_oh is None:
            first_oh = r1["oh"].copy()
        else:
            assert np.array_equal(first_oh.view(np.uint32), r1["oh"].view(np.uint32)), "registered run1 differs"
    out["run1_rows_predicted"] = int(im * jm * (r1["k2"] - r1["k1"] + 1))
    out["run1_bytes_in"] = int(sum(a.nbytes for k, a in call["keep"].items() if k != "sca") + sum(a.nbytes for a in call["keep"]["sca"]))
    b.set_param("ohx_register_host", "0")
    free_after, _ = hbm_free_bytes()
    same = bool(np.array_equal(oh.view(np.uint32), p.view(np.uint32)))      # both paths, same margins
    out.update({"ticks_s": ticks, "tick_ends": ends, "window": windows,
                "paths_agree_bit_for_bit": same, "margin_sum": ref_sum,
                "hbm_free_before": free_before, "hbm_free_after": free_after, "hbm_total": total})
    print("RANK_JSON " + json.dumps(out), flush=True)
    time.sleep(max(0.0, args.hold_until - time.time()))    # keep the memory until every rank has reported its own
    b.free()


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(round(q * (len(xs) - 1))))]


def parent(args):
    from quickchem_amd import synth
    tmp = tempfile.mkdtemp(prefix="ohx_ranks_")
    model = synth.make_model()
    path = os.path.join(tmp, "oh.model")
    open(path, "wb").write(bytes(model.image))
    result = {"grid": args.grid, "block": args.block, "params": args.param, "model_file_bytes": os.path.getsize(path), "ticks": args.ticks,
              "note": "P processes on one GPU, each a rank owning C360/(8P) gridcells in pageable host arrays; "
                      "aggregate = the gridcells of the ticks that end inside the interval in which every rank is "
                      "past its first tick and none has finished, over that interval",
              "by_ranks": {}}
    for P in args.ranks:
        start_at = time.time() + args.prep_s
        hold_until = start_at + 600
        procs = []
        for r in range(P):
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--rank", str(r), "--nranks", str(P), "--model", path,
                   "--grid", args.grid, "--ticks", str(args.ticks), "--start-at", repr(start_at), "--hold-until", repr(hold_until)]
            if args.block:
                cmd += ["--block", "%d,%d" % tuple(args.block)]
            for kv in args.param:
                cmd += ["--param", kv]
            procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        ranks = []
        for p in procs:                                     # children hold their memory until killed below
            line = ""
            for line in p.stdout:
                if line.startswith("RANK_JSON "):
                    ranks.append(json.loads(line[len("RANK_JSON "):]))
                    break
            else:
                raise SystemExit(f"a rank died: {line[-500:]}")
        for p in procs:
            p.kill()
            p.wait()
        entry = {"rows_per_rank": ranks[0]["rows"], "paths_agree_bit_for_bit": all(r["paths_agree_bit_for_bit"] for r in ranks),
                 "hip_init_s_max": max(r["hip_init_s"] for r in ranks)}
        held = [r["hbm_free_before"] - r["hbm_free_after"] for r in ranks if r["hbm_free_before"] is not None]
        if held:
            entry["hbm_held_all_ranks_bytes"] = max(r["hbm_free_before"] for r in ranks) - min(r["hbm_free_after"] for r in ranks)
        entry["run1_bytes_in_per_rank"] = ranks[0]["run1_bytes_in"]
        entry["run1_rows_predicted_per_rank"] = ranks[0]["run1_rows_predicted"]
        entry["copy_engine_auto_chose"] = [{-1: "undecided", 0: "kernel", 1: "dma"}[r["copy_engine_choice"]["choice"]] for r in ranks]
        for mode in ("reference", "fused") + RUN1_MODES:
            first = [r["ticks_s"][mode][0] for r in ranks]
            later = [t for r in ranks for t in r["ticks_s"][mode][1:]]
            # the interval in which EVERY rank is ticking steadily: from the last rank's first tick's end to the
            # first rank's last tick's end; a tick counts if it ends inside
            c0 = max(r["tick_ends"][mode][0] for r in ranks)
            c1 = min(r["tick_ends"][mode][-1] for r in ranks)
            cells = sum(r["rows"] * sum(1 for e in r["tick_ends"][mode][1:] if c0 < e <= c1) for r in ranks)
            entry[mode] = {"first_tick_s": {"max": max(first), "min": min(first)},
                           "tick_ms": {"p50": pct(later, 0.5) * 1e3, "p95": pct(later, 0.95) * 1e3, "max": max(later) * 1e3,
                                       "mean": sum(later) / len(later) * 1e3,
                                       "deciles": [round(pct(later, q / 10) * 1e3, 3) for q in range(1, 10)],
                                       "p99": pct(later, 0.99) * 1e3},
                           # the shape of the tail: rank 0's ticks in order (the first sixty after the first)
                           "rank0_ticks_ms": [round(t * 1e3, 3) for t in ranks[0]["ticks_s"][mode][1:61]],
                           "common_window_s": c1 - c0,
                           "aggregate_gridcells_per_s": cells / (c1 - c0) if c1 > c0 else None}
        result["by_ranks"][str(P)] = entry
        print(f"# P={P}: " + json.dumps(entry), file=sys.stderr, flush=True)
    print(json.dumps(result))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,2,3,6")
    ap.add_argument("--grid", default="C360")
    ap.add_argument("--ticks", type=int, default=100)
    ap.add_argument("--prep-s", type=float, default=60.0, help="time the ranks get to build their host arrays before the common start")
    ap.add_argument("--block", default="", help="im,jm: every rank owns an (im, jm, 72) block of its own instead of C360/(8P)")
    ap.add_argument("--param", action="append", default=[], help="XGBoosterSetParam name=value for every rank's booster (repeatable)")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--nranks", type=int, default=1)
    ap.add_argument("--model", default="")
    ap.add_argument("--start-at", type=float, default=0.0)
    ap.add_argument("--hold-until", type=float, default=0.0)
    args = ap.parse_args()
    args.block = [int(x) for x in args.block.split(",")] if args.block else None
    if args.child:
        return child(args)
    args.ranks = [int(x) for x in args.ranks.split(",")]
    if max(args.ranks) > 6:
        raise SystemExit("at most 6 processes may use the GPU at once on this pool (gpurun's process guard)")
    parent(args)


if __name__ == "__main__":
    main()
