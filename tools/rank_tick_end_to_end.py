#!/usr/bin/env python3
"""A GEOS rank's OH tick end to end, Boost ticks and the ticks that skip Boost timed SEPARATELY, the reference's own
OH child against the product's shell, P ranks (processes) sharing the one GPU.

Under the reference's shipped configuration (OH_instance_OH.rc:38, compute_once_per_day: T) need_to_call_BOOST is true
on the first OH tick of a model day only (OH_GridCompMod.F90:1189-1193); the other 23 ticks of a day with OH_DT one hour
do three pointwise passes - NDWET (:1247-1257), the tropopause mask (:1579-1587), the conversion (:1595).  This tool
runs the mock GEOS cap of the tests (tests/fortran/oh_gridcomp_driver.F90, OHX_DRIVER_TIMING=1: wall time of the parent's
two run phases per tick) over
   reference_child      oracle/_ref/refchild/oh_refchild_driver_hip - the reference's unmodified OH_GridCompMod.F90, its
                        five xgboost calls served by libohxgb.so on the GPU
   product_shell        quickchem_amd/lib/oh_gridcomp_driver_hip with "skip_tick: host" (the default): the skip tick is one
                        fused pass on the rank's core, no GPU call
   product_shell_device the same with "skip_tick: device": the skip tick goes over PCIe to OHXOHPostProcess
on one 48 x 24 x 72 block per rank (NOTES.wiki's rank size), --days model days of 24 hourly ticks, HISTORY asking for no
DIAG export, and prints ONE JSON object: per P and per driver the median / p90 / p99 / max of the Boost ticks and of the
skip ticks, and the day's total.  Tick 0 (model load, first-touch of every buffer) is reported apart.  With
--check the three drivers' INTERNAL OH of every tick are compared as tests/test_reference_child.py compares them.

This is a measurement, not a test: the GPU suite asserts bits, never durations.  The GPU boxes admit 6 processes on a
card; this parent never touches the GPU.  usage (GPU box): python3 tools/rank_tick_end_to_end.py --ranks 1,6"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def quantiles(xs):
    xs = sorted(xs)
    if not xs:
        return None
    at = lambda q: xs[min(len(xs) - 1, int(len(xs) * q))]
    return {"n": len(xs), "median": at(0.5), "p10": at(0.1), "p90": at(0.9), "p99": at(0.99), "max": xs[-1],
            "mean": sum(xs) / len(xs)}


def parse_ticks(stdout):
    """-> [(tick, us, nhms)] from the driver's TICK_US lines."""
    out = []
    for ln in stdout.splitlines():
        if ln.startswith("TICK_US "):
            w = ln.split()
            out.append((int(w[1]), float(w[2]), int(w[3])))
    return out


def measure(ranks=(1, 6), block=(48, 24, 72), days=2, once=True, check=False, trees=100, depth=18,
            arms=("reference_child", "product_shell", "product_shell_device"), meet_after_s=4.0, log=None):
    """-> the record described in the module docstring.  Starts only child processes; touches no GPU itself."""
    from quickchem_amd import synth
    from tests import helpers, test_gridcomp as tg
    refchild = os.path.join(ROOT, "oracle", "_ref", "refchild", "oh_refchild_driver_hip")
    grid = tuple(block)
    nticks = 24 * days + 1
    # the booster of the tests' deep_model fixture (tests/conftest.py), so the figures continue profiles/r05_rank_tick_end_to_end.json
    model = synth.make_model(num_trees=trees, max_depth=depth, sample_log2=16, min_leaf=2, grid=synth.GRIDS["C12"])
    record = {"block": list(grid), "model_days": days, "oh_dt_s": 3600, "run_dt_s": 3600, "compute_once_per_day": once,
              "data_source": "ONLINE_INST", "booster": f"{model.num_trees} trees, depth <= {depth}, {model.num_nodes} nodes (synthetic)",
              "what": "wall time of ESMF_GridCompRun phase 1 + 2 of the parent per tick, us, over all ranks; ticks after tick 0; "
                      "a Boost tick is one with nhms == 0 (OH_GridCompMod.F90:1189-1193), the rest skip Boost "
                      "(NDWET :1247-1257, mask :1579-1587, conversion :1595 only)",
              "ranks": {}}
    drivers = [(t, e, k) for t, e, k in (("reference_child", refchild, None), ("product_shell", tg.DRIVER_HIP, "host"),
                                         ("product_shell_device", tg.DRIVER_HIP, "device")) if t in arms]
    if "reference_child" in arms and not os.path.exists(refchild):
        drivers = [d for d in drivers if d[0] != "reference_child"]
        record["reference_child"] = "oracle/_ref/refchild not built"
    with tempfile.TemporaryDirectory() as tmp:
        imports, lats, lons = tg.mock_imports(grid, "ONLINE_INST", seed=21)
        open(os.path.join(tmp, "oh_M01.model"), "wb").write(model.image.tobytes())
        state = os.path.join(tmp, "state.bin")
        tg.write_state_file(state, grid, imports, lats, lons)
        for P in ranks:
            assert 1 <= P <= 6, "the GPU boxes admit six processes on a card"
            per_p = {}
            kept = {}
            for tag, exe, skip in drivers:
                rundir = os.path.join(tmp, f"run_{tag}")
                tg.write_rundir(rundir, source="ONLINE_INST", model_pattern=os.path.join(tmp, "oh_M01.model"), policy="reference",
                                exports=[], once_per_day=once, spinup=False, run_dt=3600, oh_dt=3600, avg24_tick=-1,
                                ohscale=0.85, ref_time="000000", beg="20240131 000000", skip_tick=skip)
                now = time.gmtime(time.time() + meet_after_s + 0.5 * P)
                # (a profiler wrapped around the caller - rocprofv3 -- python3 bench.py - must not follow the ranks: they are timed)
                env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCPROF", "ROCP_", "ROCTX"))}
                env.update(OHX_DRIVER_TIMING="1", OHX_DRIVER_MEET_AT=str(now.tm_hour * 3600 + now.tm_min * 60 + now.tm_sec))
                outs = [os.path.join(tmp, f"{tag}_{r}.bin") for r in range(P)]
                procs = [subprocess.Popen([exe, rundir, state, outs[r], str(nticks)], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True) for r in range(P)]
                boost, skipt, first = [], [], []
                for p in procs:
                    so, _ = p.communicate(timeout=900)
                    if p.returncode != 0:
                        raise RuntimeError(f"{tag} failed at P={P}:\n{so[-3000:]}")
                    for tick, us, nhms in parse_ticks(so):
                        if tick == 0:
                            first.append(us)
                        elif once and nhms > 0:
                            skipt.append(us)
                        else:
                            boost.append(us)
                if check and P == ranks[0]:
                    kept[tag] = tg.parse_output(outs[0], grid, [("OH", False)], [])
                for o in outs:
                    os.remove(o)
                day = None
                if once and boost and skipt:
                    day = quantiles(boost)["median"] + 23 * quantiles(skipt)["median"]
                per_p[tag] = {"boost_tick_us": quantiles(boost), "skip_tick_us": quantiles(skipt), "first_tick_us": quantiles(first),
                              "model_day_us_median_ticks": day}
                if log:
                    print(f"# P={P} {tag}: boost {per_p[tag]['boost_tick_us']} skip {per_p[tag]['skip_tick_us']}", file=log, flush=True)
            if kept:
                ref = kept.get("reference_child")
                base = "reference_child" if ref else "product_shell"
                chk = {}
                for tag in kept:
                    if tag == base:
                        continue
                    worst_skip, worst_boost = 0, 0
                    for a, b in zip(kept[base], kept[tag]):
                        u = int(helpers.ulp_diff(a["OH"]["OH"], b["OH"]["OH"]).max())
                        if once and a["nhms"] > 0:
                            worst_skip = max(worst_skip, u)
                        else:
                            worst_boost = max(worst_boost, u)
                    chk[tag] = {"against": base, "internal_oh_max_ulp_boost_ticks": worst_boost,
                                "internal_oh_max_ulp_skip_ticks": worst_skip}
                per_p["check"] = chk
            record["ranks"][str(P)] = per_p
    return record


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="1,6")
    ap.add_argument("--block", default="48,24,72")
    ap.add_argument("--days", type=int, default=2)
    ap.add_argument("--every-tick-boosts", action="store_true", help="compute_once_per_day: F (round 5's measurement)")
    ap.add_argument("--check", action="store_true", help="compare INTERNAL OH of the drivers tick by tick (first P)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--trees", type=int, default=100)
    ap.add_argument("--depth", type=int, default=18)
    args = ap.parse_args()
    record = measure(ranks=[int(p) for p in args.ranks.split(",")], block=[int(x) for x in args.block.split(",")],
                     days=args.days, once=not args.every_tick_boosts, check=args.check, trees=args.trees, depth=args.depth,
                     log=sys.stderr)
    print(json.dumps(record))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(record, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
