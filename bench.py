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
the node bytes once per step.  `cpu_baseline` is the reference's CPU path as far as it
can be had here ("port"): the Fortran host (predict_OH_with_XGB: SoA->AoS gather,
XGDMatrixCreateFromMat, XGBoosterPredict, 10**pred; OH_GridCompMod.F90:308-374)
linked against the CPU oracle, timed on this node's host cores at one thread (how a
GEOS rank runs it) and at all of them, on the first levels of the same batch.

After the timed steps every rank predicts its rows once more with a kernel of different
design (`wide` nodes, no LDS tile, no grid hint: 64 consecutive rows per wave) and
compares bit for bit: `config.verified`.
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
    ap.add_argument("--cpu-seconds", type=float, default=15.0,
                    help="cpu_baseline leg: 0 = skip, < 10 = a one-level sample, else the fixed sample (8 levels, 3 ticks)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the pcie_inclusive measurement (host-pointer entry points)")
    ap.add_argument("--no-full-oracle", action="store_true",
                    help="skip the comparison of EVERY timed margin with the CPU oracle (the first 2^20 rows and 2^20 drawn rows stay)")
    ap.add_argument("--no-rank-ticks", action="store_true",
                    help="skip the GEOS-rank tick measurement (the shell's Boost tick and skip tick as child processes)")
    ap.add_argument("--shuffle", action="store_true", help="permute the rows (destroys spatial coherence)")
    ap.add_argument("--infer-grid", action="store_true",
                    help="rows path: no hint either, but let the library look for the level size in the rows "
                         "(OHXDMatrixInferGrid), as XGDMatrixCreateFromMat does unasked for host matrices")
    ap.add_argument("--verify", action="store_true", help="(default; kept for older command lines)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the whole-batch cross-check against the `wide` kernel after the timed steps")
    ap.add_argument("--no-grid", action="store_true",
                    help="rows path: say nothing about the rows (no OHXDMatrixSetGrid, no OHXDMatrixInferGrid): what a "
                         "device-resident caller gets who knows nothing of the extensions; the first predict looks "
                         "for the level size by itself")
    ap.add_argument("--consecutive", action="store_true",
                    help="rows path: tell the library there is no grid (OHXDMatrixSetGrid(0,0)): 64 consecutive rows per wave")
    ap.add_argument("--gather-chunks", type=int, default=6,
                    help="N > 1: cut the shard into this many pieces so that a piece's all-gather overlaps the next "
                         "piece's prediction (1 = predict everything, then one all-gather)")
    ap.add_argument("--gather-gbps", type=float, default=None,
                    help="the planner's price of the all-gather, GB/s taken in per rank (default: OHX_GATHER_GBPS, else "
                         "shard.GATHER_BYTES_PER_S - a guess until an N > 1 run has been measured)")
    ap.add_argument("--piece-rounds", type=float, default=None,
                    help="the planner's price of one more piece, in rounds of the chip (default: OHX_PIECE_ROUNDS, else "
                         "shard.PIECE_ROUNDS)")
    ap.add_argument("--gather", default="both", choices=["both", "torch", "native", "none"],
                    help="N > 1: the all-gather through torch.distributed (RCCL process group), through the C ABI's "
                         "own OHXAllGatherOH (what a Fortran/MPI host would call), or not at all (a control: predict "
                         "only, so that a scaling curve can be split into prediction and exchange; not a result).  "
                         "both (default) = torch, with the same number of predict-only steps timed in front of the "
                         "timed steps in the same launch (phases.predict_only), so that ONE run yields the split")
    ap.add_argument("--rows", type=int, default=0,
                    help="use only the first N rows of the grid's batch (tests of ragged shards: N % gpus != 0)")
    ap.add_argument("--path", default="rows", choices=["rows", "fields", "run1"],
                    help="rows: AoS xx_carr -> margins (the headline); fields: the fused SoA call, 27 MAPL fields -> "
                         "10**pred*OHscale; run1: OHXBoosterRun1Device, imports -> INTERNAL OH (both 1 GPU only)")
    args = ap.parse_args()
    args.verify = not args.no_verify
    if args.path != "rows" and args.gpus != 1:
        ap.error(f"--path {args.path} is a single-GPU measurement")       # every rank, before any collective
    return args


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_topology():
    """-> (hardware threads this process may run on, physical cores among them, sockets), from /proc/cpuinfo and the
    affinity mask: os.cpu_count() counts hardware THREADS, and a line that calls them cores is off by the SMT factor."""
    try:
        allowed = set(os.sched_getaffinity(0))
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores, sockets, cpu, phys, core = set(), set(), None, None, None
    try:
        for line in list(open("/proc/cpuinfo")) + ["\n"]:
            key, _, val = line.partition(":")
            key = key.strip()
            if key == "processor":
                cpu, phys, core = int(val), None, None
            elif key == "physical id":
                phys = int(val)
            elif key == "core id":
                core = int(val)
            elif not line.strip() and cpu is not None:
                if cpu in allowed:
                    cores.add((phys, core if core is not None else cpu))
                    sockets.add(phys)
                cpu = None
    except (OSError, ValueError):
        pass
    threads = len(allowed)
    return threads, (len(cores) or threads), (len(sockets) or 1)


def cpu_quota(root="/"):
    """CPUs' worth of time the container's cgroup grants this process (cgroup v2 cpu.max, v1 cpu.cfs_quota_us), or None
    when there is no limit to be seen.  A GPU box of this pool shows all of its host's 256 hardware threads in the
    affinity mask and grants a share of them: past that many threads a team only queues for the quota.
    `root`: where /proc and /sys hang (the tests hand it a directory of their own)."""
    def read(*parts):
        try:
            return open(os.path.join(root, *parts)).read().split()
        except OSError:
            return None
    rel = ""
    for line in (read("proc", "self", "cgroup") or []):
        if line.startswith("0::"):
            rel = line[3:].strip().lstrip("/")
    for base in (os.path.join("sys", "fs", "cgroup", rel), os.path.join("sys", "fs", "cgroup")):
        w = read(base, "cpu.max")
        if w and len(w) == 2 and w[0] != "max":
            return float(w[0]) / float(w[1])
    q, per = read("sys", "fs", "cgroup", "cpu", "cpu.cfs_quota_us"), read("sys", "fs", "cgroup", "cpu", "cpu.cfs_period_us")
    if q and per and float(q[0]) > 0:
        return float(q[0]) / float(per[0])
    return None


def kernel_source_hash():
    """Identifies the kernels a committed PMC measurement was taken on (profiles/*_traffic.json)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("kernels.hip", "kernels.hpp", "flatten.cpp", "flatten.hpp"):
        h.update(open(os.path.join(ROOT, "quickchem_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def fortran_cpu_leg(model_image, grid, fields, levels, threads, workdir, ncalls):
    """oracle/lib/oh_mock_driver_oracle on the first `levels` levels of the batch's MAPL fields: the Fortran host's
    predict_OH_with_XGB (gather + XGDMatrixCreateFromMat + XGBoosterPredict + 10**) over the oracle library, `ncalls`
    ticks in one process.  The first tick holds what the reference pays once per run (the model file is loaded at
    the first call, OH_GridCompMod.F90:242-271; OpenMP team start-up, first touch of the buffers); the later ones are
    what it pays per OH alarm tick (:308-374).  `fields` = the 27 fields in HBM (torch tensors, Fortran order).
    Returns (OH_ML[levels*plane] in row order, seconds of every call)."""
    import struct
    import subprocess
    from quickchem_amd import synth
    im, jm, km = grid
    plane = im * jm
    state = os.path.join(workdir, f"state_L{levels}.bin")
    if not os.path.exists(state):
        with open(state, "wb") as f:
            f.write(struct.pack("<iiiiff", im, jm, levels, 1, 4000.0, 1.0))
            f.write(fields[synth.PL_FEATURE][:plane * levels].cpu().numpy().tobytes())     # pl (Pa): the slab test
            f.write(np.zeros(plane, dtype=np.float32).tobytes())                            # tropp below every level: all predicted
            for feat in range(synth.NFEAT):
                f.write(fields[feat][:plane if synth.IS2D[feat] else plane * levels].cpu().numpy().tobytes())
    model = os.path.join(workdir, "oh.model")
    if not os.path.exists(model):
        open(model, "wb").write(bytes(model_image))
    out = os.path.join(workdir, f"out_L{levels}_T{threads}.bin")
    exe = os.path.join(ROOT, "oracle", "lib", "oh_mock_driver_oracle")
    # threads pinned and spread over the cores: unpinned teams gave 0.89 / 0.94 / 1.04 M gridcells/s in three runs
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="spread", OMP_PLACES="cores")
    r = subprocess.run([exe, state, model, out, "compat", str(ncalls)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, env=env)
    if r.returncode != 0:
        raise SystemExit(f"bench: the Fortran CPU leg failed: {r.stdout[-500:]}")
    raw = open(out, "rb").read()
    rc, k1, k2 = struct.unpack_from("<iii", raw, 0)
    n = plane * levels
    oh = np.frombuffer(raw, dtype="<f4", count=n, offset=12)
    if rc != 0 or (k1, k2) != (1, levels):
        raise SystemExit(f"bench: the Fortran CPU leg returned rc={rc}, slab {k1}..{k2}")
    times = [float(x) for x in open(out + ".times").read().split()]
    os.remove(out)
    os.remove(out + ".times")
    return oh, times


def cpu_baseline(model_image, grid, booster, out_dev, budget_s):
    """The reference-equivalent CPU path next to the GPU number: Fortran host + oracle, one
    thread and all host cores, on the first levels of the batch's 27 MAPL fields; its OH_ML is compared with the
    GPU's fused call on the same fields.  `value` is a STEADY-STATE tick (the fastest of the ticks after the first,
    all in one process); the first tick and the one-time share of it are reported beside it.  The sample is fixed:
    8 levels at all cores (3 ticks), 1 level at one thread (2 ticks) - a shorter one only for a small budget.
    A real libxgboost, if this machine has one, is timed and compared as well."""
    import shutil
    import tempfile
    from quickchem_amd import capi, synth
    im, jm, km = grid
    plane = im * jm
    cores, phys_cores, sockets = cpu_topology()      # `cores` below = hardware threads used (OMP_NUM_THREADS)
    dev = out_dev.device
    fields = []
    for feat in range(synth.NFEAT):
        t = torch.empty(plane * (1 if synth.IS2D[feat] else km), dtype=torch.float32, device=dev)
        synth.field_device(grid, feat, t)
        fields.append(t)
    full = budget_s >= 10
    # the thread sweep: one thread (how a GEOS rank runs), 16, 64, 128, 256 as far as the host has them,
    # every physical core and every hardware thread; the sample grows with the team so that a tick stays seconds long
    quota = cpu_quota()
    share = max(1, int(round(quota))) if quota else None
    points = sorted({t for t in (1, 16, 64, 128, 256, phys_cores, cores, share or 1) if 1 <= t <= cores}) if full else sorted({1, cores})
    workdir = tempfile.mkdtemp(prefix="ohx_cpu_leg_")
    sweep = []
    try:
        for t in points:
            lv = 1 if (t == 1 or not full) else (2 if t < 64 else min(km, 8))
            oh_t, ticks = fortran_cpu_leg(model_image, grid, fields, lv, t, workdir, 2)
            sweep.append({"threads": t, "levels": lv, "gridcells": plane * lv, "value": plane * lv / min(ticks[1:]),
                          "ticks_s": [round(x, 4) for x in ticks], "oh": oh_t})
    finally:
        shutil.rmtree(workdir, ignore_errors=True)
    best = max(sweep, key=lambda e: e["value"])
    one = sweep[0]
    oh, t_all, levels, t_one = best["oh"], best["ticks_s"], best["levels"], one["ticks_s"]
    n = plane * levels
    steady, steady_one = min(t_all[1:]), min(t_one[1:])
    # the checker's verdict: the same fields through the GPU's fused call (gather, PL/100, walk, 10**) vs the
    # Fortran host + oracle; 10.0**x is libm-specific, hence 2 ulp (the raw margins are compared bit for bit below)
    oh_gpu = torch.zeros(plane * km, dtype=torch.float32, device=dev)
    booster.predict_fields_device([t.data_ptr() for t in fields], synth.IS2D, synth.PL_FEATURE, im, jm, km, 1, levels,
                                  synth.XX_MISS, oh_gpu.data_ptr(), apply_pow10=True, ohscale=1.0,
                                  stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    booster.check()
    got = oh_gpu[:n].cpu().numpy()
    ulp = int(np.abs(got.view(np.int32).astype(np.int64) - oh.view(np.int32).astype(np.int64)).max())
    if ulp > 2:
        raise SystemExit(f"bench: GPU OH differs from the Fortran CPU path by {ulp} ulp on the cpu_baseline sample")
    del fields, oh_gpu
    used = best["threads"]
    # cores = what the team really ran on: its threads, capped by the host's cores and by the CPUs the cgroup grants
    base = {"value": n / steady, "unit": "gridcells/s", "cores": int(min(used, phys_cores, quota or used)), "threads": used,
            "sockets": sockets,
            "host_cores": phys_cores, "host_threads": cores, "cgroup_cpu_quota": quota,
            "kind": "port", "cpu_model": cpu_model_name(),
            "sample": f"the BEST point of the thread sweep in `scaling`: OMP_NUM_THREADS={used} on the first {levels} of {km} "
                      f"levels ({n} gridcells) of the batch: oracle/lib/oh_mock_driver_oracle = the "
                      f"Fortran host's predict_OH_with_XGB (SoA->AoS gather, XGDMatrixCreateFromMat, XGBoosterPredict, "
                      f"10**pred; OH_GridCompMod.F90:308-374) linked against oracle/xgb_oracle.c (host: {phys_cores} cores / "
                      f"{cores} hardware threads on {sockets} socket(s)), threads "
                      f"pinned (OMP_PROC_BIND=spread OMP_PLACES=cores; gather and 10** single-threaded as in the "
                      f"reference); {len(t_all)} ticks in one process, value = the fastest tick after the first "
                      f"({steady:.2f} s); libxgboost 1.6.0 itself is not available here.  The oracle's raw nodes (112 MB, far "
                      f"beyond the caches) live in one huge-page arena replicated per NUMA node, and a block's rows are "
                      f"walked sixteen side by side, so the walk's cache misses overlap (round 6; xgboost 1.6.0 walks them "
                      f"one after the other: the same leaves, the same float32 sums).  What is left of a many-thread tick is "
                      f"largely the reference's own single-threaded gather and 10**; and a team larger than the CPU share "
                      f"the box grants this job (cgroup_cpu_quota: {quota}; 16 on a one-GPU box of this pool, whatever the "
                      f"affinity mask shows) only queues for it, which is why the sweep falls off beyond that many threads "
                      f"- a baseline, not a measure of the GPU kernel (that is roofline.frac)",
            "scaling": [{k: v for k, v in e.items() if k != "oh"} for e in sweep],
            "ticks_s": [round(t, 4) for t in t_all],
            "first_tick_s": round(t_all[0], 4),
            "load_s": round(max(t_all[0] - steady, 0.0), 4),
            "load_s_is": "first tick minus a steady tick: the model file parse of the first call (OH_GridCompMod.F90:"
                         "242-271), OpenMP team start-up, first touch",
            "oh_max_ulp_vs_gpu_fused_call": ulp,
            "one_thread": {"value": plane / steady_one, "unit": "gridcells/s", "cores": 1,
                           "sample": f"first level ({plane} gridcells), same executable, OMP_NUM_THREADS=1, "
                                     f"{len(t_one)} ticks, the fastest after the first ({steady_one:.2f} s)",
                           "ticks_s": [round(t, 4) for t in t_one]},
            "libxgboost": None}
    # opportunistic: a real libxgboost on this machine
    try:
        from oracle import real_xgboost
        real, where = real_xgboost.find_libxgboost()
    except Exception:
        real, where = None, "probe failed"
    if real is not None:
        import tempfile as _tf
        with _tf.NamedTemporaryFile(suffix=".model") as mf:
            mf.write(bytes(model_image))
            mf.flush()
            rb = capi.Booster(mf.name, lib=real)
        nreal = min(n, 4 * plane)
        rows = torch.empty((nreal, synth.NFEAT), dtype=torch.float32, device=out_dev.device)
        synth.rows_device(grid, 0, nreal, rows)
        host = rows.cpu().numpy()
        t0 = time.perf_counter()
        d = capi.DMatrix(host, missing=synth.XX_MISS, lib=real)
        pred = rb.predict(d)
        d.free()
        dt = time.perf_counter() - t0
        same = bool(np.array_equal(pred.view(np.uint32), out_dev[:nreal].cpu().numpy().view(np.uint32)))
        base["libxgboost"] = {"version": real_xgboost.version_of(real), "where": where, "value": nreal / dt,
                              "unit": "gridcells/s", "sample": f"first {nreal} rows, XGDMatrixCreateFromMat + "
                              f"XGBoosterPredict, all threads", "bit_identical_to_gpu": same}
        if not same:
            raise SystemExit("bench: GPU margins differ from the REAL libxgboost found on this machine")
    return base


def pcie_inclusive(grid, booster, dev):
    """What the reference's own call sequence costs when the batch starts in PAGEABLE HOST memory, as a GEOS rank
    hands it over (never `value`): one C360/8-sized row shard through XGDMatrixCreateFromMat -> XGBoosterPredict ->
    XGDMatrixFree (OH_GridCompMod.F90:347,356,377), and the same gridcells as 27 SoA fields through the fused
    OHXBoosterPredictFields.  Ticks after the first (the first pays hipMalloc of the matrix and the staging buffers)."""
    from quickchem_amd import capi, synth
    im, jm_all, km = grid
    jm = max(1, jm_all // 8)
    sub = (im, jm, km)
    n = im * jm * km
    t = torch.empty((n, synth.NFEAT), dtype=torch.float32, device=dev)
    synth.rows_device(sub, 0, n, t)
    rows = t.cpu().numpy()
    del t
    ticks = []
    for _ in range(4):
        t0 = time.perf_counter()
        d = capi.DMatrix(rows, missing=synth.XX_MISS)
        t1 = time.perf_counter()
        booster.predict(d, copy=False)         # the C call alone: the caller reads the booster's buffer in place (:362-374)
        t2 = time.perf_counter()
        d.free()
        ticks.append((time.perf_counter() - t0, t1 - t0, t2 - t1))
    ref = min(ticks[1:])
    fields = []
    for f in range(synth.NFEAT):
        ft = torch.empty(im * jm * (1 if synth.IS2D[f] else km), dtype=torch.float32, device=dev)
        synth.field_device(sub, f, ft)
        fields.append(ft.cpu().numpy())
        del ft
    oh = np.zeros(n, dtype=np.float32)
    fused = []
    for _ in range(4):
        t0 = time.perf_counter()
        booster.predict_fields(fields, synth.IS2D, synth.PL_FEATURE, im, jm, km, 1, km, synth.XX_MISS, oh, ohscale=0.85)
        fused.append(time.perf_counter() - t0)
    booster.lib.OHXReleaseScratch()
    return {"rows": n, "bytes_host_to_device": int(rows.nbytes), "host_memory": "pageable",
            "reference_calls": {"tick_ms": ref[0] * 1e3, "XGDMatrixCreateFromMat_ms": ref[1] * 1e3,
                                "XGBoosterPredict_ms": ref[2] * 1e3, "gridcells_per_s": n / ref[0],
                                "first_tick_ms": ticks[0][0] * 1e3},
            "fused_call": {"tick_ms": min(fused[1:]) * 1e3, "gridcells_per_s": n / min(fused[1:]),
                           "first_tick_ms": fused[0] * 1e3},
            "note": "one process, one C360/8-sized block; several ranks sharing the GPU overlap each other's copies: "
                    "profiles/r04_ranks_per_gpu.json"}


def run1_host(booster):
    """What a GEOS rank pays per OH tick through the GridComp shell (never `value`): OHXBoosterRun1, the HOST form -
    the one call quickchem_amd/fortran/oh_gridcomp.F90 makes on a Boost tick (37 import arrays in; INTERNAL OH, OH_boost
    and NDWET out; OH_GridCompMod.F90:1444-1595) - on two rank-sized blocks whose arrays stay at their addresses from
    tick to tick, pageable and with ohx_register_host (registered once, then moved by one copy launch).  The same for
    P ranks sharing the GPU and for a C360/8 block: tools/ranks_per_gpu.py, profiles/r04_ranks_per_gpu*.json."""
    from quickchem_amd import synth
    out = {"what": "OHXBoosterRun1 host form (and the fused predict-only call, host form), steady-state tick (median of 50 after 13), one process", "blocks": []}
    for block in ((48, 24, 72), (96, 48, 72)):
        st = synth.run1_state(block, seed=5)
        call = booster.run1_prepare(st, dynamic_k_range=True)
        entry = {"block": list(block), "gridcells": block[0] * block[1] * block[2],
                 "bytes_in": int(sum(a.nbytes for k, a in call["keep"].items() if k != "sca") + sum(a.nbytes for a in call["keep"]["sca"]))}
        ref = None
        for mode, key in (("0", "pageable_ms"), ("1", "registered_ms")):
            booster.set_param("ohx_register_host", mode)
            ticks = []
            for _ in range(63):
                t0 = time.perf_counter()
                r = booster.run1_call(call)
                ticks.append(time.perf_counter() - t0)
            entry[key] = float(np.median(ticks[13:])) * 1e3
            if ref is None:
                ref = r["oh"].copy()
                entry["levels_predicted"] = int(r["k2"] - r["k1"] + 1)
            elif not np.array_equal(ref.view(np.uint32), r["oh"].view(np.uint32)):
                raise SystemExit("bench: OHXBoosterRun1 on registered arrays differs from the pageable run")
        # ... and the fused predict-only call (OHXBoosterPredictFields: 27 fields in, OH_ML out, all levels), host form
        im, jm, km = block
        fields = [np.ascontiguousarray(synth.field_cpu(block, f).T) for f in range(synth.NFEAT)]
        oh = np.zeros((km, jm, im), dtype=np.float32)
        for mode, key in (("0", "fused_call_pageable_ms"), ("1", "fused_call_registered_ms")):
            booster.set_param("ohx_register_host", mode)
            ticks = []
            for _ in range(63):
                t0 = time.perf_counter()
                booster.predict_fields(fields, synth.IS2D, synth.PL_FEATURE, im, jm, km, 1, km, synth.XX_MISS, oh, ohscale=0.85)
                ticks.append(time.perf_counter() - t0)
            entry[key] = float(np.median(ticks[13:])) * 1e3
        out["blocks"].append(entry)
    booster.set_param("ohx_register_host", "0")
    booster.lib.OHXReleaseScratch()
    return out


def rank_ticks():
    """What a GEOS rank pays per OH tick under the reference's shipped compute_once_per_day: T (OH_instance_OH.rc:38):
    one Boost tick a model day and 23 ticks that skip it (OH_GridCompMod.F90:1189-1193, 1247-1257, 1579-1595), through
    the product's shell (quickchem_amd/fortran/oh_gridcomp.F90 under the mock GEOS cap) and through the reference's own
    child compiled in place (its xgboost calls served by libohxgb.so; present only where oracle/_ref was built), one
    rank alone and six sharing the GPU, a 48 x 24 x 72 block each.  Child processes only, started BEFORE this process
    opens the GPU (the box admits six processes on a card).  tools/rank_tick_end_to_end.py; never `value`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import rank_tick_end_to_end as rt
        rec = rt.measure(ranks=(1, 6), days=1, once=True, arms=("reference_child", "product_shell"))
    except Exception as exc:                      # an extra, not the headline: say so and go on
        return {"error": f"{type(exc).__name__}: {exc}"[:400]}
    brief = lambda q: None if q is None else {k: round(q[k], 1) for k in ("median", "p90", "p99", "max")} | {"n": q["n"]}
    out = {"block": rec["block"], "what": rec["what"], "booster": rec["booster"], "model_days": rec["model_days"], "ranks": {}}
    for P, arms in rec["ranks"].items():
        out["ranks"][P] = {tag: {"boost_tick_us": brief(a["boost_tick_us"]), "skip_tick_us": brief(a["skip_tick_us"]),
                                 "model_day_us": a["model_day_us_median_ticks"]} for tag, a in arms.items() if tag != "check"}
    if "reference_child" in rec:
        out["reference_child"] = rec["reference_child"]
    return out


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
    # ALGORITHMIC bytes of a Run1 tick: every import read once (3 edge fields, 28 layer fields - T, Q, TAUCLW, TAUCLI, seven
    # scattering coefficients, sixteen features used as they are, the default OH - and six 2-D fields), INTERNAL OH
    # and OH_boost written once, the booster's nodes once
    info = booster.info()
    algo = 4 * (3 * edge + 28 * vol + 6 * plane + 2 * vol) + info["node_bytes"]
    achieved = algo / per_step / 1e9
    print(json.dumps({
        "metric": "OH gridcells/sec (XGBoost predict), C360 L72 batch", "value": nslab / per_step,
        "unit": "gridcells/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": per_step * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.grid} L{km}: OH Run1 imports -> INTERNAL OH in HBM (feature engineering, "
                               f"k-slab {k1.value}..{k2.value}, predict, tropopause mask, unit conversion)",
                   "grid": list(grid), "rows_predicted": nslab, "kernel": args.kernel, "params": args.param},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "algorithmic_bytes": algo, "kernel_ms": per_step * 1e3,
                     "kernel": "OHXBoosterRun1Device: feature_pointwise_kernel, feature_column_sums_reg_kernel<72>, k_slab_kernel, "
                               + booster.fields_kernel_symbol(nslab) + ", post_process_kernel",
                     "per": "tick = all kernels of one OH Run1 from the imports to INTERNAL OH (wall clock over the timed ticks)"},
        "cpu_baseline": None}), flush=True)


def measured_traffic(grid_name, kernel, model_nodes):
    """HBM-side bytes per step from committed rocprofv3 --pmc passes (profiles/*_traffic.json) - PMC cannot be
    collected from in here.  Only a measurement taken on this workload AND on these very kernel sources (hash of
    csrc/kernels.hip, kernels.hpp, flatten.*) counts; returns (bytes, source file) or (None, why)."""
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None, "no profiles/"
    have = kernel_source_hash()
    why = "no profiles/*_traffic.json for this workload"
    for name in sorted(os.listdir(pdir), reverse=True):
        if not name.endswith("_traffic.json"):
            continue
        try:
            t = json.load(open(os.path.join(pdir, name)))
        except Exception:
            continue
        if t.get("workload") != grid_name or t.get("model_nodes") != model_nodes or kernel not in ("auto", t.get("kernel")):
            continue
        if t.get("kernel_source_hash") != have:
            if not why.startswith("profiles/"):       # name the newest such file, not the oldest
                why = f"profiles/{name} was measured on other kernel sources ({t.get('kernel_source_hash')} != {have})"
            continue
        measured_traffic.counters = t.get("counters_per_step")
        return t.get("traffic_bytes_per_step"), f"profiles/{name}"
    return None, why


measured_traffic.counters = None       # of the file measured_traffic() accepted: what unit_busy() prices

# what a wave64 vector instruction of the walk's step costs a SIMD to issue with four waves on it, in cycles of the
# nominal 2.4 GHz: measured, tools/valu_issue_microbench.hip (profiles/r05_valu_issue_microbench.txt)
VALU_CYCLES_PER_INSTRUCTION = 3.90
VALU_CYCLES_CLOCK_HZ = 2.4e9


def unit_busy(kernel_s, dev):
    """roofline.valu_issue and roofline.ta_busy: how busy the two units are that the walk is bound by, from the counters
    of the committed --pmc passes (same file and same source-hash rule as roofline.traffic; per step) and this run's
    kernel time.  valu_issue = vector instructions x the MEASURED issue cost of the step's instruction mix / (SIMDs x
    time); ta_busy = the texture addressers' busy cycles / the CUs' busy cycles.  (None, None) without a matching file."""
    import torch
    c = measured_traffic.counters
    if not c or not c.get("SQ_INSTS_VALU") or not c.get("TA_TA_BUSY_sum") or not c.get("SQ_BUSY_CU_CYCLES"):
        return None, None
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    valu_s = c["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INSTRUCTION / VALU_CYCLES_CLOCK_HZ / (4 * cus)
    valu = {"instructions_per_step": c["SQ_INSTS_VALU"], "cycles_per_instruction": VALU_CYCLES_PER_INSTRUCTION,
            "cycles_per_instruction_source": "measured: the walk's 14-instruction step, four waves per SIMD "
                                             "(tools/valu_issue_microbench.hip, profiles/r05_valu_issue_microbench.txt)",
            "simds": 4 * cus, "busy_frac": valu_s / kernel_s}
    ta = {"busy_cycles_per_step": c["TA_TA_BUSY_sum"], "cu_busy_cycles_per_step": c["SQ_BUSY_CU_CYCLES"],
          "busy_frac": c["TA_TA_BUSY_sum"] / c["SQ_BUSY_CU_CYCLES"],
          "vector_memory_instructions_per_step": c.get("SQ_INSTS_VMEM_RD"),
          "busy_cycles_per_instruction": (c["TA_TA_BUSY_sum"] / c["SQ_INSTS_VMEM_RD"]) if c.get("SQ_INSTS_VMEM_RD") else None}
    return valu, ta


def gather_issue(info, nrows, kernel_s, dev, row_loads_per_tile):
    """The roofline that really bounds the walk: wave64 vector-memory instructions through the texture
    addresser, priced at the 14 cycles one costs at the very least.  Instructions per tile = what the library says a
    wave issues to walk the forest once (OHXBoosterGetInfo[7]: per tree one coalesced tree-top load and one gather
    per step after the third) + the loads of the tile's rows (7 sixteen-byte pieces per lane when the wave fetches
    them together, 9 when every lane loads its own 27 floats).  The clock is the device's own maximum."""
    import torch
    props = torch.cuda.get_device_properties(dev)
    tiles = -(-nrows // 64)
    gathers = tiles * (info.get("gathers_per_wave", 0) + row_loads_per_tile)
    clock_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3      # kHz -> Hz; MI355X: 2.4 GHz
    peak = props.multi_processor_count * clock_hz / 14.0
    return {"unit": "wave64 gathers/s", "achieved": gathers / kernel_s, "peak": peak,
            "frac": gathers / kernel_s / peak,
            "gathers_per_wave_and_tree": info.get("gathers_per_wave", 0) / max(info["num_trees"], 1),
            "row_loads_per_tile": row_loads_per_tile,
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
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": booster.fields_kernel_symbol(n_total),
                     "kernel_ms": per_step * 1e3, "algorithmic_bytes": algo},
        "cpu_baseline": None}), flush=True)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_argv(ngpus, argv, port):
    """The command `python bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) turns itself into: N fresh
    rank processes under torch.distributed.run on this node, rendezvous on 127.0.0.1, every rank running this very
    file with the caller's own arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def launch_ranks(args):
    """`python bench.py --gpus N` started plainly (the way --gpus 1 is): start the N ranks as CHILD processes and pass
    on rank 0's JSON line and their return code.  Nothing in this process has touched the GPU (importing torch does
    not; torch.cuda.device_count() does not on this image), and nothing here replaces it (no exec): the children are
    fresh interpreters, one per GPU."""
    import subprocess
    share = os.environ.get("OHX_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        raise SystemExit(f"bench: --gpus {args.gpus} but this node shows {have} GPU(s)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    for k in ("RANK", "LOCAL_RANK", "MASTER_PORT"):          # a stale one would send the ranks elsewhere
        env.pop(k, None)
    cmd = launcher_argv(args.gpus, sys.argv[1:], free_port())
    print("bench: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    lines = 0
    for ln in proc.stdout:                                   # the ranks' stderr goes straight through
        sys.stdout.write(ln)
        sys.stdout.flush()
        lines += ln.startswith("{")
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench: the ranks ended with rc 0 but printed {lines} JSON lines", file=sys.stderr)
        rc = 1
    raise SystemExit(rc)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world} (start `python bench.py --gpus N` plainly, or under "
                         f"python -m torch.distributed.run --nproc-per-node N)")
    # the GEOS-rank ticks run as child processes and must be over before this process opens the GPU
    shell_ticks = None
    if (world == 1 and args.path == "rows" and not args.no_rank_ticks and not args.no_pcie and args.cpu_seconds > 0
            and not args.shuffle and not args.missing_ppm and not args.rows and not args.param
            and (args.trees, args.depth, args.grid) == (100, 18, "C360")):
        shell_ticks = rank_ticks()
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
    if args.rows:
        n_total = min(n_total, args.rows)
    row0, n_local = shard.row_shard(n_total, world, rank)

    # ---- what an 8-GPU run is diagnosed from afterwards: who ran where, over which RCCL ----
    dist_info = None
    if world > 1 or force_dist:
        assert dist.get_world_size() == args.gpus or force_dist, (dist.get_world_size(), args.gpus)
        ident = [None] * dist.get_world_size()
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "device": dev_index, "host": os.uname().nodename,
                "uuid": str(getattr(props, "uuid", "")), "name": props.name}
        dist.all_gather_object(ident, mine)
        places = {(i["host"], i["device"]) for i in ident}
        if len(places) != len(ident) and os.environ.get("OHX_BENCH_SHARE_GPU") != "1":
            raise SystemExit(f"bench: two ranks share a GPU: {ident}")
        try:
            rccl = capi.Communicator.rccl_version() if backend == "nccl" else None
        except capi.OhxError:
            rccl = None
        dist_info = {"backend": backend, "world_size": dist.get_world_size(), "ranks": ident, "rccl_version": rccl,
                     "torch_nccl_version": ".".join(map(str, torch.cuda.nccl.version())) if backend == "nccl" else None,
                     "env": {k: os.environ[k] for k in sorted(os.environ)
                             if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC", "OHX_ALLGATHER"))}}

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
    plane = grid[0] * grid[1]
    # pieces of the shard: one DMatrix view each (device pointers into `rows`)
    gather = (world > 1 or force_dist) and args.gather != "none"
    use_grid = not (args.no_grid or args.shuffle or args.infer_grid or args.consecutive)
    # pieces are whole levels when the shard is (bricks then have no idle lanes), else whole launches
    granule = plane if (use_grid and row0 % plane == 0 and n_local % plane == 0) else 64 * 256 * 20 * 2
    # one residency of the chip: 20 waves per CU for the tile kernels, one 16-wave block per CU for the ring kernels
    waves_per_cu = 16 if "ring" in booster.kernel_symbol(27) else 20
    reserve = int({kv.partition("=")[0]: kv.partition("=")[2] for kv in args.param}.get("ohx_reserve_cus", 0)) if waves_per_cu == 16 else 0
    round_rows = (torch.cuda.get_device_properties(dev).multi_processor_count - reserve) * waves_per_cu * 64
    prices = shard.price_list(args.gather_gbps, args.piece_rounds)
    pieces, plan_table = shard.plan_pieces_priced(n_local, args.gather_chunks if (gather and even) else 1, granule,
                                                  round_rows, world, prices)
    dmats = [capi.DMatrix(device_ptr=rows.data_ptr() + lo * synth.NFEAT * 4, nrow=hi - lo, ncol=synth.NFEAT,
                          missing=synth.XX_MISS) for lo, hi in pieces]
    if use_grid:
        for (lo, hi), dm in zip(pieces, dmats):
            dm.set_grid(grid[0], grid[1], row0 + lo)
    elif args.infer_grid:
        for dm in dmats:
            dm.infer_grid()
    elif args.consecutive:
        for dm in dmats:
            dm.set_grid(0, 0, 0)
    out_full = torch.empty(n_total, dtype=torch.float32, device=dev) if gather else out_local
    stream = torch.cuda.current_stream()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    chunks = shard.ChunkGather(out_full, n_local, world, pieces) if (gather and len(pieces) > 1) else None
    native = None
    control_loop = gather and args.gather == "both"
    if gather and args.gather == "native":
        # the C-ABI route a Fortran/MPI host would take (include/ohxgb.h part 4): RCCL through libohxgb.so itself;
        # the 128-byte id travels over torch.distributed here, over MPI_Bcast there
        ids = [capi.Communicator.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        native = capi.Communicator(ids[0], world, rank)

    def step(i=None, with_gather=True, e0=None, e1=None):
        e0 = ev0 if e0 is None else e0
        e1 = ev1 if e1 is None else e1
        if i is not None:
            e0[i].record(stream)
        for q, ((lo, hi), dm) in enumerate(zip(pieces, dmats)):
            booster.predict_device(dm, out_local.data_ptr() + lo * 4, stream=stream.cuda_stream)
            if with_gather and chunks is not None and native is None:
                chunks.start(q, out_local)
        if i is not None:
            e1[i].record(stream)      # with N > 1 this spans the predict launches of all pieces
        if not with_gather:
            return
        if native is not None:
            native.all_gather_oh(out_local.data_ptr(), n_local, n_total, out_full.data_ptr(), stream=stream.cuda_stream)
        elif chunks is not None:
            chunks.finish()
        elif gather:
            shard.all_gather_rows(out_full, out_local, n_total, world, even)

    def fence():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # --gather both: the same K steps without the all-gather, in front of the timed ones (in effect K more warm-up
    # steps): what the gather costs is then the difference of two loops of ONE launch, on the same ranks and clocks
    control = None
    if control_loop:
        c0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        c1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, with_gather=False, e0=c0, e1=c1)
        fence()
        control = [(time.perf_counter() - t0) / args.steps * 1e3,
                   float(np.mean([a.elapsed_time(b) for a, b in zip(c0, c1)]))]
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    booster.check()
    my_step_ms = elapsed / args.steps * 1e3
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # ---- parity of the timed output, on every rank: the same rows through a kernel of different design ----
    verified = None
    if args.verify:
        other = capi.Booster(model_buffer=model.image)
        other.set_param("ohx_kernel", "wide" if args.kernel != "wide" else "packed2")
        ref = torch.empty(n_local, dtype=torch.float32, device=dev)
        dm_all = capi.DMatrix(device_ptr=rows.data_ptr(), nrow=n_local, ncol=synth.NFEAT, missing=synth.XX_MISS)
        dm_all.set_grid(0, 0, 0)                        # one piece, no hint: 64 consecutive rows per wave
        other.predict_device(dm_all, ref.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize()
        other.check()
        ok = int(torch.equal(ref.view(torch.int32), out_local.view(torch.int32)))
        if gather:                                      # and the gathered field holds this rank's shard at its rows
            ok &= int(torch.equal(out_full[row0:row0 + n_local].view(torch.int32), out_local.view(torch.int32)))
        del ref, dm_all, other
        if world > 1:
            t = torch.tensor([ok], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        verified = bool(ok)
        if not verified:
            raise SystemExit("bench: the timed output differs from the cross-check kernel's (or the gathered field is misplaced)")
    kernel_ms = [a.elapsed_time(b) for a, b in zip(ev0, ev1)]
    kernel_s = float(np.mean(kernel_ms)) * 1e-3
    # every rank's two numbers on every rank (and the control loop's): `phases` prints them per rank
    per_rank = control_by_rank = rows_by_rank = None
    if world > 1 or force_dist:
        per_rank = shard.rank_times([my_step_ms, kernel_s * 1e3])
        control_by_rank = shard.rank_times(control) if control is not None else None
        rows_by_rank = [shard.row_shard(n_total, world, r)[1] for r in range(world)]
    if world > 1:
        t = torch.tensor([kernel_s], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kernel_s = float(t.item())

    info = booster.info()
    symbol = booster.kernel_symbol(27)
    # tiles per launch: the ring kernel takes capi.RING_ROUNDS_DEFAULT rounds (ohx_ring_rounds) of one 16-wave block per CU, the tile kernel
    # 2 (ohx_launches_per_residency) of 20 waves per CU
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    knob = {kv.partition("=")[0]: kv.partition("=")[2] for kv in args.param}
    rounds = int(knob.get("ohx_ring_rounds", capi.RING_ROUNDS_DEFAULT)) if "ring" in symbol else int(knob.get("ohx_launches_per_residency", 2))
    if "ring" in symbol and (args.shuffle or args.consecutive):      # rows not known to lie on a grid keep short launches
        clustered = args.shuffle and "ohx_cluster=off" not in args.param
        cap = capi.RING_ROUNDS_PERMUTED if clustered else capi.RING_ROUNDS_NO_GRID
        rounds = min(rounds, cap) if rounds > 0 else cap
    ring_cus = cus - int(knob.get("ohx_reserve_cus", 0))
    per_launch = (ring_cus * 16 if "ring" in symbol else cus * 20) * rounds if rounds > 0 else 1 << 62
    launches_per_step = sum(-(-((hi - lo + 63) // 64) // per_launch) for lo, hi in pieces)
    ms_per_step = elapsed / args.steps * 1e3
    value = n_total / (elapsed / args.steps)
    algo_bytes = BYTES_PER_CELL * n_local + info["node_bytes"]
    achieved = algo_bytes / kernel_s / 1e9

    # ---- CPU baseline (rank 0, N = 1): Fortran host + oracle on the first levels of the same batch; it also
    #      checks the timed GPU output against the checker (10**margin within 2 ulp) ----
    cpu = None
    plain = (not args.shuffle and not args.missing_ppm and args.trees == 100 and args.depth == 18)
    if rank == 0 and world == 1 and args.cpu_seconds > 0 and plain:
        cpu = cpu_baseline(model.image, grid, booster, out_local, args.cpu_seconds)
        # and bit for bit on the raw margins, oracle through ctypes, first 2**20 rows
        lib = capi.declare_xgb_api(C.CDLL(os.path.join(ROOT, "oracle", "lib", "liboracle_xgb.so")))
        n_chk = min(n_local, 1 << 20)
        ob = capi.Booster(model_buffer=model.image, lib=lib)
        od = capi.DMatrix(rows[:n_chk].cpu().numpy(), missing=synth.XX_MISS, lib=lib)
        want = ob.predict(od)
        od.free()
        if not np.array_equal(out_local[:n_chk].cpu().numpy().view(np.uint32), want.view(np.uint32)):
            raise SystemExit("bench: GPU margins differ from the oracle on the first rows of the batch")
        cpu["margins_bit_identical_on_first_rows"] = n_chk
        # ... and on as many rows drawn from the whole batch (every level, every launch of the train)
        pick = torch.randint(n_local, (n_chk,), device=dev, generator=torch.Generator(device=dev).manual_seed(2024))
        od = capi.DMatrix(rows[pick].cpu().numpy(), missing=synth.XX_MISS, lib=lib)
        want = ob.predict(od)
        od.free()
        if not np.array_equal(out_local[pick].cpu().numpy().view(np.uint32), want.view(np.uint32)):
            raise SystemExit("bench: GPU margins differ from the oracle on rows drawn from the whole batch")
        cpu["margins_bit_identical_on_random_rows"] = n_chk
        # (r6) ... and, now that the oracle walks sixteen rows side by side, EVERY row of the timed output: the whole
        # batch in pieces of 2^22 rows through the oracle's XGBoosterPredict, on as many threads as the box grants
        if not args.no_full_oracle and args.cpu_seconds >= 10:
            quota = cpu_quota()
            lib.oracle_set_num_threads.argtypes = [C.c_int]
            lib.oracle_set_num_threads(max(1, int(round(quota))) if quota else min(32, os.cpu_count() or 8))
            t_full = time.perf_counter()
            piece = 1 << 22
            for lo in range(0, n_local, piece):
                hi = min(n_local, lo + piece)
                od = capi.DMatrix(rows[lo:hi].cpu().numpy(), missing=synth.XX_MISS, lib=lib)
                want = ob.predict(od, copy=False)
                same = np.array_equal(out_local[lo:hi].cpu().numpy().view(np.uint32), want.view(np.uint32))
                od.free()
                if not same:
                    raise SystemExit(f"bench: GPU margins differ from the oracle in rows {lo} .. {hi - 1} of the batch")
            cpu["margins_bit_identical_on_the_whole_batch"] = n_local
            cpu["whole_batch_oracle_s"] = round(time.perf_counter() - t_full, 2)

    pcie = host_tick = None
    if rank == 0 and world == 1 and plain and not args.no_pcie and args.cpu_seconds > 0 and use_grid:
        pcie = pcie_inclusive(grid, booster, dev)
        host_tick = run1_host(booster)
    if rank == 0:
        traffic, traffic_src = (None, "not the default single-GPU workload")
        valu_issue = ta_busy = None
        if world == 1 and use_grid and plain and not args.param:
            traffic, traffic_src = measured_traffic(args.grid, args.kernel, info["num_nodes"])
            if traffic:
                valu_issue, ta_busy = unit_busy(kernel_s, dev)
        line = {
            "metric": f"OH gridcells/sec (XGBoost predict), {args.grid.split('L')[0]} L{grid[2]} batch",
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
                "kernel": args.kernel, "params": args.param, "missing_ppm": args.missing_ppm, "shuffled": bool(args.shuffle), "grid_hint": bool(use_grid), "grid_known_to_library": list(dmats[0].grid()), "verified": verified,
                "verified_against": ("the whole timed output, bit for bit, against the `wide` kernel (other node format, no LDS tile, no grid hint)"
                                     + ("; and against the CPU oracle, every row (cpu_baseline.margins_bit_identical_on_the_whole_batch)"
                                        if cpu and cpu.get("margins_bit_identical_on_the_whole_batch") == n_local else
                                        "; the CPU oracle sees the first 2^20 rows and 2^20 rows drawn from the whole batch "
                                        "(cpu_baseline.margins_bit_identical_on_first_rows / _on_random_rows)")) if verified else None,
                "parallelism": f"rows{world}", "gather_pieces": len(pieces) if gather else 0,
                "gather_via": (args.gather if (world > 1 or force_dist) else None),
                "planner_prices": prices if (world > 1 or force_dist) else None,
            },
            # N > 1: where a step's time goes.  predict_ms = first launch to last launch of the rank's pieces (events
            # on the launching stream, max over ranks); exposed_gather_ms = what the step takes beyond that: the part
            # of the all-gather (and of the copies into place) that the prediction of the next piece does not hide
            "phases": (shard.phases_record(per_rank, rows_by_rank, n_total, pieces, even, plan_table, prices, control_by_rank)
                       if (world > 1 or force_dist) else None),
            "distributed": dist_info,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "measured_frac": (traffic / kernel_s / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "kernel": symbol, "kernel_ms": kernel_s * 1e3,
                         "per": "step = the train of launches of one pass over the batch",
                         "launches_per_step": launches_per_step,
                         "avg_launch_us": kernel_s * 1e6 / launches_per_step,
                         "algorithmic_bytes": algo_bytes,
                         # what actually bounds the walk: gather instructions through the
                         # texture addresser, priced at the 14 cycles a wave64 gather costs at the very least
                         "gather_issue": gather_issue(info, n_local, kernel_s, dev, 7 if (use_grid or args.consecutive or args.infer_grid or args.no_grid) and not args.shuffle else 9),
                         # (r5) the two units the walk is bound by, from the committed counters of these very kernels
                         "valu_issue": valu_issue, "ta_busy": ta_busy},
            "cpu_baseline": cpu,
            "pcie_inclusive": pcie,
            "run1_host": host_tick,
            "rank_ticks": shell_ticks,
        }
        # the shell's own ticks on the rank-sized block, beside the library call's (run1_host.blocks[0] is that block)
        if shell_ticks and "ranks" in shell_ticks and host_tick and host_tick["blocks"][0]["block"] == shell_ticks.get("block"):
            for P, arms in shell_ticks["ranks"].items():
                mine = arms.get("product_shell")
                if mine and mine["skip_tick_us"] and mine["boost_tick_us"]:
                    host_tick["blocks"][0].setdefault("shell_ticks_ms", {})[f"ranks_{P}"] = {
                        "skip_tick_ms": round(mine["skip_tick_us"]["median"] / 1e3, 4),
                        "boost_tick_ms": round(mine["boost_tick_us"]["median"] / 1e3, 4)}
            host_tick["blocks"][0]["shell_ticks_ms_are"] = ("medians of rank_ticks: the GridComp shell's whole tick under compute_once_per_day: T "
                                                          "(skip tick: the fused host pass, no GPU call; Boost tick: OHXBoosterRun1 with OH_boost kept)")
        # the reference's own child is a BASELINE (oracle/_ref, the reference compiled in place): its figures go where
        # the baselines are, the product's stay in rank_ticks
        if shell_ticks and "ranks" in shell_ticks and cpu is not None:
            ref = {P: arms.pop("reference_child") for P, arms in shell_ticks["ranks"].items() if "reference_child" in arms}
            if ref:
                cpu["rank_ticks_reference_child"] = {
                    "what": "the same ticks (rank_ticks.what) through the reference's own OH_GridCompMod.F90, compiled in place "
                            "into oracle/_ref/refchild: its feature engineering, gather, 10**x, mask and conversion on the "
                            "rank's core, its five xgboost calls served by libohxgb.so on the GPU",
                    "ranks": ref}
        print(json.dumps(line), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
