"""SURVEY.md §8(f)-4 / BASELINE.json config #1 as worded: "a synthetic MAPL state through OH_GridComp Run".

The OH GridComp shell (quickchem_amd/fortran/oh_gridcomp.F90, on the mapl_lite mock) runs as the child of two parents
in turn (fixture `drivers`): the product's own minimal one (quickchem_amd/fortran/oh_standalone_cap.F90; always built,
its absence fails) and the REFERENCE'S OWN parent, QuickChem_GridCompMod.F90 and Shared/QuickChem_Generic.F90, unmodified,
compiled in place from /root/reference into oracle/_ref/ by oracle/Makefile (target `ref`; that half is skipped where
neither the reference nor a prebuilt oracle/_ref is there) - driven over model days by a mock GEOS cap
(tests/fortran/oh_gridcomp_driver.F90).  Linked against the oracle it is the CPU
plumbing case; linked against libohxgb.so (-m gpu) the same shell runs its arithmetic on the MI355X.  What the
shell decides - alarm gate, need_to_call_BOOST, which import feeds which input (OH_data_source, spin-up), the month
in the model file name, what persists between ticks - is restated here tick by tick in Python, and the numbers come
from the oracle library's OHXBoosterRun1 / OHXOHPostProcess on the inputs so chosen.
Reference: OH_GridComp/OH_GridCompMod.F90:475-802 (SetServices), :810-949 (Initialize), :957-1010 (Run),
:1017-1741 (Run1), :1749-1824 (Run2), :1831-1891 (Run_data); QuickChem_GridCompMod.F90:78-196, 292-421, 432-538."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

from quickchem_amd import capi, synth
from tests import helpers

# the mock GEOS cap (tests/fortran/oh_gridcomp_driver.F90) over the OH shell, under two parents:
#   "cap"     the product's own minimal parent, quickchem_amd/fortran/oh_standalone_cap.F90 - built by
#             quickchem_amd/fortran/Makefile and oracle/Makefile wherever the product builds; its absence FAILS the tests
#   "parent"  the reference's own QuickChem_GridCompMod.F90, compiled in place by oracle/Makefile `ref` - an additional
#             cross-check, skipped where neither /root/reference nor a prebuilt oracle/_ref is there
CAP_ORACLE = os.path.join(helpers.ROOT, "oracle", "lib", "oh_gridcomp_driver_oracle")
CAP_HIP = os.path.join(helpers.ROOT, "quickchem_amd", "lib", "oh_gridcomp_driver_hip")
DRIVER_ORACLE = os.path.join(helpers.ROOT, "oracle", "_ref", "oh_gridcomp_driver_oracle")
DRIVER_HIP = os.path.join(helpers.ROOT, "oracle", "_ref", "oh_gridcomp_driver_hip")


@pytest.fixture(params=["cap", "parent"])
def drivers(request):
    if request.param == "cap":
        assert os.path.exists(CAP_ORACLE) and os.path.exists(CAP_HIP), \
            "the product's GridComp drivers are missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
        return {"oracle": CAP_ORACLE, "hip": CAP_HIP, "which": "cap"}
    if not os.path.exists(DRIVER_ORACLE):
        pytest.skip("oracle/_ref not built: the reference's own parent (/root/reference/QuickChem_GridCompMod.F90) is "
                    "compiled in place by oracle/Makefile")
    return {"oracle": DRIVER_ORACLE, "hip": DRIVER_HIP, "which": "parent"}


F32 = np.float32
EPSILON = float(F32(18.015) / F32(28.965))        # MAPL_H2OMW / MAPL_AIRMW in real32, as mapl_lite computes it
AVOGAD, RUNIV = 6.023e26, 8314.47
DEG2RAD = F32(F32(np.pi) / F32(180.0))
RAD2DEG = F32(F32(180.0) / F32(np.pi))
WAVELENGTHS = [470, 550, 870]
AEROSOLS = ["BC", "OC", "BR", "DU", "SU", "SS", "NI"]
# import name of every "3 options" input (OH_GridCompMod.F90:1326-1436) by the key of OHXRun1Args it feeds
ONLINE = {"t_bst": "T", "qv": "Q", "ple_bst": "PLE", "zle_bst": "ZLE", "tauclw": "TAUCLW", "taucli": "TAUCLI",
          "ch4": "CH4", "co": "CO", "cloud": "FCLD"}
CLIMATOLOGY = {"no2": "oh_NO2", "o3": "oh_O3", "isop": "oh_ISOP", "acet": "oh_ACET", "c2h6": "oh_C2H6",
               "c3h8": "oh_C3H8", "prpe": "oh_PRPE", "alk4": "oh_ALK4", "mp": "oh_MP", "h2o2": "oh_H2O2",
               "ch2o": "oh_CH2O", "default_oh": "oh_OH", "gmito3": "oh_GMITO3", "gmitto3": "oh_GMITTO3",
               "albuv": "oh_ALBUV"}


def mock_imports(grid, source, seed=3):
    """Every import the OH instance declares for `source`, [i,j(,k)]-indexed float32, by MAPL name."""
    im, jm, km = grid
    st = helpers.run1_state(grid, seed=seed)
    rng = np.random.default_rng(seed + 100)
    imp = {name: st[key] for key, name in CLIMATOLOGY.items()}
    imp.update({"TROPP": st["tropp_mod"], "T": st["t_mod"], "Q": st["q_mod"], "PLE": st["ple_mod"]})
    pre = "oh_" if source == "PRECOMPUTED" else ""
    # the Boost inputs: in the ONLINE modes T, Q and PLE are the model's own fields
    online = {"ZLE": st["zle_bst"], "TAUCLW": st["tauclw"], "TAUCLI": st["taucli"], "CH4": st["ch4"], "CO": st["co"],
              "FCLD": st["cloud"]}
    if source == "PRECOMPUTED":
        online.update({"T": st["t_bst"], "Q": st["qv"], "PLE": st["ple_bst"]})
    for name, a in online.items():
        imp[pre + name] = a
    for a, name in zip(st["scacoef"], AEROSOLS):
        if source == "PRECOMPUTED":
            imp[f"oh_{name}SCACOEF"] = a                       # archived: 3-D
        else:                                                   # online: the wavelength is a 4th dimension
            cube = (rng.random((im, jm, km, len(WAVELENGTHS))) * 5e-6).astype(F32)
            cube[..., 1] = a                                    # 550 nm is the one OH asks for
            imp[f"{name}SCACOEF"] = cube
    lats = (rng.random((im, jm)) * np.pi - np.pi / 2).astype(F32)
    lons = (rng.random((im, jm)) * 2 * np.pi - np.pi).astype(F32)
    return imp, lats, lons


def write_state_file(path, grid, imports, lats, lons):
    im, jm, km = grid
    with open(path, "wb") as f:
        f.write(struct.pack("<5i", im, jm, km, len(WAVELENGTHS), len(imports)))
        f.write(helpers.fortran_flat(lats).tobytes())
        f.write(helpers.fortran_flat(lons).tobytes())
        for name, a in imports.items():
            kind = 2 if a.ndim == 2 else 5 if a.ndim == 4 else 4 if a.shape[2] == km + 1 else 3
            f.write(name.encode().ljust(32))
            f.write(struct.pack("<i", kind))
            f.write(helpers.fortran_flat(a).tobytes())


def write_rundir(d, *, source, model_pattern, once_per_day=True, spinup=True, policy="reference", run_dt=1800,
                 oh_dt=3600, ref_time="003000", beg="20240131 000000", exports=(), avg24_tick=-1,
                 active="OH", passive="", ohscale=0.85, wavelength=550, register=False, skip_tick=None):
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "AGCM.rc"), "w").write(
        f"# mock of the GEOS AGCM.rc keys OH reads\nRUN_DT: {run_dt}\nQUICKCHEM_DT: {run_dt}\nOH_DT: {oh_dt}\n"
        f"OH_REFERENCE_TIME: {ref_time}\nBEG_DATE: {beg}\nOH_EXPORTS: {' '.join(exports)}\nAVG24_READY_TICK: {avg24_tick}\n")
    open(os.path.join(d, "QuickChem_GridComp.rc"), "w").write(
        f"ACTIVE_INSTANCES_OH:  {active} # OH.1  OH.2\nPASSIVE_INSTANCES_OH:  {passive}\n")
    open(os.path.join(d, "GOCART2G_GridComp.rc"), "w").write(
        "wavelengths_for_profile_aop_in_nm: " + " ".join(map(str, WAVELENGTHS)) + "  # must hold OH's wavelength\n")
    open(os.path.join(d, "OH_instance_OH.rc"), "w").write(
        f"nbins: 1\nXGBoostFile: {model_pattern}\nOH_data_source: {source}\nspinup_24hr_imports: {'T' if spinup else 'F'}\n"
        f"wavelength_for_scacoef: {wavelength}\ncompute_once_per_day: {'T' if once_per_day else 'F'}\nOHscale: {ohscale}\n"
        f"XGBoost_model_policy: {policy}\n" + ("register_host_arrays: T\n" if register else "")
        + (f"skip_tick: {skip_tick}\n" if skip_tick else ""))


def run_driver(exe, rundir, state, out, nticks):
    r = subprocess.run([exe, str(rundir), str(state), str(out), str(nticks)], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    return r


def parse_output(path, grid, instances, exports):
    """-> list of ticks; tick["OH"] etc. per instance name.  `instances` = [(name, is_data)], `exports` = [(name, is2d)]."""
    im, jm, km = grid
    raw = open(path, "rb").read()
    nticks, ninst, nexp = struct.unpack_from("<3i", raw, 0)
    assert ninst == len(instances) and nexp == len(exports)
    off = 12
    vol, plane = im * jm * km, im * jm

    def take(n, shape):
        nonlocal off
        a = np.frombuffer(raw, dtype="<f4", count=n, offset=off).reshape(shape[::-1]).T
        off += 4 * n
        return np.ascontiguousarray(a)
    out = []
    for _ in range(nticks):
        tick, nymd, nhms = struct.unpack_from("<3i", raw, off)
        off += 12
        rec = {"tick": tick, "nymd": nymd, "nhms": nhms}
        for name, is_data in instances:
            if is_data:
                rec[name] = {"OH": take(vol, (im, jm, km))}
                continue
            ran, boosted, k1, k2 = struct.unpack_from("<4i", raw, off)
            off += 16
            model = raw[off:off + 256].decode().rstrip()
            off += 256
            inst = {"ran": bool(ran), "boosted": bool(boosted), "k1": k1, "k2": k2, "model": model,
                    "OH": take(vol, (im, jm, km))}
            (inst["parent_export_ok"],) = struct.unpack_from("<i", raw, off)
            off += 4
            for ename, is2d in exports:
                inst[ename] = take(plane, (im, jm)) if is2d else take(plane * (km + 1 if ename == "DIAG_ZLE" else km),
                                                                       (im, jm, km + 1 if ename == "DIAG_ZLE" else km))
            rec[name] = inst
        out.append(rec)
    assert off == len(raw)
    return out


def oracle_booster(image):
    return capi.Booster(model_buffer=image, lib=helpers.oracle_lib())


def oracle_solar(jday, lats, lons):
    lib = helpers.oracle_lib()
    lib.oracle_solar_geometry.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                          C.c_void_p, C.c_void_p]
    la, lo = helpers.fortran_flat(lats), helpers.fortran_flat(lons)
    jm, im = la.shape
    lat_deg, sza = np.empty_like(la), np.empty_like(la)
    assert lib.oracle_solar_geometry(jday, la.ctypes.data, lo.ctypes.data, im, jm, float(DEG2RAD), float(RAD2DEG),
                                     lat_deg.ctypes.data, sza.ctypes.data) == 0
    return lat_deg.T.copy(), sza.T.copy()


def emulate(grid, imports, lats, lons, nticks, *, source, models, policy, once_per_day, spinup, run_dt, oh_dt,
            ref_hms, beg, avg24_tick, ohscale, solar=None):
    """The shell's decisions restated; numbers from the oracle library.  `models` maps month -> image.
    `solar` (optional) maps tick -> (lat_deg, sza) to use instead of the host libm's (GPU runs: the device's trig)."""
    import datetime
    imp = {k: v.copy() for k, v in imports.items()}
    t0 = datetime.datetime.strptime(beg, "%Y%m%d %H%M%S")
    ref = t0.replace(hour=ref_hms // 10000, minute=ref_hms % 10000 // 100, second=ref_hms % 100)
    first_ring = ref - datetime.timedelta(seconds=run_dt)
    oh_ml = np.zeros(grid, dtype=F32)                   # zero-filled at Initialize (documented deviation from :893)
    first_model = None
    out = []
    for tick in range(nticks):
        now = t0 + datetime.timedelta(seconds=tick * run_dt)
        if tick > 0:
            imp["T"] = (imp["T"] * F32(1.0005)).astype(F32)
            imp["TROPP"] = (imp["TROPP"] * F32(1.002)).astype(F32)
        if tick == avg24_tick:
            for name in list(imp):
                if name + "_avg24" in imp:
                    imp[name + "_avg24"] = (imp[name] * F32(0.99)).astype(F32)
        # MAPL's run alarm: a ring time in (now - dt, now]
        k = (now - first_ring).total_seconds() // oh_dt
        ringing = (first_ring + datetime.timedelta(seconds=k * oh_dt)) > now - datetime.timedelta(seconds=run_dt)
        rec = {"tick": tick, "ran": bool(ringing), "boosted": False}
        out.append(rec)
        if not ringing:
            continue
        nhms = now.hour * 10000 + now.minute * 100 + now.second
        need_boost = not (once_per_day and nhms > 0)
        if need_boost:
            use_inst = bool(source == "ONLINE_AVG24" and imp["T_avg24"][0, 0, 0] == 0.0)

            def pick(base):
                if source == "PRECOMPUTED":
                    return imp["oh_" + base]
                if source == "ONLINE_INST" or use_inst:
                    return imp[base]
                return imp[base + "_avg24"]
            st = {key: pick(name) for key, name in ONLINE.items()}
            st.update({key: imp[name] for key, name in CLIMATOLOGY.items()})
            st.update({"ple_mod": imp["PLE"], "t_mod": imp["T"], "q_mod": imp["Q"], "tropp_mod": imp["TROPP"]})
            st["scacoef"] = [pick(a + "SCACOEF") if source == "PRECOMPUTED" else pick(a + "SCACOEF")[..., 1].copy()
                             for a in AEROSOLS]
            jday = now.timetuple().tm_yday
            st["lat_deg"], st["sza"] = solar[tick] if solar else oracle_solar(jday, lats, lons)
            month = now.month
            if policy == "reference":
                first_model = first_model if first_model is not None else month
                month = first_model
            rec["month"] = month
            b = oracle_booster(models[month])
            r = b.run1(st, dynamic_k_range=not once_per_day, tropp_min=4000.0, ohscale=ohscale, avogad=AVOGAD,
                       runiv=RUNIV, epsilon=EPSILON, want_diag=True)
            b.free()
            oh_ml = r["oh_boost"].copy()
            rec.update({"boosted": True, "OH": r["oh"], "OH_boost": r["oh_boost"], "DIAG_NDWET": r["ndwet"],
                        "k1": r["k1"], "k2": r["k2"], "use_inst": use_inst, "DIAG_LAT": st["lat_deg"],
                        "DIAG_SZA": st["sza"], "DIAG_TAUCLWDN": r["diag_tauclwdn"], "DIAG_AODUP": r["diag_aodup"],
                        "DIAG_PL": r["diag_pl_bst"], "DIAG_GMISTRATO3": r["diag_strato3"], "DIAG_T": st["t_bst"],
                        "DIAG_SC_DU": st["scacoef"][3], "DIAG_ZLE": st["zle_bst"], "DIAG_AOD": r["diag_aod"]})
        else:
            oh, ndwet = capi.oh_post_process(imp["PLE"], imp["T"], imp["Q"], imp["TROPP"], imp["oh_OH"], oh_ml,
                                             avogad=AVOGAD, runiv=RUNIV, epsilon=EPSILON, lib=helpers.oracle_lib())
            rec.update({"OH": oh, "DIAG_NDWET": ndwet})
    return out


EXPORTS = [("OH_boost", False), ("DIAG_NDWET", False), ("DIAG_SZA", True), ("DIAG_LAT", True), ("DIAG_TAUCLWDN", False),
           ("DIAG_AODUP", False), ("DIAG_PL", False), ("DIAG_GMISTRATO3", True), ("DIAG_T", False), ("DIAG_SC_DU", False),
           ("DIAG_ZLE", False), ("DIAG_AOD", False), ("DIAG_OH_M2G", False)]


def two_day_case(tmp_path, small_model, exe, on_gpu):
    """Two model days, half-hour heartbeat, OH_DT one hour, ONLINE_AVG24 with spin-up, compute_once_per_day,
    month roll-over from January to February with one resident booster per file name."""
    grid = (5, 4, 24)
    imports, lats, lons = mock_imports(grid, "ONLINE_AVG24")
    # ONLINE_AVG24 with spin-up declares both the instantaneous and the daily-mean imports; the means start at zero
    for name in ["CH4", "CO", "T", "FCLD", "Q", "TAUCLW", "TAUCLI", "PLE", "ZLE"] + [a + "SCACOEF" for a in AEROSOLS]:
        imports[name + "_avg24"] = np.zeros_like(imports[name])
    other = synth.make_model(num_trees=20, max_depth=10, sample_log2=15, min_leaf=4, grid=synth.GRIDS["C12"], model_seed=77)
    (tmp_path / "oh_M01.model").write_bytes(small_model.image.tobytes())
    (tmp_path / "oh_M02.model").write_bytes(other.image.tobytes())
    models = {1: small_model.image, 2: other.image}
    cfg = dict(source="ONLINE_AVG24", policy="by_name", once_per_day=True, spinup=True, run_dt=1800, oh_dt=3600,
               avg24_tick=40, ohscale=0.85)
    rundir = tmp_path / "run"
    # on the GPU with register_host_arrays: the shell's arrays are registered at their first tick and moved by one copy
    # launch from then on, 98 heartbeats long (the emulation below does not know the key)
    write_rundir(rundir, model_pattern=str(tmp_path / "oh_M%m2.model"), ref_time="003000", beg="20240131 000000",
                 exports=[e for e, _ in EXPORTS], register=on_gpu, **cfg)
    state, out = tmp_path / "state.bin", tmp_path / "out.bin"
    write_state_file(state, grid, imports, lats, lons)
    nticks = 98
    r = run_driver(exe, rundir, state, out, nticks)
    assert r.returncode == 0, r.stdout[-3000:]
    got = parse_output(out, grid, [("OH", False)], EXPORTS)
    solar = None
    if on_gpu:       # the device's trig for SZA (tolerance tested in tests/test_run1.py): feed the oracle what the GPU used
        solar = {t["tick"]: (t["OH"]["DIAG_LAT"], t["OH"]["DIAG_SZA"]) for t in got if t["OH"]["boosted"]}
    want = emulate(grid, imports, lats, lons, nticks, models=models, ref_hms=3000, beg="20240131 000000", solar=solar, **cfg)
    return grid, got, want, r.stdout


def check_two_days(grid, got, want, log, on_gpu):
    boosts = [t["tick"] for t in got if t["OH"]["boosted"]]
    assert boosts == [0, 48, 96], boosts                           # once per model day, at nhms == 0
    assert [t["tick"] for t in got if t["OH"]["ran"]] == list(range(0, 98, 2))      # the alarm: every other heartbeat
    assert got[0]["nymd"] == 20240131 and got[48]["nymd"] == 20240201 and got[48]["nhms"] == 0
    assert got[0]["OH"]["model"].endswith("oh_M01.model") and got[48]["OH"]["model"].endswith("oh_M02.model")
    assert want[0]["use_inst"] is True and want[48]["use_inst"] is False          # daily means arrived at tick 40
    assert "OH is in the SPINUP period" in log and "OH is *NOT* in the SPINUP period" in log
    last = None
    for g, w in zip(got, want):
        inst = g["OH"]
        assert inst["ran"] == w["ran"] and inst["boosted"] == w["boosted"], g["tick"]
        assert inst["parent_export_ok"] == 1                       # QuickChem's export OH IS the first instance's INTERNAL OH
        if not w["ran"]:
            assert last is not None and np.array_equal(helpers.bits(inst["OH"]), helpers.bits(last)), g["tick"]   # untouched
            continue
        if w["boosted"]:
            assert (inst["k1"], inst["k2"]) == (w["k1"], w["k2"])
            for name in ("DIAG_NDWET", "DIAG_LAT", "DIAG_SZA", "DIAG_TAUCLWDN", "DIAG_AODUP", "DIAG_PL", "DIAG_GMISTRATO3",
                         "DIAG_T", "DIAG_SC_DU", "DIAG_ZLE", "DIAG_AOD"):
                assert np.array_equal(helpers.bits(inst[name]), helpers.bits(w[name])), (g["tick"], name)
            k1 = w["k1"]
            assert np.all(inst["OH_boost"][:, :, :k1 - 1] == 0)
            assert helpers.ulp_diff(inst["OH_boost"][:, :, k1 - 1:], w["OH_boost"][:, :, k1 - 1:]).max() <= (2 if on_gpu else 0)
            assert helpers.ulp_diff(inst["OH"], w["OH"]).max() <= (3 if on_gpu else 0)
        else:
            assert np.array_equal(helpers.bits(inst["DIAG_NDWET"]), helpers.bits(w["DIAG_NDWET"])), g["tick"]
            if on_gpu:       # OH_ML persisted from a Boost that may differ by 2 ulp in 10**x
                assert helpers.ulp_diff(inst["OH"], w["OH"]).max() <= 3
            else:
                assert np.array_equal(helpers.bits(inst["OH"]), helpers.bits(w["OH"])), g["tick"]
            # OH_boost keeps the last Boost's values on the ticks that skip it (the export is only written by Boost)
        last = inst["OH"]
    # the second day's Boost really used other inputs and another model
    assert not np.array_equal(got[0]["OH"]["OH_boost"], got[48]["OH"]["OH_boost"])


def test_config1_two_model_days_through_the_gridcomp_on_the_cpu(tmp_path, small_model, drivers):
    grid, got, want, log = two_day_case(tmp_path, small_model, drivers["oracle"], on_gpu=False)
    check_two_days(grid, got, want, log, on_gpu=False)


@pytest.mark.gpu
def test_two_model_days_through_the_gridcomp_on_the_gpu(tmp_path, small_model, drivers):
    grid, got, want, log = two_day_case(tmp_path, small_model, drivers["hip"], on_gpu=True)
    check_two_days(grid, got, want, log, on_gpu=True)


@pytest.mark.parametrize("source", ["PRECOMPUTED", "ONLINE_INST"])
def test_data_sources_and_the_reference_model_policy(tmp_path, small_model, source, drivers):
    """The other two OH_data_source settings, Boost at every alarm (compute_once_per_day: F -> dynamic k range),
    default alarm phase (no OH_REFERENCE_TIME: it rings during the LAST heartbeat of each OH_DT interval), and the
    reference's model policy: the file of the first call for good, whatever month the name says."""
    grid = (4, 3, 20)
    imports, lats, lons = mock_imports(grid, source, seed=9)
    other = synth.make_model(num_trees=20, max_depth=10, sample_log2=15, min_leaf=4, grid=synth.GRIDS["C12"], model_seed=78)
    (tmp_path / "oh_M01.model").write_bytes(small_model.image.tobytes())
    (tmp_path / "oh_M02.model").write_bytes(other.image.tobytes())
    cfg = dict(source=source, policy="reference", once_per_day=False, spinup=False, run_dt=1800, oh_dt=3600,
               avg24_tick=-1, ohscale=1.0)
    rundir = tmp_path / "run"
    ex = [("OH_boost", False), ("DIAG_NDWET", False), ("DIAG_SZA", True), ("DIAG_LAT", True)]
    write_rundir(rundir, model_pattern=str(tmp_path / "oh_M%m2.model"), ref_time="000000", beg="20240131 220000",
                 exports=[e for e, _ in ex], **cfg)
    state, out = tmp_path / "state.bin", tmp_path / "out.bin"
    write_state_file(state, grid, imports, lats, lons)
    r = run_driver(drivers["oracle"], rundir, state, out, 8)
    assert r.returncode == 0, r.stdout[-3000:]
    got = parse_output(out, grid, [("OH", False)], ex)
    want = emulate(grid, imports, lats, lons, 8, models={1: small_model.image, 2: other.image}, ref_hms=0,
                   beg="20240131 220000", **cfg)
    assert [t["tick"] for t in got if t["OH"]["ran"]] == [1, 3, 5, 7]          # 22:30, 23:30, 00:30, 01:30
    assert got[5]["nymd"] == 20240201
    for g, w in zip(got, want):
        assert g["OH"]["boosted"] == w["boosted"] == g["OH"]["ran"]
        if w["boosted"]:
            assert w["month"] == 1                                              # February's file is never opened
            assert np.array_equal(helpers.bits(g["OH"]["OH"]), helpers.bits(w["OH"])), g["tick"]
            assert np.array_equal(helpers.bits(g["OH"]["OH_boost"]), helpers.bits(w["OH_boost"]))
    assert got[7]["OH"]["model"].endswith("oh_M02.model")     # the expanded name rolled over; the booster did not


def test_setservices_refuses_what_the_reference_refuses(tmp_path, small_model, drivers):
    grid = (4, 3, 12)
    imports, lats, lons = mock_imports(grid, "ONLINE_INST")
    state, out = tmp_path / "state.bin", tmp_path / "out.bin"
    write_state_file(state, grid, imports, lats, lons)
    (tmp_path / "m.model").write_bytes(small_model.image.tobytes())
    base = dict(model_pattern=str(tmp_path / "m.model"), exports=[])
    # an OH_data_source that is none of the three: VERIFY_(99) (:549-559)
    write_rundir(tmp_path / "a", source="ONLINE", **base)
    r = run_driver(drivers["oracle"], tmp_path / "a", state, out, 1)
    assert r.returncode != 0 and "Invalid OH_data_source: ONLINE" in r.stdout
    # the wavelength OH wants is not among GOCART2G's (:586-590)
    write_rundir(tmp_path / "b", source="ONLINE_INST", wavelength=532, **base)
    r = run_driver(drivers["oracle"], tmp_path / "b", state, out, 1)
    assert r.returncode != 0 and "Did not find OH wavelength_for_scacoef" in r.stdout
    # a model file that is not there: the first Boost fails, Run returns an error
    write_rundir(tmp_path / "c", source="ONLINE_INST", model_pattern=str(tmp_path / "nope_%m2.model"), exports=[])
    r = run_driver(drivers["oracle"], tmp_path / "c", state, out, 2)
    assert r.returncode != 0 and "Run phase 1 failed" in r.stdout
    # a tropopause at or below 40 hPa with the static k range (:287-288)
    low = dict(imports)
    low["TROPP"] = imports["TROPP"].copy()
    low["TROPP"][1, 1] = 3900.0
    write_state_file(tmp_path / "low.bin", grid, low, lats, lons)
    write_rundir(tmp_path / "d", source="ONLINE_INST", once_per_day=True, **base)
    r = run_driver(drivers["oracle"], tmp_path / "d", tmp_path / "low.bin", out, 1)
    assert r.returncode != 0 and "Minimum tropopause pressure is not low enough" in r.stdout


def test_passive_data_instance_and_a_start_after_midnight(tmp_path, small_model, drivers):
    """QuickChem_GridComp.rc lists: an active computational instance and a passive data-driven one
    (its name contains 'data': Run_data copies its import climoh001 into INTERNAL OH, no phase 2).  The run starts at 06:00 with
    compute_once_per_day: Boost is not due until midnight, and INTERNAL OH is built from the zero-filled OH_ML
    (the reference reads uninitialised memory here, :893,1582)."""
    grid = (3, 3, 16)
    imports, lats, lons = mock_imports(grid, "ONLINE_INST", seed=21)
    (tmp_path / "m.model").write_bytes(small_model.image.tobytes())
    rundir = tmp_path / "run"
    write_rundir(rundir, source="ONLINE_INST", model_pattern=str(tmp_path / "m.model"), once_per_day=True, run_dt=3600,
                 oh_dt=3600, ref_time="010000", beg="20240310 060000", exports=["OH_boost"], passive="OH.data")
    # the data instance reads OH_instance_OH.data.rc if there is one, else OH_instance_OH.rc (:533-538)
    state, out = tmp_path / "state.bin", tmp_path / "out.bin"
    clim = (np.random.default_rng(4).random(grid) * 1e-13).astype(F32)
    write_state_file(state, grid, dict(imports, climoh001=clim), lats, lons)
    r = run_driver(drivers["oracle"], rundir, state, out, 3)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "OH_instance_OH.data.rc does not exist" in r.stdout
    got = parse_output(out, grid, [("OH", False), ("OH.data", True)], [("OH_boost", False)])
    pl = (imports["PLE"][:, :, :-1] + imports["PLE"][:, :, 1:]) * F32(0.5)
    for t in got:
        assert t["OH"]["ran"] and not t["OH"]["boosted"]
        assert np.array_equal(t["OH.data"]["OH"], clim)
        assert np.all(t["OH"]["OH_boost"] == 0)
    below = pl > imports["TROPP"][:, :, None]                     # tick 0: the model has not moved yet
    assert below.any() and np.all(got[0]["OH"]["OH"][below] == 0)
    assert np.all(got[0]["OH"]["OH"][~below] > 0)                  # default_OH * NDWET * 1e-6 above the tropopause


def test_mapl_lite_config_reader(tmp_path, drivers):
    """The ESMF_Config subset through a tiny Fortran-free check: the driver refuses an rc without BEG_DATE, reads
    '#' comments, logical spellings and vectors (exercised by every case above); here the odd spellings."""
    grid = (3, 3, 8)
    imports, lats, lons = mock_imports(grid, "ONLINE_INST")
    state, out = tmp_path / "state.bin", tmp_path / "out.bin"
    write_state_file(state, grid, imports, lats, lons)
    rundir = tmp_path / "run"
    write_rundir(rundir, source="ONLINE_INST", model_pattern="/nonexistent", exports=[])
    rc = (rundir / "OH_instance_OH.rc").read_text()
    (rundir / "OH_instance_OH.rc").write_text(
        "   # leading comment\n" + rc.replace("compute_once_per_day: T", "   compute_once_per_day:   .TRUE.   # indented, Fortran spelling")
        .replace("spinup_24hr_imports: F", "spinup_24hr_imports: no"))
    agcm = (rundir / "AGCM.rc").read_text().replace("BEG_DATE: 20240131 000000", "BEG_DATE: 20240131 003000")
    (rundir / "AGCM.rc").write_text(agcm)
    r = run_driver(drivers["oracle"], rundir, state, out, 2)
    assert r.returncode == 0, r.stdout[-2000:]               # nhms = 003000 > 0 and once per day: no model file needed
    got = parse_output(out, grid, [("OH", False)], [])
    # ring times are on the hour (OH_REFERENCE_TIME = one heartbeat): silent at 00:30, ringing at 01:00
    assert got[0]["nhms"] == 3000 and not got[0]["OH"]["ran"]
    assert got[1]["nhms"] == 10000 and got[1]["OH"]["ran"] and not got[1]["OH"]["boosted"]
    (rundir / "AGCM.rc").write_text(agcm.replace("BEG_DATE", "BEGIN"))
    r = run_driver(drivers["oracle"], rundir, state, out, 1)
    assert r.returncode != 0 and "BEG_DATE" in r.stdout


# ------------------------------------------------------------------ the spec table

MAPL_CODE = {"MAPL_DimsHorzOnly": 2, "MAPL_DimsHorzVert": 3, "MAPL_VLocationNone": 0, "MAPL_VLocationCenter": 1,
             "MAPL_VLocationEdge": 2, "MAPL_RestartOptional": 0, "MAPL_RestartSkip": 1, "MAPL_RestartRequired": 2}
MAPL_CODE.update({k.lower(): v for k, v in list(MAPL_CODE.items())})


def expected_specs(golden, source, spinup, nbins=1, n4=len(WAVELENGTHS)):
    """What the reference's SetServices registers for a computational instance with this OH_data_source (None: a
    data-driven instance), from tests/golden/oh_specs.json: {(state, short name): (dims, vloc, restart, refresh,
    averaging, ungridded, add2export, units, long name)}."""
    dflt = golden["mapl_defaults"]

    def row(state, r, ungridded=0):
        if "ungridded_dims" in r:
            ungridded = {"[self%nbins]": nbins, "[self%n_wavelengths_profile]": n4}[r["ungridded_dims"]]
        return ((state, r["short_name"]),
                (MAPL_CODE[r["dims"].lower()], MAPL_CODE[r.get("vlocation", dflt["vlocation"]).lower()],
                 MAPL_CODE[r.get("restart", dflt["restart"]).lower()], r.get("refresh_interval", 0),
                 r.get("averaging_interval", 0), ungridded, bool(r.get("add2export", False)) and state == "INTERNAL",
                 r.get("units", ""), r.get("long_name", "").rstrip()))
    out = []
    if source is None:
        for r in golden["data_instance"]:
            if "per" in r:
                for b in range(1, nbins + 1):
                    out.append(row(r["state"], dict(r, short_name=r["short_name"].format(bin=b),
                                                    long_name=r["long_name"].format(bin=b))))
            else:
                out.append(row(r["state"], r))
        return dict(out)
    for state in ("IMPORT", "EXPORT", "INTERNAL"):
        out += [row(state, r) for r in golden["state_specs"][state]]
    blocks = ["always"]
    if source == "ONLINE_INST" or (source == "ONLINE_AVG24" and spinup):
        blocks.append("IMPORT_INST")
    if source == "ONLINE_AVG24":
        blocks.append("IMPORT_24")
    if source == "PRECOMPUTED":
        blocks.append("IMPORT_PRECOMPUTED")
    for b in blocks:
        out += [row("IMPORT", r) for r in golden["conditional_imports"][b]]
    assert len(dict(out)) == len(out)
    return dict(out)


def registered_specs(path):
    got = {}
    for line in open(path):
        inst, state, name, dims, vloc, restart, refresh, averaging, ungridded, a2e, units, long_name = line.rstrip("\n").split("|")
        got.setdefault(inst, {})[(state, name)] = (int(dims), int(vloc), int(restart), int(refresh), int(averaging),
                                                   int(ungridded), a2e.strip() == "T", units, long_name.rstrip())
    return got


@pytest.mark.parametrize("source,spinup", [("PRECOMPUTED", False), ("ONLINE_INST", False), ("ONLINE_AVG24", False),
                                           ("ONLINE_AVG24", True)])
def test_setservices_registers_the_reference_spec_table(tmp_path, small_model, source, spinup, drivers):
    """Every field the reference's OH SetServices registers - the rows of OH_StateSpecs.rc and the ADD_IMPORT_* lines of
    OH_GridCompMod.F90:611-634,693-783, as tests/golden/oh_specs.json holds them - is registered by the product's with
    the same short name, dims, vlocation, restart, refresh / averaging interval, ungridded dimension, units and long
    name, for a computational instance under every OH_data_source (with and without spin-up) and for a data-driven
    instance beside it; and nothing else is."""
    import json
    golden = json.load(open(os.path.join(helpers.GOLDEN, "oh_specs.json")))
    grid = (3, 3, 8)
    imports, lats, lons = mock_imports(grid, "ONLINE_INST")
    state, out = tmp_path / "state.bin", tmp_path / "out.bin"
    write_state_file(state, grid, imports, lats, lons)
    rundir = tmp_path / "run"
    write_rundir(rundir, source=source, spinup=spinup, model_pattern="/nonexistent", exports=[], passive="OH.data")
    with open(rundir / "AGCM.rc", "a") as f:
        f.write(f"SPEC_DUMP: {tmp_path / 'specs.txt'}\n")
    r = run_driver(drivers["oracle"], rundir, state, out, 0)
    assert r.returncode == 0, r.stdout[-2000:]
    got = registered_specs(tmp_path / "specs.txt")
    assert set(got) == {"OH", "OH.data"}
    for inst, want in (("OH", expected_specs(golden, source, spinup)), ("OH.data", expected_specs(golden, None, False))):
        assert set(got[inst]) == set(want), (inst, sorted(set(got[inst]) ^ set(want)))
        for key in want:
            assert got[inst][key] == want[key], (inst, key, got[inst][key], want[key])
