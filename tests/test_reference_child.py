"""The reference's OWN OH child as the checker of the product's.

oracle/Makefile (target `ref`) compiles, in place from /root/reference and without changing a character,
  OH_GridComp/OH_GridCompMod.F90     SetServices / Initialize / Run / Run1 (feature engineering :1444-1482, CALL_BOOST
                                     :1557-1574 with predict_OH_with_XGB :123-398, mask :1579-1587, conversion :1595,
                                     the DIAG dumps :1598-1728) / Run2
  QC_Environment/QC_EnvironmentMod.F90, Shared/QuickChem_Generic.F90, Shared/xgb_fortran_api.F90, QuickChem_GridCompMod.F90
against the mapl_lite mock and the five OH_*___.h headers tools/acg_lite.py writes from OH_StateSpecs.rc, and links
them under the same mock GEOS cap as the product's shell (tests/fortran/oh_gridcomp_driver.F90 -DOHX_REFERENCE_CHILD)
  * against liboracle_xgb.so  -> oracle/_ref/refchild/oh_refchild_driver_oracle   (CPU)
  * against libohxgb.so       -> oracle/_ref/refchild/oh_refchild_driver_hip      (the real caller on the MI355X, through
                                                                                   the eleven C symbols: the drop-in)
Here the same synthetic MAPL state runs through the reference's child and through the product's shell, and INTERNAL OH,
OH_boost and every DIAG export both fill are compared tick by tick.  So the arithmetic AROUND the tree walk - the six
SUM(x(k:km)) / SUM(x(1:k)) optical depths, AOD, stratO3, PL, local-noon SZA, NDWET, the WHERE mask, mol/mol -> molec/cm3 -
is checked against the reference's own lines, not against the builder's reading of them.

What this does NOT do: pin xgboost.  Under both children the trees are walked by the oracle or by the product; the
reference holds no xgboost arithmetic.  PARITY WITH libxgboost 1.6.0 STAYS UNPINNED.

Skipped where oracle/_ref/refchild is not built (no /root/reference and no prebuilt files)."""
import os

import numpy as np
import pytest

from quickchem_amd import synth
from tests import helpers
from tests import test_gridcomp as tg

REFCHILD = os.path.join(helpers.ROOT, "oracle", "_ref", "refchild")
REF_ORACLE = os.path.join(REFCHILD, "oh_refchild_driver_oracle")
REF_HIP = os.path.join(REFCHILD, "oh_refchild_driver_hip")

pytestmark = pytest.mark.skipif(not os.path.exists(REF_ORACLE),
                                reason="oracle/_ref/refchild not built: the reference's OH_GridCompMod.F90 is compiled "
                                       "in place from /root/reference by oracle/Makefile")

# every export both children fill (OH_StateSpecs.rc:41-75 less DIAG_T_avg24 / DIAG_T_in_OH, which the reference only
# mentions in comments, :1333-1334,1610-1613); True = 2-D
EXPORTS = [("OH_boost", False), ("DIAG_OH_M2G", False), ("DIAG_NDWET", False), ("DIAG_LAT", True), ("DIAG_SZA", True),
           ("DIAG_TAUCLWDN", False), ("DIAG_TAUCLIDN", False), ("DIAG_TAUCLIUP", False), ("DIAG_TAUCLWUP", False),
           ("DIAG_GMISTRATO3", True), ("DIAG_ALBUV", True), ("DIAG_AODUP", False), ("DIAG_AODDN", False),
           ("DIAG_PL", False), ("DIAG_T", False), ("DIAG_CH4", False), ("DIAG_CO", False), ("DIAG_CLOUD", False),
           ("DIAG_QV", False), ("DIAG_ZLE", False), ("DIAG_AOD", False), ("DIAG_C2H6", False), ("DIAG_ISOP", False),
           ("DIAG_SC_BC", False), ("DIAG_SC_OC", False), ("DIAG_SC_BR", False), ("DIAG_SC_DU", False),
           ("DIAG_SC_SU", False), ("DIAG_SC_SS", False), ("DIAG_SC_NI", False)]
# what the GPU's fused kernel computes with its own 10**x (tests/test_run1.py: <= 2 ulp of powf; x NDWET x 1e-6: <= 3)
POW10 = {"OH": 3, "OH_boost": 2}


def avg24_imports(grid, seed=3):
    imports, lats, lons = tg.mock_imports(grid, "ONLINE_AVG24", seed=seed)
    for name in ["CH4", "CO", "T", "FCLD", "Q", "TAUCLW", "TAUCLI", "PLE", "ZLE"] + [a + "SCACOEF" for a in tg.AEROSOLS]:
        imports[name + "_avg24"] = np.zeros_like(imports[name])
    return imports, lats, lons


def run_both(tmp_path, small_model, ref_exe, product_exe, *, grid, source, nticks, seed=3, exports=None, instances=None,
             extra_imports=None, **cfg):
    """The same run directory and state file through the reference's child and the product's shell."""
    imports, lats, lons = avg24_imports(grid, seed) if source == "ONLINE_AVG24" else tg.mock_imports(grid, source, seed=seed)
    imports.update(extra_imports or {})
    other = synth.make_model(num_trees=20, max_depth=10, sample_log2=15, min_leaf=4, grid=synth.GRIDS["C12"], model_seed=77)
    (tmp_path / "oh_M01.model").write_bytes(small_model.image.tobytes())
    (tmp_path / "oh_M02.model").write_bytes(other.image.tobytes())
    rundir, state = tmp_path / "run", tmp_path / "state.bin"
    # policy "reference": the reference's child loads the first file for good (:209,269), so the product is told to as well
    want = EXPORTS if exports is None else exports
    tg.write_rundir(rundir, source=source, model_pattern=str(tmp_path / "oh_M%m2.model"), policy="reference",
                    exports=[e for e, _ in want], **cfg)
    tg.write_state_file(state, grid, imports, lats, lons)
    out = {}
    for tag, exe in (("reference", ref_exe), ("product", product_exe)):
        r = tg.run_driver(exe, rundir, state, tmp_path / f"{tag}.bin", nticks)
        assert r.returncode == 0, (tag, r.stdout[-3000:])
        out[tag] = (tg.parse_output(tmp_path / f"{tag}.bin", grid, instances or [("OH", False)], want), r.stdout)
    return out, imports, lats, lons, {1: small_model.image, 2: other.image}


def compare(ref_ticks, prod_ticks, tolerance=None, exports=None):
    """Tick by tick, field by field; `tolerance` maps a field to the ulps it may differ by (default: bit for bit)."""
    tolerance = tolerance or {}
    exports = EXPORTS if exports is None else exports
    assert len(ref_ticks) == len(prod_ticks)
    boosts = 0
    for a, b in zip(ref_ticks, prod_ticks):
        assert (a["tick"], a["nymd"], a["nhms"]) == (b["tick"], b["nymd"], b["nhms"])
        ra, pb = a["OH"], b["OH"]
        assert ra["ran"] is True and ra["k1"] == -1            # -1 on file: the reference's child does not say what it did
        assert ra["parent_export_ok"] == pb["parent_export_ok"] == 1
        boosts += pb["boosted"]
        for name in ["OH"] + [e for e, _ in exports]:
            x, y = ra[name], pb[name]
            if name in tolerance:
                worst = int(helpers.ulp_diff(x, y).max())
                assert worst <= tolerance[name], (a["tick"], name, worst)
            else:
                differ = int(np.count_nonzero(helpers.bits(x) != helpers.bits(y)))
                assert differ == 0, (a["tick"], name, differ, int(helpers.ulp_diff(x, y).max()))
    return boosts


@pytest.mark.parametrize("skip_tick", [None, "device"])
def test_two_model_days_reference_child_against_product_shell_on_the_cpu(tmp_path, small_model, skip_tick):
    """BASELINE config #1's two model days (half-hour heartbeat, OH_DT one hour, ONLINE_AVG24 with spin-up,
    compute_once_per_day -> static k range, daily means arriving at tick 40, January -> February): every tick's INTERNAL
    OH and all thirty exports of the reference's own child and of the product's shell, both over the oracle: bit for bit.
    46 of the 49 OH ticks skip Boost (:1189-1193): the shell's fused host pass (skip_tick: host, the default) and its
    OHXOHPostProcess call (skip_tick: device; here the oracle's) against the reference's :1247-1257,1579-1595."""
    out, *_ = run_both(tmp_path, small_model, REF_ORACLE, tg.DRIVER_ORACLE, grid=(5, 4, 24), source="ONLINE_AVG24",
                       nticks=98, once_per_day=True, spinup=True, run_dt=1800, oh_dt=3600, avg24_tick=40, ohscale=0.85,
                       ref_time="003000", beg="20240131 000000", skip_tick=skip_tick)
    (ref, ref_log), (prod, prod_log) = out["reference"], out["product"]
    assert compare(ref, prod) == 3                                                # Boost at ticks 0, 48, 96
    for log in (ref_log, prod_log):
        assert "OH is in the SPINUP period" in log and "OH is *NOT* in the SPINUP period" in log
    assert np.abs(ref[0]["OH"]["OH"]).sum() > 0 and not np.array_equal(ref[0]["OH"]["OH_boost"], ref[48]["OH"]["OH_boost"])
    # the non-trivial engineered features really are non-trivial
    assert ref[0]["OH"]["DIAG_AODUP"].max() > 0 and ref[0]["OH"]["DIAG_TAUCLWDN"].max() > 0
    assert 0 < ref[0]["OH"]["DIAG_SZA"].min() and ref[0]["OH"]["DIAG_SZA"].max() < 180


@pytest.mark.parametrize("source", ["PRECOMPUTED", "ONLINE_INST"])
def test_other_data_sources_reference_child_against_product_shell_on_the_cpu(tmp_path, small_model, source):
    """The other two OH_data_source settings with Boost at every alarm (dynamic k range, :275-285), default alarm phase,
    a run across midnight: bit for bit again."""
    out, *_ = run_both(tmp_path, small_model, REF_ORACLE, tg.DRIVER_ORACLE, grid=(4, 3, 20), source=source, nticks=8,
                       seed=9, once_per_day=False, spinup=False, run_dt=1800, oh_dt=3600, avg24_tick=-1, ohscale=1.0,
                       ref_time="000000", beg="20240131 220000")
    (ref, _), (prod, _) = out["reference"], out["product"]
    # the alarm rings on ticks 1, 3, 5, 7; on the others neither child touches anything
    assert compare(ref, prod) == 4
    assert [t["tick"] for t in prod if t["OH"]["ran"]] == [1, 3, 5, 7]


def test_the_oracle_restatement_of_run1_against_the_reference_child(tmp_path, small_model):
    """oracle/xgb_oracle.c's OHXBoosterRun1 / OHXOHPostProcess - the builder's reading of OH_GridCompMod.F90:1240-1257,
    1444-1482, 1557-1595, which every other Run1 test checks the GPU against - produce what the reference's own lines
    produce: the Python emulation of tests/test_gridcomp.py (numbers from the oracle library) against the reference's
    child, bit for bit, two model days."""
    grid, cfg = (5, 4, 24), dict(source="ONLINE_AVG24", once_per_day=True, spinup=True, run_dt=1800, oh_dt=3600,
                                 avg24_tick=40, ohscale=0.85)
    out, imports, lats, lons, models = run_both(tmp_path, small_model, REF_ORACLE, tg.DRIVER_ORACLE, grid=grid, nticks=98,
                                                ref_time="003000", beg="20240131 000000", **cfg)
    ref, _ = out["reference"]
    want = tg.emulate(grid, imports, lats, lons, 98, models=models, policy="reference", ref_hms=3000,
                      beg="20240131 000000", **cfg)
    checked = 0
    for g, w in zip(ref, want):
        if not w["ran"]:
            continue
        for name in ("OH", "OH_boost", "DIAG_NDWET", "DIAG_LAT", "DIAG_SZA", "DIAG_TAUCLWDN", "DIAG_AODUP", "DIAG_PL",
                     "DIAG_GMISTRATO3", "DIAG_T", "DIAG_SC_DU", "DIAG_ZLE", "DIAG_AOD"):
            if name in w:
                assert np.array_equal(helpers.bits(g["OH"][name]), helpers.bits(w[name])), (g["tick"], name)
                checked += 1
    assert checked > 100


def test_reference_child_setservices_registers_its_own_spec_table(tmp_path, small_model):
    """The spec table of tests/golden/oh_specs.json (made from the reference's files by a committed script) is what the
    reference's child really registers when it runs - so the golden and the product's table are compared with the
    running reference, not only with a parse of its text."""
    import json
    golden = json.load(open(os.path.join(helpers.GOLDEN, "oh_specs.json")))
    grid = (3, 3, 8)
    imports, lats, lons = tg.mock_imports(grid, "ONLINE_INST")
    state, out = tmp_path / "state.bin", tmp_path / "out.bin"
    tg.write_state_file(state, grid, imports, lats, lons)
    for source, spinup in (("PRECOMPUTED", False), ("ONLINE_INST", False), ("ONLINE_AVG24", False), ("ONLINE_AVG24", True)):
        rundir = tmp_path / f"run_{source}_{spinup}"
        tg.write_rundir(rundir, source=source, spinup=spinup, model_pattern="/nonexistent", exports=[])
        with open(rundir / "AGCM.rc", "a") as f:
            f.write(f"SPEC_DUMP: {tmp_path / 'specs.txt'}\n")
        r = tg.run_driver(REF_ORACLE, rundir, state, out, 0)
        assert r.returncode == 0, r.stdout[-2000:]
        got = tg.registered_specs(tmp_path / "specs.txt")["OH"]
        want = tg.expected_specs(golden, source, spinup)
        assert set(got) == set(want), sorted(set(got) ^ set(want))
        for key in want:
            assert got[key] == want[key], (source, spinup, key, got[key], want[key])


def test_a_passive_data_instance_is_where_the_reference_child_stops_and_the_product_shell_does_not(tmp_path, small_model):
    """The one place the product's shell deviates from the reference's lines on purpose (header of oh_gridcomp.F90), shown
    with the reference's own code: a passive data-driven instance (OH.data) beside the active one.  The reference's
    SetServices registers, for such an instance, the import climoh001 and a 4-D INTERNAL OH (:611-634); its Run_data then
    takes a 3-D pointer to that field and reads the import oh_OH, which a data instance does not have (:1872,1880) - the
    run ends at its first tick with the traceback naming those lines.  The product's Run_data follows the registration
    (and the reference's commented-out lines, :1877-1878,1882): the same run directory goes through, INTERNAL OH of the
    data instance is the import, bit for bit."""
    instances = [("OH", False), ("OH.data", True)]
    grid = (5, 4, 24)
    imports, lats, lons = tg.mock_imports(grid, "PRECOMPUTED", seed=3)
    clim = (np.random.default_rng(4).random(grid) * 1e-13).astype(tg.F32)       # the data instance's import (bin 1)
    imports["climoh001"] = clim
    (tmp_path / "oh.model").write_bytes(small_model.image.tobytes())
    rundir, state = tmp_path / "run", tmp_path / "state.bin"
    tg.write_rundir(rundir, source="PRECOMPUTED", model_pattern=str(tmp_path / "oh.model"), policy="reference",
                    exports=["OH_boost"], once_per_day=True, spinup=False, run_dt=1800, oh_dt=3600, avg24_tick=-1,
                    ohscale=0.85, ref_time="010000", beg="20240310 060000", passive="OH.data")
    tg.write_state_file(state, grid, imports, lats, lons)
    r = tg.run_driver(REF_ORACLE, rundir, state, tmp_path / "reference.bin", 4)
    assert r.returncode != 0
    assert "OH_GridCompMod.F90 1872" in r.stdout and "Run phase 1 failed" in r.stdout
    r = tg.run_driver(tg.DRIVER_ORACLE, rundir, state, tmp_path / "product.bin", 4)
    assert r.returncode == 0, r.stdout[-3000:]
    prod = tg.parse_output(tmp_path / "product.bin", grid, instances, [("OH_boost", False)])
    assert len(prod) == 4
    for t in prod:
        assert np.array_equal(t["OH.data"]["OH"], clim)
        assert t["OH"]["ran"] in (True, False)


def test_two_active_instances_reference_child_against_product_shell(tmp_path, small_model):
    """ACTIVE_INSTANCES_OH: OH.1 OH.2 (QuickChem_GridComp.rc:22).  In the reference the booster handle and its first_time
    flag are SAVE variables of predict_OH_with_XGB (:182,209): two instances in one process share one booster, loaded by
    whichever ticks first.  The product's host keeps boosters by file name; with the same file the effect is the same.
    Both instances, every tick, every export, reference child against product shell."""
    instances = [("OH.1", False), ("OH.2", False)]
    out, *_ = run_both(tmp_path, small_model, REF_ORACLE, tg.DRIVER_ORACLE, grid=(4, 3, 20), source="ONLINE_INST", nticks=6,
                       seed=13, instances=instances, once_per_day=False, spinup=False, run_dt=1800, oh_dt=3600,
                       avg24_tick=-1, ohscale=0.9, ref_time="000000", beg="20240131 220000", active="OH.1 OH.2")
    (ref, _), (prod, _) = out["reference"], out["product"]
    assert len(ref) == len(prod) == 6
    boosts = 0
    for a, b in zip(ref, prod):
        for inst, _ in instances:
            boosts += b[inst]["boosted"]
            for name in ["OH"] + [e for e, _ in EXPORTS]:
                assert np.array_equal(helpers.bits(a[inst][name]), helpers.bits(b[inst][name])), (a["tick"], inst, name)
        assert np.array_equal(helpers.bits(b["OH.1"]["OH"]), helpers.bits(b["OH.2"]["OH"]))
    assert boosts == 12            # (AGCM.rc holds OH_DT, not OH.1_DT / OH.2_DT: either instance's alarm falls back to RUN_DT)


def test_reference_child_refuses_what_the_product_shell_refuses(tmp_path, small_model):
    """The error behaviour of the two children on the same bad inputs: an unknown OH_data_source, a
    wavelength GOCART2G does not have, a tropopause at or below 40 hPa with the static k range, a missing model file."""
    grid = (4, 3, 12)
    imports, lats, lons = tg.mock_imports(grid, "ONLINE_INST")
    state, out = tmp_path / "state.bin", tmp_path / "out.bin"
    tg.write_state_file(state, grid, imports, lats, lons)
    (tmp_path / "m.model").write_bytes(small_model.image.tobytes())
    base = dict(model_pattern=str(tmp_path / "m.model"), exports=[])
    low = dict(imports)
    low["TROPP"] = imports["TROPP"].copy()
    low["TROPP"][1, 1] = 3900.0
    tg.write_state_file(tmp_path / "low.bin", grid, low, lats, lons)
    cases = [("a", dict(source="ONLINE", **base), state, 1, "Invalid OH_data_source: ONLINE"),
             ("b", dict(source="ONLINE_INST", wavelength=532, **base), state, 1, "Did not find OH wavelength_for_scacoef"),
             ("c", dict(source="ONLINE_INST", model_pattern=str(tmp_path / "nope_%m2.model"), exports=[]), state, 2,
              "Run phase 1 failed"),
             ("d", dict(source="ONLINE_INST", once_per_day=True, **base), tmp_path / "low.bin", 1,
              "Minimum tropopause pressure is not low enough")]
    for tag, cfg, st, nticks, message in cases:
        tg.write_rundir(tmp_path / tag, **cfg)
        for exe in (REF_ORACLE, tg.DRIVER_ORACLE):
            r = tg.run_driver(exe, tmp_path / tag, st, out, nticks)
            assert r.returncode != 0 and message in r.stdout, (tag, exe, r.stdout[-1500:])


# ------------------------------------------------------------------------------------------------ on the MI355X

@pytest.mark.gpu
def test_reference_child_drives_the_gpu_through_the_c_abi(tmp_path, small_model):
    """The link-level drop-in with the REAL caller: the reference's unmodified OH child, parent and binding module
    linked against libohxgb.so instead of libxgboost run two model days on the MI355X - every
    XGDMatrixCreateFromMat / XGBoosterPredict of predict_OH_with_XGB (:347,356) is the HIP path - and give, bit for bit,
    what the same executable gives over the oracle: margins are bit-exact and 10.0**x is the host's in both."""
    out, *_ = run_both(tmp_path, small_model, REF_ORACLE, REF_HIP, grid=(5, 4, 24), source="ONLINE_AVG24", nticks=98,
                       once_per_day=True, spinup=True, run_dt=1800, oh_dt=3600, avg24_tick=40, ohscale=0.85,
                       ref_time="003000", beg="20240131 000000")
    (cpu, _), (gpu, _) = out["reference"], out["product"]
    for a, b in zip(cpu, gpu):
        for name in ["OH"] + [e for e, _ in EXPORTS]:
            assert np.array_equal(helpers.bits(a["OH"][name]), helpers.bits(b["OH"][name])), (a["tick"], name)
    assert np.abs(gpu[0]["OH"]["OH_boost"]).sum() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["two_days_avg24", "two_days_avg24_skip_tick_on_device", "precomputed_dynamic",
                                  "online_inst_dynamic"])
def test_product_shell_on_the_gpu_against_the_reference_child(tmp_path, small_model, case):
    """The product's shell on the MI355X (OHXBoosterRun1: feature kernels, k-slab, fused walk, mask, conversion) against
    the reference's own child on the same state: engineered features, NDWET, SZA and every other DIAG export bit for bit;
    OH_boost within 2 ulp and INTERNAL OH within 3 (the fused kernel's 10**x against the host's).  The two-day case has
    46 ticks that skip Boost: by default the shell does those on the rank's core (skip_tick: host, oh_post_process_host),
    in the second parametrisation through OHXOHPostProcess on the GPU."""
    if case.startswith("two_days_avg24"):
        kw = dict(grid=(5, 4, 24), source="ONLINE_AVG24", nticks=98, once_per_day=True, spinup=True, run_dt=1800,
                  oh_dt=3600, avg24_tick=40, ohscale=0.85, ref_time="003000", beg="20240131 000000", register=True,
                  skip_tick="device" if case.endswith("device") else None)
        boosts = 3
    else:
        kw = dict(grid=(6, 5, 30), source="PRECOMPUTED" if case.startswith("pre") else "ONLINE_INST", nticks=8, seed=11,
                  once_per_day=False, spinup=False, run_dt=1800, oh_dt=3600, avg24_tick=-1, ohscale=1.0,
                  ref_time="000000", beg="20240131 220000")
        boosts = 4
    out, *_ = run_both(tmp_path, small_model, REF_ORACLE, tg.DRIVER_HIP, **kw)
    (ref, _), (prod, _) = out["reference"], out["product"]
    assert compare(ref, prod, tolerance=POW10) == boosts


@pytest.mark.gpu
def test_reference_child_and_product_shell_on_a_block_the_ring_kernels_take(tmp_path, deep_model):
    """The same comparison at a size where the big-batch kernels do the work: a 96 x 72 x 72 block (497 664 gridcells; the
    slab ~350 000 rows), the OH booster's shape (100 trees of depth 18).  The reference's child reaches the MI355X through
    XGDMatrixCreateFromMat / XGBoosterPredict (predict_rows_ring_kernel behind the level-size search), the product's shell
    through OHXBoosterRun1 (feature kernels, slab count, fused walk); one Boost tick and one that skips it."""
    grid = (96, 72, 72)
    kw = dict(grid=grid, source="ONLINE_INST", nticks=3, seed=5, once_per_day=True, spinup=False, run_dt=1800, oh_dt=1800,
              avg24_tick=-1, ohscale=0.85, ref_time="003000", beg="20240131 000000")
    cpu_and_gpu, *_ = run_both(tmp_path, deep_model, REF_ORACLE, REF_HIP, **kw)
    (cpu, _), (gpu, _) = cpu_and_gpu["reference"], cpu_and_gpu["product"]
    for a, b in zip(cpu, gpu):                   # the reference's child: GPU against oracle, every field, bit for bit
        for name in ["OH"] + [e for e, _ in EXPORTS]:
            assert np.array_equal(helpers.bits(a["OH"][name]), helpers.bits(b["OH"][name])), (a["tick"], name)
    other = tmp_path / "shell"
    other.mkdir()
    out, *_ = run_both(other, deep_model, REF_HIP, tg.DRIVER_HIP, **kw)
    (ref, _), (prod, _) = out["reference"], out["product"]
    assert compare(ref, prod, tolerance=POW10) == 1
    assert np.count_nonzero(prod[0]["OH"]["OH_boost"]) > 200_000


@pytest.mark.gpu
def test_a_rank_s_oh_tick_end_to_end_reference_child_and_product_shell(tmp_path, deep_model):
    """A GEOS rank's block (48 x 24 x 72, NOTES.wiki's rank size) with HISTORY asking for no DIAG export, the way
    tools/rank_tick_end_to_end.py times it: the reference's own child (its five xgboost calls into libohxgb.so) and the
    product's shell (one OHXBoosterRun1 per Boost tick, arrays registered; the fused host pass on the ticks that skip
    Boost) over one model day and an hour of hourly ticks under compute_once_per_day - 2 Boost ticks, 23 that skip it.
    INTERNAL OH of every tick within the 10**x tolerance.  No duration is asserted here: the clock belongs to the tool."""
    kw = dict(grid=(48, 24, 72), source="ONLINE_INST", nticks=26, seed=21, once_per_day=True, spinup=False, run_dt=3600,
              oh_dt=3600, avg24_tick=-1, ohscale=0.85, ref_time="000000", beg="20240131 000000", exports=[])
    out, *_ = run_both(tmp_path, deep_model, REF_HIP, tg.DRIVER_HIP, **kw)
    (ref, _), (prod, _) = out["reference"], out["product"]
    assert compare(ref, prod, tolerance=POW10, exports=[]) == 2
    assert [t["tick"] for t in prod if t["OH"]["boosted"]] == [0, 24]


@pytest.mark.gpu
def test_two_ranks_on_the_gpu_reference_child_and_product_shell(tmp_path, deep_model):
    """Two ranks - two processes, each with a 48 x 24 x 72 block - sharing the one GPU (NOTES.wiki:14,33: a rank per core;
    tools/rank_tick_end_to_end.py runs up to six and keeps the clock): both ranks of the reference child and both of the
    product shell, started together, give what a rank alone gives - the library's streams, its registered arrays and its
    ring kernel are per process and do not see each other."""
    import subprocess
    grid, nticks, P = (48, 24, 72), 6, 2
    imports, lats, lons = tg.mock_imports(grid, "ONLINE_INST", seed=21)
    (tmp_path / "oh_M01.model").write_bytes(deep_model.image.tobytes())
    rundir, state = tmp_path / "run", tmp_path / "state.bin"
    tg.write_rundir(rundir, source="ONLINE_INST", model_pattern=str(tmp_path / "oh_M01.model"), policy="reference", exports=[],
                    once_per_day=False, spinup=False, run_dt=1800, oh_dt=1800, avg24_tick=-1, ohscale=0.85,
                    ref_time="000000", beg="20240131 000000")
    tg.write_state_file(state, grid, imports, lats, lons)
    ticks = {}
    for tag, exe in (("reference_child", REF_HIP), ("product_shell", tg.DRIVER_HIP)):
        procs = [subprocess.Popen([exe, str(rundir), str(state), str(tmp_path / f"{tag}_{r}.bin"), str(nticks)],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(P)]
        for p in procs:
            out, _ = p.communicate(timeout=600)
            assert p.returncode == 0, (tag, out[-2000:])
        ranks = [tg.parse_output(tmp_path / f"{tag}_{r}.bin", grid, [("OH", False)], []) for r in range(P)]
        for a, b in zip(*ranks):                                       # the same state on both ranks: the same bits
            assert np.array_equal(helpers.bits(a["OH"]["OH"]), helpers.bits(b["OH"]["OH"])), (tag, a["tick"])
        ticks[tag] = ranks[0]
    assert compare(ticks["reference_child"], ticks["product_shell"], tolerance=POW10, exports=[]) == nticks
