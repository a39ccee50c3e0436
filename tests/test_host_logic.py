"""Host-side logic around the kernels (CPU): slab selection, gather order, synthetic inputs,
and BASELINE.json config #1 through the Fortran host linked against the oracle."""
import os

import numpy as np
import pytest

from oracle import xgb_oracle as O
from quickchem_amd import oh_predict, synth
from tests import helpers


def test_k_slab_matches_oracle_and_bruteforce():
    grid = synth.GRIDS["C12"]
    pl, tropp, _ = helpers.synth_state(grid)
    for dynamic in (True, False):
        k1, k2 = oh_predict.k_slab(pl, tropp, dynamic, 4000.0)
        assert (k1, k2) == O.k_slab(pl, tropp, dynamic, 4000.0)
        best = 0
        for i in range(0, grid[0], 3):
            for j in range(0, grid[1], 5):
                lim = tropp[i, j] if dynamic else 4000.0
                best = max(best, int(np.count_nonzero(pl[i, j, :] > lim)))
        assert k2 == grid[2] and grid[2] - k1 + 1 >= best


def test_static_slab_asserts_on_low_tropopause():
    grid = synth.GRIDS["mock4x4"]
    pl, tropp, _ = helpers.synth_state(grid)
    tropp = tropp.copy()
    tropp[1, 2] = 3999.0
    with pytest.raises(oh_predict.AssertFailure, match="Minimum tropopause pressure"):
        oh_predict.k_slab(pl, tropp, False, 4000.0)
    with pytest.raises(AssertionError, match="Minimum tropopause pressure"):
        O.k_slab(pl, tropp, False, 4000.0)


def test_fields_and_rows_generators_agree():
    """rows[m][f] with m = i + im*(j + jm*k) equals field f at (i,j,k); PL rows are hPa = Pa/100."""
    grid = synth.GRIDS["mock4x4"]
    im, jm, km = grid
    rows = synth.rows_cpu(grid, 0, im * jm * km).reshape(km, jm, im, 27)
    pl, tropp, fields = helpers.synth_state(grid)
    for f in range(27):
        a = fields[f]
        if f == 1:
            continue
        want = a.T[None, :, :] if a.ndim == 2 else np.transpose(a, (2, 1, 0))
        assert np.array_equal(np.broadcast_to(want, (km, jm, im)), rows[..., f]), synth.FEATURE_NAMES[f]
    assert np.allclose(np.transpose(pl, (2, 1, 0)) / 100.0, rows[..., 1], rtol=1e-6)
    assert tropp.shape == (im, jm) and tropp.min() > 4000.0


def test_synthetic_features_look_like_the_survey_recipe():
    grid = synth.GRIDS["C12"]
    n = grid[0] * grid[1] * grid[2]
    rows = synth.rows_cpu(grid, 0, n)
    assert np.isfinite(rows).all()
    names = synth.FEATURE_NAMES
    col = {nm: rows[:, i] for i, nm in enumerate(names)}
    assert -90 <= col["LAT"].min() and col["LAT"].max() <= 90
    assert 0 < col["PL"].min() < 1 and 400 < col["PL"].max() < 1045
    assert 179 < col["T"].min() and col["T"].max() < 321
    assert 0 <= col["CLOUD"].min() and col["CLOUD"].max() <= 1 and (col["CLOUD"] == 0).mean() > 0.02
    assert 0 <= col["SZA"].min() and col["SZA"].max() <= 113
    # cumulative optical depths are monotone in k and hit exact zeros (ties for x < cond)
    cube = rows.reshape(grid[2], grid[1], grid[0], 27)
    for nm, sign in (("TAUCLWDN", -1), ("TAUCLIDN", -1), ("TAUCLIUP", 1), ("TAUCLWUP", 1), ("AODUP", 1), ("AODDN", -1)):
        d = np.diff(cube[..., names.index(nm)], axis=0) * sign
        assert (d >= 0).all(), nm
    assert (col["TAUCLWUP"] == 0).mean() > 0.2
    # shards are reproducible without communication
    part = synth.rows_cpu(grid, 12345, 777)
    assert np.array_equal(part, rows[12345:12345 + 777])


def test_python_mirror_gather_matches_oracle_gather():
    grid = synth.GRIDS["mock4x4"]
    pl, tropp, fields = helpers.synth_state(grid)
    k1, k2 = oh_predict.k_slab(pl, tropp, True, 4000.0)
    rows = O.gather_rows(fields, k1, k2)
    im, jm, km = grid
    full = synth.rows_cpu(grid, 0, im * jm * km)
    sl = full[(k1 - 1) * im * jm:]
    assert rows.shape == sl.shape
    keep = [f for f in range(27) if f != 1]
    assert np.array_equal(rows[:, keep], sl[:, keep])


@pytest.mark.parametrize("mode", ["compat", "fused"])
@pytest.mark.parametrize("dynamic", [True, False])
def test_config1_fortran_host_cpu_plumbing(tmp_path, small_model, mode, dynamic):
    """BASELINE.json config #1: 4x4x72 mock state through the Fortran predict_OH_with_XGB,
    linked against the CPU oracle (no GPU), against the oracle's own restatement."""
    grid = synth.GRIDS["mock4x4"]
    pl, tropp, fields = helpers.synth_state(grid)
    state, model, out = tmp_path / "state.bin", tmp_path / "oh.model", tmp_path / "out.bin"
    helpers.write_state_file(state, pl, tropp, fields, dynamic, ohscale=0.85)
    model.write_bytes(small_model.image.tobytes())
    r = helpers.run_driver(helpers.DRIVER_ORACLE, state, model, out, mode)
    assert r.returncode == 0, r.stdout
    rc, k1, k2, oh, _ = helpers.read_driver_output(out, *grid)
    oh_ref, _, k1r, k2r = helpers.oracle_predict_oh(small_model.image, pl, tropp, fields, dynamic)
    assert rc == 0 and (k1, k2) == (k1r, k2r)
    want = (oh_ref * np.float32(0.85)).astype(np.float32)
    assert np.array_equal(helpers.bits(oh), helpers.bits(want))
    assert np.all(oh[:, :, :k1 - 1] == 0) and np.all(oh[:, :, k1 - 1:] > 0)


def test_fortran_host_reports_a_bad_model_file(tmp_path):
    grid = synth.GRIDS["mock4x4"]
    pl, tropp, fields = helpers.synth_state(grid)
    state, out = tmp_path / "state.bin", tmp_path / "out.bin"
    helpers.write_state_file(state, pl, tropp, fields, True)
    r = helpers.run_driver(helpers.DRIVER_ORACLE, state, tmp_path / "missing.model", out)
    assert r.returncode != 0 and "XGBoosterLoadModel_f" in r.stdout


def test_reference_binding_module_drives_the_oracle(tmp_path, small_model):
    """The reference's unmodified xgb_fortran_api module (oracle/_ref) making the reference's call
    sequence against the oracle library."""
    if not os.path.exists(helpers.DROPIN_ORACLE):
        pytest.skip("oracle/_ref not built")
    import struct
    rows = synth.rows_cpu(synth.GRIDS["mock4x4"], 0, 1152)
    rf, mf, pf = tmp_path / "rows.bin", tmp_path / "oh.model", tmp_path / "pred.bin"
    with open(rf, "wb") as f:
        f.write(struct.pack("<qq", rows.shape[0], rows.shape[1]))
        f.write(rows.tobytes())
    mf.write_bytes(small_model.image.tobytes())
    r = helpers.run_driver(helpers.DROPIN_ORACLE, rf, mf, pf)
    assert r.returncode == 0, r.stdout
    raw = pf.read_bytes()
    n, = struct.unpack_from("<q", raw, 0)
    pred = np.frombuffer(raw, dtype="<f4", count=n, offset=8)
    want = helpers.oracle_predict(small_model.image, rows, synth.XX_MISS)
    assert np.array_equal(helpers.bits(pred), helpers.bits(want))


def _month_models(tmp_path, small_model):
    other = synth.make_model(num_trees=20, max_depth=10, sample_log2=15, min_leaf=4, grid=synth.GRIDS["C12"],
                             model_seed=77)
    (tmp_path / "oh_M01.model").write_bytes(small_model.image.tobytes())
    (tmp_path / "oh_M02.model").write_bytes(other.image.tobytes())
    return tmp_path / "oh_M%m2.model", other


@pytest.mark.parametrize("mode", ["compat", "fused"])
def test_month_roll_over_policies_fortran_host(tmp_path, small_model, mode):
    """XGBoostFile is month-templated (OH_instance_OH.rc:20) but the reference keeps the booster of its
    first call for good (first_time, OH_GridCompMod.F90:209,269).  The Fortran host does the same by
    default and, under OH_XGB_POLICY_BY_NAME, keeps one resident booster per file name instead."""
    grid = synth.GRIDS["mock4x4"]
    pl, tropp, fields = helpers.synth_state(grid)
    state, out = tmp_path / "state.bin", tmp_path / "out.bin"
    helpers.write_state_file(state, pl, tropp, fields, True, ohscale=1.0)
    pattern, other = _month_models(tmp_path, small_model)
    jan, _, _, _ = helpers.oracle_predict_oh(small_model.image, pl, tropp, fields, True)
    feb, _, _, _ = helpers.oracle_predict_oh(other.image, pl, tropp, fields, True)
    assert not np.array_equal(jan, feb)
    for policy, want, resident in (("reference", jan, 0), ("by_name", feb, 2)):
        r = helpers.run_driver(helpers.DRIVER_ORACLE, state, pattern, out, mode, 2, policy)   # calls dated Jan, Feb
        assert r.returncode == 0, r.stdout
        rc, _, _, oh, _ = helpers.read_driver_output(out, *grid)
        tail = np.frombuffer(open(out, "rb").read()[-4:], dtype="<i4")[0]
        assert rc == 0 and tail == resident
        if mode == "compat":
            assert np.array_equal(helpers.bits(oh), helpers.bits(want)), policy
        else:
            assert helpers.ulp_diff(oh, want).max() <= 2, policy


def test_month_roll_over_policies_python_host(tmp_path, small_model, oracle_lib):
    grid = synth.GRIDS["mock4x4"]
    pl, tropp, fields = helpers.synth_state(grid)
    pattern, other = _month_models(tmp_path, small_model)
    assert oh_predict.fill_grads_template(str(pattern), 20240215, 123000).endswith("oh_M02.model")
    assert oh_predict.fill_grads_template("a_%y4%m2%d2_%h2%n2.bin", 20231109, 63000) == "a_20231109_0630.bin"
    jan, _, _, _ = helpers.oracle_predict_oh(small_model.image, pl, tropp, fields, True)
    feb, _, _, _ = helpers.oracle_predict_oh(other.image, pl, tropp, fields, True)
    for policy, want, resident in (("reference", jan, 1), ("by_name", feb, 2)):
        p = oh_predict.OHPredictor(lib=oracle_lib, model_policy=policy)
        oh = np.zeros(grid, dtype=np.float32)
        for nymd in (20240115, 20240215):
            oh[:] = 0
            p.predict_OH_with_XGB(oh_predict.fill_grads_template(str(pattern), nymd), *grid, True, 4000.0, pl, tropp,
                                  oh_predict.OHBoostInputData(fields), oh)
        assert helpers.ulp_diff(oh, want).max() <= 2, policy          # numpy's 10**x against the oracle's powf
        other_month = feb if want is jan else jan
        assert helpers.ulp_diff(oh, other_month).max() > 1000
        assert len(p.boosters) == resident
    with pytest.raises(ValueError):
        oh_predict.OHPredictor(model_policy="sometimes")


def test_python_side_knows_the_librarys_ring_rounds_default():
    """bench.py turns kernel time into a per-launch duration with the number of launches a step makes; for the ring
    kernels that follows from LaunchTuning::ring_rounds, which the Python side carries as capi.RING_ROUNDS_DEFAULT."""
    import re
    from quickchem_amd import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hpp = open(os.path.join(root, "quickchem_amd", "csrc", "kernels.hpp")).read()
    assert int(re.search(r"int ring_rounds = (\d+);", hpp).group(1)) == capi.RING_ROUNDS_DEFAULT
    assert int(re.search(r"constexpr int kRingRoundsNoGrid = (\d+);", hpp).group(1)) == capi.RING_ROUNDS_NO_GRID
    assert int(re.search(r"constexpr int kRingRoundsPermuted = (\d+);", hpp).group(1)) == capi.RING_ROUNDS_PERMUTED
    header = open(os.path.join(root, "include", "ohxgb.h")).read()
    assert f'"ohx_ring_rounds" ring kernels: tiles per wavefront and launch (default {capi.RING_ROUNDS_DEFAULT};' in header


def test_bench_reads_the_cpu_share_a_cgroup_grants(tmp_path):
    """(r6) bench.py's cpu_baseline sweeps threads and says why it peaks where it does: the box's cgroup quota.  cgroup v2
    (cpu.max at the process's own group, or at the root), v1 (cfs quota / period), "max" and nothing at all."""
    import bench

    def tree(files):
        root = tmp_path / f"r{len(list(tmp_path.iterdir()))}"
        for rel, text in files.items():
            p = root / rel
            p.parent.mkdir(parents=True, exist_ok=True)
            p.write_text(text)
        root.mkdir(exist_ok=True)
        return str(root)
    assert bench.cpu_quota(tree({"proc/self/cgroup": "0::/process_api/abc\n", "sys/fs/cgroup/process_api/abc/cpu.max": "1600000 100000\n"})) == 16.0
    assert bench.cpu_quota(tree({"proc/self/cgroup": "0::/\n", "sys/fs/cgroup/cpu.max": "250000 100000\n"})) == 2.5
    assert bench.cpu_quota(tree({"proc/self/cgroup": "0::/\n", "sys/fs/cgroup/cpu.max": "max 100000\n"})) is None
    assert bench.cpu_quota(tree({"sys/fs/cgroup/cpu/cpu.cfs_quota_us": "800000\n", "sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"})) == 8.0
    assert bench.cpu_quota(tree({"sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1\n", "sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"})) is None
    assert bench.cpu_quota(tree({})) is None


def test_rank_tick_tool_sorts_ticks_into_boost_and_skip():
    """(r6) tools/rank_tick_end_to_end.py: the driver's TICK_US lines -> (tick, us, nhms); quantiles of a list."""
    import sys
    sys.path.insert(0, os.path.join(helpers.ROOT, "tools"))
    import rank_tick_end_to_end as rt
    log = "noise\nTICK_US 0 600000.5 0\nTICK_US 1 73.2 10000\n OH is *NOT*\nTICK_US 24 471.0 0\n"
    assert rt.parse_ticks(log) == [(0, 600000.5, 0), (1, 73.2, 10000), (24, 471.0, 0)]
    q = rt.quantiles([5.0, 1.0, 3.0, 2.0, 4.0])
    assert (q["n"], q["median"], q["max"], q["p10"]) == (5, 3.0, 5.0, 1.0) and abs(q["mean"] - 3.0) < 1e-12
    assert rt.quantiles([]) is None
