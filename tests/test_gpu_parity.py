"""Parity of the HIP path against the CPU oracle, through the C ABI (GPU).

Bit-exact on the raw margin (xx_pred) everywhere; `10**pred` within 2 ulp of the
oracle's powf (10.0**x is libm-specific)."""
import json
import os
import struct

import numpy as np
import pytest

from quickchem_amd import capi, oh_predict, synth
from tests import helpers

pytestmark = pytest.mark.gpu

KERNELS = ["wide", "packed1", "packed2", "packed4", "super1", "super2", "super3", "super4", "ring"]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


def gpu_predict(image, rows, missing, kernel="auto", option_mask=0, ntree_limit=0, params=None, grid=None):
    b = capi.Booster(model_buffer=image)
    b.set_param("ohx_kernel", kernel)
    for k, v in (params or {}).items():
        b.set_param(k, v)
    d = capi.DMatrix(rows, missing=missing)
    if grid is not None:
        d.set_grid(*grid)
    out = b.predict(d, option_mask=option_mask, ntree_limit=ntree_limit)
    d.free()
    b.free()
    return out


def with_missing(rows, rate, seed=3):
    rows = rows.copy()
    rng = np.random.default_rng(seed)
    mask = rng.random(rows.shape) < rate
    rows[mask] = np.where(rng.random(int(mask.sum())) < 0.5, np.float32(synth.XX_MISS), np.float32(np.nan))
    return rows


@pytest.mark.parametrize("kernel", KERNELS)
def test_hand_forest_golden(torch_cuda, kernel):
    cases, rows = helpers.load_hand_cases()
    image = open(os.path.join(helpers.GOLDEN, "hand_forest.json"), "rb").read()
    miss = cases["missing"]
    assert np.array_equal(gpu_predict(image, rows, miss, kernel), np.float32(cases["margin"]))
    assert np.array_equal(gpu_predict(image, rows, miss, kernel, ntree_limit=2), np.float32(cases["margin_ntree_limit_2"]))
    assert np.array_equal(gpu_predict(image, rows, miss, kernel, ntree_limit=3), np.float32(cases["margin_ntree_limit_3"]))
    assert np.array_equal(gpu_predict(image, rows, miss, kernel, option_mask=1), np.float32(cases["margin"]))
    leaves = gpu_predict(image, rows, miss, kernel, option_mask=16).reshape(len(rows), -1)
    assert np.array_equal(leaves, np.float32(cases["leaf_index"]))
    assert np.array_equal(gpu_predict(image, np.float32(cases["rows_2col"]), miss, kernel), np.float32(cases["margin_2col"]))


@pytest.mark.parametrize("kernel", KERNELS)
def test_config1_golden(torch_cuda, deep_model, kernel):
    g = json.load(open(os.path.join(helpers.GOLDEN, "mock4x4_T100.json")))
    grid = tuple(g["grid"])
    rows = synth.rows_cpu(grid, 0, grid[0] * grid[1] * grid[2])
    got = gpu_predict(deep_model.image, rows, synth.XX_MISS, kernel)
    assert np.array_equal(helpers.bits(got), np.array(g["margin_bits"], dtype=np.uint32))


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("nrows", [1, 63, 64, 65, 1000, 62208])
def test_rows_vs_oracle_ragged_sizes(torch_cuda, small_model, kernel, nrows):
    rows = synth.rows_cpu(synth.GRIDS["C12"], 0, nrows)
    want = helpers.oracle_predict(small_model.image, rows, synth.XX_MISS)
    got = gpu_predict(small_model.image, rows, synth.XX_MISS, kernel)
    assert np.array_equal(helpers.bits(got), helpers.bits(want))


def test_empty_matrix(torch_cuda, small_model):
    got = gpu_predict(small_model.image, np.zeros((0, 27), dtype=np.float32), synth.XX_MISS)
    assert got.shape == (0,)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("missing", [synth.XX_MISS, float("nan")])
def test_missing_values_vs_oracle(torch_cuda, deep_model, kernel, missing):
    rows = with_missing(synth.rows_cpu(synth.GRIDS["C12"], 5000, 20000), 0.01)
    want = helpers.oracle_predict(deep_model.image, rows, missing)
    got = gpu_predict(deep_model.image, rows, missing, kernel)
    assert np.array_equal(helpers.bits(got), helpers.bits(want))


@pytest.mark.parametrize("kernel", ["packed2", "super2", "super4"])
@pytest.mark.parametrize("rate", [1e-4, 3e-3, 0.05])
def test_rows_with_missing_values_left_to_the_second_launch(torch_cuda, deep_model, kernel, rate):
    """ohx_defer_missing: a wave some of whose rows hold missing values lists those rows for a second, small launch
    and walks the tile without missing-value logic.  Forced on for a batch below the size at which it is the
    default; at 5 % of the entries (three rows in four) the list overflows and the later waves walk missing-aware
    as before.  With the grid hint (bricks, rows fetched together), with 64 consecutive rows per wave, and with a
    first row inside a level."""
    grid = (96, 72, 72)
    rows = with_missing(synth.rows_cpu(grid, 1234, 96 * 72 * 6 + 777), rate, seed=11)
    want = helpers.oracle_predict(deep_model.image, rows, synth.XX_MISS)
    assert np.isfinite(want).all()
    for hint in ((96, 72, 1234), (0, 0, 0)):
        got = gpu_predict(deep_model.image, rows, synth.XX_MISS, kernel, params={"ohx_defer_missing": "on"}, grid=hint)
        assert np.array_equal(helpers.bits(got), helpers.bits(want)), hint
    off = gpu_predict(deep_model.image, rows, synth.XX_MISS, kernel, params={"ohx_defer_missing": "off"}, grid=(96, 72, 1234))
    assert np.array_equal(helpers.bits(off), helpers.bits(want))


def test_deferred_rows_by_default_on_a_big_batch(torch_cuda, small_model):
    """From 262 144 rows on the second launch is the default; NaN as the missing marker."""
    grid = (96, 72, 72)
    rows = with_missing(synth.rows_cpu(grid, 0, 96 * 72 * 40), 2e-4, seed=5)
    rows[rows == np.float32(synth.XX_MISS)] = np.nan
    want = helpers.oracle_predict(small_model.image, rows, float("nan"))
    got = gpu_predict(small_model.image, rows, float("nan"), "super2", grid=(96, 72, 0))
    assert np.array_equal(helpers.bits(got), helpers.bits(want))


@pytest.mark.parametrize("kernel", ["super1", "super2", "super4"])
def test_small_batches_with_their_trees_split_over_waves(torch_cuda, deep_model, kernel):
    """ohx_tree_split: a batch that leaves the chip's wave slots mostly empty has its trees cut into runs walked by
    different waves; a second launch sums the leaves in tree order, so the margins are the sequential sum's, bit for
    bit - whatever the number of runs, with a tree limit that does not divide by it, with missing values, with and
    without a grid hint (bricks, or 64 consecutive rows when the bricks of the hint would be mostly empty)."""
    grid = (96, 72, 72)
    rows = with_missing(synth.rows_cpu(grid, 500, 96 * 72 * 3 + 333), 0.003, seed=21)
    for lim in (0, 37):
        want = helpers.oracle_predict(deep_model.image, rows, synth.XX_MISS, ntree_limit=lim)
        for split in ("off", "auto", "2", "3", "7", "10"):
            for hint in ((96, 72, 500), (0, 0, 0), (360, 2160, 500)):
                got = gpu_predict(deep_model.image, rows, synth.XX_MISS, kernel, ntree_limit=lim,
                                  params={"ohx_tree_split": split}, grid=hint)
                assert np.array_equal(helpers.bits(got), helpers.bits(want)), (lim, split, hint)


@pytest.mark.parametrize("kernel", ["wide", "packed4", "super2"])
def test_ntree_limit_and_leaf_indices(torch_cuda, small_model, kernel):
    rows = with_missing(synth.rows_cpu(synth.GRIDS["C12"], 0, 3000), 0.005)
    for lim in (1, 7, 19, 20, 500):
        want = helpers.oracle_predict(small_model.image, rows, synth.XX_MISS, ntree_limit=lim)
        got = gpu_predict(small_model.image, rows, synth.XX_MISS, kernel, ntree_limit=lim)
        assert np.array_equal(helpers.bits(got), helpers.bits(want)), lim
    want = helpers.oracle_predict(small_model.image, rows, synth.XX_MISS, option_mask=16)
    got = gpu_predict(small_model.image, rows, synth.XX_MISS, kernel, option_mask=16)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("params", [{"ohx_line_slots": 0}, {"ohx_top_levels": 1}, {"ohx_top_levels": 4, "ohx_line_slots": 32},
                                    {"ohx_top_levels": 12, "ohx_min_chunk": 2}])
def test_layout_does_not_change_results(torch_cuda, deep_model, params):
    rows = synth.rows_cpu(synth.GRIDS["C12"], 0, 8192)
    want = helpers.oracle_predict(deep_model.image, rows, synth.XX_MISS)
    got = gpu_predict(deep_model.image, rows, synth.XX_MISS, "packed4", params=params)
    assert np.array_equal(helpers.bits(got), helpers.bits(want))
    got = gpu_predict(deep_model.image, rows, synth.XX_MISS, "super2",
                      params={"ohx_launches_per_residency": 0, "ohx_xcd_remap": 0})
    assert np.array_equal(helpers.bits(got), helpers.bits(want))


def test_fewer_columns_than_features(torch_cuda, small_model):
    rows = synth.rows_cpu(synth.GRIDS["C12"], 0, 4096)[:, :20].copy()
    want = helpers.oracle_predict(small_model.image, rows, synth.XX_MISS)
    for kernel in ("wide", "packed4", "super2"):
        got = gpu_predict(small_model.image, rows, synth.XX_MISS, kernel)
        assert np.array_equal(helpers.bits(got), helpers.bits(want))


def test_error_paths(torch_cuda, small_model):
    rows = synth.rows_cpu(synth.GRIDS["C12"], 0, 256)
    b = capi.Booster(model_buffer=small_model.image)
    # more columns than the booster has features
    wide_rows = np.concatenate([rows, rows[:, :3]], axis=1)
    with pytest.raises(capi.OhxError, match="Number of columns"):
        b.predict(capi.DMatrix(wide_rows, missing=synth.XX_MISS))
    # inf in the data is refused at matrix creation, as xgboost 1.6.0 does
    bad = rows.copy()
    bad[17, 4] = -np.inf
    with pytest.raises(capi.OhxError, match="inf"):
        capi.DMatrix(bad, missing=synth.XX_MISS)
    # ... unless missing itself is inf
    capi.DMatrix(bad, missing=float("inf")).free()
    # unsupported outputs
    d = capi.DMatrix(rows, missing=synth.XX_MISS)
    with pytest.raises(capi.OhxError, match="option_mask"):
        b.predict(d, option_mask=4)
    # a booster without a model
    with pytest.raises(capi.OhxError, match="no model"):
        capi.Booster().predict(d)
    # device-resident path: inf is caught by the kernel and surfaced by check()
    t = torch_cuda.from_numpy(bad).cuda()
    dd = capi.DMatrix(device_ptr=t.data_ptr(), nrow=bad.shape[0], ncol=27, missing=synth.XX_MISS)
    out = torch_cuda.empty(bad.shape[0], dtype=torch_cuda.float32, device="cuda")
    b.predict_device(dd, out.data_ptr())
    with pytest.raises(capi.OhxError, match="inf"):
        b.check()
    b.check()                                   # the flag is cleared once reported


def test_dmatrix_file_round_trip(torch_cuda, tmp_path, small_model):
    rows = synth.rows_cpu(synth.GRIDS["C12"], 0, 500)
    d = capi.DMatrix(rows, missing=synth.XX_MISS)
    assert (d.num_row, d.num_col) == (500, 27)
    p = str(tmp_path / "rows.ohxdmat")
    d.save_binary(p)
    d2 = capi.DMatrix.from_file(p)
    b = capi.Booster(model_buffer=small_model.image)
    assert np.array_equal(b.predict(d), b.predict(d2))
    csv = tmp_path / "rows.csv"
    np.savetxt(csv, rows[:50], delimiter=",", fmt="%.9g")
    d3 = capi.DMatrix.from_file(str(csv))
    assert np.array_equal(b.predict(d3), b.predict(capi.DMatrix(rows[:50], missing=float("nan"))))


def test_device_generator_matches_host_generator(torch_cuda):
    torch = torch_cuda
    grid = synth.GRIDS["C12"]
    n = grid[0] * grid[1] * grid[2]
    dev = torch.empty((n, 27), dtype=torch.float32, device="cuda")
    synth.rows_device(grid, 0, n, dev)
    torch.cuda.synchronize()
    host = synth.rows_cpu(grid, 0, n)
    got = dev.cpu().numpy()
    for f in range(27):
        assert np.array_equal(helpers.bits(got[:, f]), helpers.bits(host[:, f])), synth.FEATURE_NAMES[f]
    # a shard generated on its own equals the same rows of the whole
    part = torch.empty((1000, 27), dtype=torch.float32, device="cuda")
    synth.rows_device(grid, 40000, 1000, part)
    assert np.array_equal(part.cpu().numpy(), host[40000:41000])
    for f in (-1, 0, 1, 15, 26):
        two_d = f < 0 or synth.IS2D[f]
        t = torch.empty(grid[0] * grid[1] * (1 if two_d else grid[2]), dtype=torch.float32, device="cuda")
        synth.field_device(grid, f, t)
        want = helpers.fortran_flat(synth.field_cpu(grid, f)).ravel()
        assert np.array_equal(helpers.bits(t.cpu().numpy()), helpers.bits(want)), f


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dynamic", [True, False])
def test_fused_fields_vs_oracle(torch_cuda, deep_model, kernel, dynamic):
    """OHXBoosterPredictFields against the oracle's restatement of predict_OH_with_XGB's RUN section."""
    grid = synth.GRIDS["C12"]
    pl, tropp, fields = helpers.synth_state(grid)
    oh_ref, margin_ref, k1, k2 = helpers.oracle_predict_oh(deep_model.image, pl, tropp, fields, dynamic)
    p = oh_predict.OHPredictor()
    p.xx_bst = capi.Booster(model_buffer=deep_model.image)
    p.xx_bst.set_param("ohx_kernel", kernel)
    p.first_time = False
    oh = np.zeros(grid, dtype=np.float32)
    margins = []
    rc = p.predict_OH_with_XGB("unused", *grid, dynamic, 4000.0, pl, tropp, oh_predict.OHBoostInputData(fields), oh,
                               mode="fused", margin_out=margins)
    assert rc == 0
    assert np.array_equal(helpers.bits(margins[0]), helpers.bits(margin_ref))          # bit-exact traversal + sum
    assert np.all(oh[:, :, :k1 - 1] == 0)
    assert helpers.ulp_diff(oh[:, :, k1 - 1:], oh_ref[:, :, k1 - 1:]).max() <= 2       # 10**x, tolerance 2 ulp


@pytest.mark.parametrize("rate", [2e-4, 0.02])
def test_fused_fields_with_missing_values_left_to_the_second_launch(torch_cuda, deep_model, rate):
    """The fused call with -999.0 and NaN in its fields and ohx_defer_missing on (forced: the slab is smaller than the
    size from which it is the default): rows with a missing value are listed and predicted by the second launch of the
    fields kernel through the list; at 2 % of the entries the list overflows.  Margins bit for bit."""
    grid = synth.GRIDS["C12"]
    pl, tropp, fields = helpers.synth_state(grid)
    rng = np.random.default_rng(17)
    fields = [f.copy() for f in fields]
    for f in fields[2:]:                                  # LAT and PL stay (the slab is made from PL)
        mask = rng.random(f.shape) < rate
        f[mask] = np.where(rng.random(int(mask.sum())) < 0.5, np.float32(synth.XX_MISS), np.float32(np.nan))
    oh_ref, margin_ref, k1, k2 = helpers.oracle_predict_oh(deep_model.image, pl, tropp, fields, True)
    for setting in ("on", "off"):
        p = oh_predict.OHPredictor()
        p.xx_bst = capi.Booster(model_buffer=deep_model.image)
        p.xx_bst.set_param("ohx_defer_missing", setting)
        p.first_time = False
        oh = np.zeros(grid, dtype=np.float32)
        margins = []
        assert p.predict_OH_with_XGB("unused", *grid, True, 4000.0, pl, tropp, oh_predict.OHBoostInputData(fields), oh,
                                     mode="fused", margin_out=margins) == 0
        assert np.array_equal(helpers.bits(margins[0]), helpers.bits(margin_ref)), setting
        assert helpers.ulp_diff(oh[:, :, k1 - 1:], oh_ref[:, :, k1 - 1:]).max() <= 2


@pytest.mark.parametrize("mode", ["compat", "fused"])
def test_python_mirror_loads_model_file_once(torch_cuda, tmp_path, small_model, mode):
    """predict_OH_with_XGB mirror end to end, incl. ONE_TIME_SETUP and the SAVE'd booster (:242-271)."""
    grid = synth.GRIDS["mock4x4"]
    pl, tropp, fields = helpers.synth_state(grid)
    mf = tmp_path / "OH_model.bin"
    mf.write_bytes(small_model.image.tobytes())
    p = oh_predict.OHPredictor()
    oh = np.zeros(grid, dtype=np.float32)
    margins = []
    assert p.predict_OH_with_XGB(str(mf) + "   ", *grid, True, 4000.0, pl, tropp, oh_predict.OHBoostInputData(fields), oh,
                                 mode=mode, margin_out=margins) == 0
    oh_ref, margin_ref, k1, k2 = helpers.oracle_predict_oh(small_model.image, pl, tropp, fields, True)
    assert np.array_equal(helpers.bits(margins[0]), helpers.bits(margin_ref))
    assert helpers.ulp_diff(oh[:, :, k1 - 1:], oh_ref[:, :, k1 - 1:]).max() <= 2
    # later calls ignore the file name, as the reference does (first_time, :209,269)
    oh2 = np.zeros(grid, dtype=np.float32)
    assert p.predict_OH_with_XGB("/nonexistent", *grid, True, 4000.0, pl, tropp, oh_predict.OHBoostInputData(fields), oh2,
                                 mode=mode) == 0
    assert np.array_equal(oh, oh2)


@pytest.mark.parametrize("mode", ["compat", "fused"])
def test_fortran_host_on_the_gpu(torch_cuda, tmp_path, small_model, mode):
    """The Fortran predict_OH_with_XGB linked against libohxgb.so: same state, same answers as the
    same Fortran linked against the oracle (bit-exact OH_ML in compat mode: both use flang's 10.0**x)."""
    grid = synth.GRIDS["C12"]
    pl, tropp, fields = helpers.synth_state(grid)
    state, model = tmp_path / "state.bin", tmp_path / "oh.model"
    helpers.write_state_file(state, pl, tropp, fields, True, ohscale=0.85)
    model.write_bytes(small_model.image.tobytes())
    out_g, out_c = tmp_path / "gpu.bin", tmp_path / "cpu.bin"
    r = helpers.run_driver(helpers.DRIVER_HIP, state, model, out_g, mode)
    assert r.returncode == 0, r.stdout
    r = helpers.run_driver(helpers.DRIVER_ORACLE, state, model, out_c, "compat")
    assert r.returncode == 0, r.stdout
    rc, k1, k2, oh_g, _ = helpers.read_driver_output(out_g, *grid)
    rc2, k1c, k2c, oh_c, _ = helpers.read_driver_output(out_c, *grid)
    assert rc == 0 and rc2 == 0 and (k1, k2) == (k1c, k2c)
    if mode == "compat":
        assert np.array_equal(helpers.bits(oh_g), helpers.bits(oh_c))
    else:
        assert np.all(oh_g[:, :, :k1 - 1] == 0)
        assert helpers.ulp_diff(oh_g[:, :, k1 - 1:], oh_c[:, :, k1 - 1:]).max() <= 2


def test_reference_binding_module_drives_the_gpu(torch_cuda, tmp_path, small_model):
    """Drop-in: the reference's own xgb_fortran_api.F90 (compiled in place into oracle/_ref) making the
    reference's call sequence against libohxgb.so."""
    if not os.path.exists(helpers.DROPIN_HIP):
        pytest.skip("oracle/_ref not built")
    rows = synth.rows_cpu(synth.GRIDS["C12"], 0, 62208)
    rf, mf, pf = tmp_path / "rows.bin", tmp_path / "oh.model", tmp_path / "pred.bin"
    with open(rf, "wb") as f:
        f.write(struct.pack("<qq", rows.shape[0], rows.shape[1]))
        f.write(rows.tobytes())
    mf.write_bytes(small_model.image.tobytes())
    r = helpers.run_driver(helpers.DROPIN_HIP, rf, mf, pf)
    assert r.returncode == 0, r.stdout
    raw = pf.read_bytes()
    n, = struct.unpack_from("<q", raw, 0)
    pred = np.frombuffer(raw, dtype="<f4", count=n, offset=8)
    want = helpers.oracle_predict(small_model.image, rows, synth.XX_MISS)
    assert np.array_equal(helpers.bits(pred), helpers.bits(want))


def test_oversized_trees_use_the_packed_fallback(torch_cuda):
    """Two trees of ~1e6 nodes each: too many groups for the super-node format."""
    big = synth.make_model(num_trees=2, max_depth=24, sample_log2=20, min_leaf=1, grid=synth.GRIDS["C48"])
    rows = with_missing(synth.rows_cpu(synth.GRIDS["C48"], 100000, 30000), 0.002)
    want = helpers.oracle_predict(big.image, rows, synth.XX_MISS)
    b = capi.Booster(model_buffer=big.image)
    d = capi.DMatrix(rows, missing=synth.XX_MISS)
    assert np.array_equal(helpers.bits(b.predict(d)), helpers.bits(want))
    assert b.info()["packed"] == 1


@pytest.mark.parametrize("shape", [(4, 4, 72), (12, 72, 5), (10, 7, 9), (37, 3, 6), (1, 1, 130), (130, 1, 1)])
def test_grid_hint_changes_nothing_but_the_tiling(torch_cuda, shape, small_model):
    """OHXDMatrixSetGrid: a wave takes a brick of neighbouring gridcells instead of 64 consecutive rows.
    Same margins bit for bit - for grids that are not multiples of any brick, shards that start and end
    inside a level, every brick shape, both lane orders, with and without missing values."""
    im, jm, nk = shape
    n = im * jm * nk
    rows = synth.rows_cpu((im, jm, max(nk, 2)), 0, n) if im > 1 or jm > 1 else synth.rows_cpu((4, 4, 72), 0, n)
    want = helpers.oracle_predict(small_model.image, rows, synth.XX_MISS)
    for brick in ("auto", "4,4,4", "8,8,1", "8,4,2", "2,2,16", "64,1,1", "1,1,64", "0,0,0"):
        for kf in ("0", "1"):
            got = gpu_predict(small_model.image, rows, synth.XX_MISS, "super2", grid=(im, jm, 0),
                              params={"ohx_brick": brick, "ohx_brick_k_fastest": kf})
            assert np.array_equal(helpers.bits(got), helpers.bits(want)), (brick, kf)
    # a rank's shard: rows [r0, r0 + m) of the grid, starting and ending inside a level
    plane = im * jm
    for r0, m in ((plane // 3, n - plane // 3 - plane // 2), (0, max(1, plane // 2)), (n - 1, 1)):
        if m < 1 or r0 + m > n:
            continue
        for kernel in ("super2", "super3", "packed2"):
            got = gpu_predict(small_model.image, rows[r0:r0 + m], synth.XX_MISS, kernel, grid=(im, jm, r0))
            assert np.array_equal(helpers.bits(got), helpers.bits(want[r0:r0 + m])), (r0, m, kernel)
    holes = with_missing(rows, 0.01)
    got = gpu_predict(small_model.image, holes, synth.XX_MISS, "super2", grid=(im, jm, 0))
    assert np.array_equal(helpers.bits(got), helpers.bits(helpers.oracle_predict(small_model.image, holes, synth.XX_MISS)))


def test_grid_hint_is_validated(torch_cuda):
    d = capi.DMatrix(np.zeros((4, 27), np.float32), missing=synth.XX_MISS)
    for bad in ((-1, 4, 0), (4, 0, 0), (0, 4, 0)):
        with pytest.raises(capi.OhxError):
            d.set_grid(*bad)
    d.set_grid(0, 0, 0)
    d.free()


def test_month_roll_over_by_name_on_the_gpu(torch_cuda, tmp_path, small_model):
    """Two monthly boosters resident in HBM, selected by the expanded file name (Fortran host, both routes)."""
    grid = synth.GRIDS["C12"]
    pl, tropp, fields = helpers.synth_state(grid)
    state = tmp_path / "state.bin"
    helpers.write_state_file(state, pl, tropp, fields, True, ohscale=1.0)
    other = synth.make_model(num_trees=20, max_depth=10, sample_log2=15, min_leaf=4, grid=synth.GRIDS["C12"],
                             model_seed=77)
    (tmp_path / "oh_M01.model").write_bytes(small_model.image.tobytes())
    (tmp_path / "oh_M02.model").write_bytes(other.image.tobytes())
    (tmp_path / "oh_M03.model").write_bytes(small_model.image.tobytes())
    for mode in ("compat", "fused"):
        out_g, out_c = tmp_path / f"gpu_{mode}.bin", tmp_path / f"cpu_{mode}.bin"
        for exe, out in ((helpers.DRIVER_HIP, out_g), (helpers.DRIVER_ORACLE, out_c)):
            r = helpers.run_driver(exe, state, tmp_path / "oh_M%m2.model", out, mode, 3, "by_name")
            assert r.returncode == 0, r.stdout
            assert np.frombuffer(open(out, "rb").read()[-4:], dtype="<i4")[0] == 3
        _, k1, _, oh_g, _ = helpers.read_driver_output(out_g, *grid)
        _, _, _, oh_c, _ = helpers.read_driver_output(out_c, *grid)
        if mode == "compat":
            assert np.array_equal(helpers.bits(oh_g), helpers.bits(oh_c))
        else:
            assert helpers.ulp_diff(oh_g[:, :, k1 - 1:], oh_c[:, :, k1 - 1:]).max() <= 2


def test_inferred_level_size_that_is_not_a_multiple_of_four(torch_cuda, small_model):
    """A level of 97 x 71 = 6 887 cells: the runs of rows a wave fetches together (load_pieces) then start at
    addresses that are 4-byte but not 16-byte aligned, with the level size found by the library itself - through
    the compat calls (the verdict of the first tick serves the later ones) and on a device-resident matrix."""
    grid = (97, 71, 72)
    rows = synth.rows_cpu(grid, 0, 97 * 71 * 9)
    want = helpers.oracle_predict(small_model.image, rows, synth.XX_MISS)
    b = capi.Booster(model_buffer=small_model.image)
    for tick in range(3):                                  # create / predict / free, as the reference does every tick
        d = capi.DMatrix(rows, missing=synth.XX_MISS)
        assert np.array_equal(helpers.bits(b.predict(d)), helpers.bits(want)), tick
        assert d.grid() == (97 * 71, 1, 0, True)
        d.free()
    # rows of the same shape that are NOT level-stacked, after the library has seen that shape: still right
    shuffled = rows[np.random.default_rng(5).permutation(len(rows))]
    d = capi.DMatrix(shuffled, missing=synth.XX_MISS)
    assert np.array_equal(helpers.bits(b.predict(d)),
                          helpers.bits(helpers.oracle_predict(small_model.image, shuffled, synth.XX_MISS)))
    d.free()
    t = torch_cuda.from_numpy(rows).cuda()
    out = torch_cuda.empty(rows.shape[0], dtype=torch_cuda.float32, device="cuda")
    for hint in ("none", "full"):
        dd = capi.DMatrix(device_ptr=t.data_ptr(), nrow=rows.shape[0], ncol=27, missing=synth.XX_MISS)
        if hint == "full":
            dd.set_grid(97, 71, 0)
        out.zero_()
        b.predict_device(dd, out.data_ptr())
        torch_cuda.cuda.synchronize()
        b.check()
        assert np.array_equal(helpers.bits(out.cpu().numpy()), helpers.bits(want)), hint
        dd.free()
    assert b.lib.OHXReleaseScratch() == 0


def test_level_size_is_inferred_from_the_rows(torch_cuda, small_model):
    """No hint: XGDMatrixCreateFromMat finds the level size of a level-stacked gather from the exact
    repetition of its first column (LAT) and tiles by it; predictions are the same bit for bit.  An explicit
    hint overrides it; rows that are not level-stacked (shuffled, a ragged shard) are left alone."""
    # a 96 x 72 grid (a level of 6 912 cells, above the library's minimum of 4 096), 12 levels
    wide = synth.rows_cpu((96, 72, 72), 0, 96 * 72 * 12)
    want = helpers.oracle_predict(small_model.image, wide, synth.XX_MISS)
    d = capi.DMatrix(wide, missing=synth.XX_MISS)
    assert d.grid() == (96 * 72, 1, 0, True)
    b = capi.Booster(model_buffer=small_model.image)
    assert np.array_equal(helpers.bits(b.predict(d)), helpers.bits(want))
    d.set_grid(96, 72, 0)
    assert d.grid() == (96, 72, 0, False)
    assert np.array_equal(helpers.bits(b.predict(d)), helpers.bits(want))
    d.free()
    # a ragged shard (not whole levels) and shuffled rows: nothing to find, nothing breaks
    for arr in (wide[1000:-500], wide[np.random.default_rng(3).permutation(len(wide))]):
        d = capi.DMatrix(arr, missing=synth.XX_MISS)
        assert d.grid()[3] is False and d.grid()[0] == 0
        got = b.predict(d)
        assert np.array_equal(helpers.bits(got), helpers.bits(helpers.oracle_predict(small_model.image, arr, synth.XX_MISS)))
        d.free()
    # a caller whose first column is not a 2-D field: another one of the 2-D columns does
    swapped = wide.copy()
    swapped[:, [0, 2]] = swapped[:, [2, 0]]                 # T first, LAT third
    d = capi.DMatrix(swapped, missing=synth.XX_MISS)
    assert d.grid() == (96 * 72, 1, 0, True)
    d.free()
    # a matrix over device memory: nothing is looked at on creation, OHXDMatrixInferGrid looks on demand
    t = torch_cuda.from_numpy(wide).cuda()
    dd = capi.DMatrix(device_ptr=t.data_ptr(), nrow=wide.shape[0], ncol=27, missing=synth.XX_MISS)
    assert dd.grid() == (0, 0, 0, False)
    assert dd.infer_grid() is True and dd.grid() == (96 * 72, 1, 0, True)
    out = torch_cuda.empty(wide.shape[0], dtype=torch_cuda.float32, device="cuda")
    b.predict_device(dd, out.data_ptr())
    torch_cuda.cuda.synchronize()
    b.check()
    assert np.array_equal(helpers.bits(out.cpu().numpy()), helpers.bits(want))
    dd.free()
    # ... and the first predict on a matrix nobody described looks by itself, once
    dd = capi.DMatrix(device_ptr=t.data_ptr(), nrow=wide.shape[0], ncol=27, missing=synth.XX_MISS)
    out.zero_()
    b.predict_device(dd, out.data_ptr())
    torch_cuda.cuda.synchronize()
    assert dd.grid() == (96 * 72, 1, 0, True)
    assert np.array_equal(helpers.bits(out.cpu().numpy()), helpers.bits(want))
    dd.set_grid(0, 0, 0)                     # "no grid" is an answer too: 64 consecutive rows, no further look
    b.predict_device(dd, out.data_ptr())
    torch_cuda.cuda.synchronize()
    assert dd.grid() == (0, 0, 0, False)
    assert np.array_equal(helpers.bits(out.cpu().numpy()), helpers.bits(want))
    dd.free()
    # the reference frees and re-creates its matrix every tick: the parked buffer is reused, results unchanged
    for _ in range(3):
        d = capi.DMatrix(wide, missing=synth.XX_MISS)
        assert np.array_equal(helpers.bits(b.predict(d)), helpers.bits(want))
        d.free()
    small = capi.DMatrix(wide[:7], missing=synth.XX_MISS)       # a tiny matrix does not take the big parked buffer
    assert np.array_equal(helpers.bits(b.predict(small)), helpers.bits(want[:7]))
    small.free()
    assert b.lib.OHXReleaseScratch() == 0
    # too small to bother (fewer than 2 x 4096 rows)
    d = capi.DMatrix(wide[:5000], missing=synth.XX_MISS)
    assert d.grid() == (0, 0, 0, False)
    d.free()


def test_margin_of_a_non_identity_objective_starts_from_prob_to_margin(torch_cuda):
    """option_mask = 1 on a logistic / log-link model: xgboost starts the margin from
    obj->ProbToMargin(base_score), not from base_score; the transformed value (option_mask = 0)
    is not implemented and is refused, never returned untransformed."""
    import json as _json
    from oracle import xgb_oracle as O
    doc = _json.load(open(os.path.join(helpers.GOLDEN, "hand_forest.json")))
    cases, rows = helpers.load_hand_cases()
    for objective, base in (("binary:logistic", 0.25), ("count:poisson", 0.5), ("reg:squarederror", 0.5)):
        doc["learner"]["objective"]["name"] = objective
        doc["learner"]["learner_model_param"]["base_score"] = repr(base)
        text = _json.dumps(doc).encode()
        want = O.predict(O.load_model(text), rows, missing=cases["missing"])
        for kernel in ("wide", "packed2", "super2"):
            got = gpu_predict(text, rows, cases["missing"], kernel, option_mask=1)
            assert np.array_equal(helpers.bits(got), helpers.bits(want)), (objective, kernel)
        if objective != "reg:squarederror":
            with pytest.raises(capi.OhxError, match="prediction transform"):
                gpu_predict(text, rows, cases["missing"], "super2", option_mask=0)
            img = helpers.legacy_image(doc, version=(0, 0))           # pre-1.0 binary: the margin as stored
            got = gpu_predict(img, rows, cases["missing"], "super2", option_mask=1)
            m = O.load_model(img)
            assert float(m.base_score) == base
            assert np.array_equal(helpers.bits(got), helpers.bits(O.predict(m, rows, missing=cases["missing"])))
    doc["learner"]["objective"]["name"] = "my:custom"
    with pytest.raises(capi.OhxError, match="not known"):
        gpu_predict(_json.dumps(doc).encode(), rows, cases["missing"], "super2", option_mask=1)
    leaves = gpu_predict(_json.dumps(doc).encode(), rows, cases["missing"], "wide", option_mask=16)
    assert np.array_equal(leaves.reshape(len(rows), -1), np.float32(cases["leaf_index"]))


def test_rows_in_no_order_are_grouped_before_the_walk(torch_cuda, small_model, deep_model):
    """Rows whose order says nothing (shuffled, or too few of a grid for a level size to show) go through the
    clustering pass (csrc/kernels.hip): a key from the top of the first trees, a counting sort, and a walk through
    the permutation.  Same margins bit for bit, whatever the key width, the row count, the missing values; rows
    that ARE in grid order are recognised and left alone."""
    torch = torch_cuda
    grid = synth.GRIDS["C48"]
    n = 400_000
    rows = with_missing(synth.rows_cpu(grid, 123_456, n), 0.001)
    rng = np.random.default_rng(5)
    shuffled = rows[rng.permutation(n)]
    want = helpers.oracle_predict(deep_model.image, shuffled, synth.XX_MISS)

    def run(image, arr, params, rows_on_device=True):
        b = capi.Booster(model_buffer=image)
        for k, v in params.items():
            b.set_param(k, v)
        if not rows_on_device:
            d = capi.DMatrix(arr, missing=synth.XX_MISS)
            out = b.predict(d)
            d.free()
            return out, None
        t = torch.from_numpy(arr).cuda()
        d = capi.DMatrix(device_ptr=t.data_ptr(), nrow=arr.shape[0], ncol=arr.shape[1], missing=synth.XX_MISS)
        out = torch.empty(arr.shape[0], dtype=torch.float32, device="cuda")
        for _ in range(2):                                   # the second predict reuses the verdict, not the keys
            out.zero_()
            b.predict_device(d, out.data_ptr())
            torch.cuda.synchronize()
        b.check()
        d.free()
        return out.cpu().numpy(), b

    for params in ({}, {"ohx_cluster": "on"}, {"ohx_cluster": "off"}, {"ohx_cluster": "on", "ohx_cluster_trees": 1, "ohx_cluster_steps": 1},
                   {"ohx_cluster": "on", "ohx_cluster_trees": 8, "ohx_cluster_steps": 9},          # trimmed to 4 trees, 32 bits
                   {"ohx_cluster": "on", "ohx_cluster_trees": 3, "ohx_cluster_steps": 4, "ohx_cluster_zorder": 1},
                   {"ohx_cluster": "on", "ohx_kernel": "super4"}, {"ohx_cluster": "on", "ohx_kernel": "packed2"}):
        got, _ = run(deep_model.image, shuffled, params)
        assert np.array_equal(helpers.bits(got), helpers.bits(want)), params
    got, _ = run(deep_model.image, shuffled, {}, rows_on_device=False)      # the reference's own call sequence
    assert np.array_equal(helpers.bits(got), helpers.bits(want))
    # ragged sizes and tiny matrices, clustering forced
    for m in (1, 63, 64, 65, 1000, 4097):
        got, _ = run(small_model.image, shuffled[:m], {"ohx_cluster": "on"})
        assert np.array_equal(helpers.bits(got), helpers.bits(helpers.oracle_predict(small_model.image, shuffled[:m], synth.XX_MISS))), m
    # fewer columns than features, NaN as the missing marker
    narrow = shuffled[:70_000, :20].copy()
    b = capi.Booster(model_buffer=small_model.image)
    b.set_param("ohx_cluster", "on")
    d = capi.DMatrix(narrow, missing=float("nan"))
    assert np.array_equal(helpers.bits(b.predict(d)), helpers.bits(helpers.oracle_predict(small_model.image, narrow, float("nan"))))
    d.free()


@pytest.mark.parametrize("tops", ["on", "off", "auto"])
def test_tree_tops_walk_gives_the_same_margins(torch_cuda, small_model, deep_model, tops):
    """walk_super with and without the coalesced tree-top load (ohx_tree_tops): same margins on deep and shallow
    boosters, with missing values, with more trees than the first-step table holds; `auto` picks it for the deep
    booster (9 steps per tree) and not for the shallow one (5), and the library names the kernel it launches."""
    rows = with_missing(synth.rows_cpu(synth.GRIDS["C12"], 0, 12 * 72 * 30), 0.01)
    hand = open(os.path.join(helpers.GOLDEN, "hand_forest.json"), "rb").read()
    many = synth.make_model(num_trees=150, max_depth=5, sample_log2=12, min_leaf=2, grid=synth.GRIDS["C12"])
    for image, deep in ((small_model.image, False), (deep_model.image, True), (many.image, False), (hand, False)):
        if image is hand:
            cases, x = helpers.load_hand_cases()
            miss, want = cases["missing"], np.float32(cases["margin"])      # the golden vector
        else:
            x, miss = rows, synth.XX_MISS
            want = helpers.oracle_predict(image, x, miss)
        for kernel in ("super2", "super3"):
            got = gpu_predict(image, x, miss, kernel, params={"ohx_tree_tops": tops})
            assert np.array_equal(helpers.bits(got), helpers.bits(want)), (kernel, tops, deep)
        b = capi.Booster(model_buffer=image)
        b.set_param("ohx_kernel", "super2")          # (`auto` takes a deep booster's big batches to the ring kernel)
        b.set_param("ohx_tree_tops", tops)
        sym = b.kernel_symbol(x.shape[1])
        uses = {"on": True, "off": False, "auto": deep}[tops]
        assert sym.startswith("predict_rows_tile_kernel<2,2,") and sym.endswith(",true>" if uses else ",false>"), sym
        b.free()
    with pytest.raises(capi.OhxError, match="ohx_tree_tops"):
        capi.Booster(model_buffer=hand).set_param("ohx_tree_tops", "maybe")


@pytest.mark.parametrize("shape", [(12, 72, 9), (20, 7, 5), (4, 4, 72), (64, 3, 2)])
def test_rows_fetched_together_or_lane_by_lane(torch_cuda, shape, small_model):
    """ohx_coop_rows: the wave fetches its tile's rows as 16-byte pieces of the runs of consecutive rows and scatters
    them to the tile (default), or every lane fetches its own row.  Same margins - device matrices of exactly
    nrow x 27 floats (a piece must never be read from beyond them), shards that start and end anywhere, one-row
    and 63-row matrices, both lane orders, the brick shapes with runs of 4, 8 and 64 rows, no grid at all, missing
    values."""
    torch = torch_cuda
    im, jm, nk = shape
    n = im * jm * nk
    rows = with_missing(synth.rows_cpu((im, jm, max(nk, 2)), 0, n), 0.003)
    want = helpers.oracle_predict(small_model.image, rows, synth.XX_MISS)
    plane = im * jm
    cases = [(0, n), (plane // 3 + 1, n - plane // 3 - plane // 2 - 2), (n - 1, 1), (5, 63), (0, 65), (n - 130, 129)]
    for r0, m in cases:
        if m < 1 or r0 < 0 or r0 + m > n:
            continue
        d_rows = torch.from_numpy(rows[r0:r0 + m].copy()).to("cuda:0")
        out = torch.empty(m, dtype=torch.float32, device="cuda:0")
        for grid in ((im, jm, r0), None):
            for coop in ("1", "0"):
                for kf, brick in (("1", "auto"), ("0", "auto"), ("1", "8,4,2"), ("0", "8,8,1"), ("1", "64,1,1")):
                    if grid is None and (kf, brick) != ("1", "auto"):
                        continue
                    b = capi.Booster(model_buffer=small_model.image)
                    for k, v in (("ohx_coop_rows", coop), ("ohx_brick_k_fastest", kf), ("ohx_brick", brick)):
                        if not (k == "ohx_brick" and v == "auto"):
                            b.set_param(k, v)
                    d = capi.DMatrix(device_ptr=d_rows.data_ptr(), nrow=m, ncol=27, missing=synth.XX_MISS)
                    d.set_grid(*grid) if grid else d.set_grid(0, 0, 0)
                    out.fill_(7.0)
                    b.predict_device(d, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
                    torch.cuda.synchronize()
                    b.check()
                    assert np.array_equal(helpers.bits(out.cpu().numpy()), helpers.bits(want[r0:r0 + m])), \
                        (r0, m, grid, coop, kf, brick)
                    d.free()
                    b.free()


def test_two_oh_instances_from_two_threads(torch_cuda, small_model, deep_model):
    """QuickChem_GridComp.rc may list several OH instances (:22); a host that drives two of them from two threads has two
    boosters in one process.  Each thread runs the reference's five calls ten times on its own booster and its own
    rows (ctypes releases the GIL during the calls); the library's process-wide pools and caches - parked matrix
    buffers, the inf-check scratch, the level-size verdicts - are shared between them.  Every result against the
    oracle."""
    import threading
    grid = (96, 72, 72)
    jobs = []
    for k, model in enumerate((small_model, deep_model)):
        rows = with_missing(synth.rows_cpu(grid, 1000 * k, 96 * 72 * (3 + k)), 0.001, seed=30 + k)
        jobs.append((model.image, rows, helpers.oracle_predict(model.image, rows, synth.XX_MISS)))
    errors = []

    def run(image, rows, want):
        try:
            b = capi.Booster(model_buffer=image)
            for _ in range(10):
                d = capi.DMatrix(rows, missing=synth.XX_MISS)
                got = b.predict(d)
                d.free()
                if not np.array_equal(helpers.bits(got), helpers.bits(want)):
                    errors.append("margins differ")
            b.free()
        except Exception as e:                      # noqa: BLE001 - reported by the assert below
            errors.append(repr(e))
    threads = [threading.Thread(target=run, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert errors == []


def test_random_grids_shards_and_brick_shapes(torch_cuda):
    """tools/fuzz_tiles.py: random grids, shards, brick shapes, lane orders and launch shapes on device buffers of
    exactly nrow x 27 floats, against the oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_tiles", os.path.join(helpers.ROOT, "tools", "fuzz_tiles.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.run(150, 20261004) == []


# ------------------------------------------------------------------ the ring kernels (tree tops resident in LDS)

RING = {"ohx_tree_split": "off"}          # a small batch would have its trees split over waves (the super2 way) instead


def test_ring_kernel_is_the_default_for_deep_boosters(torch_cuda, small_model, deep_model):
    """`auto` sends the OH-shaped batches (27 columns) of a booster of five or more steps per tree through
    predict_rows_ring_kernel - the OH booster's depth 18 (nine steps), also depth 10 (five) - and a shallow booster's
    (depth 6: three steps) through the tile kernel; the library names what it launches."""
    shallow = synth.make_model(num_trees=20, max_depth=6, sample_log2=13, min_leaf=4, grid=synth.GRIDS["C12"])
    for image, want in ((deep_model.image, "predict_rows_ring_kernel"), (small_model.image, "predict_rows_ring_kernel"),
                        (shallow.image, "predict_rows_tile_kernel<2,2,true,false>")):
        b = capi.Booster(model_buffer=image)
        assert b.kernel_symbol(27) == want
        b.free()
    b = capi.Booster(model_buffer=small_model.image)
    b.set_param("ohx_kernel", "ring")
    assert b.kernel_symbol(27) == "predict_rows_ring_kernel"
    assert b.kernel_symbol(20).startswith("predict_rows_tile_kernel<2,2,false")      # other shapes: the super2 way
    b.free()


@pytest.mark.parametrize("nrows", [1, 63, 65, 1000, 16 * 64 + 1, 62208, 96 * 72 * 9 + 5])
def test_ring_rows_vs_oracle_ragged_sizes(torch_cuda, deep_model, small_model, nrows):
    """Batches that do not fill a block's 16 tiles, a wave's 64 lanes or a launch's rounds: waves without a tile still go
    round (the others count on their progress).  Deep booster (9 steps: four from LDS, five gathered) and the shallow one
    (5 steps), consecutive rows and bricks, one round per launch and everything in one launch."""
    grid = (96, 72, 72)
    rows = synth.rows_cpu(grid, 777, nrows)
    for image in (deep_model.image, small_model.image):
        want = helpers.oracle_predict(image, rows, synth.XX_MISS)
        for hint, extra in (((0, 0, 0), {}), ((96, 72, 777), {}), ((96, 72, 777), {"ohx_ring_rounds": 1}),
                            ((96, 72, 777), {"ohx_ring_rounds": 0}), ((96, 72, 777), {"ohx_xcd_remap": 0}),
                            ((96, 72, 777), {"ohx_reserve_cus": 16})):
            got = gpu_predict(image, rows, synth.XX_MISS, "ring", params=dict(RING, **extra), grid=hint)
            assert np.array_equal(helpers.bits(got), helpers.bits(want)), (hint, extra)


def test_ring_tree_counts_that_are_not_multiples_of_four(torch_cuda, deep_model):
    """Groups of four trees: a tree limit that leaves one, two or three trees in the last group (the spare chains walk the
    last tree again and their leaves are dropped), a single tree, a single group."""
    rows = with_missing(synth.rows_cpu(synth.GRIDS["C12"], 0, 30000), 0.002)
    for lim in (1, 2, 3, 4, 5, 37, 98, 99, 100):
        want = helpers.oracle_predict(deep_model.image, rows, synth.XX_MISS, ntree_limit=lim)
        got = gpu_predict(deep_model.image, rows, synth.XX_MISS, "ring", ntree_limit=lim, params=RING, grid=(12, 72, 0))
        assert np.array_equal(helpers.bits(got), helpers.bits(want)), lim


@pytest.mark.parametrize("missing", [synth.XX_MISS, float("nan")])
def test_ring_missing_values_and_the_second_launch(torch_cuda, deep_model, missing):
    """Missing values in the ring kernel: walked missing-aware by the wave that holds them (deferral off), or listed for
    the second launch (on; at 5 % the list overflows and the later waves walk missing-aware)."""
    grid = (96, 72, 72)
    for rate in (1e-4, 3e-3, 0.05):
        rows = with_missing(synth.rows_cpu(grid, 1234, 96 * 72 * 6 + 777), rate, seed=11)
        if missing != missing:
            rows[rows == np.float32(synth.XX_MISS)] = np.nan
        want = helpers.oracle_predict(deep_model.image, rows, missing)
        for defer in ("on", "off"):
            for hint in ((96, 72, 1234), (0, 0, 0)):
                got = gpu_predict(deep_model.image, rows, missing, "ring", params=dict(RING, ohx_defer_missing=defer), grid=hint)
                assert np.array_equal(helpers.bits(got), helpers.bits(want)), (rate, defer, hint)


def test_ring_golden_vectors_and_odd_boosters(torch_cuda):
    """The hand-made forest (five trees, one to three steps: every walk ends among fillers long before the fourth LDS
    step) and a booster of 150 shallow trees (more than the tile kernels' first-step table holds) through the ring kernel."""
    import json as _json
    doc = _json.load(open(os.path.join(helpers.GOLDEN, "hand_forest.json")))
    doc["learner"]["learner_model_param"]["num_feature"] = "27"       # the ring kernels take 27-feature boosters only:
    for t in doc["learner"]["gradient_booster"]["model"]["trees"]:     # the same trees, 24 features nobody splits on
        t["tree_param"]["num_feature"] = "27"
    hand = _json.dumps(doc).encode()
    cases, x = helpers.load_hand_cases()
    x27 = np.full((x.shape[0], 27), np.float32(cases["missing"]), dtype=np.float32)
    x27[:, :x.shape[1]] = x
    got = gpu_predict(hand, x27, cases["missing"], "ring", params=RING)
    assert np.array_equal(helpers.bits(got), helpers.bits(np.float32(cases["margin"])))      # the hand-computed answers
    many = synth.make_model(num_trees=150, max_depth=5, sample_log2=12, min_leaf=2, grid=synth.GRIDS["C12"])
    rows = with_missing(synth.rows_cpu(synth.GRIDS["C12"], 0, 12 * 72 * 30), 0.01)
    want = helpers.oracle_predict(many.image, rows, synth.XX_MISS)
    got = gpu_predict(many.image, rows, synth.XX_MISS, "ring", params=RING)
    assert np.array_equal(helpers.bits(got), helpers.bits(want))


def test_ring_rows_in_no_order(torch_cuda, deep_model):
    """Shuffled rows go through the clustering pass's permutation into the ring kernel (every lane its own row)."""
    rng = np.random.default_rng(5)
    rows = synth.rows_cpu((96, 72, 72), 0, 96 * 72 * 8)
    rows = rows[rng.permutation(rows.shape[0])]
    want = helpers.oracle_predict(deep_model.image, rows, synth.XX_MISS)
    for cluster in ("on", "off"):
        got = gpu_predict(deep_model.image, rows, synth.XX_MISS, "ring", params=dict(RING, ohx_cluster=cluster))
        assert np.array_equal(helpers.bits(got), helpers.bits(want)), cluster


@pytest.mark.parametrize("dynamic", [True, False])
def test_ring_fused_fields_vs_oracle(torch_cuda, deep_model, dynamic):
    """The fused call on a slab big enough for predict_fields_ring_kernel (>= two residencies of the chip), with -999.0 and
    NaN in the fields, the rows that hold them listed for the second launch or walked in place; margins bit for bit."""
    grid = (144, 96, 80)
    pl, tropp, fields = helpers.synth_state(grid)
    rng = np.random.default_rng(23)
    fields = [f.copy() for f in fields]
    for f in fields[2:]:
        mask = rng.random(f.shape) < 2e-4
        f[mask] = np.where(rng.random(int(mask.sum())) < 0.5, np.float32(synth.XX_MISS), np.float32(np.nan))
    oh_ref, margin_ref, k1, k2 = helpers.oracle_predict_oh(deep_model.image, pl, tropp, fields, dynamic)
    assert grid[0] * grid[1] * (k2 - k1 + 1) >= 256 * 16 * 64 * 2, (k1, k2)
    for defer in ("on", "off"):
        p = oh_predict.OHPredictor()
        p.xx_bst = capi.Booster(model_buffer=deep_model.image)
        p.xx_bst.set_param("ohx_kernel", "ring")
        p.xx_bst.set_param("ohx_defer_missing", defer)
        p.first_time = False
        oh = np.zeros(grid, dtype=np.float32)
        margins = []
        assert p.predict_OH_with_XGB("unused", *grid, dynamic, 4000.0, pl, tropp, oh_predict.OHBoostInputData(fields), oh,
                                     mode="fused", margin_out=margins) == 0
        assert np.array_equal(helpers.bits(margins[0]), helpers.bits(margin_ref)), defer
        assert np.all(oh[:, :, :k1 - 1] == 0)
        assert helpers.ulp_diff(oh[:, :, k1 - 1:], oh_ref[:, :, k1 - 1:]).max() <= 2


RING_SPIN1 = os.path.join(helpers.ROOT, "tools", "bin", "variants", "ringspin1", "libohxgb.so")


def test_a_ring_time_out_is_predicted_again_not_raised(torch_cuda, deep_model, capfd):
    """a ring block that gives up waiting used to end in rc -1 ("error flags 2") and the caller's
    _ASSERT(rc==0) (OH_GridCompMod.F90:356-358).  A scratch build whose waves give up after ONE look (tools/build_variant.sh
    ringspin1 -DOHX_EXP_RING_SPIN=1, made by __graft_entry__.build()) times out in nearly every block: the launch behind
    the train - the tile kernel, which only runs when that train's id stands in the time-out word - predicts the rows again
    and counts the event.  Margins bit for bit against the oracle: the host form, the device form without
    OHXBoosterCheck, with and without missing values, and the fused fields call."""
    if not os.path.exists(RING_SPIN1):
        pytest.skip("tools/bin/variants/ringspin1 not built")
    import ctypes as C
    import torch
    lib = capi.load_library(RING_SPIN1)
    grid = (96, 72, 72)
    nrow = 96 * 72 * 40
    for rate in (0.0, 2e-3):
        rows = synth.rows_cpu(grid, 0, nrow)
        if rate:
            rows = with_missing(rows, rate, seed=5)
        want = helpers.oracle_predict(deep_model.image, rows, synth.XX_MISS)
        b = capi.Booster(model_buffer=deep_model.image, lib=lib)
        b.set_param("ohx_kernel", "ring")
        b.set_param("ohx_tree_split", "off")
        assert b.ring_reruns() == 0
        d = capi.DMatrix(rows, missing=synth.XX_MISS, lib=lib)
        d.set_grid(96, 72, 0)
        assert "only after a ring time-out" in b.kernel_symbols_for(d)
        got = b.predict(d)                                            # rc 0: no error is raised
        assert np.array_equal(helpers.bits(got), helpers.bits(want)), rate
        first = b.ring_reruns()
        assert first > 0
        # the device form: stream-ordered, nobody calls OHXBoosterCheck
        dev_rows = torch.from_numpy(rows).cuda()
        out = torch.full((nrow,), float("nan"), dtype=torch.float32, device="cuda")
        dd = capi.DMatrix(device_ptr=dev_rows.data_ptr(), nrow=nrow, ncol=27, missing=synth.XX_MISS, lib=lib)
        dd.set_grid(96, 72, 0)
        b.predict_device(dd, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(helpers.bits(out.cpu().numpy()), helpers.bits(want)), rate
        assert b.ring_reruns() > first
        b.check()                                                     # and nothing is left to raise
        dd.free()
        d.free()
        b.free()
    assert "gave up waiting" in capfd.readouterr().err               # said once on stderr
    # the fused fields call on a slab big enough for predict_fields_ring_kernel
    fgrid = (144, 96, 80)
    pl, tropp, fields = helpers.synth_state(fgrid)
    oh_ref, margin_ref, k1, k2 = helpers.oracle_predict_oh(deep_model.image, pl, tropp, fields, True)
    assert fgrid[0] * fgrid[1] * (k2 - k1 + 1) >= 256 * 16 * 64 * 2
    p = oh_predict.OHPredictor()
    p.xx_bst = capi.Booster(model_buffer=deep_model.image, lib=lib)
    p.xx_bst.set_param("ohx_kernel", "ring")
    p.first_time = False
    oh = np.zeros(fgrid, dtype=np.float32)
    margins = []
    assert p.predict_OH_with_XGB("unused", *fgrid, True, 4000.0, pl, tropp, oh_predict.OHBoostInputData(fields), oh,
                                 mode="fused", margin_out=margins) == 0
    assert np.array_equal(helpers.bits(margins[0]), helpers.bits(margin_ref))
    assert helpers.ulp_diff(oh[:, :, k1 - 1:], oh_ref[:, :, k1 - 1:]).max() <= 2
    assert p.xx_bst.ring_reruns() > 0


def test_the_shipped_ring_kernel_does_not_time_out(torch_cuda, deep_model):
    """... and the library as shipped goes through the same batches without a single re-run."""
    grid = (96, 72, 72)
    rows = synth.rows_cpu(grid, 0, 96 * 72 * 40)
    b = capi.Booster(model_buffer=deep_model.image)
    b.set_param("ohx_kernel", "ring")
    b.set_param("ohx_tree_split", "off")
    d = capi.DMatrix(rows, missing=synth.XX_MISS)
    d.set_grid(96, 72, 0)
    want = helpers.oracle_predict(deep_model.image, rows, synth.XX_MISS)
    for _ in range(5):
        assert np.array_equal(helpers.bits(b.predict(d)), helpers.bits(want))
    assert b.ring_reruns() == 0
    d.free()
    b.free()


@pytest.mark.parametrize("kernel", ["super1", "super2", "super4", "ring", "auto"])
def test_fused_fields_small_slab_with_its_trees_split_over_waves(torch_cuda, deep_model, kernel):
    """The fused call on a slab that leaves the chip mostly empty (a GEOS rank's block): its trees are cut into runs
    walked by different waves and summed in tree order by a second launch, as small row batches are - whatever the number
    of runs, with -999.0 and NaN in the fields; margins bit for bit, OH_ML within 2 ulp (10**x), levels above the slab
    untouched."""
    grid = (24, 12, 72)
    pl, tropp, fields = helpers.synth_state(grid)
    rng = np.random.default_rng(29)
    fields = [f.copy() for f in fields]
    for f in fields[2:]:
        mask = rng.random(f.shape) < 1e-3
        f[mask] = np.where(rng.random(int(mask.sum())) < 0.5, np.float32(synth.XX_MISS), np.float32(np.nan))
    oh_ref, margin_ref, k1, k2 = helpers.oracle_predict_oh(deep_model.image, pl, tropp, fields, True)
    for split in ("off", "auto", "2", "3", "7", "10"):
        p = oh_predict.OHPredictor()
        p.xx_bst = capi.Booster(model_buffer=deep_model.image)
        p.xx_bst.set_param("ohx_kernel", kernel)
        p.xx_bst.set_param("ohx_tree_split", split)
        p.first_time = False
        oh = np.zeros(grid, dtype=np.float32)
        margins = []
        assert p.predict_OH_with_XGB("unused", *grid, True, 4000.0, pl, tropp, oh_predict.OHBoostInputData(fields), oh,
                                     mode="fused", margin_out=margins) == 0
        assert np.array_equal(helpers.bits(margins[0]), helpers.bits(margin_ref)), split
        assert np.all(oh[:, :, :k1 - 1] == 0)
        assert helpers.ulp_diff(oh[:, :, k1 - 1:], oh_ref[:, :, k1 - 1:]).max() <= 2, split


def test_host_form_predict_on_borrowed_rows_waits_for_the_caller_s_default_stream(torch_cuda, small_model, tmp_path):
    """(r6) A matrix made by OHXDMatrixCreateFromDevice borrows the caller's HBM; the host-form calls on it
    (XGBoosterPredict, XGDMatrixSaveBinary) run on the library's own non-blocking stream and must not read the rows before
    what the caller enqueued on the default stream - torch's - has filled them.  The rows start as NaN everywhere (every
    split takes its default child); a long queue of work and then the copy of the real rows are enqueued on torch's stream
    and the predict called at once, no synchronise: the margins are the real rows' margins, bit for bit against the
    oracle - which they are not when the library does not wait (seen with order_behind_caller taken out)."""
    torch = torch_cuda
    grid = synth.GRIDS["C12"]
    n = grid[0] * grid[1] * 24
    real = torch.empty((n, synth.NFEAT), dtype=torch.float32, device="cuda:0")
    synth.rows_device(grid, 0, n, real)
    want = helpers.oracle_predict(small_model.image, real.cpu().numpy(), synth.XX_MISS)
    rows = torch.full((n, synth.NFEAT), float("nan"), dtype=torch.float32, device="cuda:0")
    busy = torch.randn((4096, 4096), device="cuda:0")
    torch.cuda.synchronize()
    b = capi.Booster(model_buffer=small_model.image)
    d = capi.DMatrix(device_ptr=rows.data_ptr(), nrow=n, ncol=synth.NFEAT, missing=synth.XX_MISS)
    d.set_grid(grid[0], grid[1], 0)
    for _ in range(40):                               # tens of milliseconds in front of the copy, on torch's stream
        busy = busy @ busy
        busy = busy / busy.abs().max()
    rows.copy_(real, non_blocking=True)
    got = b.predict(d)                                # host form, straight away
    assert np.array_equal(helpers.bits(got), helpers.bits(want))
    # ... and the same for the matrix written to a file
    rows.fill_(float("nan"))
    for _ in range(40):
        busy = busy @ busy
        busy = busy / busy.abs().max()
    rows.copy_(real, non_blocking=True)
    d.save_binary(str(tmp_path / "rows.bin"))
    torch.cuda.synchronize()
    back = capi.DMatrix.from_file(str(tmp_path / "rows.bin"))
    assert np.array_equal(helpers.bits(b.predict(back)), helpers.bits(want))
    back.free()
    d.free()
    b.free()
