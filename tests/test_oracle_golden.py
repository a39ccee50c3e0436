"""The oracle against the golden vectors (CPU).  The reference pins nothing for this
path ("parity unpinned"), so the vectors are the build's own:
hand-computed answers for a hand-written booster, and the seeded config #1 case."""
import json
import os

import numpy as np

from oracle import xgb_oracle as O
from quickchem_amd import synth
from tests import helpers


def test_hand_forest_numpy_oracle():
    cases, rows = helpers.load_hand_cases()
    model = O.load_model(open(os.path.join(helpers.GOLDEN, "hand_forest.json"), "rb").read())
    miss = cases["missing"]
    assert np.array_equal(O.predict(model, rows, missing=miss), np.float32(cases["margin"]))
    assert np.array_equal(O.predict(model, rows, missing=miss, ntree_limit=2), np.float32(cases["margin_ntree_limit_2"]))
    assert np.array_equal(O.predict(model, rows, missing=miss, ntree_limit=3), np.float32(cases["margin_ntree_limit_3"]))
    assert np.array_equal(O.predict(model, rows, missing=miss, pred_leaf=True), np.float32(cases["leaf_index"]))
    assert np.array_equal(O.predict(model, np.float32(cases["rows_2col"]), missing=miss), np.float32(cases["margin_2col"]))


def test_hand_forest_c_oracle():
    """The C oracle reads legacy binary only: convert the hand-written JSON with the product's
    reader/writer (host logic), which also pins that conversion."""
    cases, rows = helpers.load_hand_cases()
    image = synth.convert_model(open(os.path.join(helpers.GOLDEN, "hand_forest.json"), "rb").read(), "binary")
    miss = cases["missing"]
    assert np.array_equal(helpers.oracle_predict(image, rows, miss), np.float32(cases["margin"]))
    assert np.array_equal(helpers.oracle_predict(image, rows, miss, ntree_limit=2), np.float32(cases["margin_ntree_limit_2"]))
    assert np.array_equal(helpers.oracle_predict(image, rows, miss, ntree_limit=3), np.float32(cases["margin_ntree_limit_3"]))
    leaves = helpers.oracle_predict(image, rows, miss, option_mask=16).reshape(len(rows), -1)
    assert np.array_equal(leaves, np.float32(cases["leaf_index"]))
    assert np.array_equal(helpers.oracle_predict(image, np.float32(cases["rows_2col"]), miss), np.float32(cases["margin_2col"]))


def test_accumulation_order_is_visible():
    """The +-1e8 stumps make the tree order observable: a float64 accumulator differs."""
    cases, rows = helpers.load_hand_cases()
    model = O.load_model(open(os.path.join(helpers.GOLDEN, "hand_forest.json"), "rb").read())
    leaves = O.predict(model, rows, missing=cases["missing"], pred_leaf=True).astype(int)
    wide = np.array([np.float64(model.base_score) + sum(float(model.trees[t].value[leaves[r, t]]) for t in range(5))
                     for r in range(len(rows))])
    assert not np.allclose(wide, cases["margin"])


def test_config1_golden_both_oracles(deep_model):
    g = json.load(open(os.path.join(helpers.GOLDEN, "mock4x4_T100.json")))
    grid = tuple(g["grid"])
    n = grid[0] * grid[1] * grid[2]
    assert deep_model.num_nodes == g["model_nodes"] and deep_model.max_depth == g["model_max_depth"]
    rows = synth.rows_cpu(grid, 0, n)
    assert int(np.bitwise_xor.reduce(rows.view(np.uint32).ravel())) == g["rows_crc"]
    want = np.array(g["margin_bits"], dtype=np.uint32)
    got_c = helpers.oracle_predict(deep_model.image, rows, synth.XX_MISS)
    assert np.array_equal(helpers.bits(got_c), want)
    got_np = O.predict(O.load_model(deep_model.image.tobytes()), rows, missing=synth.XX_MISS)
    assert np.array_equal(helpers.bits(got_np), want)


def test_oracles_agree_with_missing_values(small_model):
    grid = synth.GRIDS["C12"]
    rows = synth.rows_cpu(grid, 1000, 4096).copy()
    rng = np.random.default_rng(5)
    mask = rng.random(rows.shape) < 0.02
    rows[mask] = np.where(rng.random(mask.sum()) < 0.5, np.float32(synth.XX_MISS), np.float32(np.nan))
    a = helpers.oracle_predict(small_model.image, rows, synth.XX_MISS)
    b = O.predict(O.load_model(small_model.image.tobytes()), rows, missing=synth.XX_MISS)
    assert np.array_equal(helpers.bits(a), helpers.bits(b))
    # with missing = NaN, -999.0 is an ordinary value: the answers must change somewhere
    c = helpers.oracle_predict(small_model.image, rows, float("nan"))
    d = O.predict(O.load_model(small_model.image.tobytes()), rows, missing=float("nan"))
    assert np.array_equal(helpers.bits(c), helpers.bits(d))
    assert not np.array_equal(helpers.bits(a), helpers.bits(c))


def test_oracle_rejects_inf(small_model):
    import pytest
    from quickchem_amd import capi
    rows = synth.rows_cpu(synth.GRIDS["C12"], 0, 8).copy()
    rows[3, 5] = np.inf
    with pytest.raises(capi.OhxError, match="inf"):
        helpers.oracle_predict(small_model.image, rows, synth.XX_MISS)
    with pytest.raises(ValueError, match="inf"):
        O.predict(O.load_model(small_model.image.tobytes()), rows, missing=synth.XX_MISS)


def test_oracle_predict_oh_c_vs_numpy(small_model):
    """The restated RUN section of predict_OH_with_XGB, C against numpy, both slab rules."""
    grid = synth.GRIDS["mock4x4"]
    pl, tropp, fields = helpers.synth_state(grid)
    model = O.load_model(small_model.image.tobytes())
    for dynamic in (True, False):
        oh_c, margin_c, k1, k2 = helpers.oracle_predict_oh(small_model.image, pl, tropp, fields, dynamic)
        oh_np, margin_np, k1n, k2n = O.predict_OH_with_XGB(model, pl, tropp, fields, dynamic)
        assert (k1, k2) == (k1n, k2n) and 1 <= k1 <= k2 == grid[2]
        assert np.array_equal(helpers.bits(margin_c), helpers.bits(margin_np))
        assert np.all(oh_c[:, :, :k1 - 1] == 0)
        assert helpers.ulp_diff(oh_c[:, :, k1 - 1:], oh_np[:, :, k1 - 1:]).max() <= 2   # powf vs numpy power
