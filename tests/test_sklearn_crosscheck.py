"""An independent, third-party tree walker against the oracle and the HIP path.

libxgboost is not available (parity stays unpinned against it), but scikit-learn is, and its gradient-boosted
regression trees are walked by code nobody here wrote (sklearn/tree/_tree.pyx).  A scikit-learn booster is trained
on synthetic OH rows and transcribed, node for node, into XGBoost's JSON schema:
  * scikit-learn goes left when x <= t (x float32, t float64); XGBoost when x < c.  For float32 x the two agree
    exactly with c = the smallest float32 above t (nextafter(float32 rounded down from t, +inf));
  * nodes are renumbered breadth-first so that right child = left child + 1, as XGBoost's predictor assumes;
  * leaf value = learning_rate * value, base_score = the initial (mean) prediction.
Then, on rows the booster has not seen:  the leaf every row ends in must be THE SAME as scikit-learn's `apply`
(exact: this is the traversal, independent of any float accumulation), and the margins must agree with
`predict` to float32 accumulation error (scikit-learn sums in float64).  This pins the mechanics of the walk -
node layout, strict/non-strict compare, child order, leaf ids - to an outside implementation; it says nothing
about libxgboost's own file formats or its treatment of missing values."""
import json

import numpy as np
import pytest

from oracle import xgb_oracle as O
from quickchem_amd import capi, synth
from tests import helpers

sklearn = pytest.importorskip("sklearn")


def transcribe(gbr, nfeat, estimators=None, base=None):
    """-> (XGBoost JSON bytes, per tree: dict new node id -> scikit-learn node id)."""
    trees, maps = [], []
    lr = gbr.learning_rate if gbr is not None else 1.0
    for est in (gbr.estimators_[:, 0] if estimators is None else estimators):
        t = est.tree_
        go_left = getattr(t, "missing_go_to_left", None)
        order, new_of = [0], {0: 0}
        for old in order:                                   # breadth-first, siblings adjacent
            if t.children_left[old] != -1:
                for child in (t.children_left[old], t.children_right[old]):
                    new_of[child] = len(order)
                    order.append(child)
        n = len(order)
        left, right, parents = [-1] * n, [-1] * n, [2147483647] * n
        feat, cond, dleft = [0] * n, [0.0] * n, [0] * n
        for new, old in enumerate(order):
            if t.children_left[old] == -1:
                cond[new] = float(np.float32(lr * t.value[old, 0, 0]))
                continue
            left[new], right[new] = new_of[t.children_left[old]], new_of[t.children_right[old]]
            parents[left[new]] = parents[right[new]] = new
            feat[new] = int(t.feature[old])
            if go_left is not None:
                dleft[new] = int(go_left[old])
            thr = t.threshold[old]
            down = np.float32(thr)
            if float(down) > thr:                           # float32() rounds to nearest: step down if it went up
                down = np.nextafter(down, np.float32(-np.inf))
            cond[new] = float(np.nextafter(down, np.float32(np.inf)))     # smallest float32 strictly above thr
        trees.append({"base_weights": [0.0] * n, "categories": [], "categories_nodes": [], "categories_segments": [],
                      "categories_sizes": [], "default_left": dleft, "id": len(trees), "left_children": left,
                      "loss_changes": [0.0] * n, "parents": parents, "right_children": right, "split_conditions": cond,
                      "split_indices": feat, "split_type": [0] * n, "sum_hessian": [1.0] * n,
                      "tree_param": {"num_deleted": "0", "num_feature": str(nfeat), "num_nodes": str(n),
                                     "size_leaf_vector": "0"}})
        maps.append({new: old for new, old in enumerate(order)})
    base = float(np.float32(gbr.init_.constant_[0, 0])) if base is None else base
    doc = {"learner": {"attributes": {}, "feature_names": [], "feature_types": [],
                       "gradient_booster": {"model": {"gbtree_model_param": {"num_parallel_tree": "1",
                                                                             "num_trees": str(len(trees)),
                                                                             "size_leaf_vector": "0"},
                                                      "tree_info": [0] * len(trees), "trees": trees}, "name": "gbtree"},
                       "learner_model_param": {"base_score": "%.9g" % base, "num_class": "0", "num_feature": str(nfeat),
                                               "num_target": "1"},
                       "objective": {"name": "reg:squarederror", "reg_loss_param": {"scale_pos_weight": "1"}}},
           "version": [1, 6, 0]}
    return json.dumps(doc).encode(), maps


@pytest.fixture(scope="module")
def sk_case():
    from sklearn.ensemble import GradientBoostingRegressor
    grid = synth.GRIDS["C12"]
    rows = synth.rows_cpu(grid, 0, grid[0] * grid[1] * grid[2])
    rng = np.random.default_rng(7)
    train = rows[rng.choice(len(rows), 6000, replace=False)]
    y = (np.log10(train[:, 4] + 1e-12) + 0.01 * train[:, 2] - 0.2 * np.cos(np.deg2rad(train[:, 26])) +
         0.05 * rng.normal(size=len(train)))
    gbr = GradientBoostingRegressor(n_estimators=25, max_depth=9, learning_rate=0.3, subsample=0.7, random_state=0)
    gbr.fit(train, y)
    test = rows[rng.choice(len(rows), 20000, replace=False)]
    # rows sitting exactly on a threshold (the <= versus < boundary) and one float32 step either side of it
    t0 = gbr.estimators_[0, 0].tree_
    edge = test[:600].copy()
    picks = np.flatnonzero(t0.children_left != -1)
    for q, node in enumerate(picks[:200]):
        f, thr = t0.feature[node], np.float32(t0.threshold[node])
        edge[3 * q, f] = thr
        edge[3 * q + 1, f] = np.nextafter(thr, np.float32(np.inf))
        edge[3 * q + 2, f] = np.nextafter(thr, np.float32(-np.inf))
    test = np.ascontiguousarray(np.concatenate([test, edge]), dtype=np.float32)
    js, maps = transcribe(gbr, rows.shape[1])
    return gbr, js, maps, test


def check(leaves, margins, gbr, maps, test):
    want_leaves = np.asarray(gbr.apply(test)).reshape(len(test), -1).astype(np.int64)          # scikit-learn's own node ids
    got = leaves.reshape(len(test), -1).astype(np.int64)
    back = np.empty_like(got)
    for t, m in enumerate(maps):
        lut = np.full(max(m) + 1, -1, dtype=np.int64)
        for new, old in m.items():
            lut[new] = old
        back[:, t] = lut[got[:, t]]
    assert np.array_equal(back, want_leaves)                           # the traversal, exactly
    want = gbr.predict(test)
    assert np.max(np.abs(margins.astype(np.float64) - want)) < 2e-5 * max(1.0, np.max(np.abs(want)))


def test_scikit_learn_walks_like_the_oracle(sk_case):
    gbr, js, maps, test = sk_case
    model = O.load_model(js)
    check(O.predict(model, test, pred_leaf=True), O.predict(model, test), gbr, maps, test)
    binary = synth.convert_model(np.frombuffer(js, dtype=np.uint8), "binary")
    check(helpers.oracle_predict(binary, test, float("nan"), option_mask=16),
          helpers.oracle_predict(binary, test, float("nan")), gbr, maps, test)


@pytest.mark.gpu
def test_scikit_learn_walks_like_the_hip_path(sk_case):
    gbr, js, maps, test = sk_case
    for kernel in ("auto", "wide", "packed2"):
        b = capi.Booster(model_buffer=js)
        b.set_param("ohx_kernel", kernel)
        d = capi.DMatrix(test, missing=float("nan"))
        margins = b.predict(d)
        leaves = b.predict(d, option_mask=16)
        check(leaves, margins, gbr, maps, test)
        d.free()


@pytest.fixture(scope="module")
def sk_missing_case():
    """Missing values: scikit-learn's DecisionTreeRegressor takes NaN and records, per node, which child they go to
    (tree_.missing_go_to_left) - the counterpart of XGBoost's default_left."""
    from sklearn.tree import DecisionTreeRegressor
    grid = synth.GRIDS["C12"]
    rows = synth.rows_cpu(grid, 0, grid[0] * grid[1] * grid[2])
    rng = np.random.default_rng(11)

    def with_nan(a, rate):
        a = a.copy()
        a[rng.random(a.shape) < rate] = np.nan
        return a
    ests = []
    for q in range(8):
        train = with_nan(rows[rng.choice(len(rows), 5000, replace=False)], 0.05)
        y = np.nan_to_num(train[:, 2]) * 0.01 + np.nan_to_num(np.log10(train[:, 6] + 1e-12)) + rng.normal(size=len(train)) * 0.1
        ests.append(DecisionTreeRegressor(max_depth=9, min_samples_leaf=3, random_state=q).fit(train, y))
    if not hasattr(ests[0].tree_, "missing_go_to_left"):
        pytest.skip("this scikit-learn has no missing-value support in trees")
    test = np.ascontiguousarray(with_nan(rows[rng.choice(len(rows), 20000, replace=False)], 0.08), dtype=np.float32)
    js, maps = transcribe(None, rows.shape[1], estimators=ests, base=0.0)
    return ests, js, maps, test


def check_missing(leaves, ests, maps, test):
    got = leaves.reshape(len(test), -1).astype(np.int64)
    for t, (est, m) in enumerate(zip(ests, maps)):
        lut = np.full(max(m) + 1, -1, dtype=np.int64)
        for new, old in m.items():
            lut[new] = old
        assert np.array_equal(lut[got[:, t]], est.apply(test).astype(np.int64)), t


def test_missing_values_go_where_scikit_learn_sends_them(sk_missing_case):
    ests, js, maps, test = sk_missing_case
    assert np.isnan(test).any()
    check_missing(O.predict(O.load_model(js), test, pred_leaf=True), ests, maps, test)
    binary = synth.convert_model(np.frombuffer(js, dtype=np.uint8), "binary")
    check_missing(helpers.oracle_predict(binary, test, float("nan"), option_mask=16), ests, maps, test)
    # -999.0 as the missing marker, the OH path's own (OH_GridCompMod.F90:213): the same rows with NaN spelled -999.0
    marked = np.where(np.isnan(test), np.float32(-999.0), test)
    check_missing(helpers.oracle_predict(binary, marked, -999.0, option_mask=16), ests, maps, test)


@pytest.mark.gpu
def test_missing_values_on_the_hip_path_against_scikit_learn(sk_missing_case):
    ests, js, maps, test = sk_missing_case
    b = capi.Booster(model_buffer=js)
    d = capi.DMatrix(test, missing=float("nan"))
    check_missing(b.predict(d, option_mask=16), ests, maps, test)
    # the margin kernels take the same route: their sum of leaves equals the sum over scikit-learn's leaves
    want = np.zeros(len(test), dtype=np.float32)
    for est in ests:
        want = (want + est.tree_.value[est.apply(test), 0, 0].astype(np.float32)).astype(np.float32)
    for kernel in ("auto", "packed2", "wide"):
        b.set_param("ohx_kernel", kernel)
        assert np.array_equal(helpers.bits(b.predict(d)), helpers.bits(want)), kernel
    d.free()
