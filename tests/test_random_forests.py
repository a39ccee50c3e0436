"""Randomly shaped boosters (stumps, chains, lopsided trees, repeated and extreme thresholds)
on rows salted with NaN, -999.0, +-0.0, ties and huge values: the two oracles against each other
on CPU, every kernel family against the oracle on the GPU."""
import json

import numpy as np
import pytest

from oracle import xgb_oracle as O
from quickchem_amd import capi, synth
from tests import helpers

SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, 1e-30, -1e-30, 3.0e38, -3.0e38, 1.17549435e-38, 1e-45, 0.5, 2.5],
                   dtype=np.float32)


def random_tree(rng, nfeat, max_depth, p_leaf):
    left, right, feat, cond, dl = [], [], [], [], []

    def new():
        left.append(-1); right.append(-1); feat.append(0); cond.append(0.0); dl.append(0)
        return len(left) - 1
    todo = [(new(), 0)]
    while todo:
        n, d = todo.pop(0)
        leaf = d >= max_depth or (d > 0 and rng.random() < p_leaf)
        if leaf:
            cond[n] = float(np.float32(rng.normal(0, 0.1)))
            continue
        l, r = new(), new()
        left[n], right[n] = l, r
        feat[n] = int(rng.integers(0, nfeat))
        cond[n] = float(rng.choice(SPECIAL)) if rng.random() < 0.3 else float(np.float32(rng.normal(0, 2)))
        dl[n] = int(rng.integers(0, 2))
        todo += [(l, d + 1), (r, d + 1)]
    return left, right, feat, cond, dl


def random_booster_json(rng, ntree, nfeat, max_depth, p_leaf):
    trees = []
    for t in range(ntree):
        left, right, feat, cond, dl = random_tree(rng, nfeat, max_depth, p_leaf)
        n = len(left)
        parents = [2147483647] * n
        for i in range(n):
            if left[i] != -1:
                parents[left[i]] = i
                parents[right[i]] = i
        trees.append({"base_weights": [0.0] * n, "categories": [], "categories_nodes": [], "categories_segments": [],
                      "categories_sizes": [], "default_left": dl, "id": t, "left_children": left,
                      "loss_changes": [0.0] * n, "parents": parents, "right_children": right,
                      "split_conditions": cond, "split_indices": feat, "split_type": [0] * n,
                      "sum_hessian": [float(rng.integers(1, 100)) for _ in range(n)],
                      "tree_param": {"num_deleted": "0", "num_feature": str(nfeat), "num_nodes": str(n),
                                     "size_leaf_vector": "0"}})
    doc = {"learner": {"attributes": {}, "feature_names": [], "feature_types": [],
                       "gradient_booster": {"model": {"gbtree_model_param": {"num_parallel_tree": "1",
                                                                             "num_trees": str(ntree),
                                                                             "size_leaf_vector": "0"},
                                                      "tree_info": [0] * ntree, "trees": trees}, "name": "gbtree"},
                       "learner_model_param": {"base_score": "%.9g" % float(np.float32(rng.normal(0, 1))),
                                               "num_class": "0", "num_feature": str(nfeat), "num_target": "1"},
                       "objective": {"name": "reg:squarederror", "reg_loss_param": {"scale_pos_weight": "1"}}},
           "version": [1, 6, 0]}
    return json.dumps(doc).encode()


def random_rows(rng, n, nfeat):
    rows = rng.normal(0, 2, (n, nfeat)).astype(np.float32)
    salt = rng.random(rows.shape)
    rows[salt < 0.10] = rng.choice(SPECIAL, int((salt < 0.10).sum()))
    rows[(salt >= 0.10) & (salt < 0.13)] = np.nan
    rows[(salt >= 0.13) & (salt < 0.16)] = -999.0
    return rows


CASES = [(1, 1, 0, 0.0), (3, 3, 1, 0.0), (7, 27, 6, 0.3), (40, 27, 12, 0.15), (5, 31, 9, 0.05), (9, 5, 30, 0.45),
         (4, 32, 5, 0.2), (3, 40, 4, 0.2),
         (150, 27, 7, 0.25)]   # more trees than the kernels' first-step table holds (128)


@pytest.mark.parametrize("ntree,nfeat,depth,p_leaf", CASES)
def test_oracles_agree_on_random_boosters(ntree, nfeat, depth, p_leaf):
    rng = np.random.default_rng(ntree * 1000 + nfeat)
    js = random_booster_json(rng, ntree, nfeat, depth, p_leaf)
    rows = random_rows(rng, 3000, nfeat)
    model = O.load_model(js)
    binary = synth.convert_model(js, "binary")
    for missing in (-999.0, float("nan")):
        a = O.predict(model, rows, missing=missing)
        b = helpers.oracle_predict(binary, rows, missing)
        assert np.array_equal(helpers.bits(a), helpers.bits(b))
    assert np.array_equal(O.predict(model, rows, missing=-999.0, pred_leaf=True),
                          helpers.oracle_predict(binary, rows, -999.0, option_mask=16).reshape(len(rows), -1))


@pytest.mark.parametrize("ntree,nfeat,depth,p_leaf", CASES)
def test_super_node_layout_walks_like_the_oracle(ntree, nfeat, depth, p_leaf):
    """The layout the default kernel reads (16-byte super-nodes, fillers instead of a finished state, a fixed
    trip count per tree, start nodes in group 1), walked on the host the kernel's way: same margins as the
    oracle, bit for bit, for every shape of tree - or an honest "does not fit" (more than 31 features)."""
    rng = np.random.default_rng(ntree * 1000 + nfeat)
    js = random_booster_json(rng, ntree, nfeat, depth, p_leaf)
    rows = random_rows(rng, 3000, nfeat)
    model = O.load_model(js)
    for missing in (-999.0, float("nan")):
        got, info = synth.super_walk_cpu(js, rows, missing)
        if nfeat > 31:
            assert got is None
            continue
        assert np.array_equal(helpers.bits(got), helpers.bits(O.predict(model, rows, missing=missing))), missing
        assert info["steps"] >= ntree and info["super_nodes"] >= 8 * ntree
    if nfeat <= 31 and nfeat > 2:
        # fewer columns than features: the absent ones are missing
        got, _ = synth.super_walk_cpu(js, rows[:, :nfeat - 2], -999.0)
        assert np.array_equal(helpers.bits(got), helpers.bits(O.predict(model, rows[:, :nfeat - 2], missing=-999.0)))


def test_super_node_layout_on_the_oh_boosters(small_model, deep_model):
    for m in (small_model, deep_model):
        rows = synth.rows_cpu(synth.GRIDS["C12"], 100, 4000)
        rows[::17, 3] = np.nan
        rows[5::29, 1] = synth.XX_MISS
        got, info = synth.super_walk_cpu(m.image, rows)
        assert np.array_equal(helpers.bits(got), helpers.bits(helpers.oracle_predict(m.image, rows, synth.XX_MISS)))
        # depth-capped trees start below the root (phase 1): depth 10 -> 5 steps, depth 18 -> 9
        assert info["phase1_trees"] == m.num_trees and info["steps"] == m.num_trees * (m.max_depth // 2)


@pytest.mark.gpu
@pytest.mark.parametrize("ntree,nfeat,depth,p_leaf", CASES)
def test_gpu_kernels_on_random_boosters(ntree, nfeat, depth, p_leaf):
    rng = np.random.default_rng(ntree * 1000 + nfeat)
    js = random_booster_json(rng, ntree, nfeat, depth, p_leaf)
    rows = random_rows(rng, 5000, nfeat)
    model = O.load_model(js)
    for missing in (-999.0, float("nan")):
        want = O.predict(model, rows, missing=missing)
        for kernel in ("auto", "wide", "packed1", "packed2", "packed4", "super1", "super2", "super4"):
            b = capi.Booster(model_buffer=js)
            b.set_param("ohx_kernel", kernel)
            got = b.predict(capi.DMatrix(rows, missing=missing))
            assert np.array_equal(helpers.bits(got), helpers.bits(want)), (kernel, missing)
        # the clustering pass on trees of every shape (stumps, leaves at any depth, roots evaluated from the head or
        # not): a key from the top of up to four trees, any number of steps - same margins
        for trees, steps, z in ((1, 1, 0), (2, 7, 0), (4, 3, 1), (3, 16, 0)):
            b = capi.Booster(model_buffer=js)
            for k, v in (("ohx_cluster", "on"), ("ohx_cluster_trees", trees), ("ohx_cluster_steps", steps), ("ohx_cluster_zorder", z)):
                b.set_param(k, v)
            d = capi.DMatrix(rows, missing=missing)
            d.set_grid(0, 0, 0)
            got = b.predict(d)
            assert np.array_equal(helpers.bits(got), helpers.bits(want)), ("cluster", trees, steps, missing)
    b = capi.Booster(model_buffer=js)
    leaves = b.predict(capi.DMatrix(rows, missing=-999.0), option_mask=16).reshape(len(rows), -1)
    assert np.array_equal(leaves, O.predict(model, rows, missing=-999.0, pred_leaf=True))
