"""Writes the golden fixtures in this directory.

The reference has no golden vectors for the OH predict path, so
these are the build's own:

  hand_forest.json / hand_cases.json
      A five-tree booster written by hand in XGBoost's JSON schema and the answers
      WORKED OUT BY HAND below (not produced by any predictor).  They pin: strict
      `x < cond` (a tie goes right), -999.0 and NaN both meaning "missing", both
      default directions, columns the matrix does not have, ntree_limit, leaf
      indices, and float32 accumulation IN TREE ORDER starting from base_score
      (trees 0 and 1 are +1e8 / -1e8 stumps: 0.5 + 1e8 rounds to 1e8, so every
      correct answer has lost the 0.5 - a double accumulator or any other order
      gets a different number).

  mock4x4_T100.json
      BASELINE.json config #1: the seeded synthetic booster (100 trees, depth <= 18)
      on the seeded 4x4x72 state; raw margins as float32 bit patterns, written only
      if the C oracle and the numpy oracle agree bit for bit.

Run from the repo root:  python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import xgb_oracle as O  # noqa: E402
from quickchem_amd import capi, synth  # noqa: E402

NAN = float("nan")


def stump(value, tid):
    return {"base_weights": [0.0], "categories": [], "categories_nodes": [], "categories_segments": [],
            "categories_sizes": [], "default_left": [0], "id": tid, "left_children": [-1], "loss_changes": [0.0],
            "parents": [2147483647], "right_children": [-1], "split_conditions": [value], "split_indices": [0],
            "split_type": [0], "sum_hessian": [1.0],
            "tree_param": {"num_deleted": "0", "num_feature": "3", "num_nodes": "1", "size_leaf_vector": "0"}}


def hand_forest():
    tree_a = {  # node0: f0 < 1.0 (default left) ; node1: f1 < -2.5 (default right)
        "base_weights": [0.0] * 5, "categories": [], "categories_nodes": [], "categories_segments": [],
        "categories_sizes": [], "default_left": [1, 0, 0, 0, 0], "id": 2,
        "left_children": [1, 3, -1, -1, -1], "loss_changes": [0.0] * 5,
        "parents": [2147483647, 0, 0, 1, 1], "right_children": [2, 4, -1, -1, -1],
        "split_conditions": [1.0, -2.5, 0.25, -1.0, 2.0], "split_indices": [0, 1, 0, 0, 0],
        "split_type": [0] * 5, "sum_hessian": [4.0, 2.0, 2.0, 1.0, 1.0],
        "tree_param": {"num_deleted": "0", "num_feature": "3", "num_nodes": "5", "size_leaf_vector": "0"}}
    tree_b = {  # node0: f2 < 0.0 (default left)
        "base_weights": [0.0] * 3, "categories": [], "categories_nodes": [], "categories_segments": [],
        "categories_sizes": [], "default_left": [1, 0, 0], "id": 3, "left_children": [1, -1, -1],
        "loss_changes": [0.0] * 3, "parents": [2147483647, 0, 0], "right_children": [2, -1, -1],
        "split_conditions": [0.0, 0.125, -0.375], "split_indices": [2, 0, 0], "split_type": [0] * 3,
        "sum_hessian": [2.0, 1.0, 1.0],
        "tree_param": {"num_deleted": "0", "num_feature": "3", "num_nodes": "3", "size_leaf_vector": "0"}}
    trees = [stump(1.0e8, 0), stump(-1.0e8, 1), tree_a, tree_b, stump(0.0625, 4)]
    return {"learner": {"attributes": {}, "feature_names": [], "feature_types": [],
                        "gradient_booster": {"model": {"gbtree_model_param": {"num_parallel_tree": "1",
                                                                              "num_trees": "5",
                                                                              "size_leaf_vector": "0"},
                                                       "tree_info": [0, 0, 0, 0, 0], "trees": trees},
                                             "name": "gbtree"},
                        "learner_model_param": {"base_score": "5E-1", "num_class": "0", "num_feature": "3",
                                                "num_target": "1"},
                        "objective": {"name": "reg:squarederror", "reg_loss_param": {"scale_pos_weight": "1"}}},
            "version": [1, 6, 0]}


# rows, missing = -999.0.  After trees 0,1: (0.5 + 1e8) - 1e8 = 0.0 in float32.
HAND_ROWS = [
    [0.5, -3.0, -1.0],      # A: 0.5<1 -> n1; -3<-2.5 -> n3 (-1.0).  B: -1<0 -> n1 (0.125)
    [1.0, 0.0, 0.0],        # A: 1.0<1.0 false -> n2 (0.25).        B: 0<0 false -> n2 (-0.375)
    [-999.0, -2.5, NAN],    # A: f0 missing -> default left n1; -2.5<-2.5 false -> n4 (2.0).  B: NaN -> left n1
    [0.0, NAN, -999.0],     # A: n1; f1 missing -> default right n4 (2.0).  B: missing -> left n1 (0.125)
    [2.0, 5.0, 3.0],        # A: n2 (0.25).  B: 3<0 false -> n2 (-0.375)
]
#  0 + leafA + leafB + 0.0625, all exactly representable
HAND_MARGIN = [-1.0 + 0.125 + 0.0625, 0.25 - 0.375 + 0.0625, 2.0 + 0.125 + 0.0625, 2.0 + 0.125 + 0.0625,
               0.25 - 0.375 + 0.0625]
HAND_MARGIN_NTREE2 = [0.0] * 5
HAND_MARGIN_NTREE3 = [-1.0, 0.25, 2.0, 2.0, 0.25]
HAND_LEAVES = [[0, 0, 3, 1, 0], [0, 0, 2, 2, 0], [0, 0, 4, 1, 0], [0, 0, 4, 1, 0], [0, 0, 2, 2, 0]]
# two-column matrix: feature 2 is absent -> missing -> tree B goes default left (0.125)
HAND_ROWS_2COL = [[0.5, -3.0], [1.0, 0.0], [2.0, 5.0]]
HAND_MARGIN_2COL = [-1.0 + 0.125 + 0.0625, 0.25 + 0.125 + 0.0625, 0.25 + 0.125 + 0.0625]


def main():
    with open(os.path.join(HERE, "hand_forest.json"), "w") as f:
        json.dump(hand_forest(), f, indent=1)
    cases = {"missing": -999.0,
             "rows": [[None if x != x else x for x in r] for r in HAND_ROWS],     # null = NaN
             "margin": HAND_MARGIN, "margin_ntree_limit_2": HAND_MARGIN_NTREE2,
             "margin_ntree_limit_3": HAND_MARGIN_NTREE3, "leaf_index": HAND_LEAVES,
             "rows_2col": HAND_ROWS_2COL, "margin_2col": HAND_MARGIN_2COL}
    with open(os.path.join(HERE, "hand_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)

    # ---- config #1: seeded booster x seeded 4x4x72 state ----
    grid = synth.GRIDS["mock4x4"]
    params = dict(num_trees=100, max_depth=18, sample_log2=16, min_leaf=2, grid=synth.GRIDS["C12"])
    model = synth.make_model(**params)
    n = grid[0] * grid[1] * grid[2]
    rows = synth.rows_cpu(grid, 0, n)
    p_np = O.predict(O.load_model(model.image.tobytes()), rows, missing=synth.XX_MISS)
    olib = capi.declare_xgb_api(C.CDLL(os.path.join(ROOT, "oracle", "lib", "liboracle_xgb.so")))
    ob = capi.Booster(model_buffer=model.image, lib=olib)
    p_c = ob.predict(capi.DMatrix(rows, missing=synth.XX_MISS, lib=olib))
    assert np.array_equal(p_np.view(np.uint32), p_c.view(np.uint32)), "the two oracles disagree"
    out = {"grid": list(grid), "model": {k: (list(v) if isinstance(v, tuple) else v) for k, v in params.items()},
           "model_seed": synth.MODEL_SEED, "feature_seed": synth.FEATURE_SEED,
           "model_nodes": model.num_nodes, "model_max_depth": model.max_depth,
           "rows_crc": int(np.bitwise_xor.reduce(rows.view(np.uint32).ravel())),
           "margin_bits": [int(x) for x in p_c.view(np.uint32)]}
    with open(os.path.join(HERE, "mock4x4_T100.json"), "w") as f:
        json.dump(out, f)
    print("golden fixtures written:", model.num_nodes, "nodes, depth", model.max_depth)


if __name__ == "__main__":
    main()
