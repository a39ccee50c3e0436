#!/usr/bin/env python3
"""Tag look-ups per gather of the tree walk, predicted without a GPU (profiles/r05_sweeps.txt, "another order of a tile's
rows among the lanes").  Lives under tests/ because it walks the trees with the ORACLE (tests may; tools and the product
may not): 300 random 4 x 4 x 4 bricks of the synthetic C360 batch through the 100-tree booster, level by level; a gather
of super-step s costs one tag look-up per distinct 64-byte block among the four lanes of a quad, and a block is the four
child super-nodes of one parent (the node the lane stood on one super-step earlier: level 2 (s - 2)).  Prints the mean look-ups per gather of steps 5..9 for several orders of a tile's rows among
the lanes.  The counter TCP_TOTAL_CACHE_ACCESSES says 38.4 for the shipped order; this says 39.1.
usage: python3 tests/analysis/quad_lookups.py      (CPU only, about a minute)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from quickchem_amd import synth  # noqa: E402
import xgb_oracle as O  # noqa: E402

t0 = time.time()
model_s = synth.make_model(num_trees=100, max_depth=18, sample_log2=20)
model = O.load_model(model_s.image.tobytes())
print("model", model.total_nodes, time.time() - t0, flush=True)
grid = synth.GRIDS["C360"]
im, jm, km = grid
rng = np.random.default_rng(1)
NT = 300
tiles = []
rows = np.empty((NT, 4, 4, 4, 27), dtype=np.float32)   # [tile][k][j][i]
for t in range(NT):
    i0 = 4 * rng.integers(0, im // 4); j0 = 4 * rng.integers(0, jm // 4); k0 = 4 * rng.integers(5, km // 4)   # lower 52 levels
    for kk in range(4):
        for jj in range(4):
            m = i0 + im * ((j0 + jj) + jm * (k0 + kk))
            rows[t, kk, jj] = synth.rows_cpu(grid, m, 4)
print("rows", time.time() - t0, flush=True)
X = rows.reshape(NT * 64, 27)
X[:, 1] = X[:, 1]      # PL already /100 in rows
N = X.shape[0]
MISS = synth.XX_MISS

def walk_levels(tree, X):
    """node id per row per level 0..18 (stays at the leaf once reached; -1 marks 'finished before this level')"""
    node = np.zeros(N, dtype=np.int64)
    out = np.empty((19, N), dtype=np.int64)
    done = np.zeros(N, dtype=bool)
    for d in range(19):
        out[d] = np.where(done, -1, node)
        leaf = tree.cleft[node] < 0
        done = done | leaf
        f = tree.feature[node]
        x = X[np.arange(N), np.minimum(f, 26)]
        miss = np.isnan(x) | (x == MISS)
        go_left = np.where(miss, tree.default_left[node], x < tree.value[node])
        nxt = np.where(go_left, tree.cleft[node], tree.cright[node])
        node = np.where(leaf, node, nxt)
    return out

# lane orders: index arrays of shape (NT, 64) giving for lane l the cell (kk,jj,ii) flattened as kk*16+jj*4+ii
cells = np.arange(64)
kk, jj, ii = cells // 16, (cells // 4) % 4, cells % 4
order_k_fastest = np.lexsort((kk, ii, jj))      # k fastest, then i, then j
order_i_fastest = np.lexsort((ii, jj, kk))      # i fastest, then j, then k
def lookups(levels, order):
    """levels: (19, NT, 64) node ids in cell order; order: (64,) or (NT,64) lane -> cell.  mean over gathers (super-steps 5..9)
    of sum over quads of distinct parent-super-node ids"""
    res = []
    for s in range(5, 10):
        parent_level = 2 * (s - 2)
        if parent_level > 18: break
        blk = levels[parent_level]                      # (NT, 64)
        cur = levels[2 * (s - 1)] if 2 * (s - 1) <= 18 else None
        if order.ndim == 1:
            b = blk[:, order]
        else:
            b = np.take_along_axis(blk, order, axis=1)
        q = b.reshape(NT, 16, 4)
        qs = np.sort(q, axis=2)
        distinct = 1 + (qs[:, :, 1:] != qs[:, :, :-1]).sum(axis=2)
        res.append(distinct.sum(axis=1))               # per tile
    return np.array(res)                               # (steps, NT)

tot = {}
keys = {}
t1 = time.time()
all_levels = []
for ti, tree in enumerate(model.trees):
    lv = walk_levels(tree, X).reshape(19, NT, 64)
    all_levels.append(lv)
print("walk", time.time() - t1, flush=True)
# candidate sort keys per tile (tree-independent): by PL (feature 1), by T (2), by level kk, by first tree's node at level 8
def order_by(key):   # key (NT,64) -> lane->cell order
    return np.argsort(key, axis=1, kind="stable")
Xc = X.reshape(NT, 64, 27)
cands = {"k-fastest (shipped)": order_k_fastest, "i-fastest": order_i_fastest,
         "sorted by PL": order_by(Xc[:, :, 1]), "sorted by T": order_by(Xc[:, :, 2]),
         "sorted by SZA": order_by(Xc[:, :, 26]), "sorted by O3": order_by(Xc[:, :, 4]),
         "sorted by tree0 level-8 node": order_by(all_levels[0][8]),
         "sorted by tree0 level-12 node": order_by(all_levels[0][12])}
# a multi-tree key: concatenate first 4 trees' level-4 nodes
mk = np.zeros((NT, 64), dtype=np.int64)
for t in range(4):
    lv = all_levels[t][4]
    # rank-compress per tree
    _, inv = np.unique(lv, return_inverse=True)
    mk = mk * 64 + inv.reshape(NT, 64) % 64
cands["sorted by trees 0-3 level-4 nodes"] = order_by(mk)
for name, order in cands.items():
    acc = []
    for lv in all_levels:
        acc.append(lookups(lv, order))
    a = np.array(acc)          # (trees, steps, NT)
    print("%-36s lookups per gather, mean over trees/tiles by step 5..9: %s   all: %.1f" % (
        name, " ".join("%.1f" % v for v in a.mean(axis=(0, 2))), a.mean()), flush=True)
