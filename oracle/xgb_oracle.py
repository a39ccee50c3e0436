"""numpy restatement of the OH XGBoost-predict path.  TEST INFRASTRUCTURE.

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may
import this; the product never does.

PARITY UNPINNED: the reference (GEOS-ESM/QuickChem @ 2024-11-08) has no tests,
golden vectors or model files for this path and delegates the arithmetic to
xgboost 1.6.0 (reference ``Shared/CMakeLists.txt:8``), which is not vendored and
not installable here.  This module restates the published xgboost 1.6.0
semantics independently of ``oracle/xgb_oracle.c`` (different language, level
synchronous instead of row-at-a-time, its own file parsers) so that the two can
be checked against each other bit for bit, and against the hand-computed vectors
in ``tests/golden/``.

Restated pieces, with the upstream file each follows:
  * legacy binary model layout   - xgboost src/learner.cc, src/gbm/gbtree_model.cc,
                                   src/tree/tree_model.cc
  * JSON model layout            - xgboost doc/model.schema (1.6.0)
  * dense matrix semantics       - src/data/adapter.h + src/data/data.cc: an entry
                                   is missing iff NaN or == ``missing``; +-inf is an
                                   error unless ``missing`` is inf
  * prediction                   - src/predictor/cpu_predictor.cc +
                                   src/predictor/predict_fn.h:
                                   pred = base_score; for each tree in order
                                   pred += leaf; step = default child if missing
                                   else ``cleft + !(fvalue < split_cond)``
  * predict_OH_with_XGB RUN part - reference OH_GridComp/OH_GridCompMod.F90:275-383
"""
from __future__ import annotations

import json
import struct
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

NODE_DTYPE = np.dtype([("parent", "<i4"), ("cleft", "<i4"), ("cright", "<i4"),
                       ("sindex", "<u4"), ("info", "<f4")])
assert NODE_DTYPE.itemsize == 20

FEATURE_NAMES = ["LAT", "PL", "T", "NO2", "O3", "CH4", "CO", "ISOP", "ACET", "C2H6", "C3H8",
                 "PRPE", "ALK4", "MP", "H2O2", "TAUCLWDN", "TAUCLIDN", "TAUCLIUP", "TAUCLWUP",
                 "CLOUD", "QV", "GMISTRATO3", "ALBUV", "AODUP", "AODDN", "CH2O", "SZA"]
IS2D = [n in ("LAT", "GMISTRATO3", "ALBUV", "SZA") for n in FEATURE_NAMES]
XX_MISS = np.float32(-999.0)          # OH_GridCompMod.F90:213


@dataclass
class OracleTree:
    cleft: np.ndarray      # int32, -1 => leaf
    cright: np.ndarray     # int32
    feature: np.ndarray    # uint32
    default_left: np.ndarray  # bool
    value: np.ndarray      # float32: split_cond or leaf value


@dataclass
class OracleModel:
    base_score: np.float32   # the margin predictions start from: ProbToMargin(file's base_score)
    num_feature: int
    objective: str
    trees: List[OracleTree] = field(default_factory=list)

    @property
    def total_nodes(self) -> int:
        return int(sum(len(t.cleft) for t in self.trees))


# --------------------------------------------------------------------- loaders

_IDENTITY = ("reg:squarederror", "reg:linear", "reg:squaredlogerror", "reg:pseudohubererror", "reg:absoluteerror")
_LOGIT = ("reg:logistic", "binary:logistic", "binary:logitraw")
_LOG = ("count:poisson", "reg:gamma", "reg:tweedie", "survival:cox", "survival:aft")


def prob_to_margin(objective: str, base_score) -> np.float32:
    """ObjFunction::ProbToMargin of xgboost 1.6.0 by objective name (regression_loss.h: logistic losses
    -log(1/p - 1); regression_obj.cu / aft_obj.cu: log-link objectives log(p); otherwise identity).
    learner.cc starts every margin from it (LearnerConfiguration::ConfigureModelParam)."""
    b = np.float32(base_score)
    if objective in _IDENTITY or objective == "binary:hinge" or objective.startswith("rank:"):
        return b
    if objective in _LOGIT:
        if not (0.0 < float(b) < 1.0):
            raise ValueError("base_score must be in (0,1) for logistic loss")
        return np.float32(-np.log(np.float32(1.0) / b - np.float32(1.0), dtype=np.float32))
    if objective in _LOG:
        return np.float32(np.log(b, dtype=np.float32))
    raise ValueError(f"oracle: ProbToMargin of objective {objective!r} is not restated")


def _read_str(buf: bytes, off: int) -> Tuple[str, int]:
    (n,) = struct.unpack_from("<Q", buf, off)
    off += 8
    return buf[off:off + n].decode("utf-8", "replace"), off + n


def load_legacy_binary(buf: bytes) -> OracleModel:
    off = 4 if buf[:4] == b"binf" else 0
    base_score, num_feature, _num_class, _extra, _metrics, _maj, _min, _ntarget = struct.unpack_from(
        "<fIiiiIII", buf, off)
    off += 136
    objective, off = _read_str(buf, off)
    booster, off = _read_str(buf, off)
    if booster != "gbtree":
        raise ValueError("oracle: only gbtree is restated")
    (num_trees,) = struct.unpack_from("<i", buf, off)
    off += 160
    # a binary file written by xgboost < 1.0 already holds the transformed value (LearnerIO::Load)
    start = np.float32(base_score) if _maj < 1 else prob_to_margin(objective, base_score)
    model = OracleModel(start, int(num_feature), objective)
    for _ in range(num_trees):
        tp = struct.unpack_from("<37i", buf, off)
        off += 148
        n = tp[1]
        nodes = np.frombuffer(buf, dtype=NODE_DTYPE, count=n, offset=off)
        off += 20 * n + 16 * n
        sindex = nodes["sindex"]
        model.trees.append(OracleTree(
            cleft=nodes["cleft"].astype(np.int32),
            cright=nodes["cright"].astype(np.int32),
            feature=(sindex & np.uint32(0x7FFFFFFF)).astype(np.uint32),
            default_left=(sindex >> np.uint32(31)).astype(bool),
            value=nodes["info"].astype(np.float32)))
    info = np.frombuffer(buf, dtype="<i4", count=num_trees, offset=off)
    if np.any(info != 0):
        raise ValueError("oracle: multi-group boosters are not restated")
    return model


def load_json(text) -> OracleModel:
    if isinstance(text, (bytes, bytearray)):
        text = bytes(text).rstrip(b"\0").decode("utf-8")
    doc = json.loads(text)
    learner = doc["learner"]
    lmp = learner["learner_model_param"]
    gb = learner["gradient_booster"]
    if gb["name"] != "gbtree":
        raise ValueError("oracle: only gbtree is restated")
    model = OracleModel(prob_to_margin(learner["objective"]["name"], float(lmp["base_score"])),
                        int(lmp["num_feature"]), learner["objective"]["name"])
    for jt in gb["model"]["trees"]:
        model.trees.append(OracleTree(
            cleft=np.asarray(jt["left_children"], dtype=np.int32),
            cright=np.asarray(jt["right_children"], dtype=np.int32),
            feature=np.asarray(jt["split_indices"], dtype=np.uint32),
            default_left=np.asarray(jt["default_left"]).astype(bool),
            value=np.asarray(jt["split_conditions"], dtype=np.float64).astype(np.float32)))
    if any(g != 0 for g in gb["model"]["tree_info"]):
        raise ValueError("oracle: multi-group boosters are not restated")
    return model


def _ubj_value(buf: bytes, off: int, t: int):
    """One UBJSON (draft 12) value of type marker `t` starting at `off`; returns (value, new offset)."""
    ints = {ord("i"): ">b", ord("U"): ">B", ord("I"): ">h", ord("l"): ">i", ord("L"): ">q"}
    if t in ints:
        fmt = ints[t]
        return struct.unpack_from(fmt, buf, off)[0], off + struct.calcsize(fmt)
    if t == ord("d"):
        return float(np.frombuffer(buf, dtype=">f4", count=1, offset=off)[0]), off + 4
    if t == ord("D"):
        return struct.unpack_from(">d", buf, off)[0], off + 8
    if t in (ord("T"), ord("F"), ord("Z")):
        return {ord("T"): True, ord("F"): False, ord("Z"): None}[t], off
    if t == ord("C"):
        return chr(buf[off]), off + 1
    if t in (ord("S"), ord("H")):
        n, off = _ubj_value(buf, off + 1, buf[off])
        return buf[off:off + n].decode(), off + n
    if t in (ord("["), ord("{")):
        elem, count = None, None
        if buf[off] == ord("$"):
            elem = buf[off + 1]
            off += 2
        if buf[off] == ord("#"):
            count, off = _ubj_value(buf, off + 2, buf[off + 1])
        if t == ord("["):
            if elem is not None and elem in (ord("d"), ord("l"), ord("U"), ord("L"), ord("i"), ord("I"), ord("D")):
                dt = {ord("d"): ">f4", ord("l"): ">i4", ord("U"): "u1", ord("L"): ">i8", ord("i"): "i1", ord("I"): ">i2",
                      ord("D"): ">f8"}[elem]
                arr = np.frombuffer(buf, dtype=dt, count=count, offset=off)
                return arr, off + arr.nbytes
            out = []
            while count is None or len(out) < count:
                et = elem if elem is not None else buf[off]
                if elem is None:
                    off += 1
                if count is None and et == ord("]"):
                    break
                v, off = _ubj_value(buf, off, et)
                out.append(v)
            return out, off
        out = {}
        while count is None or len(out) < count:
            kt = buf[off]
            off += 1
            if count is None and kt == ord("}"):
                break
            n, off = _ubj_value(buf, off, kt)
            key = buf[off:off + n].decode()
            off += n
            vt = elem if elem is not None else buf[off]
            if elem is None:
                off += 1
            out[key], off = _ubj_value(buf, off, vt)
        return out, off
    raise ValueError(f"oracle: unknown UBJSON marker {t!r}")


def load_ubjson(buf: bytes) -> OracleModel:
    doc, _ = _ubj_value(buf, 1, buf[0])
    learner = doc["learner"]
    lmp, gb = learner["learner_model_param"], learner["gradient_booster"]
    model = OracleModel(prob_to_margin(learner["objective"]["name"], float(lmp["base_score"])),
                        int(lmp["num_feature"]), learner["objective"]["name"])
    for jt in gb["model"]["trees"]:
        model.trees.append(OracleTree(
            cleft=np.asarray(jt["left_children"]).astype(np.int32),
            cright=np.asarray(jt["right_children"]).astype(np.int32),
            feature=np.asarray(jt["split_indices"]).astype(np.uint32),
            default_left=np.asarray(jt["default_left"]).astype(bool),
            value=np.asarray(jt["split_conditions"]).astype(np.float32)))
    return model


def load_model(buf: bytes) -> OracleModel:
    if buf[:1] == b"{":
        return load_ubjson(buf) if buf[1:2] in (b"L", b"l", b"I", b"U", b"i", b"$", b"#") else load_json(buf)
    return load_legacy_binary(buf)


# --------------------------------------------------------------------- prediction

def check_dense(data: np.ndarray, missing) -> None:
    """data.cc SparsePage::Push: inf is refused unless `missing` is inf."""
    if not np.isinf(np.float32(missing)) and np.isinf(data).any():
        raise ValueError("Input data contains `inf` or `nan`")


def leaf_indices(tree: OracleTree, rows: np.ndarray, missing, num_feature: int) -> np.ndarray:
    """Level-synchronous walk of one tree for all rows; returns the leaf node ids."""
    n, ncol = rows.shape
    nid = np.zeros(n, dtype=np.int64)
    active = np.nonzero(tree.cleft[nid] != -1)[0]
    missing = np.float32(missing)
    while active.size:
        cur = nid[active]
        f = tree.feature[cur].astype(np.int64)
        have = (f < ncol) & (f < num_feature)
        fv = np.zeros(active.size, dtype=np.float32)
        fv[have] = rows[active[have], f[have]]
        miss = ~have | np.isnan(fv) | (fv == missing)
        step_default = np.where(tree.default_left[cur], tree.cleft[cur], tree.cright[cur])
        with np.errstate(invalid="ignore"):
            step_value = tree.cleft[cur] + (~(fv < tree.value[cur])).astype(np.int32)
        nxt = np.where(miss, step_default, step_value)
        nid[active] = nxt
        active = active[tree.cleft[nxt] != -1]
    return nid


def predict(model: OracleModel, rows: np.ndarray, missing=np.nan, ntree_limit: int = 0,
            pred_leaf: bool = False) -> np.ndarray:
    """XGBoosterPredict(option_mask in {0,1} or 16, ntree_limit) on a dense float32 matrix."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    if rows.ndim != 2:
        raise ValueError("rows must be 2-D")
    if rows.shape[1] > model.num_feature:
        raise ValueError("Number of columns does not match number of features in booster")
    check_dense(rows, missing)
    T = len(model.trees)
    tend = T if ntree_limit == 0 or ntree_limit > T else ntree_limit
    if pred_leaf:
        out = np.empty((rows.shape[0], tend), dtype=np.float32)
        for t in range(tend):
            out[:, t] = leaf_indices(model.trees[t], rows, missing, model.num_feature)
        return out
    acc = np.full(rows.shape[0], model.base_score, dtype=np.float32)
    for t in range(tend):                      # tree order matters: float32 adds do not commute
        leaf = leaf_indices(model.trees[t], rows, missing, model.num_feature)
        acc = (acc + model.trees[t].value[leaf]).astype(np.float32)
    return acc


# --------------------------------------------------------------------- the Fortran caller

def k_slab(pl: np.ndarray, tropp: np.ndarray, dynamic_k_range: bool, tropp_min: float) -> Tuple[int, int]:
    """OH_GridCompMod.F90:275-301.  pl is (im,jm,km) [i,j,k], tropp (im,jm); returns 1-based (k1,k2)."""
    km = pl.shape[2]
    if dynamic_k_range:
        counts = (pl > tropp[:, :, None]).sum(axis=2)
    else:
        if np.count_nonzero(tropp <= np.float32(tropp_min)):
            raise AssertionError("OH Prediction: Minimum tropopause pressure is not low enough!")
        counts = (pl > np.float32(tropp_min)).sum(axis=2)
    ksub = int(counts.max()) if counts.size else 0
    return km - ksub + 1, km


def gather_rows(fields: Sequence[np.ndarray], k1: int, k2: int) -> np.ndarray:
    """OH_GridCompMod.F90:303-345: xx_carr(27, N) as a C [N][27] array; i fastest, then j, then k."""
    im, jm = fields[1].shape[:2]
    nlev = k2 - k1 + 1
    out = np.empty((nlev, jm, im, len(fields)), dtype=np.float32)
    for f, a in enumerate(fields):
        if a.ndim == 2:
            out[..., f] = a.T[None, :, :]
        else:
            sl = a[:, :, k1 - 1:k2]
            if f == 1:
                sl = (sl / np.float32(100.0)).astype(np.float32)      # :314
            out[..., f] = np.transpose(sl, (2, 1, 0))
    return out.reshape(nlev * jm * im, len(fields))


def predict_OH_with_XGB(model: OracleModel, pl: np.ndarray, tropp: np.ndarray, fields: Sequence[np.ndarray],
                        dynamic_k_range: bool, tropp_min: float = 4000.0,
                        oh_ml: Optional[np.ndarray] = None):
    """RUN section of predict_OH_with_XGB (OH_GridCompMod.F90:275-383).

    Arrays are indexed [i,j,k] like the Fortran ones.  Returns (oh_ml, margin, k1, k2)."""
    im, jm, km = pl.shape
    k1, k2 = k_slab(pl, tropp, dynamic_k_range, tropp_min)
    rows = gather_rows(fields, k1, k2)
    margin = predict(model, rows, missing=XX_MISS)
    if oh_ml is None:
        oh_ml = np.zeros((im, jm, km), dtype=np.float32)               # :1559
    nlev = k2 - k1 + 1
    oh = np.power(np.float32(10.0), margin, dtype=np.float32)          # :369
    oh_ml[:, :, k1 - 1:k2] = np.transpose(oh.reshape(nlev, jm, im), (2, 1, 0))
    return oh_ml, margin, k1, k2


# --------------------------------------------------------------------- OH Run1, either side of the call

def run1(model: OracleModel, st: dict, dynamic_k_range: bool, tropp_min: float = 4000.0, ohscale: float = 0.85,
         avogad: float = 6.023e26, runiv: float = 8314.47, epsilon: float = 18.015 / 28.965):
    """OH Run1 from the imports to INTERNAL OH (OH_GridCompMod.F90:1240-1257, 1444-1478, 1488,
    1557-1595).  `st` maps the names of include/ohxgb.h's OHXRun1Args to [i,j(,k)] float32 arrays.
    Every SUM is accumulated from zero, ascending in the level index, one float add at a time."""
    f32 = np.float32
    km = st["t_mod"].shape[2]
    ple_mod, ple_bst, zle = (np.asarray(st[k], dtype=f32) for k in ("ple_mod", "ple_bst", "zle_bst"))
    pl_mod = ((ple_mod[:, :, :-1] + ple_mod[:, :, 1:]) * f32(0.5)).astype(f32)              # :1247
    pl_bst = ((ple_bst[:, :, :-1] + ple_bst[:, :, 1:]) * f32(0.5)).astype(f32)              # :1488
    thick = (zle[:, :, :-1] - zle[:, :, 1:]).astype(f32)                                    # :1451
    sc = (st["scacoef"][0] + st["scacoef"][1]).astype(f32)                                  # :1456-1457
    for i in range(2, 7):
        sc = (sc + st["scacoef"][i]).astype(f32)
    aod = (thick * sc).astype(f32)
    strato3 = (st["gmito3"] - st["gmitto3"]).astype(f32)                                    # :1446

    def sum_up(x):     # SUM(x(1:k))
        out = np.empty_like(x)
        acc = np.zeros(x.shape[:2], dtype=f32)
        for k in range(km):
            acc = (acc + x[:, :, k]).astype(f32)
            out[:, :, k] = acc
        return out

    def sum_dn(x):     # SUM(x(k:km)), each from zero
        out = np.empty_like(x)
        for k in range(km):
            acc = np.zeros(x.shape[:2], dtype=f32)
            for kk in range(k, km):
                acc = (acc + x[:, :, kk]).astype(f32)
            out[:, :, k] = acc
        return out

    tauclw, taucli = np.asarray(st["tauclw"], dtype=f32), np.asarray(st["taucli"], dtype=f32)
    fields = [st["lat_deg"], pl_bst, st["t_bst"], st["no2"], st["o3"], st["ch4"], st["co"], st["isop"], st["acet"],
              st["c2h6"], st["c3h8"], st["prpe"], st["alk4"], st["mp"], st["h2o2"], sum_dn(tauclw), sum_dn(taucli),
              sum_up(taucli), sum_up(tauclw), st["cloud"], st["qv"], strato3, st["albuv"], sum_up(aod), sum_dn(aod),
              st["ch2o"], st["sza"]]                                                        # :313-339
    fields = [np.asarray(a, dtype=f32) for a in fields]
    oh_ml, margin, k1, k2 = predict_OH_with_XGB(model, pl_mod, np.asarray(st["tropp_mod"], dtype=f32), fields,
                                                dynamic_k_range, tropp_min)
    oh_ml = (oh_ml * f32(ohscale)).astype(f32)                                              # :1569
    q, t = np.asarray(st["q_mod"], dtype=f32), np.asarray(st["t_mod"], dtype=f32)
    tv = ((t * (f32(1.0) + q / f32(epsilon))).astype(f32) / (f32(1.0) + q)).astype(f32)     # :1250
    ndwet = ((f32(avogad) * pl_mod).astype(f32) / (f32(runiv) * tv).astype(f32)).astype(f32)   # :1257
    ohv = np.where(pl_mod > np.asarray(st["tropp_mod"], dtype=f32)[:, :, None], oh_ml, st["default_oh"]).astype(f32)
    oh = ((ohv * ndwet).astype(f32) * f32(1.0e-6)).astype(f32)                              # :1595
    return {"oh": oh, "oh_boost": oh_ml, "ndwet": ndwet, "k1": k1, "k2": k2, "fields": fields, "margin": margin}


# ---- OH Run1's solar geometry (OH_GridCompMod.F90:401-466, 1444, 1905-1970) ----

def julian_day(nymd: int) -> int:
    ny, mm, dd = nymd // 10000, (nymd % 10000) // 100, nymd % 100
    leap = ny >= 0 and ((ny % 100 == 0 and ny % 400 == 0) or (ny % 4 == 0 and ny % 100 != 0))   # :1957-1964
    days = [31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31]
    ds = dd
    for m in range(1, mm):                                                                   # :1925-1932
        ds += 29 if (m == 2 and leap) else days[m - 1]
    return ds


def solar_geometry(jday: int, lats, lons, deg2rad, rad2deg):
    """(lat_deg, sza_noon) in float32, the reference's order of evaluation, numpy's float32 sin/cos/arcsin/arccos."""
    f = np.float32
    lats, lons = np.asarray(lats, dtype=f), np.asarray(lons, dtype=f)
    deg2rad, rad2deg = f(deg2rad), f(rad2deg)
    sindec = f(0.3978) * np.sin(f(0.9863) * (f(jday) - f(80.0)) * deg2rad, dtype=f)          # :427
    cosdec = np.cos(np.arcsin(sindec, dtype=f), dtype=f)                                      # :428-429
    sinlat = np.sin(lats, dtype=f)                                                            # :430
    coslat = np.cos(np.arcsin(sinlat, dtype=f), dtype=f)                                      # :431-432
    mylon = lons * rad2deg                                                                    # :439
    mylon = np.where(mylon > f(180.0), mylon - f(360.0), mylon)                               # :441
    mylon = np.where(mylon < f(-180.0), mylon + f(360.0), mylon)                              # :442
    tau = f(12.0) + (mylon / f(-180.0)) * f(12.0)                                             # :443
    loct = ((tau * f(15.0)) - f(180.0)) * deg2rad + lons                                      # :445
    cosz = cosdec * coslat * np.cos(loct, dtype=f) + sindec * sinlat                          # :446
    cosz = np.maximum(f(-1.0), np.minimum(f(1.0), cosz))                                      # :459-460
    return lats * rad2deg, np.arccos(cosz, dtype=f) * rad2deg                                 # :1444, :462
