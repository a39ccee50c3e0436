"""Shared test plumbing: library handles, oracle calls, Fortran driver files."""
import ctypes as C
import json
import os
import struct
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_SO = os.path.join(ROOT, "oracle", "lib", "liboracle_xgb.so")
PRODUCT_SO = os.path.join(ROOT, "quickchem_amd", "lib", "libohxgb.so")
SYNTH_SO = os.path.join(ROOT, "quickchem_amd", "lib", "libohx_synth.so")
DRIVER_HIP = os.path.join(ROOT, "quickchem_amd", "lib", "oh_mock_driver_hip")
DRIVER_ORACLE = os.path.join(ROOT, "oracle", "lib", "oh_mock_driver_oracle")
DROPIN_HIP = os.path.join(ROOT, "oracle", "_ref", "dropin_driver_hip")
DROPIN_ORACLE = os.path.join(ROOT, "oracle", "_ref", "dropin_driver_oracle")

_oracle = None


def ensure_built():
    need = [ORACLE_SO, PRODUCT_SO, SYNTH_SO, SYNTH_SO.replace("libohx_synth.so", "libohx_synth_gpu.so"), DRIVER_HIP, DRIVER_ORACLE]
    if all(os.path.exists(p) for p in need):
        return
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()


def oracle_lib():
    """The CPU oracle, with the same argtypes as the product library."""
    global _oracle
    if _oracle is None:
        from quickchem_amd import capi
        lib = capi.declare_xgb_api(C.CDLL(ORACLE_SO))
        lib.oracle_predict_OH_with_XGB.argtypes = [
            C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p,
            C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.oracle_set_num_threads.argtypes = [C.c_int]
        lib.oracle_num_threads.restype = C.c_int
        _oracle = lib
    return _oracle


def oracle_predict(model_image, rows, missing, option_mask=0, ntree_limit=0):
    from quickchem_amd import capi
    lib = oracle_lib()
    b = capi.Booster(model_buffer=model_image, lib=lib)
    d = capi.DMatrix(rows, missing=missing, lib=lib)
    out = b.predict(d, option_mask=option_mask, ntree_limit=ntree_limit)
    d.free()
    b.free()
    return out


def fortran_flat(a):
    return np.ascontiguousarray(a.T, dtype=np.float32)


def oracle_predict_oh(model_image, pl, tropp, fields, dynamic_k_range, tropp_min=4000.0):
    """oracle_predict_OH_with_XGB (C) on [i,j,k]-indexed arrays -> (oh_ml[i,j,k], margin, k1, k2)."""
    from quickchem_amd import capi, synth
    lib = oracle_lib()
    b = capi.Booster(model_buffer=model_image, lib=lib)
    im, jm, km = pl.shape
    flat = [fortran_flat(a) for a in fields]
    ptrs = (C.c_void_p * 27)(*[f.ctypes.data for f in flat])
    is2d = (C.c_int32 * 27)(*[1 if x else 0 for x in synth.IS2D])
    oh = np.zeros(im * jm * km, dtype=np.float32)
    margin = np.zeros(im * jm * km, dtype=np.float32)
    k1, k2 = C.c_int(), C.c_int()
    plf, trf = fortran_flat(pl), fortran_flat(tropp)
    rc = lib.oracle_predict_OH_with_XGB(b.handle, im, jm, km, 1 if dynamic_k_range else 0, tropp_min, plf.ctypes.data,
                                        trf.ctypes.data, ptrs, is2d, oh.ctypes.data, margin.ctypes.data,
                                        C.byref(k1), C.byref(k2))
    if rc != 0:
        raise RuntimeError(lib.XGBGetLastError().decode())
    n = im * jm * (k2.value - k1.value + 1)
    b.free()
    return oh.reshape(km, jm, im).transpose(2, 1, 0), margin[:n].copy(), k1.value, k2.value


def synth_state(grid, seed=None):
    """(pl[Pa], tropp[Pa], fields[27]) of the synthetic MAPL-like state, [i,j,k]-indexed."""
    from quickchem_amd import synth
    seed = synth.FEATURE_SEED if seed is None else seed
    fields = [np.ascontiguousarray(synth.field_cpu(grid, f, seed)) for f in range(27)]
    pl = fields[1]
    tropp = np.ascontiguousarray(synth.field_cpu(grid, -1, seed))
    return pl, tropp, fields


def write_state_file(path, pl, tropp, fields, dynamic_k_range, tropp_min=4000.0, ohscale=1.0):
    im, jm, km = pl.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<iiiiff", im, jm, km, 1 if dynamic_k_range else 0, tropp_min, ohscale))
        f.write(fortran_flat(pl).tobytes())
        f.write(fortran_flat(tropp).tobytes())
        for a in fields:
            f.write(fortran_flat(a).tobytes())


def read_driver_output(path, im, jm, km):
    raw = open(path, "rb").read()
    rc, k1, k2 = struct.unpack_from("<iii", raw, 0)
    oh = np.frombuffer(raw, dtype="<f4", count=im * jm * km, offset=12).reshape(km, jm, im).transpose(2, 1, 0)
    (seconds,) = struct.unpack_from("<d", raw, 12 + 4 * im * jm * km)
    return rc, k1, k2, np.ascontiguousarray(oh), seconds


def run_driver(exe, *args, env=None, timeout=600):
    e = dict(os.environ)
    if env:
        e.update(env)
    return subprocess.run([exe, *map(str, args)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e,
                          timeout=timeout)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def ulp_diff(a, b):
    """Distance in float32 ulps between two arrays of same-signed finite floats."""
    return np.abs(bits(a).astype(np.int64) - bits(b).astype(np.int64))


def load_hand_cases():
    cases = json.load(open(os.path.join(GOLDEN, "hand_cases.json")))
    rows = np.array([[np.nan if x is None else x for x in r] for r in cases["rows"]], dtype=np.float32)
    return cases, rows


def run1_state(grid, seed=17):
    """A synthetic OH import state for OHXBoosterRun1 (quickchem_amd.synth.run1_state)."""
    from quickchem_amd import synth
    return synth.run1_state(grid, seed)


RUN1_DRIVER_HIP = os.path.join(ROOT, "quickchem_amd", "lib", "oh_run1_driver_hip")
RUN1_DRIVER_ORACLE = os.path.join(ROOT, "oracle", "lib", "oh_run1_driver_oracle")
# the order quickchem_amd/fortran/oh_run1_driver.F90 reads the state in
RUN1_FILE_ORDER = ["ple_mod", "t_mod", "q_mod", "tropp_mod", "ple_bst", "zle_bst", "tauclw", "taucli", "scacoef",
                   "gmito3", "gmitto3", "lat_deg", "t_bst", "no2", "o3", "ch4", "co", "isop", "acet", "c2h6", "c3h8",
                   "prpe", "alk4", "mp", "h2o2", "cloud", "qv", "albuv", "ch2o", "sza", "default_oh"]


def write_run1_state_file(path, st, dynamic_k_range, tropp_min=4000.0, ohscale=0.85, avogad=6.023e26, runiv=8314.47,
                          epsilon=18.015 / 28.965):
    im, jm, km = st["t_mod"].shape
    with open(path, "wb") as f:
        f.write(struct.pack("<iiiifffff", im, jm, km, 1 if dynamic_k_range else 0, tropp_min, ohscale, avogad, runiv,
                            epsilon))
        for name in RUN1_FILE_ORDER:
            for a in (st[name] if name == "scacoef" else [st[name]]):
                f.write(fortran_flat(a).tobytes())


def read_run1_output(path, im, jm, km):
    raw = open(path, "rb").read()
    rc, k1, k2 = struct.unpack_from("<iii", raw, 0)
    n = im * jm * km
    out = [np.ascontiguousarray(np.frombuffer(raw, dtype="<f4", count=n, offset=12 + 4 * n * q)
                                .reshape(km, jm, im).transpose(2, 1, 0)) for q in range(3)]
    return rc, k1, k2, out[0], out[1], out[2]


def legacy_image(doc, *, version=(1, 6), binf=False, attributes=None, metrics=None, objective=None,
                 max_delta_step=None, trailer=None, num_deleted_lie=0, base_score=None):
    """An XGBoost legacy-binary model image written HERE, field by field, from a document in XGBoost's JSON
    schema - independent of the product's writer (csrc/forest_io.cpp) and of both oracles' readers.  Layout as
    published for xgboost 1.6.0: LearnerModelParamLegacy (136 B), objective and booster
    names (uint64 length + bytes), GBTreeModelParam (160 B), per tree TreeParam (148 B) + num_nodes x Node
    (20 B) + num_nodes x RTreeNodeStat (16 B), tree_info, then - in this order, learner.cc LearnerIO::Save -
    attributes, count:poisson's max_delta_step, metric names.  `version=(0, 0)` is what xgboost < 1.0 wrote
    (the reference's XGBoost_0.81_File, OH_instance_OH.rc:18): those words were still `reserved`."""
    learner = doc["learner"]
    lmp = learner["learner_model_param"]
    trees = learner["gradient_booster"]["model"]["trees"]
    obj = (objective or learner["objective"]["name"]).encode()
    attributes = attributes or []
    metrics = metrics or []

    def s(b):
        return struct.pack("<Q", len(b)) + b
    out = bytearray(b"binf" if binf else b"")
    out += struct.pack("<fIiiiIII", float(lmp["base_score"]) if base_score is None else base_score,
                       int(lmp["num_feature"]), int(lmp["num_class"]), 1 if attributes else 0, 1 if metrics else 0,
                       version[0], version[1], 1 if version[0] >= 1 else 0) + bytes(26 * 4)
    out += s(obj) + s(b"gbtree")
    out += struct.pack("<iiiiqii", len(trees), 1, int(lmp["num_feature"]), 0, 0, 1, 0) + bytes(32 * 4)
    for t in trees:
        n = len(t["left_children"])
        deleted = [int(x) == 0xFFFFFFFF for x in t["split_indices"]]
        out += struct.pack("<6i", 1, n, sum(deleted[1:]) + num_deleted_lie, 0, int(t["tree_param"]["num_feature"]), 0)
        out += bytes(31 * 4)
        is_left = {int(c): True for c in t["left_children"] if c != -1}
        for i in range(n):
            par = int(t["parents"][i])
            parent = -1 if par == 2147483647 else (par | (0x80000000 if is_left.get(i) else 0))
            sindex = 0xFFFFFFFF if deleted[i] else (int(t["split_indices"][i]) | (int(t["default_left"][i]) << 31))
            out += struct.pack("<IiiIf", parent & 0xFFFFFFFF, int(t["left_children"][i]), int(t["right_children"][i]),
                               sindex, float(t["split_conditions"][i]))
        for i in range(n):
            out += struct.pack("<fffi", float(t["loss_changes"][i]), float(t["sum_hessian"][i]),
                               float(t["base_weights"][i]), 0)
    out += struct.pack(f"<{len(trees)}i", *[0] * len(trees))
    if trailer is not None:
        return bytes(out + trailer)
    if attributes:
        out += struct.pack("<Q", len(attributes))
        for k, v in attributes:
            out += s(k.encode()) + s(v.encode())
    if obj == b"count:poisson":
        out += s((max_delta_step or "0.7").encode())
    if metrics:
        out += struct.pack("<Q", len(metrics))
        for m in metrics:
            out += s(m.encode())
    return bytes(out)
