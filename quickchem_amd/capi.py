"""ctypes binding of libohxgb.so (include/ohxgb.h) — plumbing only.

The names mirror the reference's Fortran binding module
(``Shared/xgb_fortran_api.F90``): the same eleven XGBoost C-API symbols, plus the
device-resident / fused extensions.  Nothing here computes; every numeric result
comes out of the HIP kernels in the shared library, and loading fails loudly if
the library has not been built.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libohxgb.so")
# the library's default of "ohx_ring_rounds" (csrc/kernels.hpp LaunchTuning::ring_rounds; tests/test_capi_host.py pins it)
RING_ROUNDS_DEFAULT = 64
RING_ROUNDS_NO_GRID = 16      # kRingRoundsNoGrid: at most, for rows not known to lie on a grid
RING_ROUNDS_PERMUTED = 4      # kRingRoundsPermuted: at most, for rows that come through the clustering pass

# every symbol include/ohxgb.h declares
ABI_SYMBOLS = [
    "XGBGetLastError", "XGDMatrixCreateFromMat", "XGDMatrixFree", "XGDMatrixNumRow", "XGDMatrixNumCol",
    "XGDMatrixSaveBinary", "XGDMatrixCreateFromFile", "XGBoosterCreate", "XGBoosterFree", "XGBoosterLoadModel",
    "XGBoosterSaveModel", "XGBoosterLoadModelFromBuffer", "XGBoosterPredict", "XGBoosterSetParam",
    "OHXDeviceCount", "OHXDMatrixCreateFromDevice", "OHXDMatrixSetGrid", "OHXDMatrixGetGrid", "OHXDMatrixInferGrid", "OHXBoosterPredictDevice", "OHXBoosterCheck",
    "OHXBoosterPredictFields", "OHXBoosterPredictFieldsDevice", "OHXBoosterRun1", "OHXBoosterRun1Device", "OHXOHPostProcess", "OHXOHPostProcessDevice",
    "OHXJulianDay", "OHXSolarGeometry", "OHXSolarGeometryDevice", "OHXBoosterGetInfo", "OHXBoosterKernelSymbol", "OHXBoosterKernelSymbolRows",
    "OHXBoosterRingReruns", "OHXBoosterCopyEngineChoice", "OHXUnregisterHost", "OHXReleaseScratch",
    "OHXCommGetUniqueId", "OHXCommInitRank", "OHXCommFree", "OHXCommInfo", "OHXShardRows", "OHXAllGatherOH",
]
# the subset QuickChem's xgb_fortran_api binds (Shared/xgb_fortran_api.F90:19-119)
REFERENCE_BOUND_SYMBOLS = [
    "XGBoosterLoadModel", "XGBoosterSaveModel", "XGDMatrixSaveBinary", "XGDMatrixFree", "XGDMatrixCreateFromFile",
    "XGBoosterPredict", "XGBoosterCreate", "XGDMatrixCreateFromMat", "XGDMatrixNumRow", "XGDMatrixNumCol",
    "XGBoosterFree",
]


class OhxError(RuntimeError):
    pass


class OHXRun1Args(C.Structure):
    """struct OHXRun1Args of include/ohxgb.h (part 3)."""
    _P = C.c_void_p
    _fields_ = ([("im", C.c_int32), ("jm", C.c_int32), ("km", C.c_int32), ("dynamic_k_range", C.c_int32),
                 ("tropp_min", C.c_float), ("ohscale", C.c_float), ("missing", C.c_float),
                 ("avogad", C.c_float), ("runiv", C.c_float), ("epsilon", C.c_float)] +
                [(n, C.c_void_p) for n in ("ple_mod", "t_mod", "q_mod", "tropp_mod", "ple_bst", "zle_bst", "tauclw",
                                           "taucli")] +
                [("scacoef", C.c_void_p * 7)] +
                [(n, C.c_void_p) for n in ("gmito3", "gmitto3", "lat_deg", "t_bst", "no2", "o3", "ch4", "co", "isop",
                                           "acet", "c2h6", "c3h8", "prpe", "alk4", "mp", "h2o2", "cloud", "qv", "albuv",
                                           "ch2o", "sza", "default_oh", "oh", "oh_boost", "ndwet", "k1", "k2",
                                           "diag_pl_bst", "diag_tauclwdn", "diag_tauclidn", "diag_taucliup",
                                           "diag_tauclwup", "diag_aodup", "diag_aoddn", "diag_aod", "diag_strato3")])

RUN1_DIAG_3D = ["diag_pl_bst", "diag_tauclwdn", "diag_tauclidn", "diag_taucliup", "diag_tauclwup", "diag_aodup",
                "diag_aoddn", "diag_aod"]


RUN1_INPUTS_3D = ["t_mod", "q_mod", "tauclw", "taucli", "t_bst", "no2", "o3", "ch4", "co", "isop", "acet", "c2h6", "c3h8",
                  "prpe", "alk4", "mp", "h2o2", "cloud", "qv", "ch2o", "default_oh"]
RUN1_INPUTS_EDGE = ["ple_mod", "ple_bst", "zle_bst"]
RUN1_INPUTS_2D = ["tropp_mod", "gmito3", "gmitto3", "lat_deg", "albuv", "sza"]


_lib: Optional[C.CDLL] = None


def declare_xgb_api(lib: C.CDLL) -> C.CDLL:
    """Attach argtypes for the XGBoost C-API subset (shared with the CPU oracle library)."""
    vp, u64, f32, i32 = C.c_void_p, C.c_uint64, C.c_float, C.c_int
    lib.XGBGetLastError.restype = C.c_char_p
    lib.XGBGetLastError.argtypes = []
    lib.XGDMatrixCreateFromMat.argtypes = [vp, u64, u64, f32, C.POINTER(vp)]
    lib.XGDMatrixFree.argtypes = [vp]
    lib.XGDMatrixNumRow.argtypes = [vp, C.POINTER(u64)]
    lib.XGDMatrixNumCol.argtypes = [vp, C.POINTER(u64)]
    lib.XGDMatrixSaveBinary.argtypes = [vp, C.c_char_p, i32]
    lib.XGDMatrixCreateFromFile.argtypes = [C.c_char_p, i32, C.POINTER(vp)]
    lib.XGBoosterCreate.argtypes = [vp, u64, C.POINTER(vp)]
    lib.XGBoosterFree.argtypes = [vp]
    lib.XGBoosterLoadModel.argtypes = [vp, C.c_char_p]
    lib.XGBoosterSaveModel.argtypes = [vp, C.c_char_p]
    lib.XGBoosterLoadModelFromBuffer.argtypes = [vp, vp, u64]
    lib.XGBoosterPredict.argtypes = [vp, vp, i32, C.c_uint, i32, C.POINTER(u64), C.POINTER(C.POINTER(f32))]
    return lib


def load_library(path: str = LIB_PATH) -> C.CDLL:
    global _lib
    if _lib is not None and path == LIB_PATH:
        return _lib
    if not os.path.exists(path):
        raise OhxError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    lib = declare_xgb_api(C.CDLL(path))
    vp, u64, f32, i32, u32 = C.c_void_p, C.c_uint64, C.c_float, C.c_int, C.c_uint32
    lib.XGBoosterSetParam.argtypes = [vp, C.c_char_p, C.c_char_p]
    lib.OHXDeviceCount.argtypes = [C.POINTER(i32)]
    lib.OHXDMatrixCreateFromDevice.argtypes = [vp, u64, u64, f32, C.POINTER(vp)]
    lib.OHXDMatrixSetGrid.argtypes = [vp, i32, i32, u64]
    lib.OHXDMatrixGetGrid.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(u64), C.POINTER(i32)]
    lib.OHXDMatrixInferGrid.argtypes = [vp, vp, C.POINTER(i32)]
    lib.OHXBoosterPredictDevice.argtypes = [vp, vp, i32, C.c_uint, vp, vp]
    lib.OHXBoosterCheck.argtypes = [vp, vp]
    lib.OHXBoosterPredictFields.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), i32, i32, i32, i32, i32, i32, i32,
                                            f32, i32, f32, vp, vp]
    lib.OHXBoosterPredictFieldsDevice.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), i32, i32, i32, i32, i32,
                                                  i32, i32, f32, i32, f32, vp, vp, vp]
    lib.OHXBoosterRun1.argtypes = [vp, C.POINTER(OHXRun1Args)]
    lib.OHXBoosterRun1Device.argtypes = [vp, C.POINTER(OHXRun1Args), vp]
    lib.OHXOHPostProcess.argtypes = [i32, i32, i32, f32, f32, f32] + [vp] * 8
    lib.OHXOHPostProcessDevice.argtypes = [i32, i32, i32, f32, f32, f32] + [vp] * 9
    lib.OHXJulianDay.argtypes = [i32, C.POINTER(i32)]
    lib.OHXSolarGeometry.argtypes = [i32, vp, vp, i32, i32, f32, f32, vp, vp]
    lib.OHXSolarGeometryDevice.argtypes = [i32, vp, vp, i32, i32, f32, f32, vp, vp, vp]
    lib.OHXBoosterGetInfo.argtypes = [vp, C.POINTER(u64)]
    lib.OHXBoosterKernelSymbol.argtypes = [vp, u64, C.POINTER(C.c_char_p)]
    lib.OHXBoosterKernelSymbolRows.argtypes = [vp, vp, C.POINTER(C.c_char_p)]
    lib.OHXBoosterRingReruns.argtypes = [vp, vp, C.POINTER(u64)]
    lib.OHXBoosterCopyEngineChoice.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    lib.OHXUnregisterHost.argtypes = [vp]
    lib.OHXReleaseScratch.argtypes = []
    lib.OHXCommGetUniqueId.argtypes = [vp]
    lib.OHXCommInitRank.argtypes = [vp, i32, i32, C.POINTER(vp)]
    lib.OHXCommFree.argtypes = [vp]
    lib.OHXCommInfo.argtypes = [C.POINTER(i32)]
    lib.OHXShardRows.argtypes = [u64, i32, i32, C.POINTER(u64), C.POINTER(u64)]
    lib.OHXAllGatherOH.argtypes = [vp, vp, u64, u64, vp, vp]
    if path == LIB_PATH:
        _lib = lib
    return lib


def check(lib: C.CDLL, rc: int) -> None:
    if rc != 0:
        raise OhxError(lib.XGBGetLastError().decode("utf-8", "replace"))


def _as_f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


# MAPL_DEGREES_TO_RADIANS / MAPL_RADIANS_TO_DEGREES as MAPL defines them (real32 of pi / 180 and its inverse)
DEG2RAD = np.float32(np.float32(np.pi) / np.float32(180.0))
RAD2DEG = np.float32(np.float32(180.0) / np.float32(np.pi))


def julian_day(nymd: int, lib: Optional[C.CDLL] = None) -> int:
    """OHXJulianDay: day of year of a yyyymmdd date (OH_GridCompMod.F90:1905-1936)."""
    lib = lib or load_library()
    out = C.c_int32()
    check(lib, lib.OHXJulianDay(nymd, C.byref(out)))
    return out.value


def solar_geometry(jday: int, lats, lons, deg2rad=DEG2RAD, rad2deg=RAD2DEG, lib: Optional[C.CDLL] = None):
    """OHXSolarGeometry on [i,j]-indexed (im, jm) arrays in radians -> (lat_deg, sza_noon), same indexing."""
    lib = lib or load_library()
    la = np.ascontiguousarray(np.asarray(lats, dtype=np.float32).T)
    lo = np.ascontiguousarray(np.asarray(lons, dtype=np.float32).T)
    jm, im = la.shape
    lat_deg, sza = np.empty_like(la), np.empty_like(la)
    check(lib, lib.OHXSolarGeometry(jday, la.ctypes.data, lo.ctypes.data, im, jm, float(deg2rad), float(rad2deg),
                                    lat_deg.ctypes.data, sza.ctypes.data))
    return lat_deg.T, sza.T


class DMatrix:
    """XGDMatrixCreateFromMat / OHXDMatrixCreateFromDevice handle."""

    def __init__(self, data=None, missing: float = float("nan"), *, device_ptr: int = 0, nrow: int = 0, ncol: int = 0,
                 lib: Optional[C.CDLL] = None):
        self.lib = lib or load_library()
        self.handle = C.c_void_p()
        if data is not None:
            arr = _as_f32(data)
            if arr.ndim != 2:
                raise ValueError("data must be 2-D [nrow][ncol]")
            nrow, ncol = arr.shape
            check(self.lib, self.lib.XGDMatrixCreateFromMat(arr.ctypes.data, nrow, ncol, missing, C.byref(self.handle)))
        else:
            check(self.lib, self.lib.OHXDMatrixCreateFromDevice(device_ptr, nrow, ncol, missing, C.byref(self.handle)))

    @property
    def num_row(self) -> int:
        out = C.c_uint64()
        check(self.lib, self.lib.XGDMatrixNumRow(self.handle, C.byref(out)))
        return out.value

    @property
    def num_col(self) -> int:
        out = C.c_uint64()
        check(self.lib, self.lib.XGDMatrixNumCol(self.handle, C.byref(out)))
        return out.value

    def set_grid(self, im: int, jm: int, row0: int = 0) -> "DMatrix":
        """OHXDMatrixSetGrid: the rows are rows row0.. of the (im, jm, *) gather (speed only)."""
        check(self.lib, self.lib.OHXDMatrixSetGrid(self.handle, im, jm, row0))
        return self

    def infer_grid(self, stream: int = 0) -> bool:
        """OHXDMatrixInferGrid: look for the level size in the rows (device matrices; waits for `stream`)."""
        found = C.c_int32()
        check(self.lib, self.lib.OHXDMatrixInferGrid(self.handle, stream or None, C.byref(found)))
        return bool(found.value)

    def grid(self):
        """OHXDMatrixGetGrid -> (im, jm, row0, inferred)."""
        im, jm, r0, inf = C.c_int32(), C.c_int32(), C.c_uint64(), C.c_int32()
        check(self.lib, self.lib.OHXDMatrixGetGrid(self.handle, C.byref(im), C.byref(jm), C.byref(r0), C.byref(inf)))
        return im.value, jm.value, r0.value, bool(inf.value)

    def save_binary(self, fname: str) -> None:
        check(self.lib, self.lib.XGDMatrixSaveBinary(self.handle, fname.encode(), 1))

    @classmethod
    def from_file(cls, fname: str, lib: Optional[C.CDLL] = None) -> "DMatrix":
        self = cls.__new__(cls)
        self.lib = lib or load_library()
        self.handle = C.c_void_p()
        check(self.lib, self.lib.XGDMatrixCreateFromFile(fname.encode(), 1, C.byref(self.handle)))
        return self

    def free(self) -> None:
        if self.handle:
            check(self.lib, self.lib.XGDMatrixFree(self.handle))
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Booster:
    """XGBoosterCreate / LoadModel / Predict handle."""

    def __init__(self, model_file: Optional[str] = None, *, model_buffer: Optional[bytes] = None,
                 lib: Optional[C.CDLL] = None):
        self.lib = lib or load_library()
        self.handle = C.c_void_p()
        # as the reference does: a handle by value and len == 0 (OH_GridCompMod.F90:255-256)
        check(self.lib, self.lib.XGBoosterCreate(None, 0, C.byref(self.handle)))
        if model_file is not None:
            self.load_model(model_file)
        elif model_buffer is not None:
            self.load_model_buffer(model_buffer)

    def load_model(self, fname: str) -> None:
        check(self.lib, self.lib.XGBoosterLoadModel(self.handle, fname.encode()))

    def load_model_buffer(self, buf) -> None:
        if isinstance(buf, np.ndarray):
            ptr, n = buf.ctypes.data, buf.nbytes
            check(self.lib, self.lib.XGBoosterLoadModelFromBuffer(self.handle, ptr, n))
        else:
            b = bytes(buf)
            check(self.lib, self.lib.XGBoosterLoadModelFromBuffer(self.handle, C.cast(C.c_char_p(b), C.c_void_p), len(b)))

    def save_model(self, fname: str) -> None:
        check(self.lib, self.lib.XGBoosterSaveModel(self.handle, fname.encode()))

    def set_param(self, name: str, value) -> None:
        check(self.lib, self.lib.XGBoosterSetParam(self.handle, name.encode(), str(value).encode()))

    def predict(self, dmat: DMatrix, option_mask: int = 0, ntree_limit: int = 0, training: int = 0,
                copy: bool = True) -> np.ndarray:
        """Host result, float32: a copy of the booster-owned buffer, or (copy=False) a view of it that is valid until
        the booster's next predict - what the reference's Fortran reads through its c_f_pointer (OH_GridCompMod.F90:362)."""
        n = C.c_uint64()
        ptr = C.POINTER(C.c_float)()
        check(self.lib, self.lib.XGBoosterPredict(self.handle, dmat.handle, option_mask, ntree_limit, training,
                                                   C.byref(n), C.byref(ptr)))
        if n.value == 0:
            return np.empty(0, dtype=np.float32)
        view = np.ctypeslib.as_array(ptr, shape=(n.value,))
        return view.copy() if copy else view

    def predict_device(self, dmat: DMatrix, out_ptr: int, option_mask: int = 0, ntree_limit: int = 0,
                       stream: int = 0) -> None:
        check(self.lib, self.lib.OHXBoosterPredictDevice(self.handle, dmat.handle, option_mask, ntree_limit, out_ptr,
                                                          stream))

    def check(self, stream: int = 0) -> None:
        check(self.lib, self.lib.OHXBoosterCheck(self.handle, stream))

    def predict_fields(self, fields: Sequence[np.ndarray], is2d: Sequence[bool], pl_feature: int, im: int, jm: int,
                       km: int, k1: int, k2: int, missing: float, oh_ml: np.ndarray, *, apply_pow10: bool = True,
                       ohscale: float = 1.0, margin: Optional[np.ndarray] = None) -> None:
        """Host arrays in Fortran order (pass the .T views' buffers: index i + im*(j + jm*k))."""
        nf = len(fields)
        ptrs = (C.c_void_p * nf)(*[f.ctypes.data for f in fields])
        flags = (C.c_int32 * nf)(*[1 if b else 0 for b in is2d])
        check(self.lib, self.lib.OHXBoosterPredictFields(
            self.handle, ptrs, flags, nf, pl_feature, im, jm, km, k1, k2, missing, 1 if apply_pow10 else 0, ohscale,
            oh_ml.ctypes.data, margin.ctypes.data if margin is not None else None))

    def predict_fields_device(self, field_ptrs: Sequence[int], is2d: Sequence[bool], pl_feature: int, im: int,
                              jm: int, km: int, k1: int, k2: int, missing: float, oh_ml_ptr: int, *,
                              apply_pow10: bool = True, ohscale: float = 1.0, margin_ptr: int = 0,
                              stream: int = 0) -> None:
        nf = len(field_ptrs)
        ptrs = (C.c_void_p * nf)(*field_ptrs)
        flags = (C.c_int32 * nf)(*[1 if b else 0 for b in is2d])
        check(self.lib, self.lib.OHXBoosterPredictFieldsDevice(
            self.handle, ptrs, flags, nf, pl_feature, im, jm, km, k1, k2, missing, 1 if apply_pow10 else 0, ohscale,
            oh_ml_ptr, margin_ptr or None, stream or None))

    def run1_prepare(self, state: dict, *, dynamic_k_range: bool, tropp_min: float = 4000.0, ohscale: float = 0.85,
                     missing: float = -999.0, avogad: float = 6.023e26, runiv: float = 8314.47,
                     epsilon: float = 18.015 / 28.965, want_boost: bool = True, want_ndwet: bool = True,
                     want_diag: bool = False) -> dict:
        """The OHXRun1Args of a tick, built once: `state` maps the names of OHXRun1Args to [i,j(,k)]-indexed float32
        arrays (edge fields have km+1 levels; "scacoef" is a list of seven).  The returned call keeps the flattened
        host arrays alive at fixed addresses - as MAPL's state pointers are from tick to tick - so that run1_call can
        be repeated on them (and ohx_register_host has something stable to register)."""
        def flat(a):
            return np.ascontiguousarray(np.asarray(a, dtype=np.float32).T)
        im, jm, km = state["t_mod"].shape
        keep = {}
        args = OHXRun1Args()
        args.im, args.jm, args.km = im, jm, km
        args.dynamic_k_range = 1 if dynamic_k_range else 0
        args.tropp_min, args.ohscale, args.missing = tropp_min, ohscale, missing
        args.avogad, args.runiv, args.epsilon = avogad, runiv, epsilon
        for name in RUN1_INPUTS_3D + RUN1_INPUTS_EDGE + RUN1_INPUTS_2D:
            keep[name] = flat(state[name])
            setattr(args, name, keep[name].ctypes.data)
        sca = [flat(a) for a in state["scacoef"]]
        keep["sca"] = sca
        args.scacoef = (C.c_void_p * 7)(*[a.ctypes.data for a in sca])
        oh = np.zeros(im * jm * km, dtype=np.float32)
        boost = np.zeros(im * jm * km, dtype=np.float32) if want_boost else None
        ndwet = np.zeros(im * jm * km, dtype=np.float32) if want_ndwet else None
        k1, k2 = C.c_int32(), C.c_int32()
        args.oh = oh.ctypes.data
        args.oh_boost = boost.ctypes.data if want_boost else None
        args.ndwet = ndwet.ctypes.data if want_ndwet else None
        args.k1 = C.cast(C.pointer(k1), C.c_void_p)
        args.k2 = C.cast(C.pointer(k2), C.c_void_p)
        diag = {}
        if want_diag:
            for name in RUN1_DIAG_3D:
                diag[name] = np.zeros(im * jm * km, dtype=np.float32)
                setattr(args, name, diag[name].ctypes.data)
            diag["diag_strato3"] = np.zeros(im * jm, dtype=np.float32)
            args.diag_strato3 = diag["diag_strato3"].ctypes.data
        return {"args": args, "keep": keep, "oh": oh, "boost": boost, "ndwet": ndwet, "k1": k1, "k2": k2, "diag": diag,
                "shape": (im, jm, km)}

    def run1_call(self, call: dict) -> dict:
        """OHXBoosterRun1 on the host arrays of a prepared call.  Returns {"oh", "oh_boost", "ndwet", "k1", "k2"} (+ the
        DIAG dumps) as views of the call's output arrays, indexed [i,j,k]."""
        im, jm, km = call["shape"]
        check(self.lib, self.lib.OHXBoosterRun1(self.handle, C.byref(call["args"])))
        unflat = lambda a: None if a is None else a.reshape(km, jm, im).transpose(2, 1, 0)   # noqa: E731
        out = {"oh": unflat(call["oh"]), "oh_boost": unflat(call["boost"]), "ndwet": unflat(call["ndwet"]),
               "k1": call["k1"].value, "k2": call["k2"].value}
        for name, a in call["diag"].items():
            out[name] = a.reshape(jm, im).T if name == "diag_strato3" else unflat(a)
        return out

    def run1(self, state: dict, **kw) -> dict:
        """OHXBoosterRun1 on host arrays (run1_prepare + run1_call)."""
        return self.run1_call(self.run1_prepare(state, **kw))

    def info(self) -> dict:
        arr = (C.c_uint64 * 8)()
        check(self.lib, self.lib.OHXBoosterGetInfo(self.handle, arr))
        keys = ["num_trees", "num_nodes", "num_slots", "node_bytes", "max_depth", "num_feature", "packed",
                "gathers_per_wave"]
        return {k: int(arr[i]) for i, k in enumerate(keys)}

    def ring_reruns(self, stream: int = 0) -> int:
        """How often a ring block gave up and the tile kernel predicted the batch again (include/ohxgb.h)."""
        out = C.c_uint64()
        check(self.lib, self.lib.OHXBoosterRingReruns(self.handle, stream, C.byref(out)))
        return int(out.value)

    def copy_engine_choice(self):
        """-> (choice, trials, picked_dma): what ohx_copy_engine = auto decided for this booster's Run1 host form
        (-1 trying / not in charge, 0 copy kernels, 1 DMA), and how its trials went (include/ohxgb.h)."""
        c, t, d = C.c_int(), C.c_uint(), C.c_uint()
        check(self.lib, self.lib.OHXBoosterCopyEngineChoice(self.handle, C.byref(c), C.byref(t), C.byref(d)))
        return int(c.value), int(t.value), int(d.value)

    def kernel_symbol(self, ncol: int) -> str:
        out = C.c_char_p()
        check(self.lib, self.lib.OHXBoosterKernelSymbol(self.handle, ncol, C.byref(out)))
        return out.value.decode()

    def kernel_symbols_for(self, dmat) -> str:
        """Every kernel a margin predict on `dmat` launches, in order, joined by " + " (OHXBoosterKernelSymbolRows)."""
        out = C.c_char_p()
        check(self.lib, self.lib.OHXBoosterKernelSymbolRows(self.handle, dmat.handle, C.byref(out)))
        return out.value.decode()

    def fields_kernel_symbol(self, nrow: int = 1 << 30) -> str:
        """The __global__ the fused OHXBoosterPredictFields launches for a slab of `nrow` gridcells: the fields kernel of
        the same family as the rows kernel (kernels.hip launch_predict_fields; the ring kernel from two residencies
        of the chip on)."""
        import re
        rows = self.kernel_symbol(27)
        if "ring" in rows:
            return "predict_fields_ring_kernel" if nrow >= 256 * 16 * 64 * 2 else "predict_fields_kernel<2,2,true>"
        m = re.match(r"predict_rows_tile_kernel<(\d+),(\d+),(?:true|false),(true|false)>", rows)
        return f"predict_fields_kernel<{m.group(1)},{m.group(2)},{m.group(3)}>" if m else "predict_fields_kernel<0,1,false>"

    def free(self) -> None:
        if self.handle:
            check(self.lib, self.lib.XGBoosterFree(self.handle))
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def oh_post_process(ple_mod, t_mod, q_mod, tropp_mod, default_oh, oh_ml, *, avogad: float = 6.023e26,
                    runiv: float = 8314.47, epsilon: float = 18.015 / 28.965, lib: Optional[C.CDLL] = None):
    """OHXOHPostProcess on [i,j(,k)]-indexed float32 arrays -> (oh, ndwet), indexed [i,j,k]."""
    lib = lib or load_library()
    if not hasattr(lib.OHXOHPostProcess, "argtypes") or lib.OHXOHPostProcess.argtypes is None:
        lib.OHXOHPostProcess.argtypes = [C.c_int] * 3 + [C.c_float] * 3 + [C.c_void_p] * 8
    flat = [np.ascontiguousarray(np.asarray(a, dtype=np.float32).T) for a in (ple_mod, t_mod, q_mod, tropp_mod, default_oh, oh_ml)]
    im, jm, km = np.asarray(t_mod).shape
    oh = np.zeros(im * jm * km, dtype=np.float32)
    ndwet = np.zeros(im * jm * km, dtype=np.float32)
    check(lib, lib.OHXOHPostProcess(im, jm, km, avogad, runiv, epsilon, *[a.ctypes.data for a in flat], oh.ctypes.data,
                                    ndwet.ctypes.data))
    return oh.reshape(km, jm, im).transpose(2, 1, 0), ndwet.reshape(km, jm, im).transpose(2, 1, 0)


UNIQUE_ID_BYTES = 128


def shard_rows(n_total: int, nranks: int, rank: int, lib: Optional[C.CDLL] = None):
    """OHXShardRows -> (row0, nrows)."""
    lib = lib or load_library()
    r0, n = C.c_uint64(), C.c_uint64()
    check(lib, lib.OHXShardRows(n_total, nranks, rank, C.byref(r0), C.byref(n)))
    return r0.value, n.value


class Communicator:
    """OHXCommInitRank handle: the RCCL communicator a Fortran/MPI host would build through the C ABI.
    `unique_id` = the 128 bytes rank 0 got from Communicator.unique_id(), distributed by the caller."""

    def __init__(self, unique_id: bytes, nranks: int, rank: int, lib: Optional[C.CDLL] = None):
        self.lib = lib or load_library()
        self.handle = C.c_void_p()
        self.nranks, self.rank = nranks, rank
        buf = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
        check(self.lib, self.lib.OHXCommInitRank(C.cast(buf, C.c_void_p), nranks, rank, C.byref(self.handle)))

    @staticmethod
    def unique_id(lib: Optional[C.CDLL] = None) -> bytes:
        lib = lib or load_library()
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        check(lib, lib.OHXCommGetUniqueId(C.cast(buf, C.c_void_p)))
        return buf.raw

    @staticmethod
    def rccl_version(lib: Optional[C.CDLL] = None) -> int:
        """ncclGetVersion's code of the librccl.so the library loaded (e.g. 22705 for 2.27.5)."""
        lib = lib or load_library()
        v = C.c_int()
        check(lib, lib.OHXCommInfo(C.byref(v)))
        return v.value

    def all_gather_oh(self, shard_ptr: int, n_local: int, n_total: int, full_ptr: int, stream: int = 0) -> None:
        check(self.lib, self.lib.OHXAllGatherOH(self.handle, shard_ptr, n_local, n_total, full_ptr, stream or None))

    def free(self) -> None:
        if self.handle:
            check(self.lib, self.lib.OHXCommFree(self.handle))
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_count() -> int:
    lib = load_library()
    n = C.c_int()
    check(lib, lib.OHXDeviceCount(C.byref(n)))
    return n.value
