"""Synthetic inputs of SURVEY.md §8(d): feature batches and the synthetic OH booster.

The reference ships neither data nor a model (its models sit on NCCS paths,
``OH_GridComp/OH_instance_OH.rc:17-20``), so tests and the benchmark draw both from
``libohx_synth.so`` (host, g++) and ``libohx_synth_gpu.so`` (the same generator in HBM, hipcc);
both compile the same ``csrc/synth_common.h`` and agree bit for bit.  Neither is part of
the product library.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
SYNTH_LIB_PATH = os.path.join(_HERE, "lib", "libohx_synth.so")
SYNTH_GPU_LIB_PATH = os.path.join(_HERE, "lib", "libohx_synth_gpu.so")

FEATURE_SEED = 20241108
MODEL_SEED = 1060
NFEAT = 27
FEATURE_NAMES = ["LAT", "PL", "T", "NO2", "O3", "CH4", "CO", "ISOP", "ACET", "C2H6", "C3H8",
                 "PRPE", "ALK4", "MP", "H2O2", "TAUCLWDN", "TAUCLIDN", "TAUCLIUP", "TAUCLWUP",
                 "CLOUD", "QV", "GMISTRATO3", "ALBUV", "AODUP", "AODDN", "CH2O", "SZA"]
IS2D = [n in ("LAT", "GMISTRATO3", "ALBUV", "SZA") for n in FEATURE_NAMES]
PL_FEATURE = 1
XX_MISS = -999.0             # OH_GridCompMod.F90:213

# BASELINE.json configs: cubed-sphere C{n} is im = n, jm = 6n in MAPL's layout
GRIDS: Dict[str, Tuple[int, int, int]] = {
    "mock4x4": (4, 4, 72),
    "C12": (12, 72, 72),
    "C48": (48, 288, 72),
    "C90": (90, 540, 72),
    "C180": (180, 1080, 72),
    "C360": (360, 2160, 72),
    "C720L137": (720, 4320, 137),
}

_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(SYNTH_LIB_PATH):
            raise capi.OhxError(f"{SYNTH_LIB_PATH} is missing: run __graft_entry__.build()")
        lib = C.CDLL(SYNTH_LIB_PATH)
        lib.ohx_synth_last_error.restype = C.c_char_p
        lib.ohx_synth_rows_cpu.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p]
        lib.ohx_synth_field_cpu.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.ohx_synth_model.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int,
                                        C.c_int, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.ohx_synth_free.argtypes = [C.c_void_p]
        lib.ohx_synth_set_threads.argtypes = [C.c_int]
        lib.ohx_model_convert.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p),
                                          C.POINTER(C.c_uint64)]
        lib.ohx_super_walk_cpu.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_float,
                                           C.c_void_p, C.POINTER(C.c_uint64)]
        _lib = lib
    return _lib


def set_threads(n: int) -> None:
    """Host threads the generators may use (torchrun pins OMP_NUM_THREADS=1 per rank)."""
    _load().ohx_synth_set_threads(int(n))


def _check(rc: int) -> None:
    if rc != 0:
        raise capi.OhxError(_load().ohx_synth_last_error().decode())


def rows_cpu(grid: Tuple[int, int, int], row_begin: int, nrows: int, seed: int = FEATURE_SEED) -> np.ndarray:
    """[nrows][27] float32, PL in hPa, rows m = i + im*(j + jm*k) (OH_GridCompMod.F90:309-345)."""
    im, jm, km = grid
    out = np.empty((nrows, NFEAT), dtype=np.float32)
    _check(_load().ohx_synth_rows_cpu(seed, im, jm, km, row_begin, nrows, out.ctypes.data))
    return out


def field_cpu(grid: Tuple[int, int, int], feature: int, seed: int = FEATURE_SEED) -> np.ndarray:
    """One MAPL field in Fortran order, returned as an [i,j(,k)]-indexed view.  feature -1 = TROPP (Pa); PL in Pa."""
    im, jm, km = grid
    two_d = feature < 0 or IS2D[feature]
    flat = np.empty(im * jm * (1 if two_d else km), dtype=np.float32)
    _check(_load().ohx_synth_field_cpu(seed, feature, im, jm, km, flat.ctypes.data))
    return flat.reshape((jm, im)).T if two_d else flat.reshape((km, jm, im)).transpose(2, 1, 0)


@dataclass
class SynthModel:
    image: np.ndarray          # uint8 file image (legacy binary or JSON)
    num_trees: int
    num_nodes: int
    num_leaves: int
    max_depth: int
    mean_path: float           # internal nodes visited per row per tree, over the growth sample


def _take(ptr: C.c_void_p, n: int) -> np.ndarray:
    lib = _load()
    buf = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    lib.ohx_synth_free(ptr)
    return buf


def make_model(num_trees: int = 100, max_depth: int = 18, sample_log2: int = 20, min_leaf: int = 8,
               grid: Tuple[int, int, int] = GRIDS["C90"], model_seed: int = MODEL_SEED,
               feature_seed: int = FEATURE_SEED, base_score: float = -13.0, leaf_sigma: float = 0.05,
               fmt: str = "binary") -> SynthModel:
    """The synthetic OH booster: `num_trees` trees of depth <= `max_depth`, 27 features,
    grown on 2**sample_log2 cells of `grid`, thresholds = sampled feature values."""
    lib = _load()
    out, n = C.c_void_p(), C.c_uint64()
    stats = (C.c_uint64 * 4)()
    im, jm, km = grid
    _check(lib.ohx_synth_model(model_seed, num_trees, max_depth, sample_log2, min_leaf, feature_seed, im, jm, km,
                               base_score, leaf_sigma, 1 if fmt == "json" else 0, C.byref(out), C.byref(n), stats))
    denom = float((1 << sample_log2) * max(num_trees, 1))
    return SynthModel(_take(out, n.value), num_trees, int(stats[0]), int(stats[1]), int(stats[2]),
                      float(stats[3]) / denom)


def convert_model(image, fmt: str) -> np.ndarray:
    """Legacy binary / JSON / UBJSON ("binary", "json", "ubj") through the product's own readers and
    writers (host logic)."""
    lib = _load()
    src = np.frombuffer(bytes(image), dtype=np.uint8) if not isinstance(image, np.ndarray) else image
    out, n = C.c_void_p(), C.c_uint64()
    code = {"binary": 0, "json": 1, "ubj": 2}[fmt]
    _check(lib.ohx_model_convert(src.ctypes.data, src.nbytes, code, C.byref(out), C.byref(n)))
    return _take(out, n.value)


def super_walk_cpu(image, rows: np.ndarray, missing: float = XX_MISS):
    """Host check of the super-node layout the kernels read (csrc/flatten.hpp): emit_super's arrays walked
    the kernels' way - fixed trip count per tree, no finished state, fillers - by a scalar loop.  Test
    support, not a prediction path.  Returns (margins, info) or (None, None) if the booster does not fit."""
    lib = _load()
    src = np.frombuffer(bytes(image), dtype=np.uint8) if not isinstance(image, np.ndarray) else image
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    out = np.empty(rows.shape[0], dtype=np.float32)
    info = (C.c_uint64 * 4)()
    rc = lib.ohx_super_walk_cpu(src.ctypes.data, src.nbytes, rows.ctypes.data, rows.shape[0], rows.shape[1], missing,
                                out.ctypes.data, info)
    if rc == 1:
        return None, None
    _check(rc)
    return out, {"super_nodes": int(info[0]), "phase1_trees": int(info[1]), "steps": int(info[2])}


# ---- device generators (torch tensors in HBM; libohx_synth_gpu.so, test support like the rest of this file) ----

_gpu_lib = None


def _load_gpu():
    global _gpu_lib
    if _gpu_lib is None:
        if not os.path.exists(SYNTH_GPU_LIB_PATH):
            raise capi.OhxError(f"{SYNTH_GPU_LIB_PATH} is missing: run __graft_entry__.build()")
        lib = C.CDLL(SYNTH_GPU_LIB_PATH)
        vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int
        lib.ohx_synth_gpu_last_error.restype = C.c_char_p
        lib.ohx_synth_rows_device.argtypes = [u32, i32, i32, i32, u64, u64, vp, vp]
        lib.ohx_synth_field_device.argtypes = [u32, i32, i32, i32, i32, vp, vp]
        lib.ohx_inject_missing_device.argtypes = [vp, u64, u32, u32, C.c_float, vp]
        _gpu_lib = lib
    return _gpu_lib


def _check_gpu(rc: int) -> None:
    if rc != 0:
        raise capi.OhxError(_load_gpu().ohx_synth_gpu_last_error().decode())


def rows_device(grid: Tuple[int, int, int], row_begin: int, nrows: int, out, seed: int = FEATURE_SEED,
                stream: int = 0) -> None:
    """Fill torch tensor `out` ([nrows][27] float32, on the GPU) with rows row_begin.."""
    im, jm, km = grid
    assert out.is_contiguous() and out.numel() == nrows * NFEAT
    _check_gpu(_load_gpu().ohx_synth_rows_device(seed, im, jm, km, row_begin, nrows, out.data_ptr(), stream or None))


def field_device(grid: Tuple[int, int, int], feature: int, out, seed: int = FEATURE_SEED, stream: int = 0) -> None:
    im, jm, km = grid
    _check_gpu(_load_gpu().ohx_synth_field_device(seed, feature, im, jm, km, out.data_ptr(), stream or None))


def inject_missing_device(rows, rate_per_million: int, missing: float = XX_MISS, seed: int = 7, stream: int = 0) -> None:
    _check_gpu(_load_gpu().ohx_inject_missing_device(rows.data_ptr(), rows.numel(), seed, rate_per_million, missing,
                                                     stream or None))


def run1_state(grid, seed=17):
    """A synthetic OH import state for OHXBoosterRun1: physically plausible magnitudes, arbitrary values."""
    im, jm, km = grid
    rng = np.random.default_rng(seed)
    f32 = np.float32

    def u(lo, hi, shape):
        return (lo + (hi - lo) * rng.random(shape)).astype(f32)
    ps = u(6.0e4, 1.04e5, (im, jm))
    sig = (np.arange(km + 1, dtype=f32) / f32(km)) ** 2
    ple = (f32(1.0) + (ps[:, :, None] - f32(1.0)) * sig[None, None, :]).astype(f32)         # Pa, edges 0..km
    zle = (f32(8.0e4) * (f32(1.0) - sig[None, None, :]) * u(0.9, 1.1, (im, jm))[:, :, None]).astype(f32)
    vol, plane = (im, jm, km), (im, jm)
    cloudy = rng.random(vol) < 0.25
    st = {
        "ple_mod": ple, "ple_bst": (ple * u(0.98, 1.02, (im, jm))[:, :, None]).astype(f32), "zle_bst": zle,
        "t_mod": u(190, 310, vol), "q_mod": u(1e-7, 2e-2, vol), "tropp_mod": u(9.0e3, 3.0e4, plane),
        "tauclw": np.where(cloudy, u(0, 8, vol), 0).astype(f32), "taucli": np.where(~cloudy & (rng.random(vol) < 0.2), u(0, 3, vol), 0).astype(f32),
        "scacoef": [u(0, 5e-6, vol) for _ in range(7)],
        "gmito3": u(250, 450, plane), "gmitto3": u(20, 60, plane),
        "lat_deg": u(-90, 90, plane), "t_bst": u(190, 310, vol), "cloud": np.clip(u(-0.5, 1.0, vol), 0, 1).astype(f32),
        "qv": u(1e-7, 2e-2, vol), "albuv": u(0.02, 0.9, plane), "sza": u(0, 113, plane),
        "default_oh": u(1e-15, 5e-13, vol),
    }
    for name, lo in (("no2", 1e-12), ("o3", 1e-8), ("ch4", 1.6e-6), ("co", 2e-8), ("isop", 1e-14), ("acet", 1e-11),
                     ("c2h6", 1e-11), ("c3h8", 1e-12), ("prpe", 1e-13), ("alk4", 1e-12), ("mp", 1e-11), ("h2o2", 1e-11),
                     ("ch2o", 1e-12)):
        st[name] = (f32(lo) * np.exp2(u(0, 8, vol))).astype(f32)
    return st
