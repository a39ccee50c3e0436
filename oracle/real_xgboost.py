"""Looks for a REAL libxgboost on this machine - checker infrastructure, like everything under oracle/:
only tests/ and bench.py's cpu_baseline leg use it, the product never does.

QuickChem pins xgboost 1.6.0 EXACT (reference Shared/CMakeLists.txt:8).  Neither the build container nor the
GPU image ships one, so the search normally comes back empty and says where it looked; OHX_LIBXGBOOST names
a library explicitly.  A library that is found is driven through quickchem_amd/capi.py's ctypes plumbing: it
exports the very C symbols the product replaces (Shared/xgb_fortran_api.F90:19-119)."""
import ctypes as C
import ctypes.util
import glob
import os
import sys


def find_libxgboost():
    """-> (CDLL, where) or (None, what was tried)."""
    tried = []
    cands = []
    env = os.environ.get("OHX_LIBXGBOOST")
    if env:
        cands.append(env)
    try:
        import importlib.util
        spec = importlib.util.find_spec("xgboost")
        if spec and spec.submodule_search_locations:
            for d in spec.submodule_search_locations:
                cands += glob.glob(os.path.join(d, "lib", "libxgboost*.so*"))
        else:
            tried.append("python package xgboost: not installed")
    except Exception as e:              # pragma: no cover
        tried.append(f"python package xgboost: {e}")
    name = ctypes.util.find_library("xgboost")
    if name:
        cands.append(name)
    cands += ["libxgboost.so", "libxgboost.so.1"]
    for d in os.environ.get("LD_LIBRARY_PATH", "").split(":") + ["/usr/lib", "/usr/local/lib", "/opt/conda/lib",
                                                                   os.path.join(sys.prefix, "lib")]:
        if d:
            cands += glob.glob(os.path.join(d, "libxgboost*.so*"))
    for c in cands:
        try:
            lib = C.CDLL(c)
        except OSError as e:
            tried.append(f"{c}: {str(e)[:60]}")
            continue
        if hasattr(lib, "XGBoosterPredict") and hasattr(lib, "XGDMatrixCreateFromMat"):
            from quickchem_amd import capi
            return capi.declare_xgb_api(lib), c
        tried.append(f"{c}: loaded but is not an XGBoost C API")
    return None, "; ".join(tried)




def version_of(lib) -> str:
    if not hasattr(lib, "XGBoostVersion"):
        return "?"
    a, b, c = C.c_int(), C.c_int(), C.c_int()
    lib.XGBoostVersion(C.byref(a), C.byref(b), C.byref(c))
    return f"{a.value}.{b.value}.{c.value}"
