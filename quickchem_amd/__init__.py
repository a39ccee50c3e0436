"""MI355X-native OH-chemistry predictor: the QuickChem OH XGBoost-predict hot path
(`OH_GridCompMod::predict_OH_with_XGB` -> `Shared/xgb_fortran_api` -> libxgboost)
as hand-written gfx950 HIP kernels behind the same C ABI.

Layout:
  csrc/     HIP kernels, model reader/writer, layout, the C ABI  -> lib/libohxgb.so
  fortran/  ISO_C_BINDING host side mirroring predict_OH_with_XGB
  capi.py   ctypes plumbing over the C ABI
  oh_predict.py  Python mirror of predict_OH_with_XGB for tests and the benchmark
  synth.py  synthetic feature batches and booster
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
