"""Host-side mirror of ``predict_OH_with_XGB`` (reference
``OH_GridComp/OH_GridCompMod.F90:123-398``) over the C ABI — same name, same
argument meaning, same error behaviour — used by the parity tests and the
benchmark.  The Fortran twin is ``fortran/oh_xgb_predict.F90``.

Two routes to the same numbers:
  * ``mode="compat"``: the reference's own five-call sequence
    (XGDMatrixCreateFromMat -> XGBoosterPredict -> 10**pred -> XGDMatrixFree);
    the gather and the ``10**`` stay on the host exactly as in the reference.
  * ``mode="fused"``: one call, ``OHXBoosterPredictFields``; gather, PL/100, walk
    and ``10**`` happen in one kernel.
Everything numeric that is on the path comes from ``libohxgb.so``; this module
only moves arrays and keeps the reference's bookkeeping (k-slab, SAVE'd booster).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import capi
from .synth import IS2D, PL_FEATURE, XX_MISS

XX_PARAM_COUNT = 27          # OH_GridCompMod.F90:228


@dataclass
class OHBoostInputData:
    """TYPE OH_BOOST_INPUT_DATA (OH_GridCompMod.F90:82-114): 27 arrays indexed [i,j(,k)],
    in the order the gather reads them (:313-339)."""
    fields: Sequence[np.ndarray]

    def __post_init__(self):
        if len(self.fields) != XX_PARAM_COUNT:
            raise ValueError("OH_BOOST_INPUT_DATA needs exactly 27 fields")


class AssertFailure(RuntimeError):
    """What MAPL's _ASSERT raises in the reference."""


def k_slab(pl: np.ndarray, tropp: np.ndarray, dynamic_k_range: bool, tropp_min: float):
    """OH_GridCompMod.F90:275-301 -> 1-based (k1, k2)."""
    km = pl.shape[2]
    if dynamic_k_range:
        counts = (pl > tropp[:, :, None]).sum(axis=2)
    else:
        if np.count_nonzero(tropp <= np.float32(tropp_min)) != 0:
            raise AssertFailure("OH Prediction: Minimum tropopause pressure is not low enough!")
        counts = (pl > np.float32(tropp_min)).sum(axis=2)
    ksubcount = int(counts.max()) if counts.size else 0
    return km - ksubcount + 1, km


def _fortran_flat(a: np.ndarray) -> np.ndarray:
    """[i,j(,k)]-indexed array -> contiguous buffer in Fortran element order."""
    return np.ascontiguousarray(a.T, dtype=np.float32)


def fill_grads_template(pattern: str, nymd: int, nhms: int = 0) -> str:
    """The GrADS-style tokens MAPL's fill_grads_template expands in XGBoostFile
    (OH_GridCompMod.F90:1187, OH_instance_OH.rc:20): %y4 %m2 %d2 %h2 %n2."""
    return (pattern.replace("%y4", f"{nymd // 10000:04d}").replace("%m2", f"{nymd % 10000 // 100:02d}")
            .replace("%d2", f"{nymd % 100:02d}").replace("%h2", f"{nhms // 10000:02d}")
            .replace("%n2", f"{nhms % 10000 // 100:02d}"))


class OHPredictor:
    """Holds the process-wide booster the reference keeps in SAVE variables
    (``xx_bst``, ``first_time``; OH_GridCompMod.F90:182,209).

    ``model_policy``: "reference" keeps the booster of the FIRST call for good and ignores later
    file names, as the reference does although XGBoostFile is month-templated (:209,269;
    OH_instance_OH.rc:20); "by_name" keeps one resident booster per file name (twelve monthly
    boosters are 0.56 GB of the 288 GB), so a month roll-over is a dictionary look-up."""

    def __init__(self, lib=None, model_policy: str = "reference"):
        if model_policy not in ("reference", "by_name"):
            raise ValueError("model_policy must be 'reference' or 'by_name'")
        self.lib = lib
        self.model_policy = model_policy
        self.boosters: dict = {}
        self.xx_bst: Optional[capi.Booster] = None
        self.first_time = True

    def predict_OH_with_XGB(self, xgb_fname: str, icount: int, jcount: int, kcount: int, dynamic_k_range: bool,
                            tropp_min: float, pl: np.ndarray, tropp: np.ndarray, bb: OHBoostInputData,
                            OH_ML: np.ndarray, mode: str = "compat", margin_out: Optional[list] = None) -> int:
        """Fills OH_ML[:, :, k1-1:k2] with 10**prediction (mol/mol); returns rc = 0.

        Raises AssertFailure where the reference's _ASSERT would fire."""
        name = xgb_fname.strip()
        if self.model_policy == "by_name" and name in self.boosters:
            self.xx_bst = self.boosters[name]
        elif self.first_time or self.model_policy == "by_name":       # ONE_TIME_SETUP, :242-271
            xx_carr_small = np.zeros((1, XX_PARAM_COUNT), dtype=np.float32)
            try:
                xx_dmtrx = capi.DMatrix(xx_carr_small, XX_MISS, lib=self.lib)          # :251
            except capi.OhxError as e:
                raise AssertFailure(f"Failed in XGDMatrixCreateFromMat_f: {e}")
            try:
                self.xx_bst = capi.Booster(lib=self.lib)                               # :256
            except capi.OhxError as e:
                raise AssertFailure(f"Failed in XGBoosterCreate_f: {e}")
            try:
                self.xx_bst.load_model(name)                                           # :261
            except capi.OhxError as e:
                raise AssertFailure(f"Failed in XGBoosterLoadModel_f: {e}")
            xx_dmtrx.free()                                                            # :264
            self.first_time = False
            self.boosters[name] = self.xx_bst
        assert pl.shape == (icount, jcount, kcount)
        k1, k2 = k_slab(pl, tropp, dynamic_k_range, tropp_min)
        ksubcount = k2 - k1 + 1
        xx_prediction_count = icount * jcount * ksubcount                              # :305
        if mode == "fused":
            flat = [_fortran_flat(a) for a in bb.fields]
            oh_flat = _fortran_flat(OH_ML)
            margin = np.empty(xx_prediction_count, dtype=np.float32)
            try:
                self.xx_bst.predict_fields(flat, IS2D, PL_FEATURE, icount, jcount, kcount, k1, k2, XX_MISS, oh_flat,
                                           apply_pow10=True, ohscale=1.0, margin=margin)
            except capi.OhxError as e:
                raise AssertFailure(f"Failed in OHXBoosterPredictFields: {e}")
            OH_ML[...] = oh_flat.T
            if margin_out is not None:
                margin_out.append(margin)
            return 0
        # ---- the reference's call sequence ----
        xx_carr = np.empty((ksubcount, jcount, icount, XX_PARAM_COUNT), dtype=np.float32)   # :306
        for f, a in enumerate(bb.fields):                                                   # :308-345
            if a.ndim == 2:
                xx_carr[..., f] = a.T[None, :, :]
            else:
                sl = a[:, :, k1 - 1:k2]
                if f == PL_FEATURE:
                    sl = (sl / np.float32(100.0)).astype(np.float32)                        # :314
                xx_carr[..., f] = np.transpose(sl, (2, 1, 0))
        xx_carr = xx_carr.reshape(xx_prediction_count, XX_PARAM_COUNT)
        try:
            xx_dmtrx = capi.DMatrix(xx_carr, XX_MISS, lib=self.lib)                         # :347
        except capi.OhxError as e:
            raise AssertFailure(f"Failed in XGDMatrixCreateFromMat_f: {e}")
        if hasattr(self.lib, "OHXDMatrixSetGrid"):
            xx_dmtrx.set_grid(icount, jcount, 0)         # not in the reference: layout hint, speed only
        try:
            xx_pred = self.xx_bst.predict(xx_dmtrx, option_mask=0, ntree_limit=0, training=0)   # :356
        except capi.OhxError as e:
            raise AssertFailure(f"Failed in XGBoosterPredict_f: {e}")
        if xx_pred.shape[0] != xx_prediction_count:                                         # :359
            raise AssertFailure("Wrong value returned for xx_pred_len")
        oh = np.power(np.float32(10.0), xx_pred, dtype=np.float32)                          # :369
        OH_ML[:, :, k1 - 1:k2] = np.transpose(oh.reshape(ksubcount, jcount, icount), (2, 1, 0))
        xx_dmtrx.free()                                                                     # :377
        if margin_out is not None:
            margin_out.append(xx_pred)
        return 0
