"""The device forms inside a hipGraph (GPU).

`OHXBoosterPredictDevice` and `OHXBoosterPredictFieldsDevice` enqueue launches, memsets and copies on the caller's
stream and nothing else once the booster's buffers exist and the matrix has been looked at, so a caller may capture
them into a graph and replay it on new contents of the same buffers (include/ohxgb.h part 2).  Replays are compared
with the CPU oracle bit for bit; a capture that would need the library to allocate or to wait is refused with a
message that says what to do, and leaves the capture's stream usable."""
import numpy as np
import pytest

from quickchem_amd import capi, synth
from tests import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    torch.cuda.set_device(0)
    return torch


def _with_missing(rows, rate, seed):
    rows = rows.copy()
    rng = np.random.default_rng(seed)
    mask = rng.random(rows.shape) < rate
    rows[mask] = np.where(rng.random(int(mask.sum())) < 0.5, np.float32(synth.XX_MISS), np.float32(np.nan))
    return rows


# (grid the rows are gathered from, rows, kernel parameters): a batch small enough to have its trees split over waves
# (two kernels per predict), and one the ring kernel walks, with rows left to the second launch
CASES = [
    ("split over waves", (12, 72, 72), 12 * 72 * 40, {}),
    ("ring + second launch", (96, 72, 72), 96 * 72 * 64, {"ohx_kernel": "ring", "ohx_tree_split": "off", "ohx_defer_missing": "on"}),
]


@pytest.mark.parametrize("name,grid,nrow,params", CASES, ids=[c[0] for c in CASES])
def test_device_predict_captured_and_replayed(torch_cuda, deep_model, name, grid, nrow, params):
    torch = torch_cuda
    first = _with_missing(synth.rows_cpu(grid, 0, nrow), 1e-4, seed=1)
    second = _with_missing(synth.rows_cpu(grid, 4 * grid[0] * grid[1], nrow), 3e-4, seed=2)
    rows = torch.from_numpy(first).cuda()
    out = torch.zeros(nrow, dtype=torch.float32, device="cuda")
    b = capi.Booster(model_buffer=deep_model.image)
    for k, v in params.items():
        b.set_param(k, v)
    d = capi.DMatrix(device_ptr=rows.data_ptr(), nrow=nrow, ncol=27, missing=synth.XX_MISS)
    d.set_grid(grid[0], grid[1], 0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b.predict_device(d, out.data_ptr(), stream=s.cuda_stream)          # the plain call that makes the buffers
    s.synchronize()
    b.check()
    assert np.array_equal(helpers.bits(out.cpu().numpy()), helpers.bits(helpers.oracle_predict(deep_model.image, first, synth.XX_MISS)))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        b.predict_device(d, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    for contents in (second, first):
        rows.copy_(torch.from_numpy(contents))
        out.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        b.check()
        want = helpers.oracle_predict(deep_model.image, contents, synth.XX_MISS)
        assert np.array_equal(helpers.bits(out.cpu().numpy()), helpers.bits(want)), name
    # and a plain call after the capture still works (the host's look at the second launch's count was left out of
    # the capture, not broken by it)
    with torch.cuda.stream(s):
        b.predict_device(d, out.data_ptr(), stream=s.cuda_stream)
    s.synchronize()
    b.check()
    assert np.array_equal(helpers.bits(out.cpu().numpy()), helpers.bits(helpers.oracle_predict(deep_model.image, first, synth.XX_MISS)))


def test_fused_fields_device_form_captured_and_replayed(torch_cuda, deep_model):
    torch = torch_cuda
    grid = (48, 36, 72)
    pl, tropp, fields = helpers.synth_state(grid)
    oh_ref, margin_ref, k1, k2 = helpers.oracle_predict_oh(deep_model.image, pl, tropp, fields, False)
    dev = [torch.from_numpy(helpers.fortran_flat(f).ravel().copy()).cuda() for f in fields]
    n = grid[0] * grid[1] * grid[2]
    oh = torch.zeros(n, dtype=torch.float32, device="cuda")
    margin = torch.zeros(grid[0] * grid[1] * (k2 - k1 + 1), dtype=torch.float32, device="cuda")
    b = capi.Booster(model_buffer=deep_model.image)
    s = torch.cuda.Stream()

    def call(stream):
        b.predict_fields_device([t.data_ptr() for t in dev], synth.IS2D, synth.PL_FEATURE, grid[0], grid[1], grid[2], k1, k2,
                                synth.XX_MISS, oh.data_ptr(), margin_ptr=margin.data_ptr(), stream=stream)

    with torch.cuda.stream(s):
        call(s.cuda_stream)
    s.synchronize()
    b.check()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        call(torch.cuda.current_stream().cuda_stream)
    oh.zero_()
    margin.zero_()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    b.check()
    assert np.array_equal(helpers.bits(margin.cpu().numpy()), helpers.bits(margin_ref))


def test_a_capture_that_would_allocate_is_refused_with_advice(torch_cuda, deep_model):
    """No plain call before the capture: the matrix has not been looked at (no grid said), so the library would have to
    wait for the stream - it says so instead, and the stream and the booster stay usable."""
    torch = torch_cuda
    grid = (96, 72, 72)
    nrow = 96 * 72 * 60
    rows = torch.from_numpy(synth.rows_cpu(grid, 0, nrow)).cuda()
    out = torch.zeros(nrow, dtype=torch.float32, device="cuda")
    b = capi.Booster(model_buffer=deep_model.image)
    d = capi.DMatrix(device_ptr=rows.data_ptr(), nrow=nrow, ncol=27, missing=synth.XX_MISS)
    warm = capi.DMatrix(device_ptr=rows.data_ptr(), nrow=64, ncol=27, missing=synth.XX_MISS)
    b.predict_device(warm, out.data_ptr())                  # the model is on the device; nothing about `d` is known
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(capi.OhxError, match="stream capture"):
        with torch.cuda.graph(g, stream=s, capture_error_mode="relaxed"):
            b.predict_device(d, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        b.predict_device(d, out.data_ptr(), stream=s.cuda_stream)
    s.synchronize()
    b.check()
    want = helpers.oracle_predict(deep_model.image, rows.cpu().numpy(), synth.XX_MISS)
    assert np.array_equal(helpers.bits(out.cpu().numpy()), helpers.bits(want))
