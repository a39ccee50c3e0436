import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from quickchem_amd import capi, synth
from tests import helpers
torch.cuda.set_device(0)
for gridname, trees in (("C12", 20), ("C48", 100)):
    grid = synth.GRIDS[gridname]
    n = grid[0]*grid[1]*grid[2]
    model = synth.make_model(num_trees=trees, max_depth=18, sample_log2=16, min_leaf=2, grid=grid)
    rows = torch.empty((n, synth.NFEAT), dtype=torch.float32, device="cuda:0")
    synth.rows_device(grid, 0, n, rows)
    booster = capi.Booster(model_buffer=model.image)
    dmat = capi.DMatrix(device_ptr=rows.data_ptr(), nrow=n, ncol=synth.NFEAT, missing=synth.XX_MISS)
    dmat.set_grid(grid[0], grid[1], 0)
    out = torch.zeros(n, dtype=torch.float32, device="cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):   # warm up: allocations, uploads, the look at the level size
            booster.predict_device(dmat, out.data_ptr(), stream=s.cuda_stream)
    s.synchronize(); booster.check()
    want = out.clone()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s):
            booster.predict_device(dmat, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    except Exception as e:
        print(gridname, "CAPTURE FAILED:", repr(e)[:500]); continue
    # new contents in the same buffers: the rows rotated by 64 cells
    rows2 = torch.roll(rows, 64, 0).contiguous()
    rows.copy_(rows2); out.zero_(); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    got = out.cpu().numpy()
    ref = helpers.oracle_predict(model.image, rows.cpu().numpy(), synth.XX_MISS)
    print(gridname, "replay bit-exact vs oracle:", np.array_equal(helpers.bits(got), helpers.bits(ref)), "kernel:", booster.kernel_symbol_rows(dmat) if hasattr(booster,'kernel_symbol_rows') else '')
    import time
    t=time.time()
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); tg=(time.time()-t)/20
    t=time.time()
    with torch.cuda.stream(s):
        for _ in range(20): booster.predict_device(dmat, out.data_ptr(), stream=s.cuda_stream)
    s.synchronize(); tl=(time.time()-t)/20
    print(gridname, "ms per predict: graph replay %.3f, plain launches %.3f" % (tg*1e3, tl*1e3))
