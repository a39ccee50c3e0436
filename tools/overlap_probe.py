#!/usr/bin/env python3
"""Can another kernel get on the chip while a ring launch runs?  (measurement aid; the question behind ohx_reserve_cus)
A C360/8 shard is predicted on the current stream; a millisecond into it the host enqueues, on a second stream, a
stand-in for a collective's kernel (a 12 MB device copy).  Printed: the predict's duration, and when the
copy started to wait and ended relative to the predict's start - with all CUs taken and with some left free.
usage (GPU box): python3 tools/overlap_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from quickchem_amd import capi, synth
    dev = torch.device("cuda:0")
    grid = synth.GRIDS["C360"]
    n = grid[0] * grid[1] * grid[2] // 8
    rows = torch.empty((n, synth.NFEAT), dtype=torch.float32, device=dev)
    synth.rows_device(grid, 0, n, rows)
    out = torch.empty(n, dtype=torch.float32, device=dev)
    model = synth.make_model(num_trees=100, max_depth=18, sample_log2=20)
    src = torch.empty(3 * 1024 * 1024, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    side = torch.cuda.Stream()
    main_stream = torch.cuda.current_stream()
    for reserve in (0, 8, 16):
        booster = capi.Booster(model_buffer=model.image)
        booster.set_param("ohx_reserve_cus", reserve)
        dm = capi.DMatrix(device_ptr=rows.data_ptr(), nrow=n, ncol=synth.NFEAT, missing=synth.XX_MISS)
        dm.set_grid(grid[0], grid[1], 0)
        res = []
        for it in range(6):
            e0, e1, c0, c1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
            torch.cuda.synchronize()
            e0.record(main_stream)
            side.wait_event(e0)
            booster.predict_device(dm, out.data_ptr(), stream=main_stream.cuda_stream)
            e1.record(main_stream)
            t_host = time.perf_counter()
            while time.perf_counter() - t_host < 1.0e-3:      # the predict (3 ms) is under way when the copy is enqueued
                pass
            with torch.cuda.stream(side):
                c0.record(side)
                dst.copy_(src)
                c1.record(side)
            torch.cuda.synchronize()
            if it >= 2:
                res.append((e0.elapsed_time(e1), e0.elapsed_time(c0), e0.elapsed_time(c1)))
        p = sum(r[0] for r in res) / len(res)
        a = sum(r[1] for r in res) / len(res)
        b = sum(r[2] for r in res) / len(res)
        print(f"ohx_reserve_cus={reserve:2d}: predict {p:.3f} ms; the 12 MB copy on the other stream: enqueued at {a:.3f} ms, done at {b:.3f} ms "
              f"after the predict's start ({'beside' if b < 0.8 * p else 'behind'} the predict)")
        del booster


if __name__ == "__main__":
    main()
