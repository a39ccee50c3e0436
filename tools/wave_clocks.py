#!/usr/bin/env python3
"""Per-wave clocks inside one launch of the rows kernel (measurement aid, not product code).

Builds an INSTRUMENTED copy of the library in a scratch directory - the product tree is not touched: the rows kernel
records wall_clock64() at every wave's start, after its first tile and at its end for the first launch of a step
into a __device__ array, and an extra entry point hands the array out.  Then runs the C360 step and prints where a
launch's time goes.  usage (GPU box): python3 tools/wave_clocks.py [--launch k] [name=value booster parameters ...]"""
import ctypes as C
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def patch(path, pairs):
    s = open(path).read()
    for old, new in pairs:
        if old not in s:
            raise SystemExit(f"wave_clocks: {os.path.basename(path)} no longer contains:\n{old}")
        s = s.replace(old, new, 1)
    open(path, "w").write(s)


def build_instrumented(scratch, launch):
    for d in ("quickchem_amd", "include", "oracle"):
        shutil.copytree(os.path.join(ROOT, d), os.path.join(scratch, d), ignore=shutil.ignore_patterns("__pycache__"))
    csrc = os.path.join(scratch, "quickchem_amd", "csrc")
    patch(os.path.join(csrc, "kernels.hip"), [
        ("// ------------------------------------------------------------------ kernels\n",
         "// ------------------------------------------------------------------ kernels\n"
         f"#define WAVE_CLOCKS_LAUNCH {launch}\n__device__ unsigned long long g_wave_clocks[4 * 8192];\n"),
        ("  extern __shared__ float lds[];\n  const int lane = threadIdx.x & (kWave - 1);\n  const int wave = threadIdx.x / kWave;\n"
         "  // the waves' feature tiles first",
         "  extern __shared__ float lds[];\n  const int lane = threadIdx.x & (kWave - 1);\n  const int wave = threadIdx.x / kWave;\n"
         "  const unsigned long long t_start = wall_clock64();\n  unsigned long long t_mid = 0, t_fill = 0;\n"
         "  // the waves' feature tiles first"),
        ("      const uint64_t next = tile_id + nwaves;\n      const uint64_t this_row = row;",
         "      if (t_fill == 0) t_fill = wall_clock64();\n"
         "      const uint64_t next = tile_id + nwaves;\n      const uint64_t this_row = row;"),
        ("      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n      tile_id = next;\n    }\n    return;",
         "      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, \"wavefront\");\n      tile_id = next;\n"
         "      if (t_mid == 0) t_mid = wall_clock64();\n    }\n"
         "    if (a.tile_begin == (uint64_t)WAVE_CLOCKS_LAUNCH * (a.tile_end - a.tile_begin) && lane == 0) {\n"
         "      const unsigned w = blockIdx.x * kWavesPerBlock + wave;\n"
         "      if (w < 8192) { g_wave_clocks[4 * w] = t_start; g_wave_clocks[4 * w + 1] = t_mid; g_wave_clocks[4 * w + 2] = wall_clock64(); g_wave_clocks[4 * w + 3] = t_fill; }\n"
         "    }\n    return;"),
        ("uint32_t cluster_key_bits(const ClusterArgs& a) {",
         "hipError_t debug_wave_clocks(unsigned long long* out) {\n"
         "  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_clocks), sizeof(unsigned long long) * 4 * 8192);\n}\n\n"
         "uint32_t cluster_key_bits(const ClusterArgs& a) {"),
    ])
    patch(os.path.join(csrc, "kernels.hpp"), [
        ("uint32_t cluster_key_bits(const ClusterArgs& a);",
         "hipError_t debug_wave_clocks(unsigned long long* out);\nuint32_t cluster_key_bits(const ClusterArgs& a);")])
    patch(os.path.join(csrc, "capi.cpp"), [
        ("int OHXReleaseScratch(void) {",
         "int OHXDebugWaveClocks(unsigned long long* out) {\n  API_BEGIN();\n  HIP_CHECK(hipDeviceSynchronize());\n"
         "  HIP_CHECK(debug_wave_clocks(out));\n  API_END();\n}\n\nint OHXReleaseScratch(void) {")])
    r = subprocess.run(["make", "-C", csrc, "../lib/libohxgb.so"], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("wave_clocks: the instrumented build failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])


def main():
    scratch = tempfile.mkdtemp(prefix="ohx_wave_clocks_")
    launch = 0
    if len(sys.argv) > 2 and sys.argv[1] == "--launch":          # which launch of the step's train to record
        launch = int(sys.argv[2])
        del sys.argv[1:3]
    build_instrumented(scratch, launch)
    sys.path.insert(0, scratch)
    import numpy as np
    import torch
    from quickchem_amd import capi, synth
    torch.cuda.set_device(0)
    grid = synth.GRIDS["C360"]
    n = grid[0] * grid[1] * grid[2]
    model = synth.make_model()
    b = capi.Booster(model_buffer=model.image)
    for kv in sys.argv[1:]:
        k, _, v = kv.partition("=")
        b.set_param(k, v)
    rows = torch.empty((n, synth.NFEAT), dtype=torch.float32, device="cuda:0")
    synth.rows_device(grid, 0, n, rows)
    d = capi.DMatrix(device_ptr=rows.data_ptr(), nrow=n, ncol=synth.NFEAT, missing=synth.XX_MISS)
    d.set_grid(grid[0], grid[1], 0)
    out = torch.empty(n, dtype=torch.float32, device="cuda:0")
    for _ in range(3):
        b.predict_device(d, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (4 * 8192))()
    capi.check(b.lib, b.lib.OHXDebugWaveClocks(buf))
    t = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 4).astype(np.float64)
    t = t[t[:, 2] > 0]
    t0 = t[:, 0].min()
    start, mid, end, fill = ((t[:, i] - t0) / 100.0 for i in range(4))  # microseconds: the clock runs at 100 MHz
    pct = lambda a: " ".join(f"p{p} {v:.1f}" for p, v in zip((5, 50, 95), np.percentile(a, [5, 50, 95])))  # noqa: E731
    print(f"{b.kernel_symbol(synth.NFEAT)}  params {sys.argv[1:]}")
    print(f"launch {launch} of a C360 step: {len(t)} waves, span {end.max():.1f} us; last wave started at {start.max():.1f} us")
    print(f"its rows are in LDS after mean {np.mean(fill - start):.1f} us  {pct(fill - start)}  max {np.max(fill - start):.1f}")
    print(f"first tile   mean {np.mean(mid - start):.1f} us  {pct(mid - start)}  max {np.max(mid - start):.1f}")
    print(f"later tiles  mean {np.mean(end - mid):.1f} us  {pct(end - mid)}  max {np.max(end - mid):.1f}")
    print(f"a wave ends  mean {end.mean():.1f} us  {pct(end)}  max {end.max():.1f}")
    print(f"idle slot-time at the tail: {100 * (end.max() - end.mean()) / end.max():.1f} % of the launch")
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
