#!/usr/bin/env python3
"""Per-wave clocks inside one launch of the ring kernel (measurement aid, not product code).

Builds an INSTRUMENTED copy of the library in a scratch directory - the product tree is not touched:
predict_rows_ring_kernel sums s_memtime over the ring's bookkeeping (claim, look at `filled`, a stager's wait for
`progress` and its DMA issue, any wait for `filled`) and over the group walks of every wave, counts the groups a wave
staged, reads HW_ID (SIMD and slot), and the waves of a few blocks print their line at the end of the step's first
launch.  Runs one C360 step with it and prints the lines sorted by block and wave (profiles/r04_ring_wave_clocks.txt).
usage (GPU box): python3 tools/ring_wave_clocks.py [bench.py arguments ...]"""
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
            raise SystemExit(f"ring_wave_clocks: {os.path.basename(path)} no longer contains:\n{old}")
        s = s.replace(old, new, 1)
    open(path, "w").write(s)


def main():
    scratch = tempfile.mkdtemp(prefix="ohx_ring_clocks_")
    for d in ("quickchem_amd", "include", "oracle", "tests"):
        shutil.copytree(os.path.join(ROOT, d), os.path.join(scratch, d), ignore=shutil.ignore_patterns("__pycache__", "_ref"))
    for f in ("bench.py", "BASELINE.json"):
        shutil.copy(os.path.join(ROOT, f), scratch)
    os.makedirs(os.path.join(scratch, "profiles"), exist_ok=True)
    patch(os.path.join(scratch, "quickchem_amd", "csrc", "kernels.hip"), [
        ("struct TopRing {\n  lds_cptr ring;",
         "struct TopRing {\n  uint64_t cyc_book = 0, cyc_walk = 0;\n  uint32_t staged = 0;\n  lds_cptr ring;"),
        ("    uint32_t won = 0;\n    uint32_t filled_now = 0;\n    if (lane == 0) {",
         "    const uint64_t c0 = __builtin_readcyclecounter();\n    uint32_t won = 0;\n    uint32_t filled_now = 0;\n    if (lane == 0) {"),
        ("    ring_order();       // the buffer is read after `filled` said so, not before",
         "    ring_order();       // the buffer is read after `filled` said so, not before\n"
         "    const uint64_t c2 = __builtin_readcyclecounter();\n    rg.cyc_book += c2 - c0;\n    rg.staged += won;"),
        ("    if (lane == 0) ring_store(rg.progress + wave, g + 1u);\n  }\n  __builtin_amdgcn_s_setprio(0);\n  return acc;",
         "    if (lane == 0) ring_store(rg.progress + wave, g + 1u);\n    rg.cyc_walk += __builtin_readcyclecounter() - c2;\n  }\n"
         "  __builtin_amdgcn_s_setprio(0);\n  return acc;"),
        ("  if (rg.gave_up && a.flags) atomicOr(a.flags, kFlagRingTimeout);\n}\n\n// The fused path (predict_fields_kernel's fill and store) around the ring walk.",
         "  if (rg.gave_up && a.flags) atomicOr(a.flags, kFlagRingTimeout);\n"
         "  if (lane == 0 && (blockIdx.x % 85) == 3 && a.tile_begin == 0) {\n"
         "    uint32_t hw;\n    asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\" : \"=s\"(hw));\n"
         "    printf(\"RING block %u wave %2d hwid %08x simd %u: book %7llu walk %8llu staged %3u\\n\", blockIdx.x, wave, hw, (hw >> 4) & 3u,\n"
         "           (unsigned long long)rg.cyc_book, (unsigned long long)rg.cyc_walk, rg.staged);\n  }\n}\n\n"
         "// The fused path (predict_fields_kernel's fill and store) around the ring walk."),
    ])
    r = subprocess.run(["make", "-C", os.path.join(scratch, "quickchem_amd", "csrc"), "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stdout[-2000:] + r.stderr[-2000:])
    r = subprocess.run([sys.executable, os.path.join(scratch, "bench.py"), "--steps", "1", "--warmup", "0", "--cpu-seconds", "0",
                        "--no-verify"] + sys.argv[1:], capture_output=True, text=True, cwd=scratch)
    lines = sorted((ln for ln in (r.stdout + r.stderr).splitlines() if ln.startswith("RING ")),
                   key=lambda ln: (int(ln.split()[2]), int(ln.split()[4])))
    print("\n".join(lines) if lines else r.stdout[-2000:] + r.stderr[-2000:])
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
