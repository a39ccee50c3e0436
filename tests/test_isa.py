"""Guards on the generated gfx950 code of the walk (CPU: hipcc cross-compiles).  Each of these was once lost
without a test noticing, and each costs measurable time on the MI355X (profiles/r02_sweeps.txt):
  * the block's first-step table must be read with ds_read_b128; when hipcc loses the LDS address space of the
    pointer it emits flat_load_dwordx4 and every tree's first step goes through the texture addresser (+3.8 %);
  * no scratch (spills) and at most 84 VGPRs in the default kernel, or a CU holds fewer than 20 waves;
  * super-nodes are fetched as ONE global_load_dwordx4 each (hipcc likes to split the vector, twice the gathers);
  * steps 2 and 3 of a tree take their super-nodes from the lanes that hold the tree's top (ds_bpermute_b32), and
    the top is one global_load_dwordx4 per tree, issued before the first step and not waited for until after it."""
import os
import re
import shutil
import subprocess

import pytest

from tests import helpers

HIPCC = "/opt/rocm/bin/hipcc"
DEFAULT_KERNEL = "predict_rows_tile_kernelILi2ELi2ELb1ELb1E"      # <super-nodes, 2 chains, prefetch, tree tops>


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "kernels.s"
    src = os.path.join(helpers.ROOT, "quickchem_amd", "csrc", "kernels.hip")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only",
                        src, "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def kernel_body(text, name_part):
    m = re.search(r"^(_Z\w*" + re.escape(name_part) + r"\w*):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel", text, re.S | re.M)
    assert m, name_part
    return m.group(2)


def test_no_flat_loads_and_no_scratch_anywhere(isa):
    assert "flat_load" not in isa and "flat_store" not in isa
    assert "scratch_load" not in isa and "scratch_store" not in isa


def test_default_walk_kernel_shape(isa):
    body = kernel_body(isa, DEFAULT_KERNEL)
    assert body.count("ds_read_b128") >= 4                      # first-step table: two chains x with/without missing values
    assert body.count("global_load_dwordx4") >= 16              # the rows' pieces, the tree tops
    assert body.count("buffer_load_dwordx4") >= 4               # the gathers below the tops: one 128-bit load each
    # two chains x two steps x four dwords, with and without missing values; a few more where a tree is beyond the table
    assert body.count("ds_bpermute_b32") >= 32
    vgpr = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", body).group(1))
    assert vgpr <= 84, vgpr                                     # 6 waves per SIMD leave 85
    assert int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", body).group(1)) == 0


def test_fields_kernel_reads_its_table_from_lds_too(isa):
    body = kernel_body(isa, "predict_fields_kernelILi2ELi2ELb1E")
    assert body.count("ds_read_b128") >= 4


@pytest.mark.parametrize("kernel", ["predict_rows_ring_kernel", "predict_fields_ring_kernel"])
def test_ring_kernels_shape(isa, kernel):
    """The ring kernels (tree tops resident in LDS): the first four steps of four chains are ds_read_b128 (with and
    without missing values: 2 x 4 x 4), the deep steps one buffer_load_dwordx4 per chain, the staging is LDS-DMA
    (`buffer_load_dwordx4 ... lds`, eleven per group, for group 0 and in the loop), the claim is an LDS
    compare-and-swap, there is ONE s_barrier (behind group 0), no scratch, and at most 128 VGPRs (16 waves per CU)."""
    body = kernel_body(isa, kernel)
    assert body.count("ds_read_b128") >= 32
    dma = len(re.findall(r"buffer_load_dwordx4 [^\n]* lds", body))
    assert dma == 22, dma
    assert body.count("buffer_load_dwordx4") - dma >= 8
    assert body.count("ds_cmpst_rtn_b32") == 1
    assert body.count("s_barrier") == 1
    assert "ds_bpermute_b32" not in body or kernel == "predict_rows_ring_kernel"     # (the rows' pieces use it; the walk does not)
    vgpr = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", body).group(1))
    assert vgpr <= 128, vgpr
    assert int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", body).group(1)) == 0
