"""OHXAllGatherOH's direct exchange - the group of ncclSend / ncclRecv per peer that a multi-GPU job with unequal shards
takes (quickchem_amd/csrc/comm.cpp) - executed for real, by several ranks, on the one GPU a box has.

RCCL refuses two ranks on one device, and no multi-GPU node has been available to this build, so until round 6 that
branch had never run.  Here the ranks are processes that share the GPU, and comm.cpp's dlopen("librccl.so") finds
tests/mock_rccl/lib/librccl.so first (LD_LIBRARY_PATH): a TEST DOUBLE that moves a message as a file between the ranks
(tests/mock_rccl/mock_rccl.cpp).  Everything above RCCL's ten entry points is the product's own code: the shard
arithmetic (OHXShardRows), the offsets into the gathered field, the device copy of a rank's own rows, the group and its
guard, the handle table.  What the double cannot show is RCCL itself.  The ranks use no torch: device memory through the
HIP runtime by ctypes, so that the only librccl in the process is the one comm.cpp loads."""
import os
import subprocess
import sys

import pytest

from tests import helpers

pytestmark = pytest.mark.gpu

MOCK_DIR = os.path.join(helpers.ROOT, "tests", "mock_rccl", "lib")

RANK = r"""
import ctypes as C, os, sys, time
import numpy as np
root, rank, nranks, meet = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, root)
from quickchem_amd import capi
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
assert hip.hipSetDevice(0) == 0
lib = capi.load_library()
assert capi.Communicator.rccl_version() == 0, "this is not the test double"

def dev(n):
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), max(4 * n, 4)) == 0
    return p

def wait_for(path):
    t0 = time.time()
    while not os.path.exists(path):
        assert time.time() - t0 < 120, "the other ranks never came: " + path
        time.sleep(0.001)

# the unique id travels the way a Fortran host's MPI would carry it: rank 0 makes it, the others read it
idfile = os.path.join(meet, "unique_id")
if rank == 0:
    uid = capi.Communicator.unique_id()
    open(idfile + ".part", "wb").write(uid)
    os.rename(idfile + ".part", idfile)
wait_for(idfile)
comm = capi.Communicator(open(idfile, "rb").read(), nranks, rank)

def expected(n_total):
    return (np.arange(n_total, dtype=np.float32) * np.float32(0.5) - np.float32(7.0))

checked = 0
for n_total, in_place in [(4099, False), (4099, True), (nranks * 1000, False), (nranks - 1, False), (10 * nranks + 1, True)]:
    row0, n = capi.shard_rows(n_total, nranks, rank) if hasattr(capi, "shard_rows") else (None, None)
    if row0 is None:
        r0, nn = C.c_uint64(), C.c_uint64()
        capi.check(lib, lib.OHXShardRows(n_total, nranks, rank, C.byref(r0), C.byref(nn)))
        row0, n = r0.value, nn.value
    want = expected(n_total)
    full = dev(n_total)
    assert hip.hipMemset(full, 0xFF, 4 * n_total) == 0                 # NaN everywhere: a row nobody wrote shows
    mine = np.ascontiguousarray(want[row0:row0 + n])
    if in_place:
        shard = C.c_void_p(full.value + 4 * row0)                       # the shard already sits at its rows
    else:
        shard = dev(n)
    if n:
        assert hip.hipMemcpy(shard, mine.ctypes.data, 4 * n, 1) == 0
    comm.all_gather_oh(shard.value, n, n_total, full.value)
    assert hip.hipDeviceSynchronize() == 0
    got = np.empty(n_total, dtype=np.float32)
    assert hip.hipMemcpy(got.ctypes.data, full, 4 * n_total, 2) == 0
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (rank, n_total, in_place, int((got.view(np.uint32) != want.view(np.uint32)).sum()))
    checked += 1
# a shard of the wrong size is refused before anything is enqueued (the peers would hang otherwise)
try:
    comm.all_gather_oh(dev(5).value, 5, 4099, dev(4099).value)
    raise SystemExit("a shard that is not OHXShardRows' was accepted")
except capi.OhxError as e:
    assert "OHXShardRows" in str(e), str(e)
comm.free()
print("RANK_OK", rank, checked)
"""


@pytest.mark.skipif(not os.path.exists(os.path.join(MOCK_DIR, "librccl.so")), reason="tests/mock_rccl not built (make -C tests/mock_rccl)")
@pytest.mark.parametrize("nranks,pairs", [(2, False), (3, False), (3, True)])
def test_direct_exchange_between_ranks_sharing_the_gpu(tmp_path, nranks, pairs):
    """Unequal shards (4 099 rows over 2 or 3 ranks; fewer rows than ranks, so that a rank holds none; 10 n + 1) take the
    send / receive group; equal shards (1 000 per rank) take ncclAllGather - or, with OHX_ALLGATHER=pairs, the group as
    well.  Separate shard buffers and shards in place.  Every rank ends with every row, bit for bit, and no row is
    written that should not be."""
    env = dict(os.environ, OHX_MOCK_RCCL_DIR=str(tmp_path),
               LD_LIBRARY_PATH=MOCK_DIR + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    if pairs:
        env["OHX_ALLGATHER"] = "pairs"
    else:
        env.pop("OHX_ALLGATHER", None)
    procs = [subprocess.Popen([sys.executable, "-c", RANK, helpers.ROOT, str(r), str(nranks), str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(nranks)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out))
    for r, (rc, out) in enumerate(outs):
        assert rc == 0 and f"RANK_OK {r} 5" in out, (r, out[-3000:])
