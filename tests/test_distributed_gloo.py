"""The N > 1 path on CPU: world_size 2 over gloo.  The per-rank compute is stood in for by
the oracle; what is under test is the sharding and the reassembly bench.py uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from quickchem_amd import shard, synth
from tests import helpers


def test_row_shards_tile_the_batch():
    for n_total in (0, 1, 7, 64, 55_987_200, 426_124_800):
        for world in (1, 2, 3, 4, 8):
            pos = 0
            sizes = []
            for r in range(world):
                row0, n = shard.row_shard(n_total, world, r)
                assert row0 == pos
                pos += n
                sizes.append(n)
            assert pos == n_total and max(sizes) - min(sizes) <= 1
    assert shard.row_shard(55_987_200, 8, 3) == (3 * 6_998_400, 6_998_400)


def test_plan_pieces_never_costs_rounds():
    plane, rr = 360 * 2160, 256 * 20 * 64
    for levels, want in ((9, [5, 4]), (18, None), (36, None), (72, None), (1, [1])):
        n = levels * plane
        p = shard.plan_pieces(n, 4, plane, rr)
        assert p[0][0] == 0 and p[-1][1] == n and all(a[1] == b[0] for a, b in zip(p, p[1:]))
        assert all((hi - lo) % plane == 0 for lo, hi in p) and len(p) <= 4
        assert sum(-(-(hi - lo) // rr) for lo, hi in p) <= -(-n // rr)        # no more rounds than uncut
        if want:
            assert [(hi - lo) // plane for lo, hi in p] == want
    assert shard.plan_pieces(1000, 4, 0, rr) == [(0, 1000)] and shard.plan_pieces(10, 1, 2, 4) == [(0, 10)]
    # ragged tail: the last piece is short
    p = shard.plan_pieces(10 * plane + 17, 3, plane, rr)
    assert p[-1][1] == 10 * plane + 17


def test_plan_pieces_prices_the_gather_nothing_hides():
    """With a world to gather over, a plan is priced: its rounds, the all-gather of its last piece, a little per piece.
    C360 on the ring kernels' rounds (one 16-wave block per CU)."""
    plane, rr = 360 * 2160, 256 * 16 * 64
    for world, max_pieces, want in ((2, 4, [9, 9, 9, 9]), (4, 4, [5, 5, 4, 4]), (8, 4, [3, 2, 2, 2]), (8, 6, [2, 2, 2, 2, 1])):
        n = 72 // world * plane
        p = shard.plan_pieces(n, max_pieces, plane, rr, world)
        assert p[0][0] == 0 and p[-1][1] == n and all(a[1] == b[0] for a, b in zip(p, p[1:]))
        assert [(hi - lo) // plane for lo, hi in p] == want
        assert sum(-(-(hi - lo) // rr) for lo, hi in p) <= -(-n // rr) + 1          # at most one round dearer than uncut
    assert shard.plan_pieces(36 * plane, 4, plane, rr, 1) == [(0, 36 * plane)]      # nothing to gather: rounds decide
    assert shard.plan_pieces(plane, 6, plane, rr, 8) == [(0, plane)]


def test_the_planner_prices_can_be_given_and_every_priced_plan_is_reported():
    """the two guessed prices are not compiled in any more - argument, else environment, else default -
    and plan_pieces_priced returns the table the choice was made from, so a measured 8-GPU line can be re-planned."""
    d = shard.price_list(environ={})
    assert d["gather_bytes_per_s"] == shard.GATHER_BYTES_PER_S and d["piece_rounds"] == shard.PIECE_ROUNDS
    assert "guess" in d["source"]["gather_bytes_per_s"]
    e = shard.price_list(environ={"OHX_GATHER_GBPS": "40", "OHX_PIECE_ROUNDS": "1.5"})
    assert e["gather_bytes_per_s"] == 40e9 and e["piece_rounds"] == 1.5 and e["source"]["piece_rounds"] == "given"
    assert shard.price_list(250.0, 0.1, environ={"OHX_GATHER_GBPS": "40"})["gather_bytes_per_s"] == 250e9     # argument first
    with pytest.raises(ValueError):
        shard.price_list(0.0, None, environ={})
    plane, rr = 360 * 2160, 256 * 16 * 64
    n = 9 * plane
    best, table = shard.plan_pieces_priced(n, 6, plane, rr, 8, d)
    assert best == shard.plan_pieces(n, 6, plane, rr, 8) and [r["pieces"] for r in table] == [1, 2, 3, 4, 5, 6]
    assert sum(r["chosen"] for r in table) == 1 and [r for r in table if r["chosen"]][0]["sizes"] == [hi - lo for lo, hi in best]
    assert all(abs(r["cost_rounds"] - (r["rounds"] + r["exposed_rounds"] + d["piece_rounds"] * r["pieces"])) < 1e-9 for r in table)
    assert min(table, key=lambda r: r["cost_rounds"])["chosen"]
    # a slow link makes small last pieces worth more launches, a dear piece fewer of them
    slow = shard.plan_pieces(n, 6, plane, rr, 8, shard.price_list(10.0, 0.4, environ={}))
    dear = shard.plan_pieces(n, 6, plane, rr, 8, shard.price_list(100.0, 5.0, environ={}))
    assert len(slow) >= len(best) >= len(dear) and len(dear) < len(slow)


def test_phases_record_shows_every_rank():
    """bench.py's `phases`: per-rank step and predict times, the exposed part of the gather per rank, who was slowest,
    the control loop without the gather (--gather both) and what the gather costs by that."""
    pieces = [(0, 100), (100, 160)]
    rec = shard.phases_record([[10.0, 8.0], [12.5, 8.5]], [160, 160], 320, pieces, True, [{"pieces": 2, "chosen": True}],
                              shard.price_list(environ={}), control=[[8.2, 8.0], [8.9, 8.5]])
    assert rec["per_rank"]["step_ms"] == [10.0, 12.5] and rec["per_rank"]["exposed_gather_ms"] == [2.0, 4.0]
    assert rec["slowest_rank"] == {"step": 1, "predict": 1} and rec["predict_ms"] == 8.5 and rec["exposed_gather_ms"] == 4.0
    assert rec["predict_only"]["step_ms"] == 8.9 and abs(rec["gather_costs_ms"] - 3.6) < 1e-9
    assert rec["pieces"] == [100, 60] and rec["gather_bytes_per_rank_sent"] == [640, 640] and rec["gather_bytes_total"] == 1280
    assert "predict_only" not in shard.phases_record([[1.0, 1.0]], [5], 5, [(0, 5)], True, [], shard.price_list(environ={}))
    assert shard.rank_times([1, 2.5]) == [[1.0, 2.5]]                 # no process group: this process alone


def test_chunk_bounds_cover_the_shard():
    for n_local in (1, 639, 640, 641, 6_998_400, 53_265_600):
        for k in (1, 2, 4, 7):
            b = shard.chunk_bounds(n_local, k, 655_360)
            assert b[0][0] == 0 and b[-1][1] == n_local and len(b) <= max(k, 1)
            assert all(hi > lo for lo, hi in b) and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
            assert all((hi - lo) % 655_360 == 0 for lo, hi in b[:-1])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, image_path, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        grid = synth.GRIDS["C12"]
        image = np.fromfile(image_path, dtype=np.uint8)
        row0, n = shard.row_shard(n_total, world, rank)
        rows = synth.rows_cpu(grid, row0, n)                      # each rank generates only its shard
        local = torch.from_numpy(helpers.oracle_predict(image, rows, synth.XX_MISS))
        full = torch.empty(n_total, dtype=torch.float32)
        shard.all_gather_rows(full, local, n_total, world, n_total % world == 0)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), full.numpy())
        if n_total % world == 0:
            # the overlapped form bench.py uses: pieces gathered asynchronously into their final places
            full2 = torch.zeros(n_total, dtype=torch.float32)
            pieces = shard.chunk_bounds(n, 3, 640)
            gather = shard.ChunkGather(full2, n, world, pieces)
            for q in range(len(pieces)):
                gather.start(q, local)
            gather.finish()
            assert torch.equal(full2.view(torch.int32), full.view(torch.int32))
            gather = shard.ChunkGather(full2.zero_(), n, world, pieces)          # a second step reuses the staging
            for q in reversed(range(len(pieces))):
                gather.start(q, local)
            gather.finish()
            assert torch.equal(full2.view(torch.int32), full.view(torch.int32))
        # what bench.py's `phases` is made of: every rank's numbers on every rank, in rank order
        assert shard.rank_times([rank + 0.5, 10 * rank]) == [[r + 0.5, 10.0 * r] for r in range(world)]
        # the C ABI's shard rule is the Python one (OHXShardRows: what a Fortran/MPI host would call)
        from quickchem_amd import capi
        for r in range(world):
            assert capi.shard_rows(n_total, world, r) == shard.row_shard(n_total, world, r)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [4096, 4099])
def test_two_ranks_reassemble_the_field(tmp_path, small_model, n_total):
    image_path = tmp_path / "m.bin"
    small_model.image.tofile(image_path)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_total, str(image_path), str(tmp_path)), nprocs=2, join=True)
    rows = synth.rows_cpu(synth.GRIDS["C12"], 0, n_total)
    want = helpers.oracle_predict(small_model.image, rows, synth.XX_MISS)
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}.npy")
        assert np.array_equal(helpers.bits(got), helpers.bits(want))


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` is how the driver runs every N: with N > 1 and no WORLD_SIZE the process must turn
    into a launcher of N fresh rank processes (never an exec, never after a GPU call), with the caller's own
    arguments; under torch.distributed.run (WORLD_SIZE set) it must be a rank."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    argv = bench.launcher_argv(4, ["--gpus", "4", "--steps", "7", "--warmup", "2"], 29999)
    assert argv[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert argv[argv.index("--nproc-per-node") + 1] == "4" and "--nnodes=1" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29999"
    at = argv.index(os.path.join(root, "bench.py"))
    assert argv[at + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]      # the ranks see what the caller said
    src = open(os.path.join(root, "bench.py")).read()
    assert "os.exec" not in src and "execv" not in src
    # no GPU here: the launcher says so before starting anything (and not the old "launch with: ..." refusal)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=env, timeout=300)
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "this node shows" in r.stderr and "launch with" not in r.stderr
    # as a rank of another world size it refuses, naming both ways to start it
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr
