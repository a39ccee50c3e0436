"""Row sharding of the gridcell batch across ranks and reassembly of the OH field.

Every gridcell is independent, so rank r of W predicts the contiguous
rows [row0, row0 + n) with a replicated booster and no data-path collective; the one
exchange step is an all-gather of the float32 OH shard (RCCL over xGMI on the GPUs,
`gloo` in the CPU tests).  Inside GEOS itself the export stays distributed and no
collective is needed at all."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def row_shard(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """(row0, nrows) of rank's shard: contiguous, sizes differing by at most one row."""
    base, rem = divmod(n_total, world)
    n = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)
    return row0, n


def _gather_flat(out: torch.Tensor, inp: torch.Tensor, world: int) -> None:
    """all_gather_into_tensor, or its list form under gloo on GPU tensors (rehearsals on a shared GPU)."""
    if dist.get_backend() == "gloo" and inp.is_cuda:
        n = inp.numel()
        dist.all_gather([out[r * n:(r + 1) * n] for r in range(world)], inp)
    else:
        dist.all_gather_into_tensor(out, inp)


def all_gather_rows(out_full: torch.Tensor, out_local: torch.Tensor, n_total: int, world: int, even: bool) -> None:
    """Every rank ends with the whole field, shards in rank (= row) order."""
    if even:
        _gather_flat(out_full, out_local, world)
        return
    # ragged shards: gather fixed-size slots, then compact
    slot = (n_total + world - 1) // world
    padded = out_local.new_zeros(slot)
    padded[: out_local.numel()] = out_local
    slots = out_local.new_empty(world * slot)
    _gather_flat(slots, padded, world)
    for r in range(world):
        row0, n = row_shard(n_total, world, r)
        out_full[row0:row0 + n] = slots[r * slot: r * slot + n]


def chunk_bounds(n_local: int, nchunks: int, granule: int) -> list:
    """Cut [0, n_local) into at most `nchunks` pieces whose sizes are multiples of `granule`
    (the rows one launch covers), so that a piece's all-gather can overlap the next piece's
    prediction without leaving ragged launches behind."""
    if nchunks <= 1 or n_local <= granule:
        return [(0, n_local)]
    per = -(-n_local // nchunks)
    per = -(-per // granule) * granule
    out, lo = [], 0
    while lo < n_local:
        hi = min(n_local, lo + per)
        out.append((lo, hi))
        lo = hi
    return out


# plan_pieces' price list, in rounds of the chip (one residency's rows walked through the OH booster, ~0.11 ms on an MI355X).
# GATHER_BYTES_PER_S and PIECE_ROUNDS are guesses no N > 1 run has confirmed: bench.py takes them
# from --gather-gbps / --piece-rounds (or OHX_GATHER_GBPS / OHX_PIECE_ROUNDS in the environment), echoes what it used
# and prints every plan it priced, so that the first 8-GPU line can be re-planned from what it measured.
WALK_ROWS_PER_S = 2.3e9          # bench.py, C360 step
GATHER_BYTES_PER_S = 100e9       # what one rank takes in during an all-gather over xGMI - a guess on the safe side
PIECE_ROUNDS = 0.4               # one more launch, one more collective enqueued: 40-45 us a piece, measured on a C360/8 shard
                                 # with one-rank RCCL (profiles/r04_sweeps.txt)


def price_list(gather_gbps=None, piece_rounds=None, environ=None) -> dict:
    """The two guessed prices: argument, else environment (OHX_GATHER_GBPS, OHX_PIECE_ROUNDS), else the module's."""
    import os
    env = os.environ if environ is None else environ
    g = gather_gbps if gather_gbps is not None else (float(env["OHX_GATHER_GBPS"]) if env.get("OHX_GATHER_GBPS") else None)
    q = piece_rounds if piece_rounds is not None else (float(env["OHX_PIECE_ROUNDS"]) if env.get("OHX_PIECE_ROUNDS") else None)
    out = {"gather_bytes_per_s": GATHER_BYTES_PER_S if g is None else g * 1e9,
           "piece_rounds": PIECE_ROUNDS if q is None else q,
           "walk_rows_per_s": WALK_ROWS_PER_S,
           "source": {"gather_bytes_per_s": "default (a guess)" if g is None else "given",
                      "piece_rounds": "default (one-rank RCCL)" if q is None else "given"}}
    if out["gather_bytes_per_s"] <= 0 or out["piece_rounds"] < 0:
        raise ValueError("gather GB/s must be positive and rounds per piece not negative")
    return out


def plan_pieces_priced(n_local: int, max_pieces: int, granule: int, round_rows: int, world: int = 1, prices=None):
    """plan_pieces, and the table it chose from: -> (pieces, [{"pieces", "sizes", "rounds", "exposed_rounds",
    "cost_rounds", "chosen"}]).  `prices` = price_list(...)."""
    prices = prices or price_list(environ={})
    gbs, per_piece, walk = prices["gather_bytes_per_s"], prices["piece_rounds"], prices["walk_rows_per_s"]
    if max_pieces <= 1 or granule <= 0 or n_local <= granule:
        return [(0, n_local)], [{"pieces": 1, "sizes": [n_local], "rounds": -(-n_local // round_rows) if round_rows else 0,
                                 "exposed_rounds": None, "cost_rounds": None, "chosen": True}]
    units = -(-n_local // granule)

    def exposed(last_rows):
        if world <= 1:
            return 0.0
        return last_rows * 4.0 * (world - 1) / gbs / (round_rows / walk)

    best, best_rounds = [(0, n_local)], -(-n_local // round_rows)
    best_cost = best_rounds + exposed(n_local) + per_piece
    table = [{"pieces": 1, "sizes": [n_local], "rounds": best_rounds, "exposed_rounds": exposed(n_local),
              "cost_rounds": best_cost}]
    for k in range(2, min(max_pieces, units) + 1):
        base, rem = divmod(units, k)
        sizes = [(base + (1 if q < rem else 0)) * granule for q in range(k)]
        out, lo = [], 0
        for sz in sizes:
            hi = min(n_local, lo + sz)
            out.append((lo, hi))
            lo = hi
        rounds = sum(-(-(hi - lo) // round_rows) for lo, hi in out)
        cost = rounds + exposed(out[-1][1] - out[-1][0]) + per_piece * k
        table.append({"pieces": k, "sizes": [hi - lo for lo, hi in out], "rounds": rounds,
                      "exposed_rounds": exposed(out[-1][1] - out[-1][0]), "cost_rounds": cost})
        if world <= 1:
            if rounds <= best_rounds:
                best, best_rounds = out, rounds
        elif cost < best_cost - 1e-9:
            best, best_cost = out, cost
    for row in table:
        row["chosen"] = row["pieces"] == len(best)
    return best, table


def plan_pieces(n_local: int, max_pieces: int, granule: int, round_rows: int, world: int = 1, prices=None) -> list:
    """Cut [0, n_local) into at most `max_pieces` pieces of whole granules (a grid level when the
    waves take bricks) so that the pieces' all-gathers can overlap the next piece's prediction.
    A piece occupies the GPU for ceil(rows / round_rows) rounds (round_rows = rows one residency of
    the chip covers).  world <= 1 (nothing to gather, or unknown): the plan with the fewest rounds
    wins and, among equals, the one with the most pieces.  world > 1: every plan is priced in
    rounds - its own, plus the all-gather of its LAST piece, which nothing hides (rows x 4 B x
    (world - 1) at the gather rate of `prices`), plus a price per piece - and the cheapest wins: an
    extra round is worth paying when it shrinks what stays exposed (C360 on 2 GPUs: one piece of 36
    levels would leave 112 MB per rank unhidden; four of 9 cost one round more of 107)."""
    return plan_pieces_priced(n_local, max_pieces, granule, round_rows, world, prices)[0]


def rank_times(values) -> list:
    """Every rank's list of numbers, on every rank, in rank order (all_gather_object: works on gloo and on RCCL):
    what bench.py's `phases` prints per rank, so that a straggler or a mis-planned split shows in the first N > 1 line."""
    if not (dist.is_available() and dist.is_initialized()):
        return [list(map(float, values))]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(map(float, values)))
    return out


def phases_record(per_rank, n_local_by_rank, n_total, pieces, even, plan_table, prices, control=None) -> dict:
    """bench.py's `phases` object.  per_rank[r] = [step_ms, predict_ms] of rank r's timed loop (predict = first launch to
    last launch of its pieces, events on the launching stream); control[r] = the same two from a loop WITHOUT the gather
    (--gather both), or None.  exposed_gather_ms[r] = step - predict: what of the all-gather and of the copies into place
    the prediction of the next piece did not hide on rank r."""
    step = [r[0] for r in per_rank]
    pred = [r[1] for r in per_rank]
    rec = {
        "predict_ms": max(pred), "exposed_gather_ms": max(step) - max(pred),
        "per_rank": {"step_ms": step, "predict_ms": pred, "exposed_gather_ms": [s - p for s, p in zip(step, pred)],
                     "rows": list(n_local_by_rank)},
        "slowest_rank": {"step": step.index(max(step)), "predict": pred.index(max(pred))},
        "gather_bytes_per_rank_sent": [4 * n for n in n_local_by_rank], "gather_bytes_total": 4 * n_total,
        "pieces": [hi - lo for lo, hi in pieces], "even_shards": even,
        "plan": {"prices": prices, "priced": plan_table},
        "hardware_note": "no claim: what an N > 1 run says about xGMI is only known once one has run",
    }
    if control is not None:
        cstep = [r[0] for r in control]
        cpred = [r[1] for r in control]
        rec["predict_only"] = {"step_ms": max(cstep), "per_rank_step_ms": cstep, "per_rank_predict_ms": cpred,
                               "what": "the same steps with the all-gather left out, timed in the same launch"}
        rec["gather_costs_ms"] = max(step) - max(cstep)
    return rec


class ChunkGather:
    """Gathers of the pieces of a shard, each overlapping the prediction of the next piece, ending with every
    rank's rows at their place in `out_full` (row order).  A piece [lo, hi) of every rank's equal-sized shard is
    gathered with ONE all_gather_into_tensor into a contiguous staging block [world][hi - lo] - the form RCCL writes
    in place, with no flatten/copy-out hidden inside the backend - and a strided device copy puts it into
    out_full[r * n_local + lo : r * n_local + hi] once the step's predictions are enqueued.  The copies move the
    field once more (224 MB at C360, against 6 GB of features read per step); what they buy is that the
    collectives run while the next piece is being walked."""

    def __init__(self, out_full: torch.Tensor, n_local: int, world: int, pieces):
        self.out_full, self.n_local, self.world = out_full, n_local, world
        self.stage = [out_full.new_empty(world * (hi - lo)) for lo, hi in pieces]
        self.pieces = list(pieces)
        self.pending = []

    def start(self, index: int, out_local: torch.Tensor):
        """Call after piece `index` has been enqueued on the current stream."""
        lo, hi = self.pieces[index]
        src = out_local[lo:hi]
        if dist.get_backend() == "gloo" and src.is_cuda:       # rehearsal on a shared GPU: gloo has no flat form there
            n = hi - lo
            work = dist.all_gather([self.stage[index][r * n:(r + 1) * n] for r in range(self.world)], src, async_op=True)
        else:
            work = dist.all_gather_into_tensor(self.stage[index], src, async_op=True)
        self.pending.append((work, index))

    def finish(self):
        """Wait for the gathers and put the pieces at their rows; leaves the copies on the current stream."""
        full = self.out_full[: self.world * self.n_local].view(self.world, self.n_local)
        for work, index in self.pending:
            work.wait()
            lo, hi = self.pieces[index]
            full[:, lo:hi].copy_(self.stage[index].view(self.world, hi - lo))
        self.pending = []
