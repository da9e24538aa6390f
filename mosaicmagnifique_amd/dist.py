"""Cell-sharded multi-GPU run: one process per GPU, ``torch.distributed`` (backend
``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` on CPU for the tests).

The reference has no multi-GPU code (one ``CUDAPhotomosaicGenerator(device)``,
CUDA/CUDAPhotomosaicGenerator.h:31).  The split is new and follows the data
dependencies of the path: base scores of different cells are independent, so
valid cells (raster order) are cut into ``world`` contiguous blocks; library,
masks and target are replicated (<= ~1.1 GB even for config E, trivial against
288 GB of HBM).  The only exchange is ONE all-gather of the per-cell candidate
lists (K x 12 bytes per cell: a few KB to a few MB; indices and scores travel in one
packed byte buffer), after which every rank runs
the same deterministic resolve, so all ranks end with the identical best-fit grid.
The payload is latency-bound, far below the size where the 7 x ~153 GB/s xGMI
links matter: a single collective on one stream, no bucketing, no overlap games.
"""
import numpy as np


def shard_range(n_cells, world, rank):
    """Contiguous block [lo, hi) of the raster-ordered cells owned by ``rank``;
    blocks are ceil(n/world) long (the last ones may be short or empty)."""
    per = -(-n_cells // world) if n_cells else 0
    lo = min(rank * per, n_cells)
    return lo, min(lo + per, n_cells)


def padded_len(n_cells, world):
    return -(-n_cells // world) if n_cells else 0


def pack_layout(per, k):
    """Byte layout of one rank's candidate lists inside the single all-gather buffer:
    ``[per][k]`` int32 indices, padded to 8 bytes, then ``[per][k]`` float64 scores."""
    idx_bytes = per * k * 4
    score_off = (idx_bytes + 7) // 8 * 8
    return score_off, score_off + per * k * 8


def allgather_candidates(idx_local, score_local, n_cells, world, group=None, packed=None):
    """ONE all-gather of the shards' candidate lists.

    idx_local  int32  [per][k]   (torch tensor, rows past the shard's own count are padding)
    score_local f64   [per][k]
    packed: optional uint8 tensor [pack_layout(per, k)[1]] whose first bytes ARE idx_local and
    whose tail IS score_local (ShardedBestFit allocates them as views of it); without it the
    two are copied into a fresh buffer first.
    Returns (idx [n_cells][k], score [n_cells][k]) identical on every rank.
    Works on CUDA tensors (nccl/RCCL) and CPU tensors (gloo).
    """
    import torch
    import torch.distributed as dist
    per, k = idx_local.shape
    assert per == padded_len(n_cells, world) and score_local.shape == (per, k)
    score_off, total = pack_layout(per, k)
    if packed is None:
        packed = torch.zeros(total, dtype=torch.uint8, device=idx_local.device)
        packed[:per * k * 4].view(torch.int32).view(per, k).copy_(idx_local)
        packed[score_off:].view(torch.float64).view(per, k).copy_(score_local)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        gathered = packed.view(1, total)
    else:
        gathered = torch.empty((world, total), dtype=torch.uint8, device=packed.device)
        dist.all_gather_into_tensor(gathered.view(-1), packed, group=group)
    idx_all = gathered[:, :per * k * 4].contiguous().view(torch.int32).view(world * per, k)
    score_all = gathered[:, score_off:].contiguous().view(torch.float64).view(world * per, k)
    # shards are contiguous blocks of `per`, so trimming the tail removes exactly the padding
    return idx_all[:n_cells], score_all[:n_cells]


class ShardedBestFit:
    """generateBestFits over ``world`` GPUs (this process drives GPU ``local_rank``).

    engine: an Engine with library / masks / target already staged (replicated).
    """

    def __init__(self, engine, rank, world, group=None):
        self.engine, self.rank, self.world, self.group = engine, rank, world, group
        self._bufs = None
        self.reset_phases()

    def reset_phases(self):
        """Zero the per-phase wall clocks of generate() (ms, this rank; `rounds` = resolve passes)."""
        self.phase_ms = {"score": 0.0, "allgather": 0.0, "resolve": 0.0, "deepen": 0.0, "rounds": 0}

    def _buffers(self, per, k):
        import torch
        key = (per, k)
        if self._bufs is None or self._bufs[0] != key:
            # a stand-in engine of the CPU tests names its own device; the real one is a GPU
            dev = getattr(self.engine, "torch_device", None) or torch.device("cuda", self.engine.device)
            rows = max(per, 1)
            score_off, total = pack_layout(rows, k)
            packed = torch.zeros(total, dtype=torch.uint8, device=dev)      # what the one all-gather sends
            idx = packed[:rows * k * 4].view(torch.int32).view(rows, k)
            score = packed[score_off:].view(torch.float64).view(rows, k)
            idx.fill_(-1)
            score.fill_(float("inf"))
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)     # the engine writes these on its own stream
            self._bufs = (key, idx, score, packed)
        return self._bufs[1], self._bufs[2], self._bufs[3]

    def _same_on_all_ranks(self, todo, device):
        """Every rank must derive the SAME set of unverified cells (identical inputs, identical
        deterministic resolve) or the next all-gather would pair up different rounds and hang:
        one tiny all-reduce (count and two checksums, as v and -v under MAX) turns a divergence into an error."""
        import torch
        import torch.distributed as dist
        if self.world == 1 or not (dist.is_available() and dist.is_initialized()):
            return
        digest = [float(len(todo)), float(int(todo.sum()) % 1000003), float(int((todo * todo % 7919).sum()) % 1000003)]
        # ONE collective: the maximum of (v, -v) over the ranks is (max v, -min v); all ranks agree iff the two halves cancel
        both = torch.tensor(digest + [-v for v in digest], dtype=torch.float64, device=device)
        dist.all_reduce(both, op=dist.ReduceOp.MAX, group=self.group)
        if not torch.equal(both[:3], -both[3:]):
            raise RuntimeError("ranks disagree on the cells to deepen (rank %d: %d cells): inputs or engines differ"
                               % (self.rank, len(todo)))

    def generate(self, cells, grid_w, grid_h, diff_type, repeat_range=0, repeat_addition=0, spec_k=None):
        """cells: ALL valid cells in raster order (every rank passes the same array).
        Returns (best_fit, best_score) as Engine.generate does, identical on all ranks.

        Same speculate-and-verify loop as mm_generate (engine.hip): short lists first, one
        all-gather, a verifying resolve on every rank (identical verdicts everywhere), then
        each rank deepens the unverified cells of its own block and the lists are gathered
        and resolved again, until no cell is left unverified."""
        import time

        import torch
        n = len(cells)
        # host wall time of this rank's phases, summed over the calls since reset_phases(): the engine calls block, and
        # the collective is followed by a synchronize -- so these are what the step is made of on THIS rank
        ph = self.phase_ms
        clock = time.perf_counter
        if spec_k is None:                      # the engine's own option, as mm_generate honours it
            spec_k = getattr(self.engine, "options", {}).get("spec_k", 4)
        k = self.engine.candidates_per_cell(repeat_range, repeat_addition)
        ks = self.engine.speculative_candidates(repeat_range, repeat_addition, spec_k)
        speculate = ks < k and n > 0
        per = padded_len(n, self.world)
        lo, hi = shard_range(n, self.world, self.rank)
        idx, score, packed = self._buffers(per, k)
        if hi - lo < per:                       # short last shard: keep the padding rows inert
            idx[hi - lo:] = -1
            score[hi - lo:] = float("inf")
            if idx.device.type == "cuda":
                torch.cuda.synchronize(idx.device)
        t0 = clock()
        if hi > lo:
            self.engine.score_cells(cells[lo:hi], diff_type, k, out_index=idx.data_ptr(),
                                    out_score=score.data_ptr(), on_device=True, k_fill=ks)
        ph["score"] += (clock() - t0) * 1e3
        deep = np.zeros(n, dtype=np.uint8) if speculate else None
        while True:
            t0 = clock()
            idx_all, score_all = allgather_candidates(idx[:per], score[:per], n, self.world, self.group,
                                                      packed=packed if per >= 1 else None)
            if idx.device.type == "cuda":
                torch.cuda.synchronize(idx.device)  # the engine runs on its own stream
            t1 = clock()
            ph["allgather"] += (t1 - t0) * 1e3      # includes the wait for the slowest rank's lists
            out = self.engine.resolve(cells, grid_w, grid_h, diff_type, repeat_range, repeat_addition, k,
                                      idx_all.data_ptr(), score_all.data_ptr(), on_device=True, deep=deep, k_fill=ks)
            ph["resolve"] += (clock() - t1) * 1e3
            ph["rounds"] += 1
            if not speculate:
                return out
            best, best_score, insufficient = out
            todo = np.nonzero((insufficient != 0) & (deep == 0))[0]
            self._same_on_all_ranks(todo.astype(np.int64), idx.device)
            if len(todo) == 0:
                return best, best_score
            deep[todo] = 1
            t0 = clock()
            mine = todo[(todo >= lo) & (todo < hi)]
            if len(mine):                           # deepen this rank's own unverified cells
                di = torch.empty((len(mine), k), dtype=torch.int32, device=idx.device)
                ds = torch.empty((len(mine), k), dtype=torch.float64, device=idx.device)
                self.engine.score_cells(cells[mine], diff_type, k, out_index=di.data_ptr(),
                                        out_score=ds.data_ptr(), on_device=True)
                rows = torch.as_tensor(mine - lo, device=idx.device)
                idx[rows] = di
                score[rows] = ds
            ph["deepen"] += (clock() - t0) * 1e3


def merge_on_host(shards, n_cells):
    """Host-side equivalent of the all-gather for already-copied numpy shards
    (list of (idx, score) in rank order); used by tests."""
    idx = np.concatenate([s[0] for s in shards])[:n_cells]
    score = np.concatenate([s[1] for s in shards])[:n_cells]
    return idx, score
