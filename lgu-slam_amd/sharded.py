"""Edge sharding of the factor graph across the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests).

The reference has no multi-GPU path; this is the new design of
SURVEY.md §8(e).  Factor-graph edges are independent units for every operator of the hot
path, so they are partitioned with NO data-path collective; the one real exchange of an
update step is the all-gather of the per-edge `target` / `weight` (E,ht,wd,2) that dense
bundle adjustment consumes (reference factor_graph.py:290-300).

Backend (low-memory) path: edges are sharded by the reference's own source-frame chunks
(factor_graph.py:272-276: all edges with ii in [i, i+8)), dealt round-robin to ranks, so
that every chunk a rank processes is identical to a chunk of the single-GPU run — which
keeps lowMem_defSample's `offset[b*n]` quirk (it reads the chunk's FIRST edge's offsets)
bit-compatible — and per-edge GRU state stays on its owner across steps.
"""
import torch
import torch.distributed as dist


def balanced_shard(n_edges, rank, world):
    """Contiguous, balanced [lo, hi) range of edge indices for `rank` (frontend path)."""
    base, rem = divmod(n_edges, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chunk_shards(ii, world, chunk=8):
    """Round-robin assignment of source-frame chunks to ranks.

    ii: 1-D integer tensor of edge source frames.  Returns a list (one per rank) of lists of
    index tensors; each index tensor selects the edges of one chunk `ii in [i, i+chunk)`, in
    the reference's iteration order (factor_graph.py:272-276).  Empty chunks are skipped.
    """
    shards = [[] for _ in range(world)]
    if ii.numel() == 0:
        return shards
    k = 0
    for i in range(0, int(ii.max()) + 1, chunk):
        idx = torch.nonzero((ii >= i) & (ii < i + chunk), as_tuple=False).flatten()
        if idx.numel() == 0:
            continue
        shards[k % world].append(idx)
        k += 1
    return shards


class EdgeExchange:
    """All-gather of per-edge tensors with uneven edge counts per rank.

    counts[r] = number of edges owned by rank r (known to every rank: the partition is a
    pure function of the edge list).  `gather(x)` takes this rank's (counts[rank], ...) tensor
    and returns the (sum(counts), ...) tensor in rank order on every rank.  One collective
    per call: shards are padded to max(counts) so that all_gather_into_tensor applies.
    """

    def __init__(self, counts, group=None):
        self.counts = [int(c) for c in counts]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if len(self.counts) != self.world:
            raise ValueError("need one edge count per rank")
        self.cmax = max(self.counts) if self.counts else 0

    def gather(self, x):
        if x.shape[0] != self.counts[self.rank]:
            raise ValueError("rank %d owns %d edges, got %d" % (self.rank, self.counts[self.rank], x.shape[0]))
        if self.world == 1:
            return x
        tail = tuple(x.shape[1:])
        pad = x.new_zeros((self.cmax,) + tail)
        pad[: x.shape[0]] = x
        out = x.new_empty((self.world * self.cmax,) + tail)
        try:
            dist.all_gather_into_tensor(out, pad.contiguous(), group=self.group)
        except (RuntimeError, NotImplementedError):  # backend without the flat variant
            parts = [x.new_empty((self.cmax,) + tail) for _ in range(self.world)]
            dist.all_gather(parts, pad.contiguous(), group=self.group)
            out = torch.cat(parts, 0)
        out = out.view((self.world, self.cmax) + tail)
        return torch.cat([out[r, : self.counts[r]] for r in range(self.world)], 0)


def sharded_pyramid_sample(block_call, coords, rank, world, exchange=None):
    """Frontend helper: sample this rank's balanced shard of edges and (optionally) gather.

    block_call(coords_shard, lo, hi) -> (hi-lo, C, ht, wd) tensor for edges [lo, hi).
    coords: (E, ...) per-edge lookup coordinates, replicated on every rank.
    """
    lo, hi = balanced_shard(coords.shape[0], rank, world)
    mine = block_call(coords[lo:hi], lo, hi)
    if exchange is None:
        return mine
    return exchange.gather(mine)


class ShardedEdgeSet:
    """Ownership bookkeeping for one sharded update step of the backend (low-memory) path.

    The partition is a pure function of the edge list, so every rank builds the same object:
    `chunks[r]` = the source-frame chunks (index tensors into the edge list) rank r processes,
    `owned[r]` = their concatenation.  `gather(x_local)` all-gathers a per-edge tensor given in
    this rank's `owned` order and returns it in the ORIGINAL edge order on every rank — what
    dense BA consumes (reference factor_graph.py:290-300).
    """

    def __init__(self, ii, rank=None, world=None, chunk=8, group=None):
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world = rank, world
        self.chunks = chunk_shards(ii, world, chunk)
        dev = ii.device
        empty = torch.zeros(0, dtype=torch.long, device=dev)
        self.owned = [torch.cat(c) if len(c) else empty for c in self.chunks]
        self.counts = [int(o.numel()) for o in self.owned]
        order = torch.cat(self.owned) if self.owned else empty   # edge ids in gathered (rank-major) order
        self.inverse = torch.empty_like(order)
        self.inverse[order] = torch.arange(order.numel(), device=dev)
        self.exchange = EdgeExchange(self.counts, group) if (dist.is_initialized() or world == 1) else None

    @property
    def my_chunks(self):
        return self.chunks[self.rank]

    @property
    def my_edges(self):
        return self.owned[self.rank]

    def gather(self, x_local):
        allx = self.exchange.gather(x_local) if self.world > 1 else x_local
        return allx[self.inverse.to(allx.device)]


class ShardedAltCorr:
    """The correlation step of update_lowmem (reference factor_graph.py:262-279) over this rank's
    chunks: one AltCorrBlock over the (replicated) feature maps, looked up chunk by chunk exactly as
    the reference iterates them, so every chunk equals a chunk of the single-GPU run."""

    def __init__(self, ofsMap, ofs_residual, GA, fmaps, ii, jj, rig=1, rank=None, world=None, chunk=8, group=None):
        from .corr import AltCorrBlock
        self.block = AltCorrBlock(ofsMap, ofs_residual, GA, fmaps)
        self.ii, self.jj, self.rig = ii, jj, rig
        self.edges = ShardedEdgeSet(ii, rank, world, chunk, group)

    def lookup(self, coords1):
        """coords1 (1,E,H,W,2) for ALL edges (replicated).  Yields (edge_index_tensor, corr (1,n,196,H,W))
        for each chunk this rank owns."""
        for idx in self.edges.my_chunks:
            iis, jjs = self.ii[idx], self.jj[idx]
            yield idx, self.block(coords1[:, idx], self.rig * iis, self.rig * jjs + (iis == jjs).long())


def sharded_ba(edges, target_local, weight_local, poses, disps, intrinsics, disps_sens, eta, ii, jj, t0, t1,
               iterations, lm, ep, motion_only):
    """The dense-BA step of a sharded update (reference factor_graph.py:290-300 after update_lowmem's chunk loop):
    every rank contributes the `target` / `weight` (n_local,2,ht,wd) of the edges it owns (in `edges.my_edges`
    order); ONE all-gather each restores the full (E,2,ht,wd) tensors in the original edge order on every rank, and the
    bundle adjustment (lgu_slam_amd.ba.ba) then runs replicated — identical inputs, deterministic kernels, so every
    rank holds the same updated `poses` / `disps` without a broadcast."""
    from . import ba as _ba
    target = edges.gather(target_local).contiguous()
    weight = edges.gather(weight_local).contiguous()
    return _ba.ba(poses, disps, intrinsics, disps_sens, target, weight, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only)
