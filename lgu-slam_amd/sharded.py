"""Edge sharding of the factor graph across the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests).

The reference has no multi-GPU path; this is the new design of
SURVEY.md §8(e).  Factor-graph edges are independent units for every operator of the hot
path, so they are partitioned with NO data-path collective; the real exchanges of an
update step are, before dense bundle adjustment (reference factor_graph.py:286-300):
  * per EDGE   `target` / `weight` (E,ht,wd,2)           — factor_graph.py:290-291,295-296
  * per SOURCE FRAME `damping` (ht,wd)                    — factor_graph.py:292,294
  * per SOURCE FRAME `upmask` (8*8*9,ht,wd) when upsampling — factor_graph.py:286-287

Backend (low-memory) path: edges are sharded by the reference's own source-frame chunks
(factor_graph.py:272-276: for i in range(0, jj.max()+1, 8): edges with ii in [i, i+8)),
dealt round-robin to ranks, so that every chunk a rank processes is identical to a chunk
of the single-GPU run — which keeps lowMem_defSample's `offset[b*n]` quirk (it reads the
chunk's FIRST edge's offsets) bit-compatible — per-edge GRU state stays on its owner
across steps, and every source frame (its damping, its upmask) has exactly one owner.
"""
import os

import torch
import torch.distributed as dist


def balanced_shard(n_edges, rank, world):
    """Contiguous, balanced [lo, hi) range of edge indices for `rank` (frontend path)."""
    base, rem = divmod(n_edges, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chunk_shards(ii, world, chunk=8, jj=None):
    """Round-robin assignment of source-frame chunks to ranks.

    ii (and jj): 1-D integer tensors of edge source (target) frames.  Returns a list (one per rank) of
    lists of index tensors; each index tensor selects the edges of one chunk `ii in [i, i+chunk)`,
    in the reference's iteration order.  The loop bound is the reference's, `range(0, jj.max()+1,
    chunk)` (factor_graph.py:273): edges whose source frame lies beyond the last chunk that bound
    produces are processed by NO chunk there, and by none here (ShardedEdgeSet.unprocessed).  With
    jj=None the bound is ii.max()+1 (every edge is in some chunk).  Empty chunks are skipped.
    """
    shards = [[] for _ in range(world)]
    if ii.numel() == 0:
        return shards
    bound = int(jj.max()) + 1 if jj is not None else int(ii.max()) + 1
    k = 0
    for i in range(0, bound, chunk):
        idx = torch.nonzero((ii >= i) & (ii < i + chunk), as_tuple=False).flatten()
        if idx.numel() == 0:
            continue
        shards[k % world].append(idx)
        k += 1
    return shards


def _flat_all_gather(group):
    """Which all-gather form this process group gets, decided ONCE from the backend's name — never by trying one
    collective and falling back to another (if only some ranks raised, the ranks would issue mismatched collectives
    and hang).  RCCL: the flat form (one contiguous receive buffer).  Anything else (gloo in the CPU tests): the list
    form, which every backend implements."""
    return dist.get_backend(group) == "nccl"


# Rehearsal switch: with LGU_REHEARSE_COLLECTIVES=1 and an initialised process group, a world of ONE rank still issues every
# collective of the sharded step (the padded all-gathers, the system all-reduces of sharded_ba_split, the replica check) instead
# of short-cutting them — on a one-GPU box that runs the exact RCCL calls (argument shapes, dtypes, in-place forms) that N > 1 runs.
REHEARSE_COLLECTIVES = os.environ.get("LGU_REHEARSE_COLLECTIVES", "0") == "1"


def _single(world):
    """True when a world of this size needs no collective (one rank and no rehearsal)."""
    return world == 1 and not (REHEARSE_COLLECTIVES and dist.is_initialized())


class RowExchange:
    """All-gather of row-indexed tensors with uneven row counts per rank, returned in a caller-given global order.

    counts[r] = rows owned by rank r; ids[r] = their global row ids (both known to every rank: the partition is a pure
    function of the edge list).  `gather(x)` takes this rank's (counts[rank], ...) tensor, rows in ids[rank] order.
    One collective per call (shards padded to max(counts)); the received (world, cmax, ...) buffer is read ONCE, through
    a precomputed index, into its destination:
      gather(x)            -> (n_global, ...) tensor, row g = the row with global id g (ids must then cover 0..n-1)
      gather(x, out=buf)   -> buf[ids] = rows (rows no rank owns keep their previous value); returns buf
    """

    def __init__(self, ids, n_global, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if len(ids) != self.world:
            raise ValueError("need one id tensor per rank")
        self.ids = [i.long() for i in ids]
        self.counts = [int(i.numel()) for i in ids]
        self.n_global = int(n_global)
        self.cmax = max(self.counts) if self.counts else 0
        self.flat = _flat_all_gather(group) if dist.is_initialized() and not _single(self.world) else True
        dev = ids[0].device if ids else torch.device("cpu")
        self.all_ids = torch.cat(self.ids) if ids else torch.zeros(0, dtype=torch.long, device=dev)
        # position of every owned row in the padded receive buffer, in all_ids order
        self.src = torch.cat([r * self.cmax + torch.arange(c, device=dev) for r, c in enumerate(self.counts)]) \
            if ids else self.all_ids
        self.complete = self.all_ids.numel() == self.n_global and \
            bool((torch.sort(self.all_ids).values == torch.arange(self.n_global, device=dev)).all())
        self.src_by_id = None
        if self.complete:  # receive-buffer position of global row g
            self.src_by_id = torch.empty_like(self.src)
            self.src_by_id[self.all_ids] = self.src
        self._dev = {}

    def _index(self, dev):
        """The three index tensors on the data's device (uploaded once per device, not per call)."""
        if dev not in self._dev:
            self._dev[dev] = tuple(t.to(dev) if t is not None else None for t in (self.all_ids, self.src, self.src_by_id))
        return self._dev[dev]

    def gather(self, x, out=None):
        if x.shape[0] != self.counts[self.rank]:
            raise ValueError("rank %d owns %d rows, got %d" % (self.rank, self.counts[self.rank], x.shape[0]))
        if out is None and not self.complete:
            raise ValueError("rows without an owner: pass out= (they keep their previous value there)")
        tail = tuple(x.shape[1:])
        if _single(self.world):
            recv = x
        else:
            pad = x.new_empty((self.cmax,) + tail)
            pad[: x.shape[0]] = x
            if self.flat:
                recv = x.new_empty((self.world * self.cmax,) + tail)
                dist.all_gather_into_tensor(recv, pad, group=self.group)
            else:
                # list form; gloo has no all-gather of device tensors, so those are staged through the host (tests that
                # run several gloo ranks on one GPU — RCCL takes the flat form above)
                host = pad.cpu() if pad.is_cuda else pad
                parts = [host.new_empty((self.cmax,) + tail) for _ in range(self.world)]
                dist.all_gather(parts, host, group=self.group)
                recv = torch.cat(parts, 0).to(x.device)
        all_ids, src, src_by_id = self._index(recv.device)
        if out is None:
            return recv.index_select(0, src_by_id)
        out.index_copy_(0, all_ids, recv.index_select(0, src))
        return out


class EdgeExchange(RowExchange):
    """All-gather of per-edge tensors in RANK order (rank 0's edges, then rank 1's, ...): counts[r] = edges of rank r."""

    def __init__(self, counts, group=None):
        counts = [int(c) for c in counts]
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        super().__init__([torch.arange(offs[r], offs[r + 1]) for r in range(len(counts))], offs[-1], group)


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
    `owned[r]` = their concatenation, `frames[r]` = the source frames of those chunks (sorted: a
    chunk is a range of source frames, so every frame has one owner), `unprocessed` = edges no chunk
    of the reference's loop covers (they keep their previous target / weight, as in the reference).
      gather(x_local[, out])         per-edge tensor in this rank's `owned` order -> ORIGINAL edge order on every rank
      gather_frames(x_local, out)    per-source-frame tensor in this rank's `frames` order -> out[frame] on every rank
    — what dense BA consumes (reference factor_graph.py:290-300).
    """

    def __init__(self, ii, rank=None, world=None, chunk=8, group=None, jj=None):
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world = rank, world
        self.chunks = chunk_shards(ii, world, chunk, jj)
        dev = ii.device
        empty = torch.zeros(0, dtype=torch.long, device=dev)
        self.owned = [torch.cat(c) if len(c) else empty for c in self.chunks]
        self.counts = [int(o.numel()) for o in self.owned]
        self.frames = [torch.cat([torch.unique(ii[idx]) for idx in c]) if len(c) else empty for c in self.chunks]
        covered = torch.zeros(ii.numel(), dtype=torch.bool, device=dev)
        for o in self.owned:
            covered[o] = True
        self.unprocessed = torch.nonzero(~covered, as_tuple=False).flatten()
        live = dist.is_initialized() or world == 1
        self.exchange = RowExchange(self.owned, ii.numel(), group) if live else None
        nframes = int(ii.max()) + 1 if ii.numel() else 0
        self.frame_exchange = RowExchange(self.frames, nframes, group) if live else None

    @property
    def my_chunks(self):
        return self.chunks[self.rank]

    @property
    def my_edges(self):
        return self.owned[self.rank]

    @property
    def my_frames(self):
        return self.frames[self.rank]

    def gather(self, x_local, out=None):
        return self.exchange.gather(x_local, out)

    def gather_frames(self, x_local, out):
        return self.frame_exchange.gather(x_local, out)


class ShardedAltCorr:
    """The correlation step of update_lowmem (reference factor_graph.py:262-279) over this rank's
    chunks: one AltCorrBlock over the (replicated) feature maps, looked up chunk by chunk exactly as
    the reference iterates them, so every chunk equals a chunk of the single-GPU run."""

    def __init__(self, ofsMap, ofs_residual, GA, fmaps, ii, jj, rig=1, rank=None, world=None, chunk=8, group=None):
        from .corr import AltCorrBlock
        self.block = AltCorrBlock(ofsMap, ofs_residual, GA, fmaps)
        self.ii, self.jj, self.rig = ii, jj, rig
        self.edges = ShardedEdgeSet(ii, rank, world, chunk, group, jj=jj)

    def lookup(self, coords1):
        """coords1 (1,E,H,W,2) for ALL edges (replicated).  Yields (edge_index_tensor, corr (1,n,196,H,W))
        for each chunk this rank owns."""
        for idx in self.edges.my_chunks:
            iis, jjs = self.ii[idx], self.jj[idx]
            yield idx, self.block(coords1[:, idx], self.rig * iis, self.rig * jjs + (iis == jjs).long())

    def lookup_all(self, coords1):
        """All of this rank's chunks in ONE lookup launch (AltCorrBlock.call_many): returns (edge_index_tensor =
        edges.my_edges, corr (1,n,196,H,W), counts) with the chunks' edges back to back in chunk order; every chunk's
        slice is bit for bit what `lookup` yields for it.  The reference's loop issues one call per chunk only to bound
        memory (factor_graph.py:272-279); 250 edges of 60 x 80 lookups are 0.9 GB here."""
        idx = self.edges.my_edges
        counts = [int(c.numel()) for c in self.edges.my_chunks]
        if not counts:
            return idx, None, counts
        iis, jjs = self.ii[idx], self.jj[idx]
        return idx, self.block.call_many(coords1[:, idx], self.rig * iis, self.rig * jjs + (iis == jjs).long(), counts), counts


def run_chunks(edges, ii, chunk_fn, with_upmask=False, corr_all=None):
    """This rank's part of the chunk loop of update_lowmem (reference factor_graph.py:272-292): chunk_fn(idx, iis) ->
    (target (n,ht,wd,2), weight (n,ht,wd,2), damping (f,ht,wd)[, upmask (f,...)]) for the edges `idx` of one chunk
    (iis = their source frames, f = torch.unique(iis).numel(), frames ascending) — the caller's correlation lookup +
    update operator.  Returns the per-chunk results as lists (target, weight, damping, upmask) in chunk order.
    corr_all = the (1,n,C,H,W) result of ShardedAltCorr.lookup_all: the lookups were done in one launch and chunk_fn is
    called as chunk_fn(idx, iis, corr) with its chunk's slice (a view)."""
    t_loc, w_loc, d_loc, u_loc = [], [], [], []
    pos = 0
    for idx in edges.my_chunks:
        if corr_all is not None:
            r = chunk_fn(idx, ii[idx], corr_all[:, pos:pos + idx.numel()])
            pos += idx.numel()
        else:
            r = chunk_fn(idx, ii[idx])
        t_loc.append(r[0]); w_loc.append(r[1]); d_loc.append(r[2])
        if with_upmask:
            u_loc.append(r[3])
    return t_loc, w_loc, d_loc, u_loc


def exchange_step(edges, local, target, weight, damping, upmask=None):
    """The exchanges that make every rank's state whole again after run_chunks: one all-gather each for target, weight
    (by edge), damping and the optional upmask (by source frame) — issued by every rank, also by ranks that own no
    chunk.  target / weight (E,ht,wd,2), damping (num_frames,ht,wd), upmask (num_frames,...) are the replicated state
    tensors; on return they hold, on EVERY rank, what the single-GPU loop leaves there: rows of processed edges / owned
    frames from their owner, everything else untouched."""
    t_loc, w_loc, d_loc, u_loc = local

    def cat(parts, like):
        return torch.cat(parts, 0) if parts else like.new_zeros((0,) + tuple(like.shape[1:]))

    edges.gather(cat(t_loc, target), out=target)
    edges.gather(cat(w_loc, weight), out=weight)
    edges.gather_frames(cat(d_loc, damping), out=damping)
    if upmask is not None:
        edges.gather_frames(cat(u_loc, upmask), out=upmask)
    return target, weight, damping, upmask


def sharded_update_step(edges, ii, chunk_fn, target, weight, damping, upmask=None):
    """run_chunks + exchange_step: one sharded pass of update_lowmem's chunk loop."""
    return exchange_step(edges, run_chunks(edges, ii, chunk_fn, upmask is not None), target, weight, damping, upmask)


def replicas_agree(*tensors, group=None):
    """True when every rank holds bit-identical copies of the given tensors (compared through an fp64 checksum and
    the element count: one small all-reduce).  The replicated BA of a sharded step silently diverges otherwise."""
    if not dist.is_initialized() or _single(dist.get_world_size(group)):
        return True
    sig = torch.stack([t.detach().double().sum() + 1e-3 * t.detach().double().abs().sum() + t.numel() for t in tensors])
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))


def sharded_ba(edges, target_local, weight_local, poses, disps, intrinsics, disps_sens, eta, ii, jj, t0, t1,
               iterations, lm, ep, motion_only, check_replicas=False, target=None, weight=None):
    """The dense-BA step of a sharded update (reference factor_graph.py:290-300 after update_lowmem's chunk loop):
    every rank contributes the `target` / `weight` (n_local,2,ht,wd) of the edges it owns (in `edges.my_edges`
    order); ONE all-gather each restores the full (E,2,ht,wd) tensors in the original edge order on every rank, and the
    bundle adjustment (lgu_slam_amd.ba.ba) then runs replicated — identical inputs, deterministic kernels, so every
    rank holds the same updated `poses` / `disps` without a broadcast.  `eta` must be the replicated per-frame damping
    (after ShardedEdgeSet.gather_frames / sharded_update_step); check_replicas=True verifies that and the gathered
    tensors across ranks before the solve (one scalar all-reduce pair) and raises on a mismatch.
    target / weight (optional, (E,2,ht,wd), replicated): the persistent per-edge state the gathered rows are written INTO.
    Needed when the edge set leaves edges unprocessed (source frames beyond the reference's `range(0, jj.max()+1, 8)`
    loop bound, factor_graph.py:273): those edges enter the BA with their PREVIOUS target / weight, as in the reference,
    and that previous value lives in these tensors; without them such a graph is refused."""
    from . import ba as _ba
    if (target is None or weight is None) and edges.unprocessed.numel():
        raise ValueError("sharded_ba: %d edges belong to no chunk of the reference's loop and keep their previous target / weight: "
                         "pass the persistent target= / weight= tensors" % edges.unprocessed.numel())
    target = edges.gather(target_local, out=target).contiguous()
    weight = edges.gather(weight_local, out=weight).contiguous()
    if check_replicas and not replicas_agree(eta, target, weight, poses, disps, group=edges.exchange.group):
        raise RuntimeError("sharded_ba: replicated inputs differ across ranks (eta / target / weight / poses / disps)")
    return _ba.ba(poses, disps, intrinsics, disps_sens, target, weight, eta, ii, jj, t0, t1, iterations, lm, ep, motion_only)


def sharded_ba_split(edges, target_local, weight_local, poses, disps, intrinsics, disps_sens, eta_by_frame, ii, jj, t0, t1,
                     iterations, lm, ep, motion_only):
    """Dense BA with its per-edge work done by the edges' OWNERS (VERDICT r1 #6b), instead of replicated as in sharded_ba:
    every rank builds the Jacobian blocks of the edges it owns, forms the depth system and the Schur products of the
    depth frames it owns — a depth frame is the source frame of its edges, and source-frame chunks are what is sharded, so
    every edge and every E block meeting in a depth frame is on that frame's owner (reference schur_block,
    src/droid_kernels.cu:1260-1290) — and assembles ITS part of the reduced camera system.  Per iteration then
      all-reduce   of the (6P)^2 double system and its right-hand side (sum of the ranks' parts),
      solve        replicated (blocked Cholesky, csrc/ba_chol.hip; identical inputs on every rank -> identical poses),
      all-gather   of the depth-frame owners' updated disparity rows.
    `target_local` / `weight_local` (n_local,2,ht,wd): this rank's edges in `edges.my_edges` order — the per-edge
    all-gathers of exchange_step are not needed for the BA at all.  `eta_by_frame` (num_frames,ht,wd): damping indexed by
    frame id; only the rows of frames this rank owns, and of frames NO rank owns (no edge anywhere: the same edge-free
    update on every rank, so those rows must be replicated), are used — the rows of frames owned by other ranks may
    hold anything (stale, zero, NaN): they are replaced by 1 before the depth system, where this rank has no block of
    such a frame (E = 0, w = 0: its contribution 0 * Q stays 0 instead of 0 * inf).  The per-frame all-gather is
    therefore not needed.  Results equal the replicated BA up to summation order (the
    system is summed per rank first); every rank holds bit-identical poses / disps afterwards.  World 1 = ba.ba."""
    from . import ba as _ba
    if edges.unprocessed.numel():
        raise RuntimeError("sharded_ba_split: %d edges belong to no chunk (the reference's jj.max() loop bound) and would drop out "
                           "of the bundle adjustment; use sharded_ba" % edges.unprocessed.numel())
    own = edges.my_edges
    ii_o, jj_o = ii[own].contiguous(), jj[own].contiguous()
    live = dist.is_initialized() and not _single(edges.world)
    group = edges.exchange.group if edges.exchange is not None else None
    on_host = live and dist.get_backend(group) != "nccl"   # gloo (tests): collectives of device tensors staged through the host

    def reduce_system(Ad, b):
        if not live:
            return
        for t in (Ad, b):
            if on_host:
                h = t.cpu()
                dist.all_reduce(h, group=group)
                t.copy_(h)
            else:
                dist.all_reduce(t, group=group)

    def after_depth(d):
        if live:
            edges.gather_frames(d[edges.my_frames].contiguous(), out=d)

    eta_safe = eta_by_frame
    if live:
        foreign = torch.zeros(eta_by_frame.shape[0], dtype=torch.bool, device=eta_by_frame.device)
        for r, fr in enumerate(edges.frames):
            if r != edges.rank and fr.numel():
                foreign[fr.to(foreign.device)] = True
        eta_safe = torch.where(foreign.view(-1, *([1] * (eta_by_frame.dim() - 1))), torch.ones_like(eta_by_frame), eta_by_frame)
    return _ba.ba(poses, disps, intrinsics, disps_sens, target_local.contiguous(), weight_local.contiguous(),
                  lambda kx: eta_safe[kx], ii_o, jj_o, t0, t1, iterations, lm, ep, motion_only,
                  _hooks=(reduce_system, after_depth))
