"""Host-side correlation blocks: this build's counterpart of the reference glue
(droid_slam/modules/corr.py) over the gfx950 operators in `ops`.

Same public surface as the reference — `CorrSampler`, `DefCorrSampler`,
`per_Corr_Normalization`, `CorrBlock(ofsMap, ofs_residual, GA, fmap1, fmap2, num_levels,
radius)`, `AltCorrBlock(ofsMap, ofs_residual, GA, fmaps, num_levels, radius)` with
`__call__`, `cat`, `__getitem__` — so droid_slam/factor_graph.py:121-123,215,262-279 can
use it unchanged.  What differs is underneath: in inference `CorrBlock.__call__` issues ONE
fused launch for the whole pyramid (ops.defcorr_pyramid_forward) instead of four sampler
launches and a torch.cat, and never reads the offset tensors of levels that are zero by
construction.
"""
import os

import torch
import torch.nn.functional as F

from . import _lib, ops


class CorrSampler(torch.autograd.Function):
    """Plain bilinear window sampler (reference corr.py:10-24)."""

    @staticmethod
    def forward(ctx, volume, coords, radius):
        ctx.save_for_backward(volume, coords)
        ctx.radius = radius
        corr, = ops.corr_index_forward(volume, coords, radius)
        return corr

    @staticmethod
    def backward(ctx, grad_output):
        volume, coords = ctx.saved_tensors
        grad_volume, = ops.corr_index_backward(volume, coords, grad_output.contiguous(), ctx.radius)
        return grad_volume, None, None


class DefCorrSampler(torch.autograd.Function):
    """Deformable sampler (reference corr.py:26-42): grads for volume and offset only."""

    @staticmethod
    def forward(ctx, volume, coords, offset, radius):
        offset = offset.float()
        volume = volume.float()
        ctx.save_for_backward(volume, coords, offset)
        ctx.radius = radius
        corr, = ops.defCorr_index_forward(volume, coords, offset, radius)
        return corr

    @staticmethod
    def backward(ctx, grad_output):
        volume, coords, offset = ctx.saved_tensors
        grad_volume, grad_offset = ops.defCorr_index_backward(volume, coords, offset, grad_output.contiguous(),
                                                              ctx.radius)
        return grad_volume, None, grad_offset, None


def per_Corr_Normalization(x, normalIndex, eps=1e-5):
    """Per-sample standardisation over `normalIndex` with biased variance (reference corr.py:44-51)."""
    mean = x.mean(dim=normalIndex, keepdim=True)
    std = torch.sqrt(x.var(dim=normalIndex, unbiased=False, keepdim=True) + eps)
    return (x - mean) / std


# Per-frame partial convolutions of the offset heads, cached per block (ops.OffsetHeadCache).  OFF by default
# since the fast path convolves only the first edge of a call (LAZY_OFFSETS, §3.4d): that edge's two frames are not met again
# within the block's life, so the cache would hold 2 x 98 x H x W floats per frame of the WHOLE buffer the block was given
# (update_lowmem hands over video.fmaps: 2.4 GB at 512 frames of 48 x 64) for nothing.  A caller that reads every edge's
# offsets over a long-lived block (LAZY_OFFSETS = False) can turn it on; it is then used while it fits HEAD_CACHE_MAX_BYTES and
# the heads are convolved per edge (ops.offset_conv_frames, no per-frame state) beyond that.  Both forms sum in different
# orders (equal to 2e-5 in the offsets): one setting serves every call of a block, so lazy and eager calls stay bit-identical.
HEAD_CACHE = os.environ.get("LGU_HEAD_CACHE", "0") == "1"
HEAD_CACHE_MAX_BYTES = 1 << 30
FUSED_OFFSETS = True   # False: the reference-shaped torch composition below also in inference (A/B and tests)


class _Snapshot:
    """Identity of the tensors a cached derivative was made from: the tensor OBJECTS themselves (held, so their ids
    cannot be recycled), their version counters (in-place writes, load_state_dict) and their storage addresses
    (`.data = ...` swaps).  A fresh Parameter that happens to land in a freed allocator block with version 0 — the same
    (data_ptr, _version) pair as the one it replaces — is a different object and does not match."""

    __slots__ = ("tensors", "marks")

    def __init__(self, tensors):
        self.tensors = tuple(tensors)
        self.marks = tuple((t._version, t.data_ptr()) for t in self.tensors)

    def matches(self, tensors):
        tensors = tuple(tensors)
        return (len(tensors) == len(self.tensors) and all(a is b for a, b in zip(tensors, self.tensors))
                and self.marks == tuple((t._version, t.data_ptr()) for t in tensors))


def _zero_offsets(like, n):
    """A structurally-zero offset level (reference corr.py:132-135) for n edges without n copies of it: one zero row
    expanded over the edge dimension (stride 0).  The samplers never read or write these levels (they get NULL), and
    CorrBlock.cat / __getitem__ re-expand instead of copying; `.contiguous()` materialises the zeros if somebody asks."""
    key = (tuple(like.shape[1:]), like.dtype, like.device)
    row = _ZERO_ROWS.get(key)
    if row is None:   # one zero row per shape / dtype / device for the life of the process (a fill launch per call otherwise);
        if len(_ZERO_ROWS) >= 16:   # nothing writes through a stride-0 expansion, so sharing the row is safe
            _ZERO_ROWS.clear()
        row = _ZERO_ROWS[key] = torch.zeros((1,) + key[0], dtype=like.dtype, device=like.device)
    return row.expand((n,) + key[0])


_ZERO_ROWS = {}


def _is_zero_expansion(t):
    return t.dim() >= 1 and t.shape[0] > 0 and t.stride(0) == 0


def generate_offsets(ofsMap, ofs_residual, feats, num_levels):
    """Learned sampling offsets (reference corr.py:117-135 / :217-235).

    feats (E,256,h,w).  Level 0: 4*tanh(PCN(ofsMap(feats))).  Level 1: the residual head on
    the 2x-pooled features, nearest-upsampled, (4*tanh(PCN(.)) + level0) / 2.  Levels >= 2
    are zero by construction.  Returned channel-last: list of (E,h,w,2*rd*rd) tensors, and
    a list of flags marking the structurally-zero levels.
    """
    _, _, h, w = feats.shape
    o0 = ofsMap(feats)
    o1_low = ofs_residual(F.avg_pool2d(feats, kernel_size=2, stride=2))
    return finish_offsets(o0, o1_low, num_levels)


def finish_offsets(o0, o1_low, num_levels, probe=None):
    """Everything of generate_offsets after the two convolutions: o0 (E,C,h,w), o1_low (E,C,h/2,w/2).  probe (optional,
    (E,1,T,h,w) fp32 plain level-1 samples): the level-1 offsets come back scaled by the uncertainty mask
    sigmoid(var(probe)) of AltCorrBlock.corr_fn (corr.py:203-207)."""
    h, w = o0.shape[2:]
    if (FUSED_OFFSETS and num_levels >= 2 and o0.is_cuda and not (torch.is_grad_enabled() and o0.requires_grad)
            and o0.dtype in (torch.float32, torch.float16) and o1_low.dtype == o0.dtype):
        # inference: standardisation, tanh, residual mix, upsampling and the channel-last transposition in one pass
        try:
            off0, off1 = ops.offsets_finalize(o0.contiguous(), o1_low.contiguous(), probe=probe)
            offsets = [off0, off1] + [_zero_offsets(off0, off0.shape[0])] * (num_levels - 2)
            return offsets[:num_levels], ([False, False] + [True] * (num_levels - 2))[:num_levels]
        except _lib.UnsupportedShape:
            pass
    o1 = F.interpolate(o1_low, (h, w))
    o0 = torch.tanh(per_Corr_Normalization(o0, [1, 2, 3])) * 4
    o1 = (torch.tanh(per_Corr_Normalization(o1, [1, 2, 3])) * 4 + o0) / 2
    offsets = [o0.permute(0, 2, 3, 1), o1.permute(0, 2, 3, 1)]
    if probe is not None:
        E_, T_ = o0.shape[0], probe.numel() // (o0.shape[0] * h * w)
        pr = probe.reshape(E_, T_, h, w).permute(0, 2, 3, 1)
        offsets[1] = offsets[1] * torch.sigmoid(torch.var(pr, dim=3)).view(E_, h, w, 1)
    zero = [False, False]
    for _ in range(2, num_levels):
        offsets.append(torch.zeros_like(offsets[0]).detach())
        zero.append(True)
    return offsets[:num_levels], zero[:num_levels]


def _uncertainty_mask(probe):
    """sigmoid of the unbiased variance over the 3x3 probe taps (reference corr.py:95-97).
    probe (E,3,3,h,w) -> (E,h,w,1)."""
    E, _, _, h, w = probe.shape
    var = torch.var(probe.permute(0, 3, 4, 1, 2), dim=[3, 4])
    return torch.sigmoid(var).view(E, h, w, 1)


class CorrBlock:
    """All-pairs correlation volume pyramid + deformable lookup (reference corr.py:52-152).

    In inference the pyramid this block owns is stored in the library's TILED slice layout (4 x 8 element
    tiles = one 128-byte HBM line each, include/lgu_corr.h LGU_PYR_TILED): the fused builder writes it at no
    extra cost and every lookup then touches ~40 % fewer HBM lines, with bit-identical results.
    `CorrBlock.TILED_PYRAMID = False` keeps the reference's row-major slices (also used whenever gradients are
    needed, for radius != 3 and for shapes the fused builder does not serve).

    `CorrBlock.OUT_FORMAT` selects the memory format of the lookup result of tiled inference blocks: "planar"
    (default) is the reference's contiguous fp32 (1,E,196,H,W); "nhwc" / "nhwc_f16" return the same logical tensor
    stored channel-last in fp32 / half (= `.half()` of the fp32 result) — what the consumer, `corr_encoder` under
    autocast (droid_net.py:76-80, factor_graph.py update), wants; `lgu_slam_amd.encoder.CorrEncoder` then runs the
    1x1 convolution as one library GEMM with bias and ReLU in its epilogue."""

    TILED_PYRAMID = True
    FUSED_BUILD = True   # fp32 maps outside autocast: the volume is built on the matrix cores straight into the tiled pyramid
                         # (ops.volume_build_pyramid); False = library GEMM + fused post-processing (A/B and tests)
    FUSED_BUILD_HALF = os.environ.get("LGU_FUSED_BUILD_HALF", "0") == "1"
                         # half maps (autocast): the same with each product sum rounded to half in the kernel, as the
                         # reference's half GEMM rounds it.  OPT-IN: a half GEMM fixes no summation order, so a raw product
                         # next to a rounding boundary can land one half ulp away from this device's library GEMM (0.03 %
                         # of the entries at 16 x 32), and the default keeps torch.matmul + the fused post-processing, which
                         # tests hold BIT-equal to the torch composition under autocast
    OUT_FORMAT = "planar"
    ENCODER = None   # a lgu_slam_amd.encoder.CorrEncoder: its first layer (1x1 convolution + ReLU) then runs INSIDE the
                     # lookup launch and __call__ returns the (1,E,128,H,W) half result of that layer; the same
                     # CorrEncoder, called on it, applies only the remaining 3x3 convolution

    def __init__(self, ofsMap, ofs_residual, GA, fmap1, fmap2, num_levels=4, radius=3):
        self.num_levels = num_levels
        self.radius = radius
        self.GA = GA
        self.ofsMap = ofsMap
        self.ofs_residual = ofs_residual

        b, n, ch, h, w = fmap1.shape
        raw = None       # the all-pairs product (half when the feature maps are, depth_video's): formed only where a path needs it
        volume = None    # raw.float() (corr.py:64), made only where needed
        feats = torch.cat((fmap1.reshape(b * n, ch, h, w), fmap2.reshape(b * n, ch, h, w)), dim=1)
        self.t = feats
        self.offset, self._zero_level = generate_offsets(ofsMap, ofs_residual, feats, num_levels)

        self.t = feats.permute(0, 2, 3, 1).contiguous()
        # the offsets are part of the autograd graph when only the offset heads train (frozen fnet / GA)
        needs_grad = torch.is_grad_enabled() and (feats.requires_grad or any(q.requires_grad for q in GA.parameters())
                                                  or any(o.requires_grad for o in self.offset))
        self._store = None       # slot-indirected level buffers (inference), see _adopt_store
        self._pyr = None         # plain list of level tensors in edge order (training / fallback)
        self._tiled = False
        self._level_hw = [(h >> i, w >> i) for i in range(num_levels)]
        if not needs_grad and hasattr(GA, "gaussian_parameters"):
            # inference: Gaussian re-weighting, "/denominator + corr" and the 3 poolings in ONE
            # pass over the volume, level 0 in place (ops.volume_pyramid)
            mean_n, cov, det = GA.gaussian_parameters(self.t)
            tiled = bool(CorrBlock.TILED_PYRAMID) and radius == 3
            if (tiled and CorrBlock.FUSED_BUILD and num_levels == 4 and fmap1.dtype == torch.float32 and fmap2.dtype == torch.float32
                    and fmap1.is_cuda and not torch.is_autocast_enabled()):
                # fp32 maps outside autocast (where the reference's matmul IS an fp32 GEMM): the product is formed on the
                # fp32 matrix cores and goes from the accumulators into the tiled pyramid — no raw volume in HBM at all
                # (ops.volume_build_pyramid).  Half maps / autocast keep the library GEMM, whose half rounding of the raw
                # volume is part of what the reference computes there.
                try:
                    dd = det.contiguous() if det.dtype in (torch.float32, torch.float16) else det.float().contiguous()
                    self._adopt_store(ops.volume_build_pyramid(fmap1.reshape(b * n, ch, h, w).contiguous(),
                                                               fmap2.reshape(b * n, ch, h, w).contiguous(),
                                                               mean_n.float().contiguous(), cov.float().contiguous(), dd,
                                                               num_levels, GA.RADIUS))
                    self._tiled = True
                    raw = None
                except _lib.UnsupportedShape:
                    pass
            elif (tiled and CorrBlock.FUSED_BUILD_HALF and num_levels == 4 and self.t.dtype == torch.float16 and self.t.is_cuda
                    and fmap1.dtype == torch.float16 and fmap2.dtype == torch.float16):
                # half maps: the reference's matmul is a half GEMM; the kernel rounds each product sum to half as it does
                try:
                    dd = det.contiguous() if det.dtype in (torch.float32, torch.float16) else det.float().contiguous()
                    self._adopt_store(ops.volume_build_pyramid(self.t, None, mean_n.float().contiguous(), cov.float().contiguous(),
                                                               dd, num_levels, GA.RADIUS))
                    self._tiled = True
                    raw = None
                except _lib.UnsupportedShape:
                    pass
        if self._store is None and not needs_grad and hasattr(GA, "gaussian_parameters"):
            if raw is None:
                raw = CorrBlock.corr(fmap1, fmap2).view(b * n, h, w, h, w)
            try:
                # a half raw volume is converted by the builder's own load, not by a .float() pass
                src = raw.contiguous() if raw.dtype == torch.float16 else raw.float().contiguous()
                # det as the Gaussian head produced it: half inside autocast, where the reference's denominator
                # 6.28 * sqrt(det) is rounded to half twice (gaussianMask_cuda.py:79-86 under factor_graph.py:90)
                dd = det.contiguous() if det.dtype in (torch.float32, torch.float16) else det.float().contiguous()
                self._adopt_store(ops.volume_pyramid(mean_n.float().contiguous(), cov.float().contiguous(), src, num_levels,
                                                     GA.RADIUS, inplace=True, tiled=tiled, det=dd))
                self._tiled = tiled
            except _lib.UnsupportedShape:
                self.corr_pyramid = None
        if self._store is None and self._pyr is None:
            if raw is None:
                raw = CorrBlock.corr(fmap1, fmap2).view(b * n, h, w, h, w)
            volume, mean_n, det = GA(self.t, raw.float())
            # pyramid over the TARGET dims: level i is (E,h,w,h/2^i,w/2^i) (reference corr.py:79-86)
            self.corr_pyramid = []
            lvl = volume.reshape(b * n * h * w, 1, h, w)
            for i in range(num_levels):
                self.corr_pyramid.append(lvl.view(b * n, h, w, h // 2 ** i, w // 2 ** i))
                lvl = F.avg_pool2d(lvl, 2, stride=2)
        self.mean_n = mean_n.view(b, n, h, w, 2)
        self.theta = 2 * det.view(b, n, h, w)

    # ---- pyramid storage ----
    # Inference blocks keep their pyramid in SLOT-INDIRECTED level buffers: `_store[l]` holds `capacity` edge slots,
    # `_slot_list[e]` is the slot of logical edge e, the lookup kernel follows that indirection
    # (lgu_defcorr_pyramid_slots_fwd_f32).  `cat` then copies only the NEW edges into free slots and `__getitem__`
    # only edits the index list, where the reference (corr.py:111-121) re-materialises the whole multi-GB pyramid
    # with torch.cat / boolean indexing on every keyframe.  `corr_pyramid` remains readable in edge order (a gathered
    # copy) and assignable (which leaves the slot form).
    def _adopt_store(self, levels):
        self._store = list(levels)
        self._pyr = None
        n = levels[0].shape[0]
        self._slot_list = list(range(n))
        self._free = []
        self._slots_dev = None
        self._plan_key = None

    def _slots(self):
        if self._slots_dev is None or self._slots_dev.shape[0] != len(self._slot_list):
            self._slots_dev = torch.tensor(self._slot_list, dtype=torch.int32, device=self._store[0].device)
        return self._slots_dev

    def _edge_order_levels(self):
        if self._store is None:
            return self._pyr
        if self._slot_list == list(range(self._store[0].shape[0])):
            return self._store
        idx = self._slots().long()
        return [st[idx] for st in self._store]

    @property
    def corr_pyramid(self):
        return self._edge_order_levels()

    @corr_pyramid.setter
    def corr_pyramid(self, levels):
        self._pyr = levels
        self._store = None
        self._plan_key = None

    def _reserve(self, extra):
        """Make room for `extra` more edges: grow every level buffer (amortised doubling)."""
        cap = self._store[0].shape[0]
        if len(self._free) >= extra:
            return
        newcap = max(2 * cap, cap + extra - len(self._free))
        for l, st in enumerate(self._store):
            grown = torch.empty((newcap,) + tuple(st.shape[1:]), dtype=st.dtype, device=st.device)
            grown[:cap] = st
            self._store[l] = grown
        self._free += list(range(cap, newcap))
        self._plan_key = None

    # ---- lookup ----
    def __call__(self, coords):
        batch, num, ht, wd, _ = coords.shape
        E = batch * num
        rd = 2 * self.radius + 1
        coords_xy = coords.reshape(E, ht, wd, 2)   # as handed over: x, y interleaved (the fused kernel reads this form)
        coords = None                               # (E,2,ht,wd) planes, made only where an operator needs them

        needs_grad = torch.is_grad_enabled() and (any(o.requires_grad for o in self.offset) or (
            self._store is None and any(v.requires_grad for v in self._pyr)))
        if needs_grad:
            if self._store is not None:   # an inference store with trainable offsets: leave it, the fused launch
                self._to_reference_layout()   # returns no grad_fn and writes the offsets through raw pointers
            coords = coords_xy.permute(0, 3, 1, 2).contiguous()
            # training: reference-shaped composition through the autograd Functions.
            # Uncertainty probe on level 1; the mask is folded into offset[1] and PERSISTS
            # across calls, exactly like the reference (corr.py:94-99)
            probe = CorrSampler.apply(self.corr_pyramid[1], coords / 2, 1)
            self.offset[1] = self.offset[1] * _uncertainty_mask(probe)
            out = [DefCorrSampler.apply(self.corr_pyramid[i], coords / 2 ** i,
                                        self.offset[i].contiguous().view(E, ht, wd, rd, rd, 2), self.radius)
                   .view(batch, num, -1, ht, wd) for i in range(self.num_levels)]
            return torch.cat(out, dim=2), self.mean_n, self.theta

        offs = []
        for i in range(self.num_levels):
            if self._zero_level[i]:
                offs.append(None)
                continue
            o = self.offset[i]
            if o.dtype != torch.float32 or not o.is_contiguous():
                o = o.float().contiguous()
                self.offset[i] = o  # keep the buffer the kernel zeroes the centre of
            offs.append(o.view(E, ht, wd, rd, rd, 2))
        if self._store is not None:
            pyr, slots = self._store, self._slots()
        else:
            pyr, slots = [v if v.is_contiguous() else v.contiguous() for v in self._pyr], None
        # inference: probe + mask + all levels + concatenation in ONE launch; offset[1] is
        # scaled in place by the kernel (the same persistent state as above).  The prepared
        # launch is rebuilt only when the pyramid / offset buffers change (cat, __getitem__).
        fmt = CorrBlock.OUT_FORMAT if self._tiled else "planar"
        enc = CorrBlock.ENCODER.fused_operands(self.num_levels * rd * rd) if (CorrBlock.ENCODER is not None and self._tiled) else None
        key = tuple(t.data_ptr() for t in pyr) + tuple(o.data_ptr() if o is not None else 0 for o in offs) + \
            ((slots.data_ptr(), slots.shape[0]) if slots is not None else ()) + (fmt, enc[0].data_ptr() if enc else 0)
        try:
            if getattr(self, "_plan_key", None) != key:
                self._plan = ops.DefcorrPyramidPlan(pyr, offs, self.radius, probe=True, tiled=self._tiled,
                                                    level_hw=self._level_hw, coords_last=True, slots=slots,
                                                    out_format=fmt, encoder=enc)
                self._plan_key = key
            out = self._plan(coords_xy if coords_xy.is_contiguous() else coords_xy.contiguous())
        except _lib.UnsupportedShape:
            # shapes the fused probe does not serve (e.g. W2 % 4 != 0): separate probe ops
            self._plan_key = None
            self._to_reference_layout()
            coords = coords_xy.permute(0, 3, 1, 2).contiguous()
            pyr = [v if v.is_contiguous() else v.contiguous() for v in self._pyr]
            probe, = ops.corr_index_forward(pyr[1], (coords / 2).contiguous(), 1)
            self.offset[1] = (self.offset[1] * _uncertainty_mask(probe)).contiguous()
            offs[1] = self.offset[1].view(E, ht, wd, rd, rd, 2)
            out = ops.defcorr_pyramid_forward(pyr, coords, offs, self.radius)
        return out.view(batch, num, -1, ht, wd), self.mean_n, self.theta

    def _to_reference_layout(self):
        """Plain row-major level tensors in edge order (only needed to mix with a block built without the fused
        builder, or for shapes the fused lookup does not serve)."""
        levels = self._edge_order_levels()
        if self._tiled:
            levels = [ops.volume_retile(v.contiguous(), to_tiled=False, hw=self._level_hw[i]) for i, v in enumerate(levels)]
            self._tiled = False
        self.corr_pyramid = levels

    def cat(self, other):
        if self._store is not None and other._store is not None and self._tiled == other._tiled:
            # append: only the new edges move (into free slots of this block's buffers)
            src = other._edge_order_levels()
            n_new = src[0].shape[0]
            self._reserve(n_new)
            take, self._free = self._free[:n_new], self._free[n_new:]
            if n_new:
                idx = torch.tensor(take, dtype=torch.long, device=src[0].device)
                for l in range(self.num_levels):
                    self._store[l][idx] = src[l]
            self._slot_list = self._slot_list + take
            self._slots_dev = None
        else:
            if self._tiled != other._tiled or self._store is not None or other._store is not None:
                self._to_reference_layout()
                other._to_reference_layout()
            self.corr_pyramid = [torch.cat([a, b], 0) for a, b in zip(self._pyr, other._pyr)]
        for i in range(self.num_levels):
            a, b = self.offset[i], other.offset[i]
            if self._zero_level[i] and other._zero_level[i] and _is_zero_expansion(a) and _is_zero_expansion(b):
                self.offset[i] = _zero_offsets(a, a.shape[0] + b.shape[0])   # still zero: nothing to copy
            else:
                self.offset[i] = torch.cat([a, b], 0)
            self._zero_level[i] = self._zero_level[i] and other._zero_level[i]
        return self

    def __getitem__(self, index):
        if self._store is not None:
            # drop / reorder edges: edit the slot list, return the dropped slots to the free list; no volume moves
            n = len(self._slot_list)
            keep = torch.arange(n, device=self._store[0].device)[index].reshape(-1).tolist()
            new_list = [self._slot_list[i] for i in keep]
            kept = set(new_list)
            self._free = self._free + [sl for sl in self._slot_list if sl not in kept]
            self._slot_list = new_list
            self._slots_dev = None
            # the kept positions are known on the host now: index the offsets with them (a boolean mask would cost one
            # device -> host round trip per level)
            index = torch.tensor(keep, dtype=torch.long, device=self._store[0].device)
        else:
            self.corr_pyramid = [v[index] for v in self._pyr]
        for i in range(self.num_levels):
            o = self.offset[i]
            if self._zero_level[i] and _is_zero_expansion(o) and index.dtype == torch.long and index.dim() == 1:
                self.offset[i] = _zero_offsets(o, index.shape[0])
            else:
                self.offset[i] = o[index]
        return self

    @staticmethod
    def corr(fmap1, fmap2):
        """All-pairs correlation (reference corr.py:144-152): (b,n,1,h*w,h,w), both maps / 4."""
        batch, num, dim, ht, wd = fmap1.shape
        f1 = fmap1.reshape(batch * num, dim, ht * wd) / 4.0
        f2 = fmap2.reshape(batch * num, dim, ht * wd) / 4.0
        return torch.matmul(f1.transpose(1, 2), f2).view(batch, num, 1, ht * wd, ht, wd)


class AltCorrBlock:
    """Low-memory lookup: correlations recomputed from feature maps (reference corr.py:155-249).

    `offset` (the per-level learned offsets of the last call, reference attribute) is materialised on access: with one
    sample per pixel the reference's sampler reads `offset[b * n]` with n = 0 for every edge b (lowMem_defSample.cu:80-83
    — all edges of a call sample with the FIRST edge's offsets), so the fast path computes the offset heads, their
    post-processing and the level-1 probe's mask for that edge only (LAZY_OFFSETS) and forms the other edges' rows — which
    nothing on the path reads — only if somebody asks for the attribute."""

    LAZY_OFFSETS = True

    @property
    def offset(self):
        if getattr(self, "_lazy", None) is not None:
            self._offset = self._materialise_offsets()
            self._lazy = None
        return self._offset

    @offset.setter
    def offset(self, value):
        self._offset = value
        self._lazy = None

    def __init__(self, ofsMap, ofs_residual, GA, fmaps, num_levels=4, radius=3):
        self.num_levels = num_levels
        self.radius = radius
        self.GA = GA  # stored, never applied on this path (reference corr.py:156-158,174-215)
        self.ofsMap = ofsMap
        self.ofs_residual = ofs_residual
        self.offset = []

        B, N, C, H, W = fmaps.shape
        lvl = fmaps.view(B * N, C, H, W) / 4.0
        self.pyramid = []
        for i in range(num_levels):
            self.pyramid.append(lvl.permute(0, 2, 3, 1).contiguous().view(B, N, H // 2 ** i, W // 2 ** i, C))
            lvl = F.avg_pool2d(lvl, 2, stride=2)

    def _frame_operands(self):
        """The stored pyramid as the fused launches read it: per level (N,Hl,Wl,C) views (float copies of a float
        pyramid's levels) and, in self._chunked, the chunk-planar form the matrix-core sweep reads the target maps in
        (ops.lowmem_chunked: 16 x-adjacent positions of a 16-byte channel chunk are contiguous) — made once per block, like
        the pyramid itself."""
        frames = [p_[0].contiguous() for p_ in self.pyramid]
        if self.pyramid[0].dtype != torch.float16:
            frames = [f.float() for f in frames]
        if (getattr(self, "_chunked", None) is None or self._chunked[0].dtype != frames[0].dtype
                or not self._chunked_snap.matches(self.pyramid)):    # rewritten / replaced levels: re-derive
            self._chunked = [ops.lowmem_chunked(f) for f in frames]
            self._chunked_snap = _Snapshot(self.pyramid)
        return frames

    def call_many(self, coords, ii, jj, counts):
        """`torch.cat([self(coords[:, s:e], ii[s:e], jj[s:e]) for the consecutive edge ranges of lengths counts], dim=1)`
        — the chunk loop of update_lowmem (reference factor_graph.py:272-279), which issues one corr_fn call per chunk of
        source frames only to bound memory — in ONE lookup launch (ops.LowmemPyramidPlan(off_row=...)) behind ONE batched
        pass of the probe, the offset heads and their post-processing over the calls' first edges: a call's edges all
        sample with its first edge's offsets (class docstring), so each call contributes one offset row.  Every edge's
        result is bit for bit that of its own call.  `offset` afterwards is the LAST call's, as after the loop.  Inputs the
        fused launch does not serve (gradients, S > 1, float maps, ...) are issued call by call."""
        counts = [int(c) for c in counts]
        E = ii.shape[0]
        if sum(counts) != E or coords.shape[1] != E or any(c <= 0 for c in counts):
            raise RuntimeError("call_many: counts must be positive and add up to the number of edges")
        squeeze = coords.dim() == 5
        starts = [0]
        for c in counts[:-1]:
            starts.append(starts[-1] + c)

        def one_by_one():
            return torch.cat([self(coords[:, s:s + c], ii[s:s + c], jj[s:s + c]) for s, c in zip(starts, counts)], dim=1)

        B, H, W = coords.shape[0], coords.shape[2], coords.shape[3]
        S = 1 if squeeze else coords.shape[4]
        rd = 2 * self.radius + 1
        if not (len(counts) > 1 and self.LAZY_OFFSETS and not torch.is_grad_enabled() and S == 1 and B == 1
                and self.num_levels >= 2 and self.pyramid[0].dtype == torch.float16 and ii.dtype == torch.int64
                and jj.dtype == torch.int64):
            return one_by_one()
        try:
            frames = self._frame_operands()
            K = len(counts)
            ckey = (tuple(counts), ii.device)
            if getattr(self, "_calls_key", None) != ckey:   # edge -> call tables, uploaded once per partition
                self._calls_first = torch.tensor(starts, device=ii.device)
                self._calls_row = torch.repeat_interleave(torch.arange(K, dtype=torch.int32, device=ii.device),
                                                          torch.tensor(counts, device=ii.device)).contiguous()
                self._calls_key = ckey
            first, off_row = self._calls_first, self._calls_row
            iic, jjc = ii.contiguous(), jj.contiguous()
            c0 = coords.reshape(E, 1, H, W, 2).contiguous()
            i0, j0 = iic[first], jjc[first]

            def probe_of(cs, i_, j_):   # the plain r = 1 samples of level 1 (corr.py:201-202)
                return ops.lowmem_pyramid_forward_mixed(frames[0], [self._chunked[1]], cs, [None], 1, ii=i_, jj=j_, lbase=1,
                                                        chunked=True)

            firsts = self._offsets_from_frames(1, i0, j0, probe=probe_of(c0[first], i0, j0), store=False)
            if firsts is False:
                return one_by_one()
            rows, zero_level = firsts
            offs = [None if zero_level[i] else rows[i].contiguous().view(K, H, W, rd, rd, 2).float()
                    for i in range(self.num_levels)]
            fused = ops.lowmem_pyramid_forward_mixed(frames[0], self._chunked, c0, offs, self.radius, ii=iic, jj=jjc,
                                                     chunked=True, off_row=off_row)
        except _lib.UnsupportedShape:
            return one_by_one()
        self.one_launch_calls = getattr(self, "one_launch_calls", 0) + 1   # (tests: the one-launch path ran, not the loop)
        sl, nl = starts[-1], counts[-1]
        il, jl, cl = iic[sl:], jjc[sl:], c0[sl:]
        self._offset, self._zero_level = None, zero_level
        self._lazy = (il, jl, lambda n: probe_of(cl[:n], il[:n], jl[:n]), nl, [None if o is None else o[K - 1:] for o in offs])
        out = fused.view(1, E, -1, H, W)
        return out if squeeze else out.unsqueeze(-1)

    def _offsets_from_frames(self, B, ii, jj, probe=None, store=True):
        """Inference fast path of the offset heads for a half pyramid (update_lowmem's case, autocast off): both heads
        run on the matrix cores straight from stored frames (ops.offset_conv_frames: no gather / x 4 / cat / cast of a
        (E,256,H,W) tensor, fp32-accurate split-half weights).  The residual head's input, the 2 x 2 average of the
        frames, is pooled ONCE per block instead of per call (the same fp32 averages of the same numbers: pooling
        commutes with the per-edge gather) and split into two half parts.  The rest is finish_offsets.  Sets
        self.offset; returns False when the general composition has to run."""
        conv = self.ofsMap
        C = self.pyramid[0].shape[-1]
        if not (FUSED_OFFSETS and B == 1 and self.num_levels >= 2 and self.pyramid[0].dtype == torch.float16
                and self.pyramid[0].is_cuda and not torch.is_autocast_enabled() and ii.dtype == torch.int64
                and jj.dtype == torch.int64 and isinstance(conv, torch.nn.Conv2d) and conv.bias is not None
                and conv.in_channels == 2 * C and C % 32 == 0 and conv.out_channels <= 112
                and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.stride == (1, 1)
                and conv.dilation == (1, 1) and conv.groups == 1 and conv.weight.dtype == torch.float32
                and not (torch.is_grad_enabled() and (conv.weight.requires_grad or any(q.requires_grad for q in self.ofs_residual.parameters())))):
            return False
        res = self.ofs_residual
        res_fast = (isinstance(res, torch.nn.Conv2d) and res.bias is not None and res.in_channels == 2 * C
                    and res.out_channels <= 112 and res.kernel_size == (3, 3) and res.padding == (1, 1)
                    and res.stride == (1, 1) and res.dilation == (1, 1) and res.groups == 1
                    and res.weight.dtype == torch.float32)
        heads = [conv.weight, conv.bias] + list(res.parameters())
        if getattr(self, "_ofs_snap", None) is None or not self._ofs_snap.matches(heads):
            self._ofs_packed = ops.pack_offset_conv(conv.weight, conv.bias)
            self._res_packed = ops.pack_offset_conv(res.weight, res.bias) if res_fast else None
            self._ofs_snap = _Snapshot(heads)
            self._head_snap = None       # the per-frame partial convolutions were made with the old weights
        level0 = self.pyramid[0]
        frames0 = level0[0]
        if (getattr(self, "_pooled", None) is None or not self._pooled_snap.matches([level0])
                or self._pooled_fast != res_fast):
            self._pooled_snap, self._pooled_fast = _Snapshot([level0]), res_fast
            self._head_snap = None       # ... or from the old frames
            # 2 x 2 averages of the frames in fp32, as avg_pool2d of the reference's fp32 input gives them (x 4 is a power
            # of two and moves to the weights exactly), channel-last, split into two half parts: hi + lo == the average
            # to 2^-22.  The general path keeps the fp32 averages (x 4) in NCHW.
            pooled = F.avg_pool2d(frames0.permute(0, 3, 1, 2).float(), kernel_size=2, stride=2)
            if res_fast:
                pl_ = pooled.permute(0, 2, 3, 1).contiguous()
                hi = pl_.half()
                self._pooled = (hi, (pl_ - hi.float()).half())
            else:
                self._pooled = (pooled * 4.0,)
        iic, jjc = ii.contiguous(), jj.contiguous()
        try:
            nf, hh, ww = frames0.shape[0], frames0.shape[1], frames0.shape[2]
            cache_bytes = 2 * 4 * nf * conv.out_channels * (hh * ww + (hh // 2) * (ww // 2))
            if HEAD_CACHE and res_fast and C % 64 == 0 and cache_bytes <= HEAD_CACHE_MAX_BYTES:
                # per-frame partial convolutions, kept for the life of this block (ops.OffsetHeadCache): the heads are
                # linear in cat(frame ii, frame jj), a frame is source / target of ~10 edges each and the block serves
                # every chunk of an update_lowmem pass, so each frame is convolved once and an edge costs a sum
                if getattr(self, "_head_snap", None) is None:   # new weights or a rewritten pyramid: the partials are stale
                    self._head0 = ops.OffsetHeadCache(frames0, ops.pack_offset_conv_parts(conv.weight, conv.bias))
                    self._head1 = ops.OffsetHeadCache(self._pooled[0], ops.pack_offset_conv_parts(res.weight, res.bias),
                                                      frames_lo=self._pooled[1])
                    self._head_snap = True
                E = iic.shape[0]
                work = self._head0.mark(iic, jjc)   # one claim pass for both heads: they need the same frames
                self._head0.convolve(work, E)
                self._head1.convolve(work, E)
                o0 = self._head0.combine(iic, jjc)
                o1_low = self._head1.combine(iic, jjc, reset=work[1])
            else:
                o0 = ops.offset_conv_frames(frames0, iic, jjc, self._ofs_packed)
                if len(self._pooled) == 2:
                    o1_low = ops.offset_conv_frames(self._pooled[0], iic, jjc, self._res_packed, frames_lo=self._pooled[1])
                else:
                    o1_low = res(torch.cat((self._pooled[0][ii], self._pooled[0][jj]), dim=1))
        except _lib.UnsupportedShape:
            return False
        res_ = finish_offsets(o0, o1_low, self.num_levels, probe=probe)   # probe: level 1 comes back masked
        if not store:
            return res_
        self.offset, self._zero_level = res_
        return True

    def _materialise_offsets(self):
        """Every edge's offsets of the last (lazy) call, as the reference's attribute holds them: the full heads and
        post-processing (same kernels per sample), with the first edge's rows taken from the tensors the lookup actually
        used (their centre taps zeroed by the sampler, lowMem_defSample.cu:80-81)."""
        iic, jjc, run_probe, E, offs = self._lazy
        full = self._offsets_from_frames(1, iic, jjc, probe=run_probe(E), store=False)
        if full is False:   # cannot happen: the same path just served the first edge
            raise RuntimeError("AltCorrBlock: offsets of the last call are no longer available")
        rows, _ = full
        rows = list(rows)
        for i, o in enumerate(offs):
            if o is not None:
                rows[i] = rows[i].contiguous()
                rows[i][:1] = o.view(rows[i][:1].shape)
        return rows

    def corr_fn(self, coords, ii, jj):
        B, N, H, W, S, _ = coords.shape
        rd = 2 * self.radius + 1
        coords = coords.permute(0, 1, 4, 2, 3, 5)

        def source_maps():   # the per-edge gather of the source frames: only the per-level operators below need it
            f = self.pyramid[0][:, ii]
            return f.reshape((B * N,) + f.shape[2:])

        def make_feats():
            # offsets come from the un-scaled level-0 maps (reference corr.py:177-189)
            # (standard NCHW strides: the cat of permuted views would come out channel-last, which sends the fp32
            # convolutions below to MIOpen's NHWC implicit-GEMM kernels — 0.89 ms against 0.48 ms for the NCHW ones)
            f1_0 = source_maps()
            f2_0 = self.pyramid[0][:, jj]
            f2_0 = f2_0.reshape((B * N,) + f2_0.shape[2:])
            return torch.cat(((f1_0 * 4.0).permute(0, 3, 1, 2), (f2_0 * 4.0).permute(0, 3, 1, 2)), dim=1).float() \
                .contiguous(memory_format=torch.contiguous_format)

        # Features stored in half precision (as the SLAM system keeps them) stay half: the mixed
        # operators take exact half products, accumulate in fp32 and equal the reference's `.float()`
        # call sites up to fp32 summation order (<= 1e-5).
        mixed = self.pyramid[0].dtype == torch.float16
        # One sample per pixel (the SLAM system's case): the level-1 probe and then ALL levels in one launch each, written
        # straight into the concatenated tensor (ops.LowmemPyramidPlan).  Both read the frame buffers in place at ii / jj
        # (no per-edge gathers of the pyramid), half buffers stay half, and levels whose offsets are zero by construction
        # read no offset tensor at all.  The probe does not depend on the offsets, so it runs FIRST and the offset
        # post-processing applies its uncertainty mask while it writes level 1 (no separate pass over that tensor).
        fused_ok = S == 1 and B == 1 and self.num_levels >= 2 and ii.dtype == torch.int64 and jj.dtype == torch.int64
        probe = frames = iic = jjc = c0 = None
        if fused_ok:
            try:
                frames = self._frame_operands()
                iic, jjc = ii.contiguous(), jj.contiguous()
                c0 = coords.reshape(B * N, S, H, W, 2).contiguous()
            except _lib.UnsupportedShape:
                fused_ok = False

        def run_probe(n_edges):   # the plain r = 1 samples of level 1 (corr.py:201-202) for the first n_edges edges
            return ops.lowmem_pyramid_forward_mixed(frames[0], [self._chunked[1]], c0[:n_edges], [None], 1, ii=iic[:n_edges],
                                                    jj=jjc[:n_edges], lbase=1, chunked=True)

        if fused_ok and self.LAZY_OFFSETS and not torch.is_grad_enabled():
            # Only the first edge's offsets are ever read (class docstring): probe, heads, post-processing for that edge.
            try:
                i0, j0 = iic[:1], jjc[:1]
                first = self._offsets_from_frames(B, i0, j0, probe=run_probe(1), store=False)
                if first is not False:
                    rows, zero_level = first
                    offs = [None if zero_level[i] else rows[i].contiguous().view(1, H, W, rd, rd, 2).float()
                            for i in range(self.num_levels)]
                    fused = ops.lowmem_pyramid_forward_mixed(frames[0], self._chunked, c0, offs, self.radius, ii=iic, jj=jjc, chunked=True)
                    self._offset, self._zero_level = None, zero_level
                    self._lazy = (iic, jjc, run_probe, B * N, offs)   # what `offset` needs to form every edge's rows on demand
                    return fused.view(B, N, -1, H, W).unsqueeze(-1)   # (1,E,L*rd*rd,H,W,S=1)
            except _lib.UnsupportedShape:
                pass
        if fused_ok:
            try:
                probe = run_probe(B * N)
            except _lib.UnsupportedShape:
                fused_ok, probe = False, None

        feats = None
        masked = False
        if self._offsets_from_frames(B, ii, jj, probe=probe):
            zero_level = self._zero_level
            masked = probe is not None
        else:
            feats = make_feats()
            self.offset, zero_level = generate_offsets(self.ofsMap, self.ofs_residual, feats, self.num_levels)

        if fused_ok:
            try:
                if not masked:
                    o1 = self.offset[1]
                    if o1.dtype == torch.float32 and o1.is_contiguous() and not o1.requires_grad:
                        ops.probe_mask_scale_(probe, o1)   # variance, sigmoid and the scaling in one pass, in place
                    else:
                        pr = probe.permute(0, 1, 3, 4, 2).contiguous().view(N, H, W, 3, 3)
                        mask = torch.sigmoid(torch.var(pr, dim=[3, 4])).view(B * N, H, W, 1)
                        self.offset[1] = self.offset[1] * mask
                offs = [None if zero_level[i] else self.offset[i].contiguous().view(B * N, H, W, rd, rd, 2).float()
                        for i in range(self.num_levels)]
                fused = ops.lowmem_pyramid_forward_mixed(frames[0], self._chunked, c0, offs, self.radius, ii=iic, jj=jjc, chunked=True)
                return fused.view(B, N, -1, H, W).unsqueeze(-1)   # (1,E,L*rd*rd,H,W,S=1)
            except _lib.UnsupportedShape:
                # channel counts / radii the matrix-core kernel does not serve: per-level operators below, which apply the
                # mask themselves to freshly generated offsets
                self.offset, _ = generate_offsets(self.ofsMap, self.ofs_residual, feats if feats is not None else make_feats(),
                                                  self.num_levels)
        out = []
        f1 = source_maps()
        f1 = f1.contiguous() if mixed else f1.float().contiguous()
        for i in range(self.num_levels):
            f2 = self.pyramid[i][:, jj]
            f2 = f2.reshape((B * N,) + f2.shape[2:])
            f2 = f2.contiguous() if mixed else f2.float().contiguous()
            coords_i = (coords / 2 ** i).reshape(B * N, S, H, W, 2).contiguous()
            use_mixed = mixed
            if i == 1:
                if use_mixed:
                    try:
                        probe, = ops.altcorr_forward_mixed(f1, f2, coords_i, 1)
                    except _lib.UnsupportedShape:
                        use_mixed = False
                if not use_mixed:
                    probe, = ops.altcorr_forward(f1.float(), f2.float(), coords_i, 1)
                probe = probe.permute(0, 1, 3, 4, 2).contiguous().view(N, H, W, 3, 3)  # needs B = S = 1, as in the reference
                mask = torch.sigmoid(torch.var(probe, dim=[3, 4])).view(B * N, H, W, 1)
                self.offset[1] = self.offset[1] * mask
            off = self.offset[i].contiguous().view(B * N, H, W, rd, rd, 2).float()
            corr = None
            if use_mixed:
                try:
                    corr, = ops.lowMem_defSample_mixed(f1, f2, coords_i, off, self.radius)
                except _lib.UnsupportedShape:
                    corr = None
            if corr is None:
                corr, = ops.lowMem_defSample(f1.float(), f2.float(), coords_i, off, self.radius)
            out.append(corr.view(B, N, S, -1, H, W).permute(0, 1, 3, 4, 5, 2))
        return torch.cat(out, dim=2)

    def __call__(self, coords, ii, jj):
        squeeze = coords.dim() == 5
        if squeeze:
            coords = coords.unsqueeze(dim=-2)
        corr = self.corr_fn(coords, ii, jj)
        if squeeze:
            corr = corr.squeeze(dim=-1)
        return corr.contiguous()
