"""Operator layer: the reference's Python-visible operator signatures on top of the C ABI.

Each function mirrors one pybind11 entry of the reference (names, positional arguments,
list-of-tensors return, in-place side effects, error text):
  defCorrSample.*            offersample_LGS/droid.cpp:53-147
  droid_backends.altcorr_*   src/droid.cpp:193-217,246-247
Tensors must live on a HIP device ("cuda" in PyTorch-ROCm) — there is no CPU path — and
be float32 (what every reference call site passes: corr.py:30-31,64,202,209;
gaussianMask_cuda.py:11-12).  Kernels are enqueued on torch's current stream.
"""
import ctypes

import torch

from . import _lib

_vp = ctypes.c_void_p


def _check_dtype(t, name, dtype):
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must be a HIP device tensor: lgu_slam_amd has no CPU fallback" % name)
    if t.dtype != dtype:
        raise RuntimeError("expected scalar type %s but found %s (%s)" % (str(dtype).replace("torch.", ""), str(t.dtype).replace("torch.", ""), name))


def _check(*named):
    """_check(volume, "volume", coords, "coords", ...): the reference's CHECK_INPUT on every
    argument first (TORCH_CHECK(x.is_contiguous(), #x " must be contiguous"), droid.cpp:48-49),
    then what this library additionally requires (HIP device, float32)."""
    pairs = list(zip(named[0::2], named[1::2]))
    for t, name in pairs:
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous" % name)
    for t, name in pairs:
        if not t.is_cuda:
            raise RuntimeError("%s must be a HIP device tensor: lgu_slam_amd has no CPU fallback" % name)
        if t.dtype != torch.float32:
            raise RuntimeError("expected scalar type Float but found %s (%s)" % (str(t.dtype).replace("torch.", ""), name))


_TORCH_NAME = {torch.float32: "Float", torch.float16: "Half", torch.float64: "Double", torch.bfloat16: "BFloat16",
               torch.int64: "Long", torch.int32: "Int"}


def _via_float(fn, scalar_named, coords, tail, inout=()):
    """Half / double operands of the volume-path operators.  The reference dispatches those kernels with
    AT_DISPATCH_FLOATING_TYPES_AND_HALF on the first tensor's type (defCorrSample_kernel.cu:185,219;
    corrSample_kernel.cu:158,189; gaussianAttn.cu:152,187): every scalar_t accessor must then have that dtype
    ("expected scalar type Half but found Float") while `coords` stays a float accessor.  The HIP kernels are fp32, so
    these dtypes are served through fp32 copies: computed in fp32 (the arithmetic of the float dispatch), results cast to
    the operands' dtype, in-place side effects (centre zeroing) copied back.  Every reference call site passes float
    (corr.py:30-31,64; gaussianMask_cuda.py:11-12) — this is interface completeness, not a fast path.
    scalar_named = [t0, "name0", t1, "name1", ...] (t0 decides the dtype); inout = indices into the scalar tensors that
    the operator modifies."""
    ts, names = list(scalar_named[0::2]), list(scalar_named[1::2])
    dt = ts[0].dtype
    for t, n in zip(ts, names):
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous" % n)
    if coords is not None:
        if not coords.is_contiguous():
            raise RuntimeError("coords must be contiguous")
        if coords.dtype != torch.float32:
            raise RuntimeError("expected scalar type Float but found %s" % _TORCH_NAME.get(coords.dtype, str(coords.dtype)))
    for t, n in zip(ts, names):
        if t.dtype != dt:
            raise RuntimeError("expected scalar type %s but found %s" % (_TORCH_NAME.get(dt, str(dt)), _TORCH_NAME.get(t.dtype, str(t.dtype))))
    f32 = [t.float() for t in ts]
    outs = fn(*f32, *([coords] if coords is not None else []), *tail)
    for i in inout:
        ts[i].copy_(f32[i])
    return [o.to(dt) for o in outs]


_OTHER_FLOATS = (torch.float16, torch.float64)


def _stream(t):
    return _vp(torch.cuda.current_stream(t.device).cuda_stream)


class _CurrentDevice:
    """Launch guard of the prepared plans: the library launches on the CURRENT device, so a plan whose buffers live on
    another device switches for the call — and costs one integer comparison when it already is current (the usual
    one-process-per-GPU case)."""

    def __init__(self, device):
        self.idx = device.index if device.index is not None else torch.cuda.current_device()
        self.prev = -1

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
            self.prev = -1
        return False


def _ptr(t):
    return _vp(t.data_ptr())


def defCorr_index_forward(volume, coords, offset, radius):
    if volume.dtype in _OTHER_FLOATS:
        return _via_float(lambda v, o, c, r: defCorr_index_forward(v, c, o, r), [volume, "volume", offset, "offset"], coords,
                          (radius,), inout=(1,))
    _check(volume, "volume", coords, "coords", offset, "offset")
    E, H1, W1, H2, W2 = volume.shape
    rd = 2 * radius + 1
    if tuple(coords.shape) != (E, 2, H1, W1) or offset.numel() != E * H1 * W1 * rd * rd * 2:
        raise RuntimeError("defCorr_index_forward: shape mismatch between volume, coords and offset")
    corr = torch.empty((E, rd, rd, H1, W1), dtype=volume.dtype, device=volume.device)
    if E == 0:
        return [corr]  # no edges: nothing to launch (empty tensors have null data pointers)
    with torch.cuda.device(volume.device):
        rc = _lib.load().lgu_defcorr_fwd_f32(_ptr(volume), _ptr(coords), _ptr(offset), _ptr(corr),
                                             E, H1, W1, H2, W2, radius, _stream(volume))
    _lib.check(rc, "defCorr_index_forward")
    return [corr]


def defCorr_index_backward(volume, coords, offset, corr_grad, radius):
    if volume.dtype in _OTHER_FLOATS:
        return _via_float(lambda v, o, g, c, r: defCorr_index_backward(v, c, o, g, r),
                          [volume, "volume", offset, "offset", corr_grad, "corr_grad"], coords, (radius,), inout=(1,))
    _check(volume, "volume", coords, "coords", offset, "offset", corr_grad, "corr_grad")
    E, H1, W1, H2, W2 = volume.shape
    volume_grad = torch.zeros_like(volume)
    offset_grad = torch.empty_like(offset)
    if E == 0:
        return [volume_grad, offset_grad]
    with torch.cuda.device(volume.device):
        rc = _lib.load().lgu_defcorr_bwd_f32(_ptr(volume), _ptr(coords), _ptr(offset), _ptr(corr_grad),
                                             _ptr(volume_grad), _ptr(offset_grad), E, H1, W1, H2, W2, radius,
                                             _stream(volume))
    _lib.check(rc, "defCorr_index_backward")
    return [volume_grad, offset_grad]


def corr_index_forward(volume, coords, radius):
    if volume.dtype in _OTHER_FLOATS:
        return _via_float(lambda v, c, r: corr_index_forward(v, c, r), [volume, "volume"], coords, (radius,))
    _check(volume, "volume", coords, "coords")
    E, H1, W1, H2, W2 = volume.shape
    rd = 2 * radius + 1
    if tuple(coords.shape) != (E, 2, H1, W1):
        raise RuntimeError("corr_index_forward: coords must be (E,2,H1,W1)")
    corr = torch.empty((E, rd, rd, H1, W1), dtype=volume.dtype, device=volume.device)
    if E == 0:
        return [corr]
    with torch.cuda.device(volume.device):
        rc = _lib.load().lgu_corridx_fwd_f32(_ptr(volume), _ptr(coords), _ptr(corr), E, H1, W1, H2, W2, radius,
                                             _stream(volume))
    _lib.check(rc, "corr_index_forward")
    return [corr]


def corr_index_backward(volume, coords, corr_grad, radius):
    if volume.dtype in _OTHER_FLOATS:
        return _via_float(lambda v, g, c, r: corr_index_backward(v, c, g, r), [volume, "volume", corr_grad, "corr_grad"], coords,
                          (radius,))
    _check(volume, "volume", coords, "coords", corr_grad, "corr_grad")
    E, H1, W1, H2, W2 = volume.shape
    volume_grad = torch.zeros_like(volume)
    if E == 0:
        return [volume_grad]
    with torch.cuda.device(volume.device):
        rc = _lib.load().lgu_corridx_bwd_f32(_ptr(volume), _ptr(coords), _ptr(corr_grad), _ptr(volume_grad),
                                             E, H1, W1, H2, W2, radius, _stream(volume))
    _lib.check(rc, "corr_index_backward")
    return [volume_grad]


def gaussianMask(means, covs, volume, radius):
    if volume.dtype in _OTHER_FLOATS:   # gaussianAttn.cu:152 dispatches on volume; means / covs are scalar_t accessors too
        return _via_float(lambda v, m, c, r: gaussianMask(m, c, v, r), [volume, "volume", means, "means", covs, "covs"], None, (radius,))
    _check(volume, "volume", means, "means", covs, "covs")
    E, H1, W1, H2, W2 = volume.shape
    volume1 = torch.empty_like(volume)
    if E == 0:
        return [volume1]
    with torch.cuda.device(volume.device):
        rc = _lib.load().lgu_gaussmask_fwd_f32(_ptr(means), _ptr(covs), _ptr(volume), _ptr(volume1),
                                               E, H1, W1, H2, W2, radius, _stream(volume))
    _lib.check(rc, "gaussianMask")
    return [volume1]


def gaussianMask_backward(means, covs, volume, volume_grad, radius):
    if volume.dtype in _OTHER_FLOATS:
        return _via_float(lambda v, m, c, g, r: gaussianMask_backward(m, c, v, g, r),
                          [volume, "volume", means, "means", covs, "covs", volume_grad, "volume_grad"], None, (radius,))
    _check(volume, "volume", means, "means", covs, "covs", volume_grad, "volume_grad")
    E, H1, W1, H2, W2 = volume.shape
    means_grad = torch.empty_like(means)
    covs_grad = torch.empty_like(covs)
    if E == 0:
        return [means_grad, covs_grad]
    with torch.cuda.device(volume.device):
        rc = _lib.load().lgu_gaussmask_bwd_f32(_ptr(means), _ptr(covs), _ptr(volume), _ptr(volume_grad),
                                               _ptr(means_grad), _ptr(covs_grad), E, H1, W1, H2, W2, radius,
                                               _stream(volume))
    _lib.check(rc, "gaussianMask_backward")
    return [means_grad, covs_grad]


def lowMem_defSample(fmap1, fmap2, coords, offset, radius):
    _check(fmap1, "fmap1", fmap2, "fmap2", coords, "coords", offset, "offset")
    B, S, H1, W1, _ = coords.shape
    _, H2, W2, C = fmap2.shape
    rd = 2 * radius + 1
    corr = torch.empty((B, S, rd, rd, H1, W1), dtype=fmap1.dtype, device=fmap1.device)
    if B == 0:
        return [corr]
    with torch.cuda.device(fmap1.device):
        rc = _lib.load().lgu_lowmem_defsample_fwd_f32(_ptr(fmap1), _ptr(fmap2), _ptr(coords), _ptr(offset), _ptr(corr),
                                                      B, S, H1, W1, H2, W2, C, offset.shape[0], radius,
                                                      _stream(fmap1))
    _lib.check(rc, "lowMem_defSample")
    return [corr]


def altcorr_forward(fmap1, fmap2, coords, radius):
    _check(fmap1, "fmap1", fmap2, "fmap2", coords, "coords")
    B, S, H1, W1, _ = coords.shape
    _, H2, W2, C = fmap2.shape
    rd = 2 * radius + 1
    corr = torch.empty((B, S, rd * rd, H1, W1), dtype=fmap1.dtype, device=fmap1.device)
    if B == 0:
        return [corr]
    with torch.cuda.device(fmap1.device):
        rc = _lib.load().lgu_altcorr_fwd_f32(_ptr(fmap1), _ptr(fmap2), _ptr(coords), _ptr(corr),
                                             B, S, H1, W1, H2, W2, C, radius, _stream(fmap1))
    _lib.check(rc, "altcorr_forward")
    return [corr]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    _check(fmap1, "fmap1", fmap2, "fmap2", coords, "coords", corr_grad, "corr_grad")
    B, S, H1, W1, _ = coords.shape
    _, H2, W2, C = fmap2.shape
    fmap1_grad = torch.empty_like(fmap1)
    fmap2_grad = torch.zeros_like(fmap2)
    coords_grad = torch.zeros_like(coords)  # allocated, never written by the reference (altcorr_kernel.cu:336)
    if B == 0:
        return [fmap1_grad, fmap2_grad, coords_grad]
    with torch.cuda.device(fmap1.device):
        rc = _lib.load().lgu_altcorr_bwd_f32(_ptr(fmap1), _ptr(fmap2), _ptr(coords), _ptr(corr_grad),
                                             _ptr(fmap1_grad), _ptr(fmap2_grad), B, S, H1, W1, H2, W2, C, radius,
                                             _stream(fmap1))
    _lib.check(rc, "altcorr_backward")
    return [fmap1_grad, fmap2_grad, coords_grad]


def lowMem_defSample_mixed(fmap1, fmap2, coords, offset, radius):
    """lowMem_defSample on HALF-precision feature maps with fp32 accumulation and output: equal to
    `lowMem_defSample(fmap1.float(), fmap2.float(), coords, offset, radius)` — what the reference call site
    does (corr.py:209) — up to fp32 summation order (the contraction runs on the matrix cores with exact half
    products), without materialising the float copies."""
    _check_dtype(fmap1, "fmap1", torch.float16); _check_dtype(fmap2, "fmap2", torch.float16)
    _check(coords, "coords", offset, "offset")
    B, S, H1, W1, _ = coords.shape
    _, H2, W2, C = fmap2.shape
    rd = 2 * radius + 1
    corr = torch.empty((B, S, rd, rd, H1, W1), dtype=torch.float32, device=fmap1.device)
    if B == 0:
        return [corr]
    with torch.cuda.device(fmap1.device):
        rc = _lib.load().lgu_lowmem_defsample_fwd_h16(_ptr(fmap1), _ptr(fmap2), _ptr(coords), _ptr(offset), _ptr(corr),
                                                      B, S, H1, W1, H2, W2, C, offset.shape[0], radius, _stream(fmap1))
    _lib.check(rc, "lowMem_defSample_mixed")
    return [corr]


def altcorr_forward_mixed(fmap1, fmap2, coords, radius):
    """altcorr_forward on HALF-precision feature maps, fp32 accumulation/output (= the reference call
    site corr.py:202 on `.float()` copies, up to fp32 summation order)."""
    _check_dtype(fmap1, "fmap1", torch.float16); _check_dtype(fmap2, "fmap2", torch.float16)
    _check(coords, "coords")
    B, S, H1, W1, _ = coords.shape
    _, H2, W2, C = fmap2.shape
    rd = 2 * radius + 1
    corr = torch.empty((B, S, rd * rd, H1, W1), dtype=torch.float32, device=fmap1.device)
    if B == 0:
        return [corr]
    with torch.cuda.device(fmap1.device):
        rc = _lib.load().lgu_altcorr_fwd_h16(_ptr(fmap1), _ptr(fmap2), _ptr(coords), _ptr(corr), B, S, H1, W1, H2, W2, C,
                                             radius, _stream(fmap1))
    _lib.check(rc, "altcorr_forward_mixed")
    return [corr]


class LowmemPyramidPlan:
    """The per-level loop of AltCorrBlock.corr_fn (reference corr.py:192-213) as ONE launch over half or float
    feature maps (lgu_lowmem_pyramid_fwd_h16 / _f32): level l samples fmap2s[l] at coords / 2^(lbase+l) with
    offsets[l] (None = zero offsets) and writes channels l*rd*rd.. of the concatenated output
    (B, S, L*rd*rd, H1, W1).

    Two forms.  Per-edge maps: fmap1 (B,H1,W1,C), fmap2s[l] (B,H2l,W2l,C).  Frame buffers + indices (ii, jj int64
    device tensors of length B): fmap1 (F,H1,W1,C), fmap2s[l] (F,H2l,W2l,C) are read in place at frames ii[b] /
    jj[b] — what `self.pyramid[i][:, jj]` gathers in the reference, without the per-edge copies.
    The pointer tables are built once; a call costs one ctypes invocation.  Raises UnsupportedShape (at the first
    call) for channel counts / radii the matrix-core kernel does not serve."""

    def __init__(self, fmap1, fmap2s, offsets, radius, ii=None, jj=None, lbase=0, chunked=False, off_row=None):
        """chunked=True: every fmap2s[l] is in the chunk-planar form of lowmem_chunked() — (F, C/k, H2l, W2l, k) — instead
        of channel-last (F, H2l, W2l, C); same results bit for bit, the sweep's loads become line-friendly.
        off_row (int32 device tensor of length B, values < rows of the offset tensors): SEVERAL reference calls in one
        launch — edge b samples with offset row off_row[b], the first edge of its call (lgu_lowmem_pyramid_calls_fwd_h16:
        half maps, one sample per pixel); every edge's result is that of its own call."""
        L = len(fmap2s)
        if len(offsets) != L or not 1 <= L <= 4:
            raise RuntimeError("LowmemPyramidPlan: need 1..4 levels and one offset entry (tensor or None) per level")
        dt = torch.float16 if fmap1.dtype == torch.float16 else torch.float32
        _check_dtype(fmap1, "fmap1", dt)
        for l, f in enumerate(fmap2s):
            _check_dtype(f, "fmap2[%d]" % l, dt)
            if offsets[l] is not None:
                _check(offsets[l], "offset[%d]" % l)
        if (ii is None) != (jj is None):
            raise RuntimeError("LowmemPyramidPlan: pass both ii and jj or neither")
        if ii is not None:
            _check_dtype(ii, "ii", torch.int64); _check_dtype(jj, "jj", torch.int64)
            if ii.dim() != 1 or ii.shape != jj.shape:
                raise RuntimeError("LowmemPyramidPlan: ii and jj must be 1-D and of equal length")
        self._keep = (fmap1, list(fmap2s), list(offsets), ii, jj)
        self.L, self.radius, self.lbase = L, radius, lbase
        _, self.H1, self.W1, self.C = fmap1.shape
        self.B = fmap1.shape[0] if ii is None else ii.shape[0]
        self.device = fmap1.device
        self._guard = _CurrentDevice(self.device)
        self.NO = max([o.shape[0] for o in offsets if o is not None] or [max(self.B, 1)])
        self._f2 = (_vp * L)(*[f.data_ptr() for f in fmap2s])
        self._op = (_vp * L)(*[(o.data_ptr() if o is not None else None) for o in offsets])
        if chunked:
            k = 8 if dt == torch.float16 else 4
            for f in fmap2s:
                if f.dim() != 5 or f.shape[4] != k or f.shape[1] * k != self.C:
                    raise RuntimeError("chunked fmap2 levels must be (F, C/%d, H2, W2, %d)" % (k, k))
        self._h2 = (ctypes.c_int * L)(*[f.shape[2 if chunked else 1] for f in fmap2s])
        self._w2 = (ctypes.c_int * L)(*[f.shape[3 if chunked else 2] for f in fmap2s])
        self._ii = ii.data_ptr() if ii is not None and self.B else None
        self._jj = jj.data_ptr() if jj is not None and self.B else None
        lib = _lib.load()
        self._rows = None
        if off_row is not None:
            _check_dtype(off_row, "off_row", torch.int32)
            if dt != torch.float16 or off_row.dim() != 1 or off_row.shape[0] != self.B or not off_row.is_contiguous():
                raise RuntimeError("LowmemPyramidPlan: off_row needs half feature maps and one int32 entry per edge")
            rows = {o.shape[0] for o in offsets if o is not None}
            if len(rows) > 1:
                raise RuntimeError("LowmemPyramidPlan: with off_row every offset tensor holds one row per call")
            self.NO = rows.pop() if rows else 1
            self._keep = self._keep + (off_row,)
            self._rows = off_row.data_ptr() if self.B else None
            self._chunked = 1 if chunked else 0
            self._fn = lib.lgu_lowmem_pyramid_calls_fwd_h16
        elif chunked:
            self._fn = lib.lgu_lowmem_pyramid_chunked_fwd_h16 if dt == torch.float16 else lib.lgu_lowmem_pyramid_chunked_fwd_f32
        else:
            self._fn = lib.lgu_lowmem_pyramid_fwd_h16 if dt == torch.float16 else lib.lgu_lowmem_pyramid_fwd_f32

    def __call__(self, coords, out=None):
        _check(coords, "coords")
        B, S, H1, W1, _ = coords.shape
        if (B, H1, W1) != (self.B, self.H1, self.W1):
            raise RuntimeError("coords must be (B,S,H1,W1,2) for the planned feature maps")
        ch = self.L * (2 * self.radius + 1) ** 2
        if out is None:
            out = torch.empty((B, S, ch, H1, W1), dtype=torch.float32, device=self.device)
        if B == 0:
            return out
        if self._rows is not None:
            if S != 1:
                raise RuntimeError("LowmemPyramidPlan: off_row serves one sample per pixel")
            with self._guard:
                rc = self._fn(self._keep[0].data_ptr(), self._f2, coords.data_ptr(), self._op, out.data_ptr(), self.L, self.lbase, B,
                              H1, W1, self._h2, self._w2, self.C, self.NO, self._rows, self.radius, self._ii, self._jj,
                              self._chunked, torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(rc, "lowmem_pyramid_forward (several calls)")
            return out
        with self._guard:
            rc = self._fn(self._keep[0].data_ptr(), self._f2, coords.data_ptr(), self._op, out.data_ptr(), self.L, self.lbase, B, S,
                          H1, W1, self._h2, self._w2, self.C, self.NO, self.radius, self._ii, self._jj,
                          torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "lowmem_pyramid_forward")
        return out


def lowmem_chunked(fmap):
    """Channel-last feature map (F,H,W,C), half or float -> the chunk-planar form (F, C/k, H, W, k), k = 16 bytes of
    channels (8 halves / 4 floats), that LowmemPyramidPlan(chunked=True) reads.  A permuted copy (setup, once per map)."""
    k = 8 if fmap.dtype == torch.float16 else 4
    F, H, W, C = fmap.shape
    if C % k != 0:
        raise _lib.UnsupportedShape("lowmem_chunked: C must be a multiple of %d" % k)
    return fmap.view(F, H, W, C // k, k).permute(0, 3, 1, 2, 4).contiguous()


def lowmem_pyramid_forward_mixed(fmap1, fmap2s, coords, offsets, radius, out=None, ii=None, jj=None, lbase=0, chunked=False,
                                 off_row=None):
    """One-shot form of LowmemPyramidPlan (half or float feature maps)."""
    with torch.cuda.device(fmap1.device):
        return LowmemPyramidPlan(fmap1, fmap2s, offsets, radius, ii=ii, jj=jj, lbase=lbase, chunked=chunked,
                                 off_row=off_row)(coords, out=out)


PYR_PROBE, PYR_TILED, PYR_COORDS_LAST = 1, 2, 4  # flags of lgu_defcorr_pyramid_fwd_f32 (include/lgu_corr.h)
PYR_OUT_NHWC, PYR_OUT_F16 = 8, 16
OUT_FORMATS = {"planar": 0, "nhwc": PYR_OUT_NHWC, "nhwc_f16": PYR_OUT_NHWC | PYR_OUT_F16}


def _pyr_out(out_format, E, C, H1, W1, device, out):
    """Output tensor of the fused pyramid sampler, logical shape (E, C, H1, W1) in every format.
    "planar": contiguous fp32, the reference's tensor (corr.py:109).  "nhwc" / "nhwc_f16": the same values stored
    channel-last (torch.channels_last strides) in fp32 / half — what the corr_encoder 1x1 convolution that consumes
    the lookup under autocast wants (droid_net.py:76-80); half = Tensor.half() of the fp32 result, bit for bit."""
    if out_format not in OUT_FORMATS:
        raise RuntimeError("out_format must be one of %s" % sorted(OUT_FORMATS))
    if out_format == "planar":
        if out is None:
            return torch.empty((E, C, H1, W1), dtype=torch.float32, device=device)
        _check(out, "out")
        return out
    dt = torch.float16 if out_format == "nhwc_f16" else torch.float32
    if out is None:
        return torch.empty((E, H1, W1, C), dtype=dt, device=device).permute(0, 3, 1, 2)
    if not out.is_cuda:
        raise RuntimeError("out must be a CUDA tensor")
    if out.dtype != dt or tuple(out.shape) != (E, C, H1, W1) or not out.permute(0, 2, 3, 1).is_contiguous():
        raise RuntimeError("out must be a channels-last (E,C,H1,W1) %s tensor for out_format=%r" % (dt, out_format))
    return out
TILE_H, TILE_W = 4, 8        # tiled slice layout: 4 x 8 element tiles, one 128-byte line each


def tiled_shape(E, H1, W1, H2, W2):
    """Shape of a pyramid level in the tiled slice layout (include/lgu_corr.h, LGU_PYR_TILED)."""
    return (E, H1, W1, -(-H2 // TILE_H), -(-W2 // TILE_W), TILE_H, TILE_W)


def volume_retile(volume, to_tiled=True, hw=None):
    """Layout conversion of a pyramid level.  to_tiled: (E,H1,W1,H2,W2) row-major slices -> tiled
    (E,H1,W1,ceil(H2/4),ceil(W2/8),4,8) (padding zero-filled).  Otherwise the inverse; `hw` = the logical
    (H2, W2) when the tiled form is padded.  Setup / test utility, not on the lookup path."""
    _check(volume, "volume")
    if to_tiled:
        E, H1, W1, H2, W2 = volume.shape
        out = torch.empty(tiled_shape(E, H1, W1, H2, W2), dtype=volume.dtype, device=volume.device)
    else:
        E, H1, W1, nty, ntx, th, tw = volume.shape
        if (th, tw) != (TILE_H, TILE_W):
            raise RuntimeError("volume_retile: not a tiled pyramid level")
        H2, W2 = hw if hw is not None else (nty * TILE_H, ntx * TILE_W)
        out = torch.empty((E, H1, W1, H2, W2), dtype=volume.dtype, device=volume.device)
    n = E * H1 * W1
    if n == 0:
        return out
    with torch.cuda.device(volume.device):
        rc = _lib.load().lgu_volume_retile_f32(_ptr(volume), _ptr(out), n, H2, W2, 1 if to_tiled else 0, _stream(volume))
    _lib.check(rc, "volume_retile")
    return out


def _level_dims(volumes, tiled, level_hw):
    if not tiled:
        return [v.shape[3] for v in volumes], [v.shape[4] for v in volumes]
    if level_hw is None:
        level_hw = [(v.shape[3] * TILE_H, v.shape[4] * TILE_W) for v in volumes]
    for v, (h, w) in zip(volumes, level_hw):
        if v.dim() != 7 or tuple(v.shape[3:]) != (-(-h // TILE_H), -(-w // TILE_W), TILE_H, TILE_W):
            raise RuntimeError("tiled pyramid level must be (E,H1,W1,ceil(H2/4),ceil(W2/8),4,8)")
    return [h for h, _ in level_hw], [w for _, w in level_hw]


def defcorr_pyramid_forward(volumes, coords, offsets, radius, probe=False, out=None, tiled=False, level_hw=None,
                            coords_last=False, out_format="planar"):
    """Fused CorrBlock.__call__ body (reference droid_slam/modules/corr.py:88-109): all
    pyramid levels in ONE launch, written straight into the concatenated tensor.

    volumes: list of L tensors (E,H1,W1,H2l,W2l); coords (E,2,H1,W1) in level-0 units
    (each level samples at coords / 2^l); offsets: list of L tensors (E,H1,W1,rd,rd,2)
    or None for a structurally-zero level.  Offsets are modified in place (centre zeroing).
    tiled=True: volumes are in the tiled slice layout (volume_retile / volume_pyramid(tiled=True));
    level_hw = their logical (H2, W2) when padded.  Same results, fewer HBM lines touched.
    coords_last=True: coords is (E,H1,W1,2), x and y interleaved (no permute pass in front of the lookup).
    out_format: "planar" (default, the reference's tensor), "nhwc" or "nhwc_f16" (see _pyr_out; tiled pyramids only).
    Returns (E, L*rd*rd, H1, W1).
    """
    L = len(volumes)
    if len(offsets) != L:
        raise RuntimeError("defcorr_pyramid_forward: need one offset entry (tensor or None) per level")
    _check(coords, "coords")
    for l, v in enumerate(volumes):
        _check(v, "volume[%d]" % l)
        if offsets[l] is not None:
            _check(offsets[l], "offset[%d]" % l)
    E, H1, W1 = volumes[0].shape[:3]
    rd = 2 * radius + 1
    hs, ws = _level_dims(volumes, tiled, level_hw)
    out = _pyr_out(out_format, E, L * rd * rd, H1, W1, coords.device, out)
    if E == 0:
        return out
    vp = (_vp * L)(*[v.data_ptr() for v in volumes])
    op = (_vp * L)(*[(o.data_ptr() if o is not None else None) for o in offsets])
    h2 = (ctypes.c_int * L)(*hs)
    w2 = (ctypes.c_int * L)(*ws)
    flags = ((PYR_PROBE if probe else 0) | (PYR_TILED if tiled else 0) | (PYR_COORDS_LAST if coords_last else 0)
             | OUT_FORMATS[out_format])
    if tuple(coords.shape) != ((E, H1, W1, 2) if coords_last else (E, 2, H1, W1)):
        raise RuntimeError("defcorr_pyramid_forward: coords must be %s" % ("(E,H1,W1,2)" if coords_last else "(E,2,H1,W1)"))
    with torch.cuda.device(coords.device):
        rc = _lib.load().lgu_defcorr_pyramid_fwd_f32(vp, _ptr(coords), op, _ptr(out), L, E, H1, W1, h2, w2, radius,
                                                     flags, _stream(coords))
    _lib.check(rc, "defcorr_pyramid_forward")
    return out


def volume_pyramid(means, covs, volume, num_levels, radius=4, inplace=False, tiled=False, det=None):
    """Fused volume post-processing of CorrBlock.__init__ (reference gaussianMask_cuda.py:84-86
    + corr.py:79-86): level0 = gaussianMask(means, covs, volume, radius) / (6.28*sqrt(det)) +
    volume, levels 1.. by 2x2 average pooling of the target dims — one pass over the volume.
    Returns the list of pyramid levels; with inplace=True level 0 reuses `volume`'s storage.
    tiled=True: the levels are written in the tiled slice layout (tiled_shape); same values.
    A HALF volume (the matmul of half feature maps) is converted by the kernel's own load — the levels are fp32 and
    equal those of volume.float(); inplace is then ignored.
    det: the (E, H1*W1) determinant GaussianMask.gaussian_parameters returned, fp32 or half.  Inside autocast the
    reference's det is half and its denominator `6.28 * torch.sqrt(det)` carries two half roundings (factor_graph.py:90,
    gaussianMask_cuda.py:79-86); passing that det reproduces them.  None: det = cov0 * cov1 in fp32."""
    half_in = volume.dtype == torch.float16
    if half_in:
        _check(means, "means", covs, "covs")
        if not (volume.is_cuda and volume.is_contiguous()):
            raise RuntimeError("volume must be a contiguous CUDA tensor")
        inplace = False
    else:
        _check(volume, "volume", means, "means", covs, "covs")
    E, H1, W1, H2, W2 = volume.shape
    f32 = torch.float32
    if tiled:
        alias = inplace and H2 % TILE_H == 0 and W2 % TILE_W == 0
        levels = [volume.view(tiled_shape(E, H1, W1, H2, W2)) if alias
                  else torch.empty(tiled_shape(E, H1, W1, H2, W2), dtype=f32, device=volume.device)]
        for l in range(1, num_levels):
            levels.append(torch.empty(tiled_shape(E, H1, W1, H2 >> l, W2 >> l), dtype=f32, device=volume.device))
    else:
        levels = [volume if inplace else torch.empty(volume.shape, dtype=f32, device=volume.device)]
        for l in range(1, num_levels):
            levels.append(torch.empty((E, H1, W1, H2 >> l, W2 >> l), dtype=f32, device=volume.device))
    if E == 0:
        return levels
    lp = (_vp * num_levels)(*[t.data_ptr() for t in levels])
    if det is not None:
        if det.dtype not in (torch.float32, torch.float16) or not (det.is_cuda and det.is_contiguous()) or det.numel() != E * H1 * W1:
            raise RuntimeError("det must be a contiguous fp32 or half CUDA tensor of E*H1*W1 elements")
        with torch.cuda.device(volume.device):
            rc = _lib.load().lgu_volume_pyramid_det(_ptr(means), _ptr(covs), _ptr(det), 1 if det.dtype == torch.float16 else 0,
                                                    _ptr(volume), 1 if half_in else 0, lp, num_levels, E, H1, W1, H2, W2, radius,
                                                    1 if tiled else 0, _stream(volume))
        _lib.check(rc, "volume_pyramid")
        return levels
    with torch.cuda.device(volume.device):
        if half_in:
            rc = _lib.load().lgu_volume_pyramid_h16(_ptr(means), _ptr(covs), _ptr(volume), lp, num_levels, E, H1, W1, H2, W2,
                                                    radius, 1 if tiled else 0, _stream(volume))
        else:
            fn = _lib.load().lgu_volume_pyramid_tiled_f32 if tiled else _lib.load().lgu_volume_pyramid_f32
            rc = fn(_ptr(means), _ptr(covs), _ptr(volume), lp, num_levels, E, H1, W1, H2, W2, radius, _stream(volume))
    _lib.check(rc, "volume_pyramid")
    return levels


def volume_build_pyramid(fmap1, fmap2, means, covs, det=None, num_levels=4, radius=4):
    """CorrBlock.__init__'s volume built straight into the TILED pyramid (reference corr.py:145-152 matmul of the feature
    maps / 4 each + :64 .float() + gaussianMask_cuda.py:84-86 + corr.py:79-86) in one launch on the matrix cores
    (csrc/volbuild.hip): the raw all-pairs volume never reaches HBM.
      fp32:  fmap1, fmap2 (E, C, H, W) fp32 contiguous (un-scaled)                       -> lgu_volume_build_pyramid_f32
      half:  fmap1 = the channel-last pair (E, H, W, 2C) half (CorrBlock's `t`), fmap2 None -> lgu_volume_build_pyramid_h16
             (the product is rounded to half as the reference's half GEMM rounds it)
    means, covs (E, H, W, 2); det as for volume_pyramid.  Returns the levels in tiled_shape form.  Raises UnsupportedShape
    for sizes the kernels do not serve (the caller then takes matmul + volume_pyramid): H % 8, W not in {16, 32, 64},
    C % 16 (fp32) / C % 32 (half), num_levels != 4."""
    half = fmap2 is None
    if half:
        if fmap1.dtype != torch.float16 or not (fmap1.is_cuda and fmap1.is_contiguous()) or fmap1.dim() != 4 or fmap1.shape[3] % 2:
            raise RuntimeError("volume_build_pyramid: the half form takes one contiguous (E,H,W,2C) half CUDA tensor")
        _check(means, "means", covs, "covs")
        E, H, W, C2 = fmap1.shape
        C = C2 // 2
    else:
        _check(fmap1, "fmap1", fmap2, "fmap2", means, "means", covs, "covs")
        E, C, H, W = fmap1.shape
        if tuple(fmap2.shape) != (E, C, H, W):
            raise RuntimeError("volume_build_pyramid: fmap1 / fmap2 (E,C,H,W)")
    if means.numel() != E * H * W * 2 or covs.numel() != E * H * W * 2:
        raise RuntimeError("volume_build_pyramid: means / covs (E,H,W,2)")
    if num_levels != 4:
        raise _lib.UnsupportedShape("volume_build_pyramid: four levels")
    levels = [torch.empty(tiled_shape(E, H, W, H >> l, W >> l), dtype=torch.float32, device=fmap1.device) for l in range(num_levels)]
    if E == 0:
        return levels
    dptr, dhalf = None, 0
    if det is not None:
        if det.dtype not in (torch.float32, torch.float16) or not (det.is_cuda and det.is_contiguous()) or det.numel() != E * H * W:
            raise RuntimeError("det must be a contiguous fp32 or half CUDA tensor of E*H*W elements")
        dptr, dhalf = _ptr(det), 1 if det.dtype == torch.float16 else 0
    lp = (_vp * num_levels)(*[t.data_ptr() for t in levels])
    with torch.cuda.device(fmap1.device):
        if half:
            work = torch.empty_like(fmap1)   # the maps in MFMA fragment order (written by the entry's first launch)
            rc = _lib.load().lgu_volume_build_pyramid_h16(_ptr(fmap1), _ptr(work), _ptr(means), _ptr(covs), dptr, dhalf, lp,
                                                          num_levels, E, C, H, W, radius, _stream(fmap1))
        else:
            rc = _lib.load().lgu_volume_build_pyramid_f32(_ptr(fmap1), _ptr(fmap2), _ptr(means), _ptr(covs), dptr, dhalf, lp,
                                                          num_levels, E, C, H, W, radius, _stream(fmap1))
    _lib.check(rc, "volume_build_pyramid")
    return levels


def gaussian_params(mean_ofs, cov_raw, h, w, eps=1e-5):
    """Tail of GaussianMask.gaussian_parameters after the two linear heads (reference gaussianMask_cuda.py:69-83) in one
    launch: mean_ofs, cov_raw (E, h*w, 2)-shaped, both fp32 or both half.  Returns mean (E,h,w,2) fp32, cov (E,h,w,2) fp32,
    det (E, h*w) in the input dtype."""
    if mean_ofs.dtype not in (torch.float32, torch.float16) or cov_raw.dtype != mean_ofs.dtype:
        raise RuntimeError("gaussian_params: fp32 or half inputs of one dtype")
    for t, nm in ((mean_ofs, "mean_ofs"), (cov_raw, "cov_raw")):
        if not (t.is_cuda and t.is_contiguous()) or t.numel() % (h * w * 2) != 0:
            raise RuntimeError("%s must be a contiguous CUDA tensor of (E, h*w, 2) elements" % nm)
    E = mean_ofs.numel() // (h * w * 2)
    if cov_raw.numel() != mean_ofs.numel():
        raise RuntimeError("gaussian_params: mean_ofs and cov_raw differ in size")
    mean = torch.empty((E, h, w, 2), dtype=torch.float32, device=mean_ofs.device)
    cov = torch.empty_like(mean)
    det = torch.empty((E, h * w), dtype=mean_ofs.dtype, device=mean_ofs.device)
    if E:
        with torch.cuda.device(mean.device):
            rc = _lib.load().lgu_gaussian_params(_ptr(mean_ofs), _ptr(cov_raw), _ptr(mean), _ptr(cov), _ptr(det), E, h, w,
                                                 1 if mean_ofs.dtype == torch.float16 else 0, float(eps), _stream(mean))
        _lib.check(rc, "gaussian_params")
    return mean, cov, det


def probe_mask_scale_(probe, offset):
    """offset *= sigmoid(var(probe over its taps)) in place (reference corr.py:203-207): probe (E,1,T,H,W) or (E,T,H,W)
    fp32 from altcorr_forward / the fused probe launch, offset (E,H,W,C) fp32."""
    _check(probe, "probe", offset, "offset")
    E, H, W, C = offset.shape
    T = probe.numel() // max(E * H * W, 1)
    if probe.numel() != E * T * H * W or T < 2:
        raise RuntimeError("probe_mask_scale_: probe must hold T >= 2 samples per pixel of offset")
    if E:
        with torch.cuda.device(offset.device):
            rc = _lib.load().lgu_probe_mask_scale_f32(_ptr(probe), _ptr(offset), E, H * W, T, C, _stream(offset))
        _lib.check(rc, "probe_mask_scale")
    return offset


def pack_offset_conv(weight, bias, scale=4.0):
    """Operands of lgu_offset_conv_frames_h16 from a Conv2d(2C, Cout, 3, padding=1)'s parameters: the weight times
    `scale` (AltCorrBlock feeds the convolution 4 x the stored frames; a power of two, exact) split into two half parts
    hi + lo and laid out in MFMA fragment order.  Returns (wpack, bias_fp32, Cout, C)."""
    Cout, C2, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or C2 % 64 != 0 or Cout > 112 or bias is None:
        raise RuntimeError("pack_offset_conv: expected a 3x3 convolution with bias, <= 112 output and 64 k input channels")
    w = torch.zeros((112, C2, 9), dtype=torch.float32, device=weight.device)
    w[:Cout] = (weight.detach().float() * scale).reshape(Cout, C2, 9)
    hi = w.half()
    lo = (w - hi.float()).half()
    ks = C2 // 32

    def frag(t):   # (112, C2, 9) -> [tap][kstep][ntile][kg][nl][8]
        return t.view(7, 16, ks, 4, 8, 9).permute(5, 2, 0, 3, 1, 4)

    wpack = torch.stack((frag(hi), frag(lo)), dim=2).contiguous()   # [tap][kstep][part][ntile][kg][nl][8]
    return wpack, bias.detach().float().contiguous(), Cout, C2 // 2


def pack_offset_conv_parts(weight, bias, scale=4.0):
    """The two input halves of a Conv2d(2C, Cout, 3, padding=1) packed separately (pack_offset_conv of weight[:, :C] and
    weight[:, C:]) for the per-frame partial convolutions of OffsetHeadCache.  Returns (wpack_a, wpack_b, bias, Cout, C)."""
    Cout, C2 = weight.shape[:2]
    C = C2 // 2
    if C % 64 != 0:
        raise _lib.UnsupportedShape("pack_offset_conv_parts: C must be a multiple of 64")
    wa = pack_offset_conv(weight[:, :C], bias, scale)
    wb = pack_offset_conv(weight[:, C:], bias, scale)
    return wa[0], wb[0], wa[1], Cout, C


class OffsetHeadCache:
    """One offset head of AltCorrBlock (a 3x3 convolution of cat(frames[ii], frames[jj])) through per-frame partial
    convolutions kept for the life of the block: P_A[f] + P_B[f'] for an edge (f, f').  Frames whose partials are missing
    are found and convolved on the device (lgu_offset_heads_mark / lgu_offset_conv_worklist_h16), so a call costs no host
    round trip; every later chunk that meets a frame again pays only the sum."""

    def __init__(self, frames, parts, frames_lo=None):
        wa, wb, bias, Cout, C = parts
        _check_dtype(frames, "frames", torch.float16)
        NF, H, W, Cf = frames.shape
        if Cf != C or not frames.is_contiguous():
            raise RuntimeError("OffsetHeadCache: frames must be contiguous (NF,H,W,%d)" % C)
        if (Cout * H * W) % 4 != 0:
            raise _lib.UnsupportedShape("OffsetHeadCache: Cout*H*W must be a multiple of 4")
        self.frames, self.frames_lo, self.parts = frames, frames_lo, parts
        self.NF, self.H, self.W, self.C, self.Cout = NF, H, W, C, Cout
        dev = frames.device
        self.PA = torch.empty((NF, Cout, H, W), dtype=torch.float32, device=dev)
        self.PB = torch.empty((NF, Cout, H, W), dtype=torch.float32, device=dev)
        self.done = torch.zeros((2, NF), dtype=torch.int32, device=dev)
        self.count = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.worklist = None

    def mark(self, ii, jj):
        """Device-side claim of the frames of ii / jj that have no partials yet -> (worklist, count) tensors.  Two heads
        over the same frames (AltCorrBlock's full-resolution and residual head) share ONE mark pass: the flags of the
        cache that marks stand for both."""
        _check_dtype(ii, "ii", torch.int64)
        _check_dtype(jj, "jj", torch.int64)
        E = ii.shape[0]
        if self.worklist is None or self.worklist.numel() < 2 * E:
            self.worklist = torch.empty(max(2 * E, 64), dtype=torch.int32, device=self.frames.device)
        if E:
            with torch.cuda.device(self.frames.device):
                _lib.check(_lib.load().lgu_offset_heads_mark(_ptr(ii), _ptr(jj), E, _ptr(self.done), self.NF, _ptr(self.worklist),
                                                             _ptr(self.count), _stream(self.frames)), "offset_heads_mark")
        return self.worklist, self.count

    def convolve(self, work, E):
        """Partial convolutions of the frames on the worklist `work` = (worklist, count) of a mark pass over E edges."""
        if E == 0:
            return
        wa, wb, bias, Cout, C = self.parts
        wl, cnt = work
        with torch.cuda.device(self.frames.device):
            _lib.check(_lib.load().lgu_offset_conv_worklist_h16(
                _ptr(self.frames), _ptr(self.frames_lo) if self.frames_lo is not None else None, _ptr(wl), _ptr(cnt), 2 * E, _ptr(wa),
                _ptr(wb), _ptr(bias), _ptr(self.PA), _ptr(self.PB), self.H, self.W, C, Cout, _stream(self.frames)), "offset_conv_worklist")

    def combine(self, ii, jj, reset=None):
        """(E, Cout, H, W) fp32 = P_A[ii] + P_B[jj]; reset: the count tensor of the mark pass, zeroed by the LAST combine
        that follows it."""
        E = ii.shape[0]
        out = torch.empty((E, self.Cout, self.H, self.W), dtype=torch.float32, device=self.frames.device)
        if E:
            with torch.cuda.device(self.frames.device):
                _lib.check(_lib.load().lgu_offset_heads_combine_f32(_ptr(self.PA), _ptr(self.PB), _ptr(ii), _ptr(jj), _ptr(out), E,
                                                                    self.Cout * self.H * self.W, _ptr(reset) if reset is not None else None,
                                                                    _stream(self.frames)), "offset_heads_combine")
        return out

    def __call__(self, ii, jj):
        """One head on its own: mark, convolve what is missing, sum."""
        work = self.mark(ii, jj)
        self.convolve(work, ii.shape[0])
        return self.combine(ii, jj, reset=work[1])


def offset_conv_frames(frames, ii, jj, packed, frames_lo=None):
    """ofsMap(cat(frames[ii] * 4, frames[jj] * 4).float()) of AltCorrBlock.corr_fn (reference corr.py:174-189, :220)
    without materialising its input: frames (NF,H,W,C) half channel-last, ii / jj (E) int64, packed from
    pack_offset_conv.  frames_lo: optional second half part of the input (input = frames + frames_lo, same shape).
    Returns (E, Cout, H, W) fp32."""
    wpack, bias, Cout, C = packed
    _check_dtype(frames, "frames", torch.float16)
    if frames_lo is not None:
        _check_dtype(frames_lo, "frames_lo", torch.float16)
        if frames_lo.shape != frames.shape:
            raise RuntimeError("offset_conv_frames: frames_lo must have the shape of frames")
    _check_dtype(ii, "ii", torch.int64)
    _check_dtype(jj, "jj", torch.int64)
    NF, H, W, Cf = frames.shape
    if Cf != C or not frames.is_contiguous():
        raise RuntimeError("offset_conv_frames: frames must be contiguous (NF,H,W,%d)" % C)
    E = ii.shape[0]
    out = torch.empty((E, Cout, H, W), dtype=torch.float32, device=frames.device)
    if E == 0:
        return out
    with torch.cuda.device(frames.device):
        rc = _lib.load().lgu_offset_conv_frames_h16(_ptr(frames), _ptr(frames_lo) if frames_lo is not None else None, _ptr(ii),
                                                    _ptr(jj), _ptr(wpack), _ptr(bias), _ptr(out), E, H, W, C, Cout, _stream(frames))
    _lib.check(rc, "offset_conv_frames")
    return out


def offsets_finalize(o0, o1_lowres, eps=1e-5, autocast=None, probe=None):
    """Post-processing of the two offset convolutions' outputs (reference corr.py:117-135 / :217-235) in one pass:
    per-sample standardisation, 4*tanh, the level-1 residual mix, nearest upsampling of the pooled-resolution head and
    the transposition to the samplers' channel-last fp32 layout.  o0 (E,C,H,W), o1_lowres (E,C,Hl,Wl), both fp32 or
    both half.  Half inputs: every step rounded to half like torch's half kernels; with autocast (default: whether
    autocast is enabled now) level 1 is evaluated in fp32, because autocast promotes the nearest upsampling to fp32.
    probe (optional, (E,T,H,W) or (E,1,T,H,W) fp32, T >= 2 samples per pixel): off1 additionally scaled by the uncertainty
    mask sigmoid(var(probe)) of AltCorrBlock.corr_fn (corr.py:203-207) — what probe_mask_scale_ would do in a second pass.
    Returns (off0, off1), each (E,H,W,C) fp32."""
    if autocast is None:
        autocast = torch.is_autocast_enabled()
    if o0.dtype not in (torch.float32, torch.float16) or o1_lowres.dtype != o0.dtype:
        raise RuntimeError("offsets_finalize: fp32 or half inputs of one dtype")
    for t, n in ((o0, "o0"), (o1_lowres, "o1")):
        if not (t.is_cuda and t.is_contiguous() and t.dim() == 4):
            raise RuntimeError("%s must be a contiguous 4-d CUDA tensor" % n)
    E, C, H, W = o0.shape
    if o1_lowres.shape[:2] != (E, C):
        raise RuntimeError("offsets_finalize: o1 must have o0's (E, C)")
    Hl, Wl = o1_lowres.shape[2:]
    out0 = torch.empty((E, H, W, C), dtype=torch.float32, device=o0.device)
    out1 = torch.empty_like(out0)
    if E == 0:
        return out0, out1
    lib = _lib.load()
    scratch = torch.empty(int(lib.lgu_offsets_finalize_scratch_bytes(E)), dtype=torch.uint8, device=o0.device)
    mode = (2 if autocast else 1) if o0.dtype == torch.float16 else 0
    with torch.cuda.device(o0.device):
        if probe is not None:
            _check(probe, "probe")
            T = probe.numel() // max(E * H * W, 1)
            if probe.numel() != E * T * H * W or T < 2:
                raise RuntimeError("offsets_finalize: probe must hold T >= 2 samples per pixel")
            rc = lib.lgu_offsets_finalize_masked(_ptr(o0), _ptr(o1_lowres), _ptr(probe), T, _ptr(out0), _ptr(out1), _ptr(scratch), E, C,
                                                 H, W, Hl, Wl, mode, float(eps), _stream(o0))
        else:
            rc = lib.lgu_offsets_finalize(_ptr(o0), _ptr(o1_lowres), _ptr(out0), _ptr(out1), _ptr(scratch), E, C, H, W, Hl, Wl,
                                          mode, float(eps), _stream(o0))
    _lib.check(rc, "offsets_finalize")
    return out0, out1


ENC_N = 128  # output channels of the fused corr_encoder layer (csrc/defcorr.hip)


def pack_encoder_layer(weight, bias):
    """Operands of the fused lookup + first corr_encoder layer (lgu_defcorr_pyramid_enc_fwd_f32) from the 1x1
    convolution's parameters (reference droid_net.py:76-77): the (128, K, 1, 1) weight as half rows zero-padded to a
    multiple of 32 entries, the bias as half.  Returns (w (128, Kp) half, b (128,) half)."""
    n, k = weight.shape[0], weight.shape[1]
    if n != ENC_N or weight.numel() != n * k or bias is None or bias.numel() != n:
        raise RuntimeError("pack_encoder_layer: expected a (%d, K, 1, 1) convolution weight with bias" % ENC_N)
    kp = -(-k // 32) * 32
    w = torch.zeros((n, kp), dtype=torch.float16, device=weight.device)
    w[:, :k] = weight.detach().reshape(n, k).half()
    return w, bias.detach().half().contiguous()


class DefcorrPyramidPlan:
    """Prepared launch of the fused pyramid sampler for a fixed pyramid / offset set: the
    pointer and size tables are built once, a call costs one ctypes invocation.  Used by
    CorrBlock (the pyramid is fixed between `cat`/`__getitem__` calls while `coords` changes
    every update) and by bench.py so the step is not bound by Python argument handling.
    """

    def __init__(self, volumes, offsets, radius, probe=False, tiled=False, level_hw=None, coords_last=False, slots=None,
                 out_format="planar", encoder=None):
        """out_format: see _pyr_out.  encoder = (w, b) from pack_encoder_layer: the first corr_encoder layer runs in the
        same launch and the call returns relu(W1 . half(samples) + b1) as a channel-last half (E,128,H1,W1) tensor
        (tiled pyramids only; out_format is then ignored).  coords_last=True: calls take coords as (E,H1,W1,2) (x, y interleaved) instead of (E,2,H1,W1).
        slots: int32 device tensor (E,): edge e's volume slices live at volumes[l][slots[e]] (the level buffers may
        hold more slots than E); offsets / coords / out stay indexed by e."""
        L = len(volumes)
        if len(offsets) != L:
            raise RuntimeError("DefcorrPyramidPlan: need one offset entry (tensor or None) per level")
        named = []
        for l, v in enumerate(volumes):
            named += [v, "volume[%d]" % l]
            if offsets[l] is not None:
                named += [offsets[l], "offset[%d]" % l]
        _check(*named)
        if slots is not None:
            _check_dtype(slots, "slots", torch.int32)
        self._keep = (list(volumes), list(offsets), slots)  # keep the buffers alive
        self.L, self.radius = L, radius
        if out_format not in OUT_FORMATS:
            raise RuntimeError("out_format must be one of %s" % sorted(OUT_FORMATS))
        self.flags = ((PYR_PROBE if probe else 0) | (PYR_TILED if tiled else 0) | (PYR_COORDS_LAST if coords_last else 0)
                      | OUT_FORMATS[out_format])
        self.coords_last, self.out_format = coords_last, out_format
        self._enc = None
        if encoder is not None:
            w, b = encoder
            kp = -(-(L * (2 * radius + 1) ** 2) // 32) * 32
            if (w.dtype != torch.float16 or b.dtype != torch.float16 or tuple(w.shape) != (ENC_N, kp)
                    or tuple(b.shape) != (ENC_N,) or not (w.is_cuda and b.is_cuda and w.is_contiguous() and b.is_contiguous())):
                raise RuntimeError("encoder must be pack_encoder_layer(...) operands for %d input channels" % (L * (2 * radius + 1) ** 2))
            self._enc = (w, b)
            self.flags &= ~(PYR_OUT_NHWC | PYR_OUT_F16)
            self.out_format = "nhwc_f16"
            self._fn_enc = _lib.load().lgu_defcorr_pyramid_enc_fwd_f32
        self.E, self.H1, self.W1 = volumes[0].shape[:3]
        if slots is not None:
            self.E = slots.shape[0]
        self._slots = slots.data_ptr() if slots is not None and slots.numel() else None
        self.device = volumes[0].device
        self._guard = _CurrentDevice(self.device)
        self.channels = L * (2 * radius + 1) ** 2
        self._vp = (_vp * L)(*[v.data_ptr() for v in volumes])
        self._op = (_vp * L)(*[(o.data_ptr() if o is not None else None) for o in offsets])
        hs, ws = _level_dims(volumes, tiled, level_hw)
        self._h2 = (ctypes.c_int * L)(*hs)
        self._w2 = (ctypes.c_int * L)(*ws)
        self._fn = _lib.load().lgu_defcorr_pyramid_fwd_f32
        self._fn_slots = _lib.load().lgu_defcorr_pyramid_slots_fwd_f32

    def __call__(self, coords, out=None):
        want = (self.E, self.H1, self.W1, 2) if self.coords_last else (self.E, 2, self.H1, self.W1)
        if tuple(coords.shape) != want:
            raise RuntimeError("coords must be %s" % ("(E,H1,W1,2)" if self.coords_last else "(E,2,H1,W1)"))
        _check(coords, "coords")
        out = _pyr_out(self.out_format, self.E, ENC_N if self._enc else self.channels, self.H1, self.W1, self.device, out)
        if self.E == 0:
            return out
        st = torch.cuda.current_stream(self.device).cuda_stream
        with self._guard:
            if self._enc is not None:
                rc = self._fn_enc(self._vp, self._slots, coords.data_ptr(), self._op, self._enc[0].data_ptr(),
                                  self._enc[1].data_ptr(), out.data_ptr(), self.L, self.E, self.H1, self.W1, self._h2, self._w2,
                                  self.radius, ENC_N, self.flags, st)
            elif self._slots is not None:
                rc = self._fn_slots(self._vp, self._slots, coords.data_ptr(), self._op, out.data_ptr(), self.L, self.E, self.H1,
                                    self.W1, self._h2, self._w2, self.radius, self.flags, st)
            else:
                rc = self._fn(self._vp, coords.data_ptr(), self._op, out.data_ptr(), self.L, self.E, self.H1, self.W1,
                              self._h2, self._w2, self.radius, self.flags, st)
        _lib.check(rc, "defcorr_pyramid_forward")
        return out
