import sys, torch
sys.path.insert(0, '/root/repo')
import lgu_slam_amd as lgu
from lgu_slam_amd import ops, _lib
torch.manual_seed(23)
H, W = 18, 22
N, C = 9, 128
fmaps = (torch.randn(1, N, C, H, W, device="cuda") * 0.5).half()
ofsMap = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
ofsRes = torch.nn.Conv2d(2 * C, 98, 3, padding=1).cuda()
ys, xs = torch.meshgrid(torch.arange(H, device="cuda").float(), torch.arange(W, device="cuda").float(), indexing="ij")
ii = torch.tensor([0, 0, 1, 1, 1, 2, 3, 3, 4, 5, 5, 6, 7, 8, 8], device="cuda")
jj = torch.tensor([1, 2, 0, 2, 3, 3, 4, 2, 5, 4, 6, 7, 8, 7, 6], device="cuda")
counts = [5, 1, 3, 4, 2]
coords = (torch.stack([xs, ys], -1)[None, None] + 2.0 * torch.randn(1, ii.numel(), H, W, 2, device="cuda")).contiguous()
with torch.no_grad():
    blk = lgu.AltCorrBlock(ofsMap, ofsRes, None, fmaps)
    frames = blk._frame_operands()
    try:
        h = ops.OffsetHeadCache(frames[0], ops.pack_offset_conv_parts(ofsMap.weight, ofsMap.bias))
        w = h.mark(ii[:1].contiguous(), jj[:1].contiguous()); h.convolve(w, 1); print("head0 ok")
    except _lib.UnsupportedShape as e:
        print("head0 unsupported:", e)
    outs = []
    for rep in range(2):
        b = lgu.AltCorrBlock(ofsMap, ofsRes, None, fmaps)
        parts, s = [], 0
        for c in counts:
            parts.append(b(coords[:, s:s + c], ii[s:s + c], jj[s:s + c])); s += c
        outs.append(torch.cat(parts, 1))
    print("loop reproducible:", torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max()))
    many = blk.call_many(coords, ii, jj, counts)
    print("many == loop:", torch.equal(many, outs[0]), float((many - outs[0]).abs().max()))
    s = 0
    for k, c in enumerate(counts):
        print(k, float((many[:, s:s+c] - outs[0][:, s:s+c]).abs().max())); s += c
