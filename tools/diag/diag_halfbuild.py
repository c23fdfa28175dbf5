import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lgu_slam_amd as lgu
torch.manual_seed(11)
h, w = 16, 32
ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).cuda()
GA = lgu.GaussianMask(h, w).cuda()
f1 = (torch.randn(1, 3, 128, h, w, device="cuda") * 0.5).half()
f2 = (torch.randn(1, 3, 128, h, w, device="cuda") * 0.5).half()
ys, xs = torch.meshgrid(torch.arange(h, device="cuda").float(), torch.arange(w, device="cuda").float(), indexing="ij")
coords = (torch.stack([xs, ys], -1)[None, None].repeat(1, 3, 1, 1, 1) + torch.randn(1, 3, h, w, 2, device="cuda")).contiguous()
blks, outs = [], []
for flag in (False, True, False):
    lgu.CorrBlock.FUSED_BUILD_HALF = flag
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        blk = lgu.CorrBlock(ofsMap, ofsRes, GA, f1, f2)
        pre = [o.clone() for o in blk.offset]
        out = blk(coords)[0].float()
    blks.append((blk, pre)); outs.append(out)
for a, b, name in ((0, 1, "lib vs build"), (0, 2, "lib vs lib")):
    print(name)
    for l in range(4):
        sa, sb = blks[a][0]._store[l], blks[b][0]._store[l]
        print(" level", l, "store max diff", float((sa - sb).abs().max()), "scale", float(sa.abs().max()), "frac", float(((sa - sb).abs() > 0).float().mean()))
        print("   offsets pre diff", float((blks[a][1][l].float() - blks[b][1][l].float()).abs().max()),
              "post diff", float((blks[a][0].offset[l].float() - blks[b][0].offset[l].float()).abs().max()))
    d = (outs[a] - outs[b]).abs()
    sc = float(outs[a].abs().max())
    print(" out max diff", float(d.max()), "frac>2e-3sc", float((d > 2e-3 * sc).float().mean()))
    dd = d.view(3, 4, 49, h, w)
    for l in range(4):
        print("   level", l, "max", float(dd[:, l].max()), "frac", float((dd[:, l] > 2e-3 * sc).float().mean()))
