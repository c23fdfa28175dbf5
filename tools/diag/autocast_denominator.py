"""DIAGNOSTIC: the fused pyramid builder against the torch composition of GaussianMask.forward under autocast, pixel by
pixel (which denominators differ and why).  Found the v_fma_mixlo_f16 single-rounding issue recorded in csrc/gaussmask.hip."""
import sys, torch
sys.path.insert(0, '/root/repo')
import lgu_slam_amd as lgu
torch.manual_seed(5)
E, h, w = 2, 48, 64
dev_ = "cuda"
GA = lgu.GaussianMask(h, w).to(dev_)
torch.nn.init.normal_(GA.meanMap.weight, 0, 0.3)
f1 = (torch.randn(1, E, 128, h, w, device=dev_) * 0.5).half()
f2 = (torch.randn(1, E, 128, h, w, device=dev_) * 0.5).half()
feats = torch.cat((f1.reshape(E, 128, h, w), f2.reshape(E, 128, h, w)), 1).permute(0, 2, 3, 1).contiguous()
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    vol = lgu.CorrBlock.corr(f1, f2).view(E, h, w, h, w).float()
    mean, cov, det = GA.gaussian_parameters(feats)   # fused
    lgu.gaussian_mask.FUSED_PARAMS = False
    mean2, cov2, det2 = GA.gaussian_parameters(feats)   # torch
    print("params equal:", torch.equal(mean, mean2), torch.equal(cov, cov2), torch.equal(det, det2), det.dtype, det2.dtype,
          (cov - cov2).abs().max().item(), (mean - mean2).abs().max().item())
    got = lgu.ops.volume_pyramid(mean.float().contiguous(), cov.float().contiguous(), vol.clone(), 1, 4, det=det.contiguous())[0]
    corr1, = lgu.ops.gaussianMask(mean.float().contiguous(), cov.float().contiguous(), vol, 4)
    den = 6.28 * torch.sqrt(det).view(E, h, w, 1, 1)
    print("den dtype", den.dtype, corr1.dtype)
    want = corr1 / den + vol
    d = (got - want).abs()
    print("same params: max diff", d.max().item(), "frac > 1e-6:", (d > 1e-6).float().mean().item())
    ref0, _, _ = GA(feats, vol)
    d2 = (got - ref0).abs().amax(dim=(3, 4))
    print("vs GA.forward: max", d2.max().item(), "pixels > 1e-6:", (d2 > 1e-6).float().mean().item())
    got32 = lgu.ops.volume_pyramid(mean.float().contiguous(), cov.float().contiguous(), vol.clone(), 1, 4)[0]
    print("fp32 den vs want:", (got32 - want).abs().max().item())
with torch.no_grad():
    den_t = (6.28 * torch.sqrt(det)).float().view(-1)
    s = torch.sqrt(det.float()).half()
    den_e = (s.float() * torch.tensor(6.28, dtype=torch.float32, device=dev_)).half().float().view(-1)
    print("emulated den equal torch:", torch.equal(den_t, den_e), (den_t != den_e).float().mean().item())
    bad = (d.amax(dim=(3, 4)) > 1e-6).view(-1).nonzero().flatten()[:5]
    print("bad pixels", bad.tolist(), "det", det.view(-1)[bad].tolist(), "den_t", den_t[bad].tolist(), "den_e", den_e[bad].tolist())
    # den implied by the fused kernel: pick an in-window element
    for b in bad[:3].tolist():
        g, wv, v = got.view(-1, h, w)[b], want.view(-1, h, w)[b], vol.view(-1, h, w)[b]
        c1 = corr1.view(-1, h, w)[b]
        idx = (c1.abs()).argmax()
        print("pix", b, "corr1", c1.view(-1)[idx].item(), "got-v", (g.view(-1)[idx] - v.view(-1)[idx]).item(), "want-v", (wv.view(-1)[idx] - v.view(-1)[idx]).item(),
              "den_from_got", (c1.view(-1)[idx] / (g.view(-1)[idx] - v.view(-1)[idx])).item())
