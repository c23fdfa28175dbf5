#!/usr/bin/env python3
"""bench.py — def-corr-sample throughput (Mpix·edges/s) of the fused deformable pyramid
sample on MI355X, with its HBM roofline and the CPU baseline timed beside it.

One STEP = one pass of the hot path (body of CorrBlock.__call__, reference
droid_slam/modules/corr.py:101-109: 4 pyramid levels x 49 taps per pixel) over one batch
of synthetic edges: BASELINE config 2 — 48x64 fmap, L=4, r=3, E=20 edges per GPU — ONE
kernel launch.  With --probe the level-1 uncertainty probe of corr.py:94-99 is timed in
the step as well.  Inputs are resident in HBM before the timed region.

Multi-GPU (--gpus N, launched by torch.distributed.run): factor-graph edges are
independent, so every rank samples its own E edges (weak scaling, no data-path
collective); `value` = all ranks' units / max-over-ranks time.  The sharded driver's one
real exchange — the all-gather of per-edge target/weight before BA
(factor_graph.py:290-300) — is timed separately and reported under "exchange".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def make_inputs(E, H1, W1, L, radius, seed, device, from_fmaps=True):
    """SURVEY §8(d) canonical synthetic inputs, generated on `device`."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    rd = 2 * radius + 1

    def randn(*s):
        return torch.randn(*s, generator=g, device=device, dtype=torch.float32)

    if from_fmaps:  # volume0 = (fmap1/4)^T (fmap2/4), pyramid by avg_pool2d (corr.py:145-152,83-86)
        f1 = randn(E, 128, H1 * W1) * 0.5 / 4
        f2 = randn(E, 128, H1 * W1) * 0.5 / 4
        v = torch.matmul(f1.transpose(1, 2), f2).view(E * H1 * W1, 1, H1, W1)
        vols = []
        for l in range(L):
            vols.append(v.view(E, H1, W1, H1 >> l, W1 >> l).contiguous())
            v = torch.nn.functional.avg_pool2d(v, 2, stride=2)
        del f1, f2, v
    else:
        vols = [randn(E, H1, W1, H1 >> l, W1 >> l) for l in range(L)]
    ys, xs = torch.meshgrid(torch.arange(H1, device=device, dtype=torch.float32),
                            torch.arange(W1, device=device, dtype=torch.float32), indexing="ij")
    coords = (torch.stack([xs, ys])[None] + 3.0 * randn(E, 2, H1, W1)).contiguous()
    o0 = 4 * torch.tanh(randn(E, H1, W1, rd, rd, 2))
    o1 = (4 * torch.tanh(randn(E, H1, W1, rd, rd, 2)) + o0) / 2
    offs = [o0, o1] + [None] * (L - 2)
    return vols, coords, offs[:L]


def algorithmic_bytes_per_unit(vols, coords, offs, radius, max_edges=4):
    """A = out bytes + coords + offsets actually read + 4*U, with U = unique in-bounds volume
    elements one pixel's taps touch, summed over levels and counted exactly from the inputs
    (SURVEY §8(d)); averaged over the first `max_edges` edges."""
    E, _, H1, W1 = coords.shape
    Es = min(E, max_edges)
    rd = 2 * radius + 1
    dev = coords.device
    d = torch.arange(-radius, radius + 1, device=dev)
    di = d.view(1, 1, 1, rd, 1)
    dj = d.view(1, 1, 1, 1, rd)
    U = torch.zeros(Es, H1, W1, device=dev)
    off_bytes = 0
    for l, v in enumerate(vols):
        H2, W2 = v.shape[3], v.shape[4]
        x0 = (coords[:Es, 0] / 2 ** l).view(Es, H1, W1, 1, 1)
        y0 = (coords[:Es, 1] / 2 ** l).view(Es, H1, W1, 1, 1)
        if offs[l] is not None:
            o = offs[l][:Es].clone()
            o[:, :, :, radius, radius] = 0
            ox, oy = o[..., 0] + x0, o[..., 1] + y0
            off_bytes += rd * rd * 2 * 4
        else:
            ox, oy = x0.expand(Es, H1, W1, rd, rd), y0.expand(Es, H1, W1, rd, rd)
        x1 = torch.floor(ox).long() + di
        y1 = torch.floor(oy).long() + dj
        valid = (x1 >= 0) & (x1 < W2) & (y1 >= 0) & (y1 < H2)
        ids = []
        for ddy in (0, 1):
            for ddx in (0, 1):
                xx, yy = x1 + ddx, y1 + ddy
                ok = valid & (xx < W2) & (yy < H2)
                ids.append(torch.where(ok, yy * W2 + xx, torch.full_like(xx, -1)))
        ids = torch.stack(ids, -1).view(Es, H1, W1, -1)
        ids, _ = torch.sort(ids, dim=-1)
        uniq = (ids[..., 1:] != ids[..., :-1]).sum(-1) + 1  # distinct values incl. possibly -1
        uniq = uniq - (ids[..., 0] < 0).long()
        U += uniq.float()
    Umean = float(U.mean())
    out_bytes = len(vols) * rd * rd * 4
    return out_bytes + 8 + off_bytes + 4.0 * Umean, Umean


def cpu_baseline(E, H1, W1, L, radius, budget_s=12.0):
    """torch-CPU F.grid_sample formulation (oracle/grid_sample_baseline.py) on all host
    cores, same input distribution, bounded sample."""
    from oracle import grid_sample_baseline as G
    # threads actually used: the cores this process may run on, capped at 32 (the torch-CPU
    # gather stops scaling well before that and collapses when oversubscribed on big hosts)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 32))
    torch.set_num_threads(cores)
    Es = min(E, 4)
    vols, coords, offs = make_inputs(Es, H1, W1, L, radius, 999, torch.device("cpu"), from_fmaps=False)
    G.defcorr_pyramid(vols, coords, offs, radius)  # warm-up
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 5000):
        t0 = time.perf_counter()
        G.defcorr_pyramid(vols, coords, offs, radius)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {
        "value": Es * H1 * W1 / med / 1e6, "unit": "Mpix·edges/s", "cores": cores, "kind": "port",
        "sample": "torch-CPU F.grid_sample formulation of the same 4-level pass, E=%d edges of %dx%d, "
                  "median of %d reps (%.1f s of CPU work)" % (Es, H1, W1, len(times), sum(times)),
    }


def dry_run_cpu(args, rank, world):
    """Same control flow as the real run (barrier, timed loop, max over ranks, exchange, one
    JSON line on rank 0) on gloo/CPU with an empty step.  Used by tests/test_host.py only."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="gloo")
    E, H1, W1 = args.edges, 48, 64

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        pass
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    barrier()
    wall = max(time.perf_counter() - t0, 1e-9)
    exchange = None
    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        mine = torch.randn(E, H1, W1, 4)
        allv = torch.empty(world * E, H1, W1, 4)
        dist.all_gather_into_tensor(allv, mine)
        exchange = {"op": "all_gather(target,weight) over gloo (dry run)", "bytes_per_rank": mine.numel() * 4, "ms": 0.0}
    if rank == 0:
        res = {"metric": "def-corr-sample Mpix·edges/s (48×64 fmap, r=3, L=4)", "value": world * E * H1 * W1 / (wall / args.steps) / 1e6,
               "unit": "Mpix·edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic", "config": {"workload": "DRY RUN (no kernel executed)"},
               "roofline": None, "cpu_baseline": None}
        if exchange:
            res["exchange"] = exchange
        emit(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


_REAL_STDOUT = None


def quiet_stdout():
    """Everything that libraries print to stdout while the benchmark runs (RCCL prints a host / library banner on
    communicator creation) goes to stderr; the ONE JSON line is written to the real stdout by emit()."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj, ensure_ascii=False) + "\n").encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        os.write(1, line)
    else:
        os.write(_REAL_STDOUT, line)


def lowmem_main(args, ops, dev, rank, world, use_dist):
    """BASELINE config 4: lowMem_defSample (no stored volume), 640x480 input -> 60x80 feature maps of 128 channels kept
    in half precision as the SLAM system stores them, L=4, r=3, one chunk of `--edges` edges per GPU.  One step = the
    per-level loop of AltCorrBlock.corr_fn (corr.py:192-213) = ONE fused launch (lgu_lowmem_pyramid_fwd_h16)."""
    B, H1, W1, C, L, R = args.edges, 60, 80, 128, 4, 3
    g = torch.Generator(device=dev)
    g.manual_seed(4321 + rank)

    def randn(*s):
        return torch.randn(*s, generator=g, device=dev, dtype=torch.float32)

    f1 = (randn(B, H1, W1, C) * 0.125).half()
    f2s = [(randn(B, H1 >> l, W1 >> l, C) * 0.125).half() for l in range(L)]
    ys, xs = torch.meshgrid(torch.arange(H1, device=dev, dtype=torch.float32),
                            torch.arange(W1, device=dev, dtype=torch.float32), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 3.0 * randn(B, 1, H1, W1, 2)).contiguous()
    o0 = (4 * torch.tanh(randn(B, H1, W1, 7, 7, 2))).contiguous()
    o1 = ((4 * torch.tanh(randn(B, H1, W1, 7, 7, 2)) + o0) / 2).contiguous()
    plan = ops.LowmemPyramidPlan(f1, f2s, [o0, o1, None, None], R)
    out = torch.empty(B, 1, L * 49, H1, W1, device=dev)
    units = B * H1 * W1

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        plan(coords, out=out)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        plan(coords, out=out)
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if use_dist:
        import torch.distributed as dist
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if rank == 0:
        flop_unit = 49 * L * 4 * C * 2  # taps x levels x corners x channels x 2
        kern_s = dev_ms * 1e-3 / args.steps
        achieved = flop_unit * units / kern_s / 1e12
        cpu = None
        if not (args.no_cpu or world > 1):
            from oracle import oracle as O
            Bs = 1
            a1, a2 = f1[:Bs].float().cpu().numpy(), [f[:Bs].float().cpu().numpy() for f in f2s]
            cc = coords[:Bs].cpu().numpy()
            oo = [o0[:Bs].cpu().numpy(), o1[:Bs].cpu().numpy(), np.zeros((Bs, H1, W1, 7, 7, 2), np.float32), np.zeros((Bs, H1, W1, 7, 7, 2), np.float32)]
            times = []
            t_start = time.perf_counter()
            while len(times) < 2 or (time.perf_counter() - t_start < 10.0 and len(times) < 50):
                t1 = time.perf_counter()
                for l in range(L):
                    O.lowMem_defSample(a1, a2[l], (cc / 2 ** l).astype(np.float32), oo[l].copy(), R)
                times.append(time.perf_counter() - t1)
            cpu = {"value": Bs * H1 * W1 / float(np.median(times)) / 1e6, "unit": "Mpix·edges/s", "cores": 1, "kind": "port",
                   "sample": "C restatement of lowMem_defSample (oracle/lgu_oracle.c), 4 levels, %d edge of %dx%dx%d, median of %d reps "
                             "(%.1f s of CPU work)" % (Bs, H1, W1, C, len(times), sum(times))}
        res = {"metric": "def-corr-sample Mpix·edges/s (60×80 fmap, on-the-fly correlation, r=3, L=4)",
               "value": world * units / (wall / args.steps) / 1e6, "unit": "Mpix·edges/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f16 features, f32 accumulate", "data": "synthetic",
               "config": {"workload": "BASELINE config 4: lowMem_defSample, 60x80x128 half feature maps, L=4, r=3, one chunk of "
                                      "%d edges per GPU, all levels in one launch" % B,
                          "edges_per_gpu": B, "units_per_step_per_gpu": units, "sharding": "edges (no data-path collective)"},
               "roofline": {"bound": "mfma", "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0,
                            "traffic": None, "algorithmic_flop_per_unit": flop_unit, "kernel": "lgu::lowmem_mfma_kernel<3,4>",
                            "device_ms_per_step": dev_ms / args.steps,
                            "note": "the contraction is 2.2 % of the dense f16 MFMA peak by design: the kernel is bound by L2 -> CU "
                                    "reads of the swept windows (DESIGN.md §3.4), the matrix cores are idle most of the time"},
               "cpu_baseline": cpu}
        emit(res)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--edges", type=int, default=20, help="edges per GPU (BASELINE config 2: 20)")
    ap.add_argument("--probe", action="store_true", help="also time the level-1 uncertainty probe in the step")
    ap.add_argument("--variant", type=int, default=0, help="LGU_DEFCORR_VARIANT (A/B only)")
    ap.add_argument("--workload", choices=["defcorr", "lowmem"], default="defcorr",
                    help="'defcorr' = BASELINE config 2 (the headline metric, stored-volume path); 'lowmem' = BASELINE config 4 "
                         "(on-the-fly correlation from half feature maps at 60x80, the backend's path), reported in the same units")
    ap.add_argument("--layout", choices=["tiled", "rowmajor"], default="tiled",
                    help="storage of the pyramid the sampler reads: 'tiled' = the 4x8-tile slice layout CorrBlock keeps "
                         "its pyramid in (production), 'rowmajor' = the reference operator's layout (drop-in operator path)")
    ap.add_argument("--out-format", choices=["planar", "nhwc", "nhwc_f16"], default="planar",
                    help="output tensor: 'planar' = the reference's contiguous (E,196,H,W) fp32; 'nhwc' / 'nhwc_f16' = the "
                         "same values channel-last in fp32 / half, the form the consumer 1x1 convolution takes (tiled layout only)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--randn-volumes", action="store_true", help="N(0,1) volumes instead of fmap products")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="TEST ONLY: exercise the launch / process-group / timing / JSON logic with gloo on CPU and an "
                         "empty step (no kernel runs, the printed value is meaningless)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    quiet_stdout()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)  # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)  # RCCL
    if args.variant:
        os.environ["LGU_DEFCORR_VARIANT"] = str(args.variant)

    import lgu_slam_amd
    lgu_slam_amd._lib.load()
    ops = lgu_slam_amd.ops
    if args.workload == "lowmem":
        return lowmem_main(args, ops, dev, rank, world, use_dist)

    E, H1, W1, L, R = args.edges, 48, 64, 4, 3
    vols, coords, offs = make_inputs(E, H1, W1, L, R, 1234 + rank, dev, from_fmaps=not args.randn_volumes)
    out = torch.empty(E, L * 49, H1, W1, device=dev)
    units = E * H1 * W1

    # prepared launch: pointer tables built once, one ctypes call per step (the kernel is
    # ~60 us; per-call Python argument handling would otherwise bound the loop)
    tiled = args.layout == "tiled"
    level_hw = [(H1 >> l, W1 >> l) for l in range(L)]
    if tiled:  # what CorrBlock.__init__ (ops.volume_pyramid(tiled=True)) leaves in HBM; conversion is setup, not timed
        rowmajor_vols = vols
        vols = [ops.volume_retile(v) for v in rowmajor_vols]
    plan = ops.DefcorrPyramidPlan(vols, offs, R, probe=args.probe, tiled=tiled, level_hw=level_hw,
                                  out_format=args.out_format)
    if args.out_format != "planar":
        out = ops._pyr_out(args.out_format, E, L * (2 * R + 1) ** 2, H1, W1, dev, None)

    def step():
        # --probe: the level-1 uncertainty probe, variance, sigmoid and the stateful
        # offset[1] *= mask of corr.py:94-99 run inside the same launch
        plan(coords, out=out)

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()  # kernels are enqueued on torch's current stream, which these events time
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if use_dist:
        import torch.distributed as dist
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    ms_per_step = wall * 1e3 / args.steps
    value = world * units / (wall / args.steps) / 1e6

    exchange = None
    if use_dist:  # the sharded driver's per-BA-step all-gather of target+weight (E,ht,wd,2)x2
        import torch.distributed as dist
        mine = torch.randn(E, H1, W1, 4, device=dev)
        allv = torch.empty(world * E, H1, W1, 4, device=dev)
        for _ in range(5):
            dist.all_gather_into_tensor(allv, mine)
        barrier()
        t1 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            dist.all_gather_into_tensor(allv, mine)
        barrier()
        ag = (time.perf_counter() - t1) / reps
        t = torch.tensor([ag], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exchange = {"op": "all_gather(target,weight) over RCCL", "bytes_per_rank": mine.numel() * 4,
                    "ms": float(t.item()) * 1e3}

    if rank == 0:
        A, U = algorithmic_bytes_per_unit(rowmajor_vols if tiled else vols, coords, offs, R)
        if args.out_format == "nhwc_f16":  # the output row is 2-byte elements
            A -= L * (2 * R + 1) ** 2 * 2
        kname = {0: "lgu::defcorr_gather_kernel<3,%s,12,2,16,LAYOUT>", 4: "lgu::defcorr_gather_kernel<3,%s,12,2,32,LAYOUT>",
                 5: "lgu::defcorr_gather_kernel<3,%s,12,2,16,LAYOUT>",
                 3: "lgu::defcorr_gather_kernel<3,%s,12,4,16,LAYOUT>", 1: "lgu::defcorr_pyr_kernel<3,%s,12>",
                 2: "lgu::defcorr_generic_kernel%s"}.get(args.variant, "?%s") % (("true" if args.probe else "false") if args.variant != 2 else "")
        kname = kname.replace("LAYOUT", "true" if tiled else "false")
        if "gather_kernel" in kname:  # trailing template argument = output form; channel-last forms run 8-pixel tiles
            om = {"planar": 0, "nhwc": 1, "nhwc_f16": 2}[args.out_format]
            kname = (kname.replace(",2,16,", ",2,8,") if om else kname)[:-1] + ",%d>" % om
        # HBM bytes per launch from the PMC passes of this same command (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE, separate runs, gfx950 x2 fetch correction): measured offline
        # because counters cannot be collected from inside the timed process; see profiles/.
        traffic, traffic_src = None, None
        tname = "traffic_r01_tiled.json" if tiled else "traffic_r01.json"
        tfile = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tfile) and E == 20 and not args.probe and args.out_format == "planar":
            t = json.load(open(tfile))
            if t.get("kernel") == kname:
                traffic, traffic_src = t["hbm_bytes_per_launch"], "profiles/" + tname
        kern_s = dev_ms * 1e-3 / args.steps  # average launch-to-launch device time of the step
        achieved = A * units / kern_s / 1e9
        res = {
            "metric": "def-corr-sample Mpix·edges/s (48×64 fmap, r=3, L=4)",
            "value": value, "unit": "Mpix·edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: TartanAir-mono shape, 48x64 fmap, L=4, r=3, "
                                   "%d edges per GPU, fused 4-level deformable sample%s" % (E, " + level-1 probe" if args.probe else ""),
                       "edges_per_gpu": E, "units_per_step_per_gpu": units, "sharding": "edges (no data-path collective)",
                       "variant": args.variant, "volumes": "N(0,1)" if args.randn_volumes else "fmap products + avg_pool pyramid",
                       "pyramid_layout": "4x8-tiled slices (CorrBlock's own storage; results bit-identical)" if tiled
                                         else "row-major slices (reference operator layout)",
                       "out_format": args.out_format},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch (PMC)",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": A * units,
                         "algorithmic_bytes_per_unit": A, "unique_volume_elements_per_unit": U,
                         "kernel": kname, "device_ms_per_step": dev_ms / args.steps},
            "cpu_baseline": None if (args.no_cpu or world > 1) else cpu_baseline(E, H1, W1, L, R),  # rank 0, N=1 only
        }
        if exchange:
            res["exchange"] = exchange
        emit(res)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
