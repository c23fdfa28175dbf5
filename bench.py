#!/usr/bin/env python3
"""bench.py — def-corr-sample throughput (Mpix·edges/s) of the fused deformable pyramid
sample on MI355X, with its HBM roofline and the CPU baseline timed beside it.

One STEP = one pass of the hot path (body of CorrBlock.__call__, reference
droid_slam/modules/corr.py:101-109: 4 pyramid levels x 49 taps per pixel) over one batch
of synthetic edges: BASELINE config 2 — 48x64 fmap, L=4, r=3, E=20 edges per GPU — ONE
kernel launch.  With --probe the level-1 uncertainty probe of corr.py:94-99 is timed in
the step as well.  Inputs are resident in HBM before the timed region.

Cache mode (--cache, default cold): the lines one launch touches (~243 MiB) would fit the
256 MiB Infinity Cache if every step replayed the same inputs, so the headline rotates over
--sets (4) disjoint input sets: every launch is HBM-served, as a lookup inside the SLAM loop
is.  `extra.warm_cache` holds the replay figure next to it.

`extra` (N=1): the production call (probe fused in), the reference-layout operator path,
BASELINE config 3 (E=40) and config 4 (lowmem) — each the median of --blocks blocks of
--steps launches, device time by HIP events on the launch stream.

Multi-GPU (--gpus N): one process per GPU over RCCL.  Under torch.distributed.run the ranks
are the launcher's; from a plain `python bench.py --gpus N` this process — before it touches
any GPU — starts the N ranks itself (torch.distributed.run as a child process), relays rank
0's one JSON line and exits non-zero if any rank failed.  `n_gpus` is the number of ranks the
process group actually initialised.  Factor-graph edges are independent, so for the headline
every rank samples its own E edges (weak scaling, no data-path collective); `value` = all
ranks' units / max-over-ranks time.  The multi-GPU DESIGN — BASELINE config 5: one global-BA
iteration over a fixed ~2000-edge graph, source-frame chunks dealt to the ranks, all-gathers
of target / weight / damping, replicated BA: STRONG scaling — is measured in the same run
and reported under "strong_scaling_config5" (N > 1) / extra.config5_backend_n1 (N = 1).
The per-BA-step all-gather alone is also timed ("exchange").

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
    n_ranks = dist.get_world_size() if world > 1 else 1
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
    strong = None
    if not args.no_backend:
        # the config-5 leg's control flow: fixed edge list, source-frame chunks dealt to the ranks, exchanged, agreement checked
        import lgu_slam_amd
        sh = lgu_slam_amd.sharded
        ii = torch.arange(64).repeat_interleave(2)
        edges = sh.ShardedEdgeSet(ii, rank=rank, world=world)
        target = torch.zeros(ii.numel(), 2)
        local = torch.full((edges.counts[rank], 2), float(rank + 1))
        edges.gather(local, out=target)
        agree = sh.replicas_agree(target) if world > 1 else True
        strong = {"workload": "DRY RUN of the config-5 control flow (no kernel executed)", "n_gpus": n_ranks,
                  "edges_total": int(ii.numel()), "edges_per_rank": edges.counts, "replicas_agree": agree,
                  "every_edge_owned": bool((target != 0).all())}
    if rank == 0:
        res = {"metric": "def-corr-sample Mpix·edges/s (48×64 fmap, r=3, L=4)", "value": n_ranks * E * H1 * W1 / (wall / args.steps) / 1e6,
               "unit": "Mpix·edges/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic", "config": {"workload": "DRY RUN (no kernel executed)"},
               "roofline": None, "cpu_baseline": None}
        if exchange:
            res["exchange"] = exchange
        if strong is not None:
            res["strong_scaling_config5" if n_ranks > 1 else "config5_backend_n1"] = strong
        emit(res)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def launch_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1) outside any launcher: start the N ranks as CHILD processes, one per GPU,
    through torch.distributed.run on 127.0.0.1 with a free port; relay rank 0's JSON line; exit with the launcher's
    status.  This parent makes no GPU / HIP call before or after (a process that has touched the GPU must not be
    replaced or forked into ranks), and nothing is re-exec'd."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)   # stderr passes through
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write(proc.stdout.decode(errors="replace"))
        sys.stderr.write("bench: the %d-rank launch failed (launcher exit code %d)\n" % (args.gpus, proc.returncode))
        sys.exit(proc.returncode or 1)
    try:
        got = json.loads(lines[-1]).get("n_gpus")
    except ValueError:
        got = None
    if got != args.gpus:
        sys.stderr.write("bench: asked for %d ranks, the line reports n_gpus = %r\n" % (args.gpus, got))
        sys.exit(1)
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()
    sys.exit(0)


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


def _events():
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


NO_GRAPH = False  # --no-graph


def capture_steps(step, first, count):
    """`count` consecutive steps (launch indices first .. first + count - 1) captured into ONE HIP graph, or None when
    capture is not possible.  The timed region then replays the graph: K launches back to back with no host code between
    them, so `value` does not depend on how fast this box's host issues a ~50 us launch (the kernels, their order and
    their inputs are exactly those of the direct loop)."""
    try:
        g = torch.cuda.CUDAGraph()
        # thread_local: calls of other threads (RCCL's watchdog polling its events under torchrun) must neither fail nor
        # invalidate this capture
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for i in range(count):
                step(first + i)
        g.replay()  # untimed: the first launch of a graph uploads it
        torch.cuda.synchronize()
        return g
    except Exception as exc:  # pragma: no cover - depends on the runtime
        sys.stderr.write("bench: HIP graph capture failed (%s); timing direct launches\n" % exc)
        torch.cuda.synchronize()
        return None


def time_blocks(step, steps, blocks, pre_block=None, use_graph=True):
    """Device time per step (ms) of `blocks` blocks of `steps` back-to-back launches, HIP events on torch's current
    stream (the stream every launch goes to).  step(i) is given the global launch index.  Returns the list of per-block
    averages; callers report the median.  Without per-block state to restore (pre_block) the block is captured once into
    a HIP graph and replayed, so that the events bracket device work only, not this host's launch rate."""
    graph = capture_steps(step, 0, steps) if (use_graph and pre_block is None and not NO_GRAPH) else None
    res, n = [], 0
    for _ in range(blocks):
        if pre_block is not None:
            pre_block()
        e0, e1 = _events()
        e0.record()
        if graph is not None:
            graph.replay()
        else:
            for _ in range(steps):
                step(n)
                n += 1
        e1.record()
        e1.synchronize()
        res.append(e0.elapsed_time(e1) / steps)
    return res


class DefcorrSets:
    """`nsets` independent input sets (volumes, coords, offsets) of E edges each and one prepared launch per set.
    Launch i uses set i % nsets: with nsets >= 3 two uses of the same line are separated by the other sets' launches
    (nsets - 1) x ~255 MB of touched lines, more than the 256 MiB Infinity Cache holds (MI355X_MICROARCH.md, Infinity
    Cache: a table stays resident only while table + everything touched in between fits) — every launch is HBM-served,
    as a lookup in the SLAM loop is (GRU / convolution / BA kernels run between lookups, factor_graph.py:204-254).
    nsets = 1 is the warm mode: every launch re-touches the same lines."""

    def __init__(self, ops, E, nsets, seed, dev, randn_volumes=False, keep_rowmajor=False):
        self.E, self.H1, self.W1, self.L, self.R = E, 48, 64, 4, 3
        self.ops, self.dev, self.nsets = ops, dev, nsets
        self.level_hw = [(self.H1 >> l, self.W1 >> l) for l in range(self.L)]
        self.rowmajor, self.tiled, self.coords, self.offs, self.offs1_saved = [], [], [], [], []
        for s in range(nsets):
            vols, coords, offs = make_inputs(E, self.H1, self.W1, self.L, self.R, seed + 7919 * s, dev, from_fmaps=not randn_volumes)
            self.tiled.append([ops.volume_retile(v) for v in vols])
            # set 0's row-major form is what algorithmic_bytes_per_unit counts from; the others only if asked for
            self.rowmajor.append(vols if (keep_rowmajor or s == 0) else None)
            self.coords.append(coords)
            self.offs.append(offs)
            self.offs1_saved.append(offs[1].clone())
        self.out = torch.empty(E, self.L * 49, self.H1, self.W1, device=dev)
        self.units = E * self.H1 * self.W1

    def plans(self, layout="tiled", probe=False, out_format="planar"):
        tiled = layout == "tiled"
        ps = []
        for s in range(self.nsets):
            ps.append(self.ops.DefcorrPyramidPlan(self.tiled[s] if tiled else self.rowmajor[s], self.offs[s], self.R, probe=probe,
                                                  tiled=tiled, level_hw=self.level_hw if tiled else None, out_format=out_format))
        return ps

    def restore_offsets(self, only=None):
        """The fused probe scales offset[1] in place on every call (corr.py:99, persistent); a CorrBlock lives for 8-16
        lookups, so timed blocks of probe-on launches restart from the original offsets."""
        for s in (range(self.nsets) if only is None else (only,)):
            self.offs[s][1].copy_(self.offs1_saved[s])

    def stepper(self, plans, out, cold=True, probe=False):
        """probe=True: a set's offsets are restored before every 16th use of that set (a 24 MB copy per 16 lookups, as a
        new CorrBlock would bring fresh offsets), so that a long run does not drive offset[1] to zero."""
        n = self.nsets if cold else 1

        def step(i):
            s = i % n
            if probe and (i // n) % 16 == 0:
                self.restore_offsets(only=s)
            plans[s](self.coords[s], out=out)
        return step


def lowmem_setup(ops, dev, B, seed):
    H1, W1, C, L, R = 60, 80, 128, 4, 3
    g = torch.Generator(device=dev)
    g.manual_seed(seed)

    def randn(*s):
        return torch.randn(*s, generator=g, device=dev, dtype=torch.float32)

    f1 = (randn(B, H1, W1, C) * 0.125).half()
    f2s = [(randn(B, H1 >> l, W1 >> l, C) * 0.125).half() for l in range(L)]
    ys, xs = torch.meshgrid(torch.arange(H1, device=dev, dtype=torch.float32),
                            torch.arange(W1, device=dev, dtype=torch.float32), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 3.0 * randn(B, 1, H1, W1, 2)).contiguous()
    o0 = (4 * torch.tanh(randn(B, H1, W1, 7, 7, 2))).contiguous()
    o1 = ((4 * torch.tanh(randn(B, H1, W1, 7, 7, 2)) + o0) / 2).contiguous()
    # the target maps in the chunk-planar form AltCorrBlock keeps them in (ops.lowmem_chunked; conversion is setup)
    plan = ops.LowmemPyramidPlan(f1, [ops.lowmem_chunked(f) for f in f2s], [o0, o1, None, None], R, chunked=True)
    out = torch.empty(B, 1, L * 49, H1, W1, device=dev)
    return dict(B=B, H1=H1, W1=W1, C=C, L=L, R=R, f1=f1, f2s=f2s, coords=coords, o0=o0, o1=o1, plan=plan, out=out,
                units=B * H1 * W1)


def lowmem_roofline(S, dev_ms):
    """Roofline record of the low-memory launch: the contraction against the dense f16 MFMA peak, and the compulsory HBM
    bytes (feature maps once + offsets + coords + out) against the HBM peak."""
    flop_unit = 49 * S["L"] * 4 * S["C"] * 2  # taps x levels x corners x channels x 2
    kern_s = dev_ms * 1e-3
    achieved = flop_unit * S["units"] / kern_s / 1e12
    hbm_unit = (1 + 1.328) * S["C"] * 2 + 784 + 784 + 8  # half maps: fmap1 + fmap2 pyramid, out, offsets of 2 levels, coords
    traffic, tsrc = None, None
    for tname in ("traffic_r03_lowmem.json", "traffic_r02_lowmem.json"):   # PMC passes of this same command (tools/gpu_lowmem_run.sh)
        tfile = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tfile) and S["B"] == 16:
            traffic, tsrc = json.load(open(tfile))["hbm_bytes_per_launch"], "profiles/" + tname
            break
    return {"bound": "mfma", "achieved": achieved, "peak": 2500.0, "unit": "TFLOP/s", "frac": achieved / 2500.0,
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)", "traffic_source": tsrc,
            "algorithmic_flop_per_unit": flop_unit, "kernel": "lgu::lowmem_coop_kernel (csrc/lowmem_coop.hip)",
            "device_ms_per_step": dev_ms,
            "hbm_compulsory_bytes_per_unit": hbm_unit, "hbm_compulsory_GBps": hbm_unit * S["units"] / kern_s / 1e9,
            "note": "on-the-fly correlation is a contraction over C=128 followed by a 49-tap bilinear sample per level; the "
                    "launch is bound by the per-pixel box / sweep-control / sampling / write-out code around the contraction, "
                    "not by a data path or the MFMA rate (DESIGN.md §7.2: ablating loads, LDS traffic and MFMAs together removes "
                    "16 %; a build with 4 waves per SIMD is no faster; one profile's instruction counters — ~56 M VALU/SALU/LDS/VMEM + "
                    "2 M MFMA wave-instructions per 16-edge launch, profiles/r02_pmc_lowmem_kernels.txt — summed at one instruction per "
                    "SIMD per 4-cycle slot would roughly fill the launch; that sum over-counts (dual issue) and is a reading aid, not a "
                    "measured fraction); the matrix cores are busy ~12 % of the time (PMC, profiles/)"}


def lowmem_cpu_baseline(S):
    from oracle import oracle as O
    Bs, H1, W1, C, L, R = 1, S["H1"], S["W1"], S["C"], S["L"], S["R"]
    a1, a2 = S["f1"][:Bs].float().cpu().numpy(), [f[:Bs].float().cpu().numpy() for f in S["f2s"]]
    cc = S["coords"][:Bs].cpu().numpy()
    oo = [S["o0"][:Bs].cpu().numpy(), S["o1"][:Bs].cpu().numpy(), np.zeros((Bs, H1, W1, 7, 7, 2), np.float32),
          np.zeros((Bs, H1, W1, 7, 7, 2), np.float32)]
    times = []
    t_start = time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_start < 10.0 and len(times) < 50):
        t1 = time.perf_counter()
        for l in range(L):
            O.lowMem_defSample(a1, a2[l], (cc / 2 ** l).astype(np.float32), oo[l].copy(), R)
        times.append(time.perf_counter() - t1)
    return {"value": Bs * H1 * W1 / float(np.median(times)) / 1e6, "unit": "Mpix·edges/s", "cores": 1, "kind": "port",
            "sample": "C restatement of lowMem_defSample (oracle/lgu_oracle.c), 4 levels, %d edge of %dx%dx%d, median of %d reps "
                      "(%.1f s of CPU work)" % (Bs, H1, W1, C, len(times), sum(times))}


def lowmem_main(args, ops, dev, rank, world, use_dist):
    """BASELINE config 4: lowMem_defSample (no stored volume), 640x480 input -> 60x80 feature maps of 128 channels kept
    in half precision as the SLAM system stores them, L=4, r=3, one chunk of `--edges` edges per GPU.  One step = the
    per-level loop of AltCorrBlock.corr_fn (corr.py:192-213) = ONE fused launch (lgu_lowmem_pyramid_fwd_h16)."""
    S = lowmem_setup(ops, dev, args.edges, 4321 + rank)
    plan, coords, out, units = S["plan"], S["coords"], S["out"], S["units"]
    n_ranks = 1
    if use_dist:
        import torch.distributed as dist
        n_ranks = dist.get_world_size()

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        plan(coords, out=out)
    graph = None if args.no_graph else capture_steps(lambda i: plan(coords, out=out), 0, args.steps)
    barrier()
    ev0, ev1 = _events()
    t0 = time.perf_counter()
    ev0.record()
    if graph is not None:
        graph.replay()
    else:
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
        blocks = time_blocks(lambda i: plan(coords, out=out), args.steps, args.blocks)
        roof = lowmem_roofline(S, float(np.median(blocks)))
        roof["device_ms_per_step_blocks"] = blocks
        res = {"metric": "def-corr-sample Mpix·edges/s (60×80 fmap, on-the-fly correlation, r=3, L=4)",
               "value": n_ranks * units / (wall / args.steps) / 1e6, "unit": "Mpix·edges/s", "n_gpus": n_ranks, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f16 features, f32 accumulate", "data": "synthetic",
               "config": {"workload": "BASELINE config 4: lowMem_defSample, 60x80x128 half feature maps, L=4, r=3, one chunk of "
                                      "%d edges per GPU, all levels in ONE launch (4 x 8 pixel tiles, four waves share the swept window), "
                                      "target maps chunk-planar as AltCorrBlock stores them" % S["B"],
                          "edges_per_gpu": S["B"], "units_per_step_per_gpu": units, "sharding": "edges (no data-path collective)"},
               "roofline": roof,
               "cpu_baseline": None if (args.no_cpu or world > 1) else lowmem_cpu_baseline(S)}
        emit(res)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def backend_main(args, lgu, dev, rank, world, use_dist):
    """--workload backend: the config-5 step as the whole run's line."""
    res = backend_run(args, lgu, dev, rank, world, use_dist, args.steps, args.warmup)
    if rank == 0:
        emit(res)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def backend_run(args, lgu, dev, rank, world, use_dist, steps, warmup):
    """BASELINE config 5: one global-BA iteration of update_lowmem (reference factor_graph.py:256-302) over a ~2000-edge
    factor graph (200 keyframes of 60x80x128 half feature maps, edges between frames at most 5 apart = 1970 edges),
    STRONG scaling: the graph is fixed and its source-frame chunks are dealt round-robin to the ranks.  One step =
      lookups   this rank's chunks through AltCorrBlock (offset heads + fused low-memory lookup per chunk) and a
                stand-in for the update operator (out of scope: target = coords + the first sample channels,
                weight = sigmoid of the next ones, damping = per-source-frame mean)
      exchange  all-gather of target / weight by edge and of damping by source frame (RCCL)
      ba        dense bundle adjustment, 2 iterations, replicated on every rank (parity unpinned, experimental)
    `value` = pixel·edges of the WHOLE graph per second of step time (max over ranks).
    Every rank calls this (it issues collectives); returns the record on rank 0, None elsewhere."""
    import torch.distributed as dist
    sh = lgu.sharded
    n_ranks = dist.get_world_size() if use_dist else 1
    N, H, W, C, span = 200, 60, 80, 128, 5
    g = torch.Generator(device=dev)
    g.manual_seed(777)  # the same graph and state on every rank (replicated, as the SLAM system holds them)

    def randn(*s):
        return torch.randn(*s, generator=g, device=dev, dtype=torch.float32)

    ii_l = [i for i in range(N) for j in range(N) if i != j and abs(i - j) <= span]
    jj_l = [j for i in range(N) for j in range(N) if i != j and abs(i - j) <= span]
    ii, jj = torch.tensor(ii_l, device=dev), torch.tensor(jj_l, device=dev)
    E = ii.numel()
    fmaps = (randn(1, N, C, H, W) * 0.5).half()
    ofsMap = torch.nn.Conv2d(2 * C, 98, 3, padding=1).to(dev)
    ofsRes = torch.nn.Conv2d(2 * C, 98, 3, padding=1).to(dev)
    with torch.no_grad():
        for m in (ofsMap, ofsRes):
            m.weight.copy_(randn(*m.weight.shape) * 0.02)
            m.bias.copy_(randn(*m.bias.shape) * 0.02)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32),
                            torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
    grid = torch.stack([xs, ys], -1)
    coords1 = (grid[None, None] + 2.0 * randn(1, E, H, W, 2)).contiguous()
    sac = sh.ShardedAltCorr(ofsMap, ofsRes, None, fmaps, ii, jj, rig=1, rank=rank, world=world)
    edges = sac.edges
    target = coords1[0].clone()
    weight = torch.zeros(E, H, W, 2, device=dev)
    damping = torch.full((N, H, W), 1e-3, device=dev)
    poses0 = torch.zeros(N, 7, device=dev)
    poses0[:, 6] = 1
    poses0[:, 0] = torch.arange(N, device=dev) * 0.05
    disps0 = 0.3 + 0.7 * torch.rand(N, H, W, generator=g, device=dev)
    intr = torch.tensor([60.0, 60.0, 40.0, 30.0], device=dev)
    sens = torch.zeros_like(disps0)
    poses, disps = poses0.clone(), disps0.clone()

    # per chunk: its edges' positions among its source frames and the frames' edge counts (graph bookkeeping, once)
    chunk_tab = {}
    for idx in edges.my_chunks:
        fr, inv = torch.unique(ii[idx], return_inverse=True)
        chunk_tab[id(idx)] = (inv, fr.numel(), torch.bincount(inv, minlength=fr.numel()).view(-1, 1, 1).float())  # idx: the set's own tensor
    one_launch = not args.chunk_loop

    def chunk_fn(idx, iis, corr=None):
        with torch.no_grad():
            if corr is None:   # the reference's loop: one corr_fn call per chunk (factor_graph.py:272-279)
                corr = sac.block(coords1[:, idx], iis, jjs_of(idx))
            corr = corr[0]                                                    # (n,196,H,W)
            t = coords1[0, idx] + 0.05 * corr[:, 0:2].permute(0, 2, 3, 1)
            w = torch.sigmoid(corr[:, 2:4].permute(0, 2, 3, 1))
            inv, nf, cnt = chunk_tab[id(idx)]
            d = torch.zeros(nf, H, W, device=dev).index_add_(0, inv, torch.sigmoid(corr[:, 4])) / cnt
        return t, w, d

    def jjs_of(idx):
        iis, jjs = ii[idx], jj[idx]
        return jjs + (iis == jjs).long()

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    phase = {"corr_block": [], "lookups": [], "update_standin": [], "exchange": [], "ba": []}
    # The reference builds ONE AltCorrBlock per update_lowmem call and runs `steps = 8` iterations over it
    # (factor_graph.py:256-265).  Per-call state — the feature pyramid, its chunk-planar / pooled forms and the per-frame
    # partial convolutions of the offset heads (ops.OffsetHeadCache) — is therefore rebuilt every CALL_STEPS-th iteration,
    # INSIDE the timed region: a step carries an eighth of it on average (phase "corr_block" = mean over the timed steps).
    CALL_STEPS = 8
    it_no = [0]
    step_wall = []

    def step(record):
        ev[4].record()
        # (every warm-up step builds a block, so that the allocator has seen a new block being built while the old one is
        # still alive — the first such construction otherwise falls on the first timed step and, on some boxes, stalled it
        # for ~120 ms in fresh device allocations: 24.9 instead of 17.5 ms per step over 16 steps)
        if it_no[0] % CALL_STEPS == 0 or not record:
            sac.block = lgu.AltCorrBlock(ofsMap, ofsRes, None, fmaps)
        it_no[0] += 1
        ev[0].record()
        if one_launch:   # all of this rank's chunks in one lookup launch (ShardedAltCorr.lookup_all), then the update stand-in per chunk
            with torch.no_grad():
                _, corr_all, _ = sac.lookup_all(coords1)
            ev[5].record()   # the lookups proper end here; what follows up to ev[1] is the stand-in for the update operator
            local = sh.run_chunks(edges, ii, chunk_fn, corr_all=corr_all)
        else:
            local = sh.run_chunks(edges, ii, chunk_fn)   # (lookup and stand-in interleaved per chunk: one phase)
            ev[5].record()
        ev[1].record()
        sh.exchange_step(edges, local, target, weight, damping)
        ev[2].record()
        eta = (0.2 * damping[torch.unique(ii)] + 1e-7).contiguous()          # factor_graph.py:294
        tg = target.permute(0, 3, 1, 2).contiguous()                          # :295-296
        wg = weight.permute(0, 3, 1, 2).contiguous()
        poses.copy_(poses0); disps.copy_(disps0)
        if args.ba_split:   # owners build / eliminate their edges and depth frames, the system is all-reduced, the solve replicated
            tl = torch.cat(local[0], 0).permute(0, 3, 1, 2).contiguous() if local[0] else tg[:0]
            wl = torch.cat(local[1], 0).permute(0, 3, 1, 2).contiguous() if local[1] else wg[:0]
            sh.sharded_ba_split(edges, tl, wl, poses, disps, intr, sens, 0.2 * damping + 1e-7, ii, jj, 1, N, 2, 1e-5, 1e-2, False)
        else:
            lgu.ba.ba(poses, disps, intr, sens, tg, wg, eta, ii, jj, 1, N, 2, 1e-5, 1e-2, False)   # :299-300
        ev[3].record()
        if record:
            ev[3].synchronize()
            phase["lookups"].append(ev[0].elapsed_time(ev[5]))
            phase["update_standin"].append(ev[5].elapsed_time(ev[1]))
            for k, name in ((1, "exchange"), (2, "ba")):
                phase[name].append(ev[k].elapsed_time(ev[k + 1]))
            phase["corr_block"].append(ev[4].elapsed_time(ev[0]))
            step_wall.append(time.perf_counter())

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step(False)
    it_no[0] = 0   # the timed region starts with a call's first iteration
    import gc
    gc.collect()
    gc.freeze()    # the graph tables, plans and modules built so far are permanent: the collector's full passes (tens of ms over the
                   # objects of ~2000 tensors per step in --chunk-loop mode) no longer rescan them inside the timed region
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    barrier()
    wall = time.perf_counter() - t0
    gc.unfreeze()
    agree = sh.replicas_agree(poses, disps, target, weight, damping) if use_dist and (world > 1 or sh.REHEARSE_COLLECTIVES) else True
    if use_dist:
        t = torch.tensor([wall] + [float(np.median(phase[k])) for k in ("lookups", "exchange", "ba", "update_standin")] +
                         [float(np.mean(phase["corr_block"]))], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ph = float(t[0]), [float(x) for x in t[1:]]
    else:
        ph = [float(np.median(phase[k])) for k in ("lookups", "exchange", "ba", "update_standin")] + [float(np.mean(phase["corr_block"]))]
    if rank == 0:
        units = E * H * W
        return {"metric": "def-corr-sample Mpix·edges/s (60×80 fmap, on-the-fly correlation, r=3, L=4; global BA step)",
              "value": units / (wall / steps) / 1e6, "unit": "Mpix·edges/s", "n_gpus": n_ranks, "steps": steps,
              "warmup": warmup, "ms_per_step": wall * 1e3 / steps, "higher_is_better": True, "scaling": "strong",
              "vs_baseline": None, "dtype": "f16 features, f32 accumulate; f32/f64 BA", "data": "synthetic",
              "config": {"workload": "BASELINE config 5: one update_lowmem iteration over a %d-edge graph (%d keyframes, 60x80x128 half "
                                     "features), source-frame chunks of 8 dealt round-robin to %d rank(s); lookups + all-gathers + "
                                     "replicated dense BA (2 iterations)" % (E, N, world),
                         "edges_total": E, "edges_this_rank": int(edges.counts[rank]), "chunks_this_rank": len(edges.my_chunks),
                         "units_per_step": units, "update_operator": "stand-in (out of scope)",
                         "lookups": "one launch for all of the rank's chunks (AltCorrBlock.call_many: every chunk's results bit-identical "
                                    "to its own call)" if one_launch else "one call per chunk (--chunk-loop: the reference's loop)",
                         "ba": "lgu_slam_amd.ba (experimental: parity unpinned); " + ("per-edge work on the owners, all-reduced system, replicated solve "
                                "(--ba-split)" if args.ba_split else "replicated on every rank")},
              "phases_ms_max_over_ranks": {"lookups": ph[0], "update_standin": ph[3], "exchange": ph[1], "ba": ph[2],
                                           "corr_block_per_step": ph[4]},
              "phases_note": "lookups = ShardedAltCorr.lookup_all (probe, offset heads of the calls' first edges, the lookup launch); "
                             "update_standin = the per-chunk stand-in for the update operator (~10 small torch ops per chunk: target, weight, "
                             "damping from the lookups; out of scope, there to feed the exchange) — with --chunk-loop the two are interleaved per "
                             "chunk and reported together under lookups; "
                             "lookups / update_standin / exchange / ba: medians over the timed steps; corr_block_per_step: MEAN over the timed steps of "
                             "the AltCorrBlock construction (pyramid) that every %d-th iteration starts with, as one update_lowmem "
                             "call = one block + 8 iterations in the reference; the first iteration over a new block also pays the "
                             "per-frame partial convolutions of the offset heads inside its lookups; `ms_per_step` / `value` are the wall "
                             "clock over the K timed steps as the contract asks — a Python-driven step of ~1000 launches and multi-GB "
                             "allocations occasionally stalls on the host (one step of 16 at 30-140 ms on some boxes): "
                             "`step_wall_ms_this_rank` shows every step, `ms_per_step_median_this_rank` is robust to it" % CALL_STEPS,
              "step_wall_ms_this_rank": [round((b_ - a_) * 1e3, 2) for a_, b_ in zip([t0] + step_wall[:-1], step_wall)],
              "ms_per_step_median_this_rank": float(np.median([(b_ - a_) * 1e3 for a_, b_ in zip([t0] + step_wall[:-1], step_wall)])),
              "replicas_agree": agree,
              "roofline": None, "cpu_baseline": None}
    return None


def kernel_name(variant, probe, tiled, out_format):
    """Name of the kernel a launch of this configuration runs (as rocprofv3 prints it, spaces removed)."""
    b = lambda v: "true" if v else "false"  # noqa: E731
    if variant == 0 and out_format == "planar":   # the production configuration: csrc/defcorr_lean.hip
        return "lgu::lean::defcorr_lean_kernel<%s,%s>" % (b(probe), b(tiled))
    if variant == 1:
        return "lgu::defcorr_pyr_kernel<3,%s,12>" % b(probe)
    if variant == 2:
        return "lgu::defcorr_generic_kernel"
    gp, tpx = {3: (4, 16), 4: (2, 32)}.get(variant, (2, 16))
    om = {"planar": 0, "nhwc": 1, "nhwc_f16": 2}[out_format]
    if om:
        tpx = 8
    pair = variant in (0, 7) and om == 0
    return "lgu::defcorr_gather_kernel<3,%s,12,%d,%d,%s,%d,%s>" % (b(probe), gp, tpx, b(tiled), om, b(pair))


def load_traffic(kname, cache, tiled):
    """HBM bytes per launch from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    separate runs, gfx950 x2 fetch correction; tools/gpu_full_run.sh): counters cannot be collected from inside the timed
    process.  Newest round first; only a file measured in the same cache mode and layout on the same kernel is used."""
    lay = "tiled" if tiled else "rowmajor"
    for tname in ("traffic_r03_%s_%s.json" % (cache, lay), "traffic_r02_%s_%s.json" % (cache, lay)):
        tfile = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tfile):
            t = json.load(open(tfile))
            if t.get("kernel") == kname and t.get("cache") == cache:
                return t["hbm_bytes_per_launch"], "profiles/" + tname
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--edges", type=int, default=20, help="edges per GPU (BASELINE config 2: 20)")
    ap.add_argument("--probe", action="store_true", help="also time the level-1 uncertainty probe in the step")
    ap.add_argument("--variant", type=int, default=0, help="LGU_DEFCORR_VARIANT (A/B only)")
    ap.add_argument("--workload", choices=["defcorr", "lowmem", "backend"], default="defcorr",
                    help="'defcorr' = BASELINE config 2 (the headline metric, stored-volume path); 'lowmem' = BASELINE config 4 "
                         "(on-the-fly correlation from half feature maps at 60x80, the backend's path), reported in the same units; "
                         "'backend' = BASELINE config 5 (one global-BA iteration over a 1970-edge graph, STRONG scaling over --gpus: "
                         "shard lookups + all-gathers + replicated BA, per-phase breakdown; use --steps 10)")
    ap.add_argument("--layout", choices=["tiled", "rowmajor"], default="tiled",
                    help="storage of the pyramid the sampler reads: 'tiled' = the 4x8-tile slice layout CorrBlock keeps "
                         "its pyramid in (production), 'rowmajor' = the reference operator's layout (drop-in operator path)")
    ap.add_argument("--out-format", choices=["planar", "nhwc", "nhwc_f16"], default="planar",
                    help="output tensor: 'planar' = the reference's contiguous (E,196,H,W) fp32; 'nhwc' / 'nhwc_f16' = the "
                         "same values channel-last in fp32 / half, the form the consumer 1x1 convolution takes (tiled layout only)")
    ap.add_argument("--cache", choices=["cold", "warm"], default="cold",
                    help="'cold' (default, the headline): consecutive launches rotate over --sets disjoint input sets so that no "
                         "line touched by a launch is still in the 256 MiB Infinity Cache when it is touched again; 'warm': "
                         "every launch re-reads the same inputs (round 1's mode)")
    ap.add_argument("--sets", type=int, default=4, help="input sets rotated in cold mode (>= 3)")
    ap.add_argument("--blocks", type=int, default=5, help="timed blocks of --steps launches per 'extra' figure (median reported)")
    ap.add_argument("--no-extra", action="store_true", help="skip the 'extra' figures (probe on, row-major operator path, "
                    "config 3, config 4, warm cache)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--chunk-loop", action="store_true", help="backend workload: one lookup call per source-frame chunk, as the "
                    "reference's update_lowmem issues them, instead of ONE launch for all of the rank's chunks (the default; same results)")
    ap.add_argument("--ba-split", action="store_true", help="backend workload: bundle adjustment with the per-edge work on the edges' "
                    "owners + an all-reduce of the reduced camera system (sharded.sharded_ba_split) instead of replicated on every rank")
    ap.add_argument("--no-graph", action="store_true", help="issue the timed steps one by one from Python instead of replaying "
                                                             "them from one HIP graph")
    ap.add_argument("--randn-volumes", action="store_true", help="N(0,1) volumes instead of fmap products")
    ap.add_argument("--no-backend", action="store_true", help="skip the BASELINE config-5 leg (one sharded global-BA iteration, strong "
                    "scaling) that the default workload reports beside the headline")
    ap.add_argument("--backend-steps", type=int, default=8, help="timed steps of the config-5 leg inside the default workload")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="TEST ONLY: exercise the launch / process-group / timing / JSON logic with gloo on CPU and an "
                         "empty step (no kernel runs, the printed value is meaningless)")
    args = ap.parse_args()
    global NO_GRAPH
    NO_GRAPH = args.no_graph

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])   # no launcher around us: be the launcher (never returns)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if args.cache == "cold" and args.sets < 3:
        raise SystemExit("--cache cold needs --sets >= 3")
    quiet_stdout()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)  # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)  # RCCL
        if dist.get_world_size() != world:
            raise SystemExit("RCCL initialised %d ranks, the launcher promised %d" % (dist.get_world_size(), world))
    if args.variant:   # A/B only: the library honours its debug variables only when asked to at load time
        os.environ["LGU_DEBUG_KNOBS"] = "1"
        os.environ["LGU_DEFCORR_VARIANT"] = str(args.variant)

    import lgu_slam_amd
    lgu_slam_amd._lib.load()
    ops = lgu_slam_amd.ops
    if args.workload == "lowmem":
        return lowmem_main(args, ops, dev, rank, world, use_dist)
    if args.workload == "backend":
        return backend_main(args, lgu_slam_amd, dev, rank, world, use_dist)

    E, H1, W1, L, R = args.edges, 48, 64, 4, 3
    cold = args.cache == "cold"
    tiled = args.layout == "tiled"
    want_extra = rank == 0 and world == 1 and not args.no_extra
    # what CorrBlock.__init__ (ops.volume_pyramid(tiled=True)) leaves in HBM; building it is setup, not timed
    sets = DefcorrSets(ops, E, args.sets if cold else 1, 1234 + rank, dev, args.randn_volumes,
                       keep_rowmajor=(not tiled) or want_extra)
    units = sets.units
    # prepared launches: pointer tables built once, one ctypes call per step (the kernel is ~50 us; per-call Python
    # argument handling would otherwise bound the loop)
    plans = sets.plans(args.layout, args.probe, args.out_format)
    out = sets.out
    if args.out_format != "planar":
        out = ops._pyr_out(args.out_format, E, L * (2 * R + 1) ** 2, H1, W1, dev, None)
    # --probe: the level-1 uncertainty probe, variance, sigmoid and the stateful offset[1] *= mask of corr.py:94-99 run
    # inside the same launch
    step = sets.stepper(plans, out, cold, probe=args.probe)

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if args.probe:
        sets.restore_offsets()
    # the K timed steps as one HIP graph (not with the stateful probe: its offsets are restored between blocks)
    graph = None if (args.no_graph or args.probe) else capture_steps(step, args.warmup, args.steps)
    barrier()
    ev0, ev1 = _events()
    t0 = time.perf_counter()
    ev0.record()  # kernels are enqueued on torch's current stream, which these events time
    if graph is not None:
        graph.replay()
    else:
        for i in range(args.steps):
            step(args.warmup + i)
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if use_dist:
        import torch.distributed as dist
        t = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    n_ranks = 1
    if use_dist:
        import torch.distributed as dist
        n_ranks = dist.get_world_size()   # the ranks RCCL actually initialised
    ms_per_step = wall * 1e3 / args.steps
    value = n_ranks * units / (wall / args.steps) / 1e6

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

    # BASELINE config 5 in the same run: the multi-GPU design proper (fixed graph, chunks dealt to the ranks, all-gathers,
    # BA) — STRONG scaling; every rank takes part (collectives)
    strong, strong_error = None, None
    if not args.no_backend:
        try:
            strong = backend_run(args, lgu_slam_amd, dev, rank, world, use_dist, args.backend_steps, 3)
        except Exception as exc:   # the headline above is measured: a failure of this leg is reported in the line, not instead of it
            strong_error = "%s: %s" % (type(exc).__name__, exc)
            print("[bench] config-5 leg failed on rank %d: %s" % (rank, strong_error), file=sys.stderr)
        torch.cuda.empty_cache()

    if rank == 0:
        A, U = algorithmic_bytes_per_unit(sets.rowmajor[0], sets.coords[0], sets.offs[0], R)
        A_planar = A
        if args.out_format == "nhwc_f16":  # the output row is 2-byte elements
            A -= L * (2 * R + 1) ** 2 * 2
        kname = kernel_name(args.variant, args.probe, tiled, args.out_format)
        traffic, traffic_src = (None, None)
        if E == 20 and not args.probe and args.out_format == "planar":
            traffic, traffic_src = load_traffic(kname, args.cache, tiled)
        # headline device time: median over blocks of --steps launches in the headline's own mode (HIP events on the
        # launch stream); the single timed region above gives `value` / `ms_per_step` per the driver contract
        pre = sets.restore_offsets if args.probe else None
        hb = time_blocks(step, min(args.steps, 16) if args.probe else args.steps, args.blocks, pre)
        kern_ms = float(np.median(hb))
        achieved = A * units / (kern_ms * 1e-3) / 1e9
        cache_note = ("cold: launches rotate over %d disjoint input sets (%d x ~255 MB of touched lines between two uses of a "
                      "line > 256 MiB Infinity Cache), every launch HBM-served" % (args.sets, args.sets - 1)) if cold else \
                     "warm: every launch re-reads the same inputs (touched set ~243 MiB can stay in the Infinity Cache)"
        res = {
            "metric": "def-corr-sample Mpix·edges/s (48×64 fmap, r=3, L=4)",
            "value": value, "unit": "Mpix·edges/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config 2: TartanAir-mono shape, 48x64 fmap, L=4, r=3, "
                                   "%d edges per GPU, fused 4-level deformable sample%s" % (E, " + level-1 probe" if args.probe else ""),
                       "probe": "on (fused in the launch)" if args.probe else
                                "off (BASELINE.json's metric: the 4-level deformable sample alone; the production call with the "
                                "level-1 probe fused in is extra.probe_on)",
                       "cache": cache_note,
                       "launch": ("the K timed steps replayed from one HIP graph (same kernels, order and inputs as the direct loop)"
                                  if graph is not None else "K direct launches from Python"),
                       "edges_per_gpu": E, "units_per_step_per_gpu": units, "sharding": "edges (no data-path collective)",
                       "variant": args.variant, "volumes": "N(0,1)" if args.randn_volumes else "fmap products + avg_pool pyramid",
                       "pyramid_layout": "4x8-tiled slices (CorrBlock's own storage; results bit-identical)" if tiled
                                         else "row-major slices (reference operator layout)",
                       "out_format": args.out_format},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes per launch (PMC)",
                         "traffic_source": traffic_src, "algorithmic_bytes_per_launch": A * units,
                         "algorithmic_bytes_per_unit": A, "unique_volume_elements_per_unit": U,
                         "kernel": kname, "cache": args.cache,
                         "device_ms_per_step": kern_ms, "device_ms_per_step_blocks": hb,
                         "device_ms_per_step_timed_region": dev_ms / args.steps},
            "cpu_baseline": None,
        }
        res["config"]["library"] = lgu_slam_amd._lib.version()
        if want_extra:
            res["extra"] = extras(args, ops, dev, sets, A_planar, kern_ms)
            res["extra"].update(glue_extras(lgu_slam_amd, dev, args))
        if strong is not None:
            keep = {k: strong[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype",
                                           "phases_ms_max_over_ranks", "ms_per_step_median_this_rank", "replicas_agree")}
            keep["workload"] = strong["config"]["workload"]
            keep["edges_total"], keep["edges_this_rank"] = strong["config"]["edges_total"], strong["config"]["edges_this_rank"]
            keep["ba"] = strong["config"]["ba"]
            keep["note"] = ("strong scaling: the graph is fixed, `value` = its pixel·edges per second of step time (max over ranks); "
                            "lookups shrink with the rank count, exchange and the replicated BA do not (DESIGN.md §6); "
                            "`python bench.py --workload backend --gpus N [--ba-split]` runs this leg alone with more steps")
            if n_ranks > 1:
                res["strong_scaling_config5"] = keep
            else:
                res.setdefault("extra", {})["config5_backend_n1"] = keep
        if strong_error is not None:
            res["strong_scaling_config5" if n_ranks > 1 else "config5_backend_error"] = {"error": strong_error}
        if not (args.no_cpu or world > 1):  # rank 0, N=1 only
            res["cpu_baseline"] = cpu_baseline(E, H1, W1, L, R)
        if exchange:
            res["exchange"] = exchange
        emit(res)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def extras(args, ops, dev, sets, A, headline_ms):
    """The figures VERDICT r01 asked to see in the driver-run line next to the headline (each the median of --blocks
    blocks of --steps launches, device ms by HIP events, cold cache unless named warm): the production call with the
    level-1 probe fused in, the reference-layout operator path, BASELINE config 3 (E = 40), BASELINE config 4 (lowmem),
    and the headline kernel with a warm cache."""
    E, units = sets.E, sets.units
    steps, blocks = args.steps, args.blocks

    def rec(ms_list, u=units, a=A, **kw):
        ms = float(np.median(ms_list))
        d = {"device_ms_per_step": ms, "Mpix_edges_per_s": u / ms / 1e3, "roofline_frac": a * u / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
             "blocks_ms": [round(x, 5) for x in ms_list]}
        d.update(kw)
        return d

    ex = {"note": "device ms per launch, median of %d blocks of %d launches; cold cache (rotating %d input sets) unless named "
                  "warm; roofline_frac = algorithmic bytes / time / 8 TB/s as for the headline" % (blocks, steps, sets.nsets)}
    cold = sets.nsets > 1
    # (a) production call: probe on (CorrBlock.__call__ always launches with the probe, corr.py:94-99)
    pp = sets.plans("tiled", True, "planar")
    sets.restore_offsets()
    pblocks = min(max(blocks, -(-steps // 16)), 25)
    ex["probe_on"] = rec(time_blocks(sets.stepper(pp, sets.out, cold), min(steps, 16), pblocks, sets.restore_offsets),
                         workload="config 2 + fused level-1 probe (production CorrBlock.__call__), blocks of <= 16 launches "
                                  "restarting from the original offsets (a CorrBlock serves 8-16 lookups)")
    sets.restore_offsets()
    # (b) reference-layout operator path (what defCorrSample.* drop-in callers get)
    if all(v is not None for v in sets.rowmajor):
        pr = sets.plans("rowmajor", False, "planar")
        ex["rowmajor_operator_path"] = rec(time_blocks(sets.stepper(pr, sets.out, cold), steps, blocks),
                                           workload="config 2 over row-major slices (reference operator layout)")
    # (e) warm cache, headline kernel
    pt = sets.plans("tiled", False, "planar")
    ex["warm_cache"] = rec(time_blocks(sets.stepper(pt, sets.out, False), steps, blocks),
                           workload="config 2, every launch on the same inputs (round 1's mode)")
    ex["cold_over_warm"] = headline_ms / ex["warm_cache"]["device_ms_per_step"] if args.cache == "cold" and not args.probe else None
    # (c) BASELINE config 3: EuRoC stereo, ~2x edge count
    del pp, pt
    s40 = DefcorrSets(ops, 2 * E, 3 if cold else 1, 4242, dev, args.randn_volumes)
    p40 = s40.plans("tiled", False, "planar")
    ex["config3_E40"] = rec(time_blocks(s40.stepper(p40, s40.out, cold), steps, blocks), u=s40.units,
                            workload="BASELINE config 3: %d edges per launch (stereo + temporal), tiled" % (2 * E))
    ex["config3_E40"]["device_us_per_edge"] = ex["config3_E40"]["device_ms_per_step"] * 1e3 / (2 * E)
    p40p = s40.plans("tiled", True, "planar")
    ex["config3_E40_probe_on"] = rec(time_blocks(s40.stepper(p40p, s40.out, cold), min(steps, 16), pblocks, s40.restore_offsets), u=s40.units,
                                     workload="config 3 with the fused probe")
    del s40, p40, p40p
    torch.cuda.empty_cache()
    # (d) BASELINE config 4: lowmem
    # 16 edges = 2 400 workgroups over 1 024 slots = 2.3 rounds (tail quantisation); 64 edges is a backend chunk's size
    # (B ~ 50-150) — both stated
    for B, key in ((16, "config4_lowmem"), (64, "config4_lowmem_B64")):
        S = lowmem_setup(ops, dev, B, 4321)
        lb = time_blocks(lambda i: S["plan"](S["coords"], out=S["out"]), steps, blocks)
        lm = float(np.median(lb))
        ex[key] = {"workload": "BASELINE config 4: lowMem_defSample, 60x80x128 half feature maps, L=4, r=3, %d edges, all "
                               "levels in one launch" % B, "device_ms_per_step": lm, "Mpix_edges_per_s": S["units"] / lm / 1e3,
                   "device_us_per_edge": lm * 1e3 / B, "blocks_ms": [round(x, 5) for x in lb], "roofline": lowmem_roofline(S, lm)}
        del S
        torch.cuda.empty_cache()
    return ex


def glue_extras(lgu, dev, args):
    """Host-glue costs in the driver-run line: CorrBlock.__init__ at E = 20 (fp32 maps, and half maps
    under autocast as factor_graph.py:90 builds it) with the HBM fraction of its volume post-processing, and
    AltCorrBlock.__call__ on a 16-edge chunk at 60x80.  Device time by HIP events around GROUPS of calls issued back to
    back (multi-launch host paths: a single isolated call would measure Python's launch latency, not the device)."""
    out = {}
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    E, h, w = 20, 48, 64
    with torch.no_grad():
        ofsMap = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
        ofsRes = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
        GA = lgu.GaussianMask(h, w).to(dev)
        f1 = torch.randn(1, E, 128, h, w, device=dev, generator=g) * 0.5
        f2 = torch.randn(1, E, 128, h, w, device=dev, generator=g) * 0.5
        units = E * h * w
        # SURVEY §8(d), a5 + f1: per pixel·edge the post-processing must read the raw slice once and write the level-0 slice
        # (12 288 B each at 48x64) and the three pooled levels (+ 32.8 %), plus 16 B of Gaussian parameters: what ONE fused
        # pass moves.  (a5 alone — read 16 + 81*4 B, write 12 288 B — is SURVEY's 12.6 KB/unit.)
        a5_bytes = 16 + 81 * 4 + h * w * 4
        f1_bytes = 16 + h * w * 4 + int(h * w * 4 * (1 + 1 / 4 + 1 / 16 + 1 / 64))
        for half in (False, True):
            a, b = (f1.half(), f2.half()) if half else (f1, f2)
            # constructions issued back to back, as the frontend issues them between other work: device time of groups
            # of 5 (the host runs ahead of the device, so this is the device's cost, not the launch latency of ~40 launches)
            times = []
            for it in range(6):
                e0, e1 = _events()
                e0.record()
                for _ in range(5):
                    with torch.autocast("cuda", dtype=torch.float16, enabled=half):
                        blk = lgu.CorrBlock(ofsMap, ofsRes, GA, a, b)
                e1.record()
                e1.synchronize()
                if it >= 1:
                    times.append(e0.elapsed_time(e1) / 5)
            # the fused volume post-processing launch alone (gaussianMask + /denominator + corr + 3 poolings: a5 + f1)
            with torch.autocast("cuda", dtype=torch.float16, enabled=half):
                mean_n, cov, det = GA.gaussian_parameters(blk.t)
            raw = lgu.CorrBlock.corr(a, b).view(E, h, w, h, w)
            raw = raw.contiguous() if half else raw.float().contiguous()
            vp = []
            for it in range(12):
                src = raw.clone()
                e0, e1 = _events()
                e0.record()
                lgu.ops.volume_pyramid(mean_n.float().contiguous(), cov.float().contiguous(), src, 4, 4, inplace=True, tiled=True,
                                       det=det.contiguous() if det.dtype in (torch.float32, torch.float16) else det.float().contiguous())
                e1.record()
                e1.synchronize()
                if it >= 2:
                    vp.append(e0.elapsed_time(e1))
            ms, vms = float(np.median(times)), float(np.median(vp))
            in_bytes = h * w * (2 if half else 4)
            fused_bytes = 16 + in_bytes + int(h * w * 4 * (1 + 1 / 4 + 1 / 16 + 1 / 64))
            out["corrblock_init_E20_%s" % ("half_autocast" if half else "fp32")] = {
                "workload": "CorrBlock.__init__, 20 edges of 48x64x128 %s maps: all-pairs matmul, two offset heads + post-processing, "
                            "Gaussian head, fused volume post-processing into the 4-level tiled pyramid" % ("half (autocast)" if half else "fp32"),
                "device_ms_per_construction": ms, "us_per_edge": ms * 1e3 / E,
                "volume_postprocessing_kernel": {
                    "device_ms": vms, "algorithmic_bytes_per_unit_fused": fused_bytes,
                    "hbm_GBps": fused_bytes * units / (vms * 1e-3) / 1e9, "hbm_frac": fused_bytes * units / (vms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "a5_alone_bytes_per_unit_SURVEY": a5_bytes,
                    "note": "one pass: reads the raw slice (%d B) + 16 B parameters, writes levels 0-3 (fp32, +32.8 %%); a5 alone by "
                            "SURVEY §8(d) is 12.6 KB/unit (write-dominated) — the fused pass moves %.1f KB/unit and replaces the reference's "
                            "memset + mask + 2 elementwise + 3 pooling passes" % (in_bytes, fused_bytes / 1024.0)}}
            del blk, raw, src
        del f1, f2, a, b
        torch.cuda.empty_cache()
        # AltCorrBlock.__call__: 16 edges over 8 frames at 60x80 (half buffer, as depth_video holds it)
        N, H, W = 8, 60, 80
        fm = (torch.randn(1, N, 128, H, W, device=dev, generator=g) * 0.5).half()
        ofsMap2 = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
        ofsRes2 = torch.nn.Conv2d(256, 98, 3, padding=1).to(dev)
        ii = torch.arange(16, device=dev) // 2
        jj = (ii + 1 + torch.arange(16, device=dev) % 2) % N
        ys, xs = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
        coords = (torch.stack([xs, ys], -1)[None, None] + 2 * torch.randn(1, 16, H, W, 2, device=dev, generator=g)).contiguous()
        blk = lgu.AltCorrBlock(ofsMap2, ofsRes2, None, fm)
        times = []
        for it in range(7):   # groups of 10 calls back to back (the chunk loop of update_lowmem issues them like this)
            e0, e1 = _events()
            e0.record()
            for _ in range(10):
                blk(coords, ii, jj)
            e1.record()
            e1.synchronize()
            if it >= 1:
                times.append(e0.elapsed_time(e1) / 10)
        ms = float(np.median(times))
        out["altcorrblock_call"] = {"workload": "AltCorrBlock.__call__, 16 edges over 8 frames of 60x80x128 half maps: level-1 probe, offset heads "
                                                "of the call's first edge + post-processing, fused 4-level low-memory lookup",
                                    "device_ms_per_call": ms, "Mpix_edges_per_s": 16 * H * W / ms / 1e3}
    return out


if __name__ == "__main__":
    main()
