"""Host-side logic that needs no GPU: operator-layer argument checking (reference error
text, loud failure without a device), drop-in module surface, offset generation, the
Gaussian-mask parameter layout, and the edge-sharding / all-gather path under gloo with
world_size 2."""
import os
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ops_fail_loudly_without_a_device(lgu):
    v = torch.zeros(1, 2, 2, 2, 4)
    c = torch.zeros(1, 2, 2, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lgu.ops.corr_index_forward(v, c, 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lgu.ops.defcorr_pyramid_forward([v], c, [None], 1)


def test_contiguity_error_text_matches_reference(lgu):
    # reference: TORCH_CHECK(x.is_contiguous(), "volume must be contiguous") (offersample_LGS/droid.cpp:48-49)
    v = torch.zeros(1, 2, 2, 4, 2).transpose(3, 4)
    with pytest.raises(RuntimeError, match="^volume must be contiguous$"):
        lgu.ops.corr_index_forward(v, torch.zeros(1, 2, 2, 2), 1)
    with pytest.raises(RuntimeError, match="^coords must be contiguous$"):
        lgu.ops.corr_index_forward(torch.zeros(1, 2, 2, 2, 4), torch.zeros(1, 2, 2, 2).transpose(2, 3), 1)


def test_missing_library_raises(lgu, monkeypatch):
    monkeypatch.setattr(lgu._lib, "_lib", None)
    monkeypatch.setattr(lgu._build, "SO_PATH", "/nonexistent/liblgu_corr.so")
    with pytest.raises(RuntimeError, match="no CPU\\s+fallback|missing"):
        lgu._lib.load()


def test_dropin_modules_expose_reference_names(lgu):
    d, b = lgu.install_dropins()
    for n in ("gaussianMask", "gaussianMask_backward", "lowMem_defSample", "corr_index_forward",
              "corr_index_backward", "defCorr_index_forward", "defCorr_index_backward"):
        assert callable(getattr(d, n))  # offersample_LGS/droid.cpp:138-147
    assert callable(b.altcorr_forward) and callable(b.altcorr_backward)  # src/droid.cpp:246-247
    with pytest.raises(NotImplementedError):
        b.ba()
    assert sys.modules["defCorrSample"] is d


def test_generate_offsets_matches_reference_formula(lgu):
    torch.manual_seed(0)
    E, h, w = 2, 12, 16
    feats = torch.randn(E, 256, h, w)
    ofsMap = nn.Conv2d(256, 98, 3, padding=1)
    ofsRes = nn.Conv2d(256, 98, 3, padding=1)
    offs, zero = lgu.corr.generate_offsets(ofsMap, ofsRes, feats, 4)
    assert [tuple(o.shape) for o in offs] == [(E, h, w, 98)] * 4 and zero == [False, False, True, True]
    assert float(offs[0].abs().max()) < 4 and float(offs[1].abs().max()) < 4
    assert not offs[2].any() and not offs[3].any()

    def pcn(x):  # reference corr.py:44-51
        m = x.mean(dim=[1, 2, 3], keepdim=True)
        return (x - m) / torch.sqrt(x.var(dim=[1, 2, 3], unbiased=False, keepdim=True) + 1e-5)
    o0 = torch.tanh(pcn(ofsMap(feats))) * 4
    o1 = nn.functional.interpolate(ofsRes(nn.functional.avg_pool2d(feats, 2, 2)), (h, w))
    o1 = (torch.tanh(pcn(o1)) * 4 + o0) / 2
    assert torch.allclose(offs[0], o0.permute(0, 2, 3, 1), atol=1e-6)
    assert torch.allclose(offs[1], o1.permute(0, 2, 3, 1), atol=1e-6)


def test_gaussian_mask_parameter_names_and_grid(lgu):
    ga = lgu.GaussianMask(6, 8)
    names = set(dict(ga.named_parameters()))
    # the reference state dict carries these keys (gaussianMask_cuda.py:38-41,58)
    assert {"meanMap.weight", "meanMap.bias", "covMap.weight", "covMap.bias", "map.weight", "map.bias"} <= names
    assert not ga.meanMap.weight.any() and not ga.meanMap.bias.any()
    assert tuple(ga.coord.shape) == (6, 8, 2) and ga.coord[2, 5].tolist() == [5.0, 2.0]  # (x, y)


def test_shard_partitions(lgu):
    sh = lgu.sharded
    for n in (0, 1, 7, 20, 2001):
        for world in (1, 2, 3, 8):
            spans = [sh.balanced_shard(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    ii = torch.tensor([0, 0, 1, 3, 8, 9, 9, 17, 17, 30, 31])
    shards = sh.chunk_shards(ii, 2, chunk=8)
    got = sorted(int(i) for r in shards for idx in r for i in idx)
    assert got == list(range(ii.numel()))  # every edge exactly once
    for r in shards:
        for idx in r:  # one source-frame chunk per index tensor, as factor_graph.py:272-276
            assert int(ii[idx].max()) // 8 == int(ii[idx].min()) // 8


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import lgu_slam_amd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        E = 7
        full = torch.arange(E * 3 * 2, dtype=torch.float32).view(E, 3, 2)
        counts = [lgu_slam_amd.sharded.balanced_shard(E, r, world)[1] - lgu_slam_amd.sharded.balanced_shard(E, r, world)[0]
                  for r in range(world)]
        ex = lgu_slam_amd.sharded.EdgeExchange(counts)
        out = lgu_slam_amd.sharded.sharded_pyramid_sample(lambda c, lo, hi: c * 2.0, full, rank, world, ex)
        q.put((rank, bool(torch.equal(out, full * 2.0)), counts))
    finally:
        dist.destroy_process_group()


def test_edge_exchange_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res) and res[0][2] == [4, 3]


def _rehearsal_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import lgu_slam_amd
    sh = lgu_slam_amd.sharded
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        ii = torch.tensor([0, 0, 1, 3, 9, 9, 10, 17, 17, 18])
        x = torch.arange(10 * 3, dtype=torch.float32).view(10, 3)
        outs = []
        for rehearse in (False, True):
            sh.REHEARSE_COLLECTIVES = rehearse
            edges = sh.ShardedEdgeSet(ii)
            assert edges.world == 1 and sh._single(1) == (not rehearse)
            outs.append((edges.gather(x[edges.my_edges]), edges.gather_frames(torch.arange(float(edges.my_frames.numel())), out=torch.zeros(20)),
                         sh.replicas_agree(x)))
        q.put(bool(torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], x)
                   and outs[0][2] and outs[1][2]))
    finally:
        dist.destroy_process_group()


def test_collectives_can_be_rehearsed_at_world_size_one():
    """LGU_REHEARSE_COLLECTIVES: a world of one rank with an initialised process group issues the sharded step's collectives
    instead of short-cutting them (how the exact RCCL calls are run on a one-GPU box); results equal the short cut."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    p = ctx.Process(target=_rehearsal_worker, args=(0, 1, port, q))
    p.start()
    ok = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0 and ok


def test_torch_library_registration(lgu):
    from lgu_slam_amd import torch_ops
    assert len(torch_ops.REGISTERED) == 9
    sch = str(torch.ops.lgu.defCorr_index_forward.default._schema)
    assert "Tensor(a2!) offset" in sch and sch.endswith("-> Tensor[]")  # in/out offset, list return
    assert "Tensor fmap1" in str(torch.ops.lgu.altcorr_forward.default._schema)


def test_bench_distributed_control_flow_gloo_world2():
    """bench.py's multi-rank launch contract (torch.distributed.run, RANK/WORLD_SIZE env, barrier +
    max-over-ranks timing, exchange, ONE JSON line from rank 0) exercised on CPU with gloo."""
    import json
    import subprocess
    port = 29600 + (os.getpid() % 1500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--dry-run-cpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["exchange"]["bytes_per_rank"] == 20 * 48 * 64 * 4 * 4
    for k in ("metric", "value", "unit", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert k in d
    sc = d["strong_scaling_config5"]
    assert sc["n_gpus"] == 2 and sc["replicas_agree"] and sc["every_edge_owned"] and sum(sc["edges_per_rank"]) == sc["edges_total"]


def _run_bench(argv, env_drop=("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=300,
                          cwd=ROOT, env=env)


def test_bench_launches_its_own_ranks_from_a_plain_command():
    """`python bench.py --gpus 2` with NO launcher around it (VERDICT r02 missing #3: it used to run one rank silently and
    print n_gpus 1): the parent starts the two ranks itself, relays rank 0's single JSON line, and the line's n_gpus is
    the number of ranks the process group initialised.  It also carries the config-5 strong-scaling leg's record."""
    import json
    out = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-cpu"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["exchange"]["bytes_per_rank"] == 20 * 48 * 64 * 4 * 4
    sc = d["strong_scaling_config5"]
    assert sc["n_gpus"] == 2 and sc["replicas_agree"] and sc["every_edge_owned"] and min(sc["edges_per_rank"]) > 0
    # N = 1: no launcher, no process group, one line
    one = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "0", "--dry-run-cpu"])
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and "exchange" not in d1 and d1["config5_backend_n1"]["n_gpus"] == 1


def test_bench_exits_nonzero_when_a_rank_fails():
    """A failing rank must fail the whole command (no line, non-zero status), whoever launched it."""
    out = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--dry-run-cpu", "--edges", "-1"])
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # and a launcher whose world disagrees with --gpus is refused instead of silently measuring something else
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"], capture_output=True,
                         text=True, timeout=120, cwd=ROOT, env=env)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)


def test_sharded_edge_set_world1_orders(lgu):
    ii = torch.tensor([9, 0, 0, 17, 1, 3, 8, 9, 17, 30, 31])
    es = lgu.sharded.ShardedEdgeSet(ii, rank=0, world=1)
    assert sorted(es.my_edges.tolist()) == list(range(ii.numel()))
    x = torch.arange(ii.numel(), dtype=torch.float32)[:, None] * 10
    assert torch.equal(es.gather(x[es.my_edges]), x)  # back in original edge order


def _gloo_edge_set_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import lgu_slam_amd
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        ii = torch.tensor([9, 0, 0, 17, 1, 3, 8, 9, 17, 30, 31, 40, 41, 2])
        es = lgu_slam_amd.sharded.ShardedEdgeSet(ii)
        full = torch.arange(ii.numel() * 2, dtype=torch.float32).view(-1, 2)
        got = es.gather(full[es.my_edges])  # each rank contributes only what it owns
        q.put((rank, bool(torch.equal(got, full)), es.counts))
    finally:
        dist.destroy_process_group()


def test_sharded_edge_set_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_gloo_edge_set_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res) and sum(res[0][2]) == 14


def _reference_chunk_loop(ii, jj, chunk_fn, target, weight, damping, upmask):
    """The single-process loop of the reference (factor_graph.py:272-292), written out: what a sharded step must equal."""
    for i in range(0, int(jj.max()) + 1, 8):
        v = torch.nonzero((ii >= i) & (ii < i + 8)).flatten()
        if v.numel() == 0:
            continue
        t, w, d, u = chunk_fn(v, ii[v])
        target[v] = t
        weight[v] = w
        damping[torch.unique(ii[v])] = d
        upmask[torch.unique(ii[v])] = u
    return target, weight, damping, upmask


def _step_inputs():
    g = torch.Generator().manual_seed(11)
    # a non-bidirectional graph: the last source frames (40, 41) lie beyond the last chunk the reference's loop bound
    # (jj.max() + 1 = 34 -> chunks start at 0, 8, 16, 24, 32) produces: processed by nobody, values stay
    ii = torch.tensor([9, 0, 0, 17, 1, 3, 8, 9, 17, 30, 31, 40, 41, 2, 33, 12, 12])
    jj = torch.tensor([8, 1, 2, 16, 0, 2, 9, 10, 18, 31, 30, 33, 33, 3, 32, 11, 13])
    E, nf, ht, wd = ii.numel(), 42, 3, 4
    state = [torch.randn(E, ht, wd, 2, generator=g), torch.randn(E, ht, wd, 2, generator=g),
             torch.randn(nf, ht, wd, generator=g), torch.randn(nf, 5, ht, wd, generator=g)]

    def chunk_fn(idx, iis):  # deterministic stand-in for lookup + update operator: depends on edge ids and frames only
        fr = torch.unique(iis).float()
        e = idx.float().view(-1, 1, 1, 1)
        return (e + torch.zeros(idx.numel(), ht, wd, 2), 2 * e + torch.ones(idx.numel(), ht, wd, 2),
                fr.view(-1, 1, 1) * 10 + torch.zeros(fr.numel(), ht, wd), fr.view(-1, 1, 1, 1) * 100 + torch.zeros(fr.numel(), 5, ht, wd))
    return ii, jj, state, chunk_fn


def _gloo_step_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import lgu_slam_amd
    sh = lgu_slam_amd.sharded
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        ii, jj, state, chunk_fn = _step_inputs()
        want = _reference_chunk_loop(ii, jj, chunk_fn, *[t.clone() for t in state])
        es = sh.ShardedEdgeSet(ii, jj=jj)
        calls = []

        def counted(idx, iis):
            calls.append(idx.numel())
            return chunk_fn(idx, iis)
        got = sh.sharded_update_step(es, ii, counted, *[t.clone() for t in state])
        ok = all(torch.equal(a, b) for a, b in zip(got, want))
        # bookkeeping: every chunk on exactly one rank, frames owned once, the unprocessed edges are the reference's
        agree = sh.replicas_agree(*got)
        differ = sh.replicas_agree(got[0] + rank)  # rank-dependent tensor: must be reported as different
        q.put((rank, ok, agree, differ, sum(calls), es.unprocessed.tolist(), [f.tolist() for f in es.frames], es.counts))
    finally:
        dist.destroy_process_group()


def test_sharded_update_step_gloo_world2():
    """A complete sharded step (chunk loop on the owners, then the target / weight all-gathers by edge and the damping /
    upmask all-gathers by source frame) leaves on BOTH ranks exactly what the reference's single-process loop leaves —
    including the edges its jj.max() loop bound never processes."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_gloo_step_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, agree, differ, ncalls, unproc, frames, counts in res:
        assert ok and agree and not differ
        assert unproc == [11, 12]                      # ii = 40, 41 > last chunk of range(0, jj.max() + 1, 8)
        assert sum(counts) == 15 and sorted(f for fr in frames for f in fr) == [0, 1, 2, 3, 8, 9, 12, 17, 30, 31, 33]
    assert res[0][4] + res[1][4] == 15 and res[0][4] > 0 and res[1][4] > 0   # edges split over the two ranks


def test_sharded_update_step_world1_equals_reference_loop(lgu):
    ii, jj, state, chunk_fn = _step_inputs()
    want = _reference_chunk_loop(ii, jj, chunk_fn, *[t.clone() for t in state])
    es = lgu.sharded.ShardedEdgeSet(ii, rank=0, world=1, jj=jj)
    got = lgu.sharded.sharded_update_step(es, ii, chunk_fn, *[t.clone() for t in state])
    assert all(torch.equal(a, b) for a, b in zip(got, want))
    with pytest.raises(ValueError, match="without an owner"):
        es.gather(state[0][es.my_edges])               # unprocessed edges: a destination buffer is required
    # jj=None keeps the old bound (every edge in some chunk)
    assert lgu.sharded.ShardedEdgeSet(ii, rank=0, world=1).unprocessed.numel() == 0


def test_volume_operator_dtype_dispatch_errors(lgu):
    """Half / double volumes follow the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF rule: every scalar_t operand
    has the volume's dtype, coords stays float (defCorrSample_kernel.cu:185-191) — checked before any device work."""
    ops = lgu.ops
    v = torch.zeros(1, 2, 2, 4, 4, dtype=torch.float16)
    c = torch.zeros(1, 2, 2, 2)
    o = torch.zeros(1, 2, 2, 3, 3, 2)
    with pytest.raises(RuntimeError, match="expected scalar type Half but found Float"):
        ops.defCorr_index_forward(v, c, o, 1)
    with pytest.raises(RuntimeError, match="expected scalar type Float but found Double"):
        ops.defCorr_index_forward(v, c.double(), o.half(), 1)
    with pytest.raises(RuntimeError, match="expected scalar type Double but found Float"):
        ops.gaussianMask(torch.zeros(1, 2, 2, 2), torch.ones(1, 2, 2, 2), v.double(), 1)
    with pytest.raises(RuntimeError, match="offset must be contiguous"):
        ops.defCorr_index_forward(v, c, o.half().transpose(1, 2), 1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):   # dtypes consistent: reaches the fp32 operator, which needs a GPU
        ops.corr_index_forward(v, c, 1)


def test_tiled_layout_host_logic(lgu):
    """Pure host side of the tiled slice layout: shapes, logical-size checks, and a numpy restatement of the
    address formula of include/lgu_corr.h (LGU_PYR_TILED) that round-trips every element of a padded slice."""
    import numpy as np
    ops = lgu.ops
    assert ops.tiled_shape(2, 48, 64, 48, 64) == (2, 48, 64, 12, 8, 4, 8)
    assert ops.tiled_shape(1, 3, 5, 6, 8) == (1, 3, 5, 2, 1, 4, 8)      # 6 rows pad to 8
    assert ops.tiled_shape(1, 3, 5, 15, 20) == (1, 3, 5, 4, 3, 4, 8)    # 15x20 pads to 16x24
    t = torch.zeros(ops.tiled_shape(1, 2, 2, 6, 8))
    assert ops._level_dims([t], True, [(6, 8)]) == ([6], [8])
    assert ops._level_dims([t], True, None) == ([8], [8])              # unpadded logical size assumed
    with pytest.raises(RuntimeError, match="tiled pyramid level"):
        ops._level_dims([torch.zeros(1, 2, 2, 6, 8)], True, [(6, 8)])   # row-major tensor passed as tiled
    with pytest.raises(RuntimeError, match="tiled pyramid level"):
        ops._level_dims([t], True, [(12, 8)])                           # logical size that does not match the tiles
    H2, W2 = 15, 20
    tpr = -(-W2 // 8)
    y, x = np.meshgrid(np.arange(H2), np.arange(W2), indexing="ij")
    pos = ((y // 4) * tpr + x // 8) * 32 + (y % 4) * 8 + x % 8
    assert len(np.unique(pos)) == H2 * W2 and pos.max() < (-(-H2 // 4)) * tpr * 32
    # separability used by the kernel: the +1 neighbours are one add away
    sx = np.where(x % 8 == 7, 25, 1)
    sy = np.where(y % 4 == 3, tpr * 32 - 24, 8)
    assert np.array_equal((pos + sx)[:, :-1], pos[:, 1:]) and np.array_equal((pos + sy)[:-1], pos[1:])


def test_lowmem_plan_argument_checks(lgu):
    ops = lgu.ops
    h = torch.zeros(1, 4, 4, 32, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="1..4 levels"):
        ops.LowmemPyramidPlan(h, [h] * 5, [None] * 5, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.LowmemPyramidPlan(h, [h], [None], 3)
    with pytest.raises(RuntimeError, match="no CPU fallback|expected scalar type"):
        ops.LowmemPyramidPlan(h.float(), [h], [None], 3)


def test_corr_encoder_host_logic():
    """CorrEncoder: structure check, and anything that is not a channel-last half CUDA tensor is handed to
    the wrapped module unchanged (the reference call, droid_net.py:116)."""
    import lgu_slam_amd as lgu
    torch.manual_seed(0)
    enc = torch.nn.Sequential(torch.nn.Conv2d(196, 128, 1), torch.nn.ReLU(inplace=True),
                              torch.nn.Conv2d(128, 128, 3, padding=1), torch.nn.ReLU(inplace=True)).eval()
    fused = lgu.CorrEncoder(enc)
    x = torch.randn(2, 196, 6, 8)
    assert not fused.takes(x) and not fused.takes(x.half().contiguous(memory_format=torch.channels_last))
    with torch.no_grad():
        assert torch.equal(fused(x), enc(x))
    with pytest.raises(RuntimeError):
        lgu.CorrEncoder(torch.nn.Sequential(torch.nn.Conv2d(196, 128, 3), torch.nn.ReLU()))
    with pytest.raises(RuntimeError):
        lgu.ops.DefcorrPyramidPlan.__init__(object.__new__(lgu.ops.DefcorrPyramidPlan), [], [], 3, out_format="nchw")
    assert lgu.CorrBlock.OUT_FORMAT == "planar" and set(lgu.ops.OUT_FORMATS) == {"planar", "nhwc", "nhwc_f16"}


def test_offset_conv_weight_packing_host_logic():
    """pack_offset_conv: hi + lo reproduces scale * W to 2^-21 relative, zero rows beyond Cout, and the MFMA fragment
    order [tap][kstep][part][ntile][kg][nl][8] maps back to W[n = ntile*16 + nl][c = kstep*32 + kg*8 + t][tap]."""
    import lgu_slam_amd as lgu
    torch.manual_seed(3)
    W = torch.randn(98, 256, 3, 3)
    b = torch.randn(98)
    wpack, bias, Cout, C = lgu.ops.pack_offset_conv(W, b, scale=4.0)
    assert (Cout, C) == (98, 128) and tuple(wpack.shape) == (9, 8, 2, 7, 4, 16, 8) and wpack.dtype == torch.float16
    assert bias.dtype == torch.float32 and torch.equal(bias, b)
    rec = (wpack[:, :, 0].float() + wpack[:, :, 1].float())          # [tap][kstep][ntile][kg][nl][t]
    rec = rec.permute(2, 4, 1, 3, 5, 0).reshape(112, 256, 9)        # [n][c][tap]
    want = torch.zeros(112, 256, 9)
    want[:98] = (W * 4.0).reshape(98, 256, 9)
    assert float((rec - want).abs().max()) <= 2.0 ** -21 * float(want.abs().max())
    assert float(rec[98:].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        lgu.ops.pack_offset_conv(torch.randn(120, 256, 3, 3), torch.randn(120))
    with pytest.raises(RuntimeError):
        lgu.ops.pack_offset_conv(torch.randn(98, 256, 1, 1), b)


def test_ba_assembly_tables_host_logic():
    """ba._csr: rows grouped by destination (ascending row inside a group), CSR over ALL destinations, rows pointing
    outside [0, ndst) dropped — the tables lgu_ba_assemble_f64 walks."""
    import numpy as np
    from lgu_slam_amd import ba as B
    dest = np.array([2, -1, 0, 2, 5, 0, 7, 2])
    ptr, idx = B._csr(dest, 6, torch.device("cpu"))
    assert ptr.tolist() == [0, 2, 2, 5, 5, 5, 6]
    assert idx.tolist() == [2, 5, 0, 3, 7, 4]
    ptr, idx = B._csr(np.array([-1, 9]), 3, torch.device("cpu"))
    assert ptr.tolist() == [0, 0, 0, 0] and idx.numel() == 1   # placeholder entry keeps the pointer valid
