"""Data-parallel path on CPU: world_size 2, gloo, one process per rank (the reference has no DP).

Parity definition: every rank evaluates the loss on ITS shard (shard-local old_mean and negatives); the post-all-reduce
head gradient equals the mean over ranks of the per-shard gradients.  The loss itself is evaluated with the CPU oracle
here (the HIP path needs a GPU; the same GradBucket is driven by bench.py on RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import depthg_oracle as O
from depthg_amd.parallel import GradBucket, shard_range

B_GLOBAL, C, D, HW, S, N = 4, 16, 8, 8, 4, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_problem():
    g = torch.Generator().manual_seed(11)
    feats = torch.randn(B_GLOBAL, C, HW, HW, generator=g)
    feats_pos = torch.randn(B_GLOBAL, C, HW, HW, generator=g)
    depth = torch.randint(0, 256, (B_GLOBAL, 1, 4 * HW, 4 * HW), generator=g).float()
    w = torch.randn(D, C, 1, 1, generator=g) * 0.3      # stand-in 1x1-conv head (cluster1, reference src/modules.py:80-81)
    bias = torch.zeros(D)
    coords1 = torch.rand(B_GLOBAL, S, S, 2, generator=g) * 2 - 1
    coords2 = torch.rand(B_GLOBAL, S, S, 2, generator=g) * 2 - 1
    return feats, feats_pos, depth, w, bias, coords1, coords2


def _shard_grads(rank, world):
    """Oracle loss on the shard of `rank`; returns (dW, dbias)."""
    feats, feats_pos, depth, w, bias, coords1, coords2 = _make_problem()
    lo, hi = shard_range(B_GLOBAL, world, rank)
    w = w.clone().requires_grad_(True)
    bias = bias.clone().requires_grad_(True)
    f, fp, d = feats[lo:hi], feats_pos[lo:hi], depth[lo:hi]
    code = torch.nn.functional.conv2d(f, w, bias)
    code_pos = torch.nn.functional.conv2d(fp, w, bias)
    g = torch.Generator().manual_seed(100 + rank)       # shard-local negatives
    perms = [O.super_perm(hi - lo, g) for _ in range(N)]
    cfg = O.default_cfg(feature_samples=S, neg_samples=N)
    out = O.forward(cfg, f, fp, code, code_pos, d, d, coords1=coords1[lo:hi], coords2=coords2[lo:hi], perms=perms)
    O.total_loss(cfg, out).backward()
    return w.grad.detach(), bias.grad.detach()


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    gw, gb = _shard_grads(rank, world)
    w = torch.nn.Parameter(torch.zeros_like(gw)); w.grad = gw.clone()
    b = torch.nn.Parameter(torch.zeros_like(gb)); b.grad = gb.clone()
    bucket = GradBucket.for_parameters([w, b], dist)
    bucket.pack()
    bucket.allreduce_mean_()
    bucket.unpack()
    # every rank holds bit-identical averaged gradients afterwards
    gathered = [torch.zeros_like(bucket.flat) for _ in range(world)]
    dist.all_gather(gathered, bucket.flat)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    if rank == 0:
        ret["w"] = w.grad.clone()
        ret["b"] = b.grad.clone()
        ret["same"] = same
        ret["numel"] = bucket.flat.numel()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_allreduce_matches_mean_of_shard_gradients():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    want_w = sum(_shard_grads(r, world)[0] for r in range(world)) / world
    want_b = sum(_shard_grads(r, world)[1] for r in range(world)) / world
    assert ret["same"]
    assert ret["numel"] == D * C + D
    assert torch.allclose(ret["w"], want_w, rtol=1e-6, atol=1e-9)
    assert torch.allclose(ret["b"], want_b, rtol=1e-6, atol=1e-9)
    assert float(want_w.abs().max()) > 0


def test_bucket_single_process_is_identity():
    p = torch.nn.Parameter(torch.arange(6.0).view(2, 3))
    p.grad = torch.ones(2, 3) * 3
    bucket = GradBucket.for_parameters([p], None)
    bucket.pack()
    bucket.allreduce_mean_()
    bucket.unpack()
    assert torch.equal(p.grad, torch.ones(2, 3) * 3)


# ---- validation metric under DP: per-rank confusion matrices, summed at compute() -----------------
def _metric_shards():
    from conftest import load_golden
    g = load_golden("metrics.npz")
    n, e, hung = (int(v) for v in g["e3_hung_cfg"])
    shards = [O.confusion_counts(torch.from_numpy(p), torch.from_numpy(t), n, e)
              for p, t in zip(g["e3_hung_preds"], g["e3_hung_target"])]
    return g, n, e, bool(hung), shards


def _metric_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from depthg_amd.metrics import UnsupervisedMetrics
    g, n, e, hung, shards = _metric_shards()
    m = UnsupervisedMetrics("test/cluster/", n, e, hung)
    # rank 0 saw batches 0 and 2, rank 1 batch 1 (the accumulation itself is the GPU kernel, covered by the -m gpu tests)
    m.stats = shards[0] + shards[2] if rank == 0 else shards[1].clone()
    local = m.stats.clone()
    out = m.compute()
    ret[rank] = (out["test/cluster/mIoU"], out["test/cluster/Accuracy"], torch.equal(m.stats, local))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_metric_sums_confusion_matrices():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_metric_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    g = _metric_shards()[0]
    for rank in range(world):
        miou, acc, untouched = ret[rank]
        assert miou == pytest.approx(float(g["e3_hung_miou"]), rel=1e-6)      # == the reference on all three batches
        assert acc == pytest.approx(float(g["e3_hung_acc"]), rel=1e-6)
        assert untouched                                                        # compute() leaves the local state alone


def _worker_recorded(rank, world, port, ret):
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "dp_hip_shards.npz"))
    w = torch.nn.Parameter(torch.zeros(D, C, 1, 1)); w.grad = torch.from_numpy(fx[f"dw{rank}"]).clone()
    b = torch.nn.Parameter(torch.zeros(D)); b.grad = torch.from_numpy(fx[f"db{rank}"]).clone()
    bucket = GradBucket.for_parameters([w, b], dist)
    bucket.pack()
    bucket.allreduce_mean_()
    bucket.unpack()
    if rank == 0:
        ret["w"] = w.grad.clone()
        ret["b"] = b.grad.clone()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_allreduce_of_recorded_hip_gradients():
    """The shard gradients the HIP path produced on an MI355X for this file's problem (tests/golden/dp_hip_shards.npz, recorded
    by tests/golden/make_dp_hip_fixture.py): two gloo ranks all-reduce them through the same GradBucket.  The result is the
    exact fp32 mean of the two recorded shards and agrees with the mean of the per-shard ORACLE gradients within the gradient
    tolerance of the HIP path (clamp-mask flips)."""
    import numpy as np
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker_recorded, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
            assert p.exitcode == 0
        got_w, got_b = ret["w"], ret["b"]
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "dp_hip_shards.npz"))
    mean_w = (torch.from_numpy(fx["dw0"]) + torch.from_numpy(fx["dw1"])) * 0.5
    mean_b = (torch.from_numpy(fx["db0"]) + torch.from_numpy(fx["db1"])) * 0.5
    assert torch.equal(got_w, mean_w) and torch.equal(got_b, mean_b)
    ow = sum(_shard_grads(r, world)[0] for r in range(world)) / world
    ob = sum(_shard_grads(r, world)[1] for r in range(world)) / world
    assert (got_w - ow).norm() <= 3e-2 * ow.norm(), float((got_w - ow).norm() / ow.norm())
    assert (got_b - ob).norm() <= 3e-2 * ob.norm() + 1e-7


# ---- the N > 1 step schedule of bench.py: DoubleBufferedExchange over gloo with a stub compute ----------
def _schedule_worker(rank, world, port, ret):
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from depthg_amd.parallel import DoubleBufferedExchange
    n = 1000
    buckets = [GradBucket(n, "cpu", dist) for _ in range(2)]
    trace, seen = [], []
    sched = None

    def compute(k):
        # the step's kernels end by filling bucket k with THIS step's gradients: here a pattern that names (rank, step)
        step_no = sched.count
        buckets[k].flat.fill_(float(1000 * rank + step_no))
        buckets[k].flat[1] = float(rank)
        return step_no

    sched = DoubleBufferedExchange(buckets, compute, comm_stream=None, trace=trace)
    # time-based clock warm-up as in bench.py: the ranks run DIFFERENT counts of it (no collective in there, or the pairs of the
    # timed steps' collectives would slip against each other and the run would deadlock or average different steps)
    t0, warm_calls = time.perf_counter(), 0
    while time.perf_counter() - t0 < (0.05 if rank == 0 else 0.25) or warm_calls < 3 + 4 * rank:
        sched.warm()
        warm_calls += 1
    dist.barrier()
    K = 7
    for i in range(K):
        out = sched.step()
        k = i & 1
        # (CPU buckets: the exchange has completed) bucket k = mean over ranks of this step's pattern on every rank
        seen.append((out, k, float(buckets[k].flat[0]), float(buckets[k].flat[1]), float(buckets[1 - k].flat[0])))
    sched.drain()
    dist.barrier()
    ret[rank] = (warm_calls, seen, [t for t in trace if t[0] != "warm"])
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_step_schedule_with_stub_compute():
    """bench.py's N > 1 schedule, the class itself, two gloo ranks: buckets alternate, the wait on a bucket's previous collective
    precedes its refill, the refill precedes its exchange, warm-up steps of different counts per rank issue no collective, every
    step's bucket ends as the mean over ranks of THAT step's gradients, and the other bucket still holds the previous step's."""
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_schedule_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
            assert p.exitcode == 0
        res = {r: ret[r] for r in range(world)}
    assert res[0][0] != res[1][0] and res[1][0] >= 7              # the ranks warmed up for different counts
    for r in range(world):
        _, seen, trace = res[r]
        for i, (out, k, v0, v1, other) in enumerate(seen):
            assert out == i and k == (i & 1)
            assert v0 == pytest.approx(500.0 + i) and v1 == pytest.approx(0.5)        # mean over the two ranks of 1000 r + i, r
            if i > 0:
                assert other == pytest.approx(500.0 + i - 1)                          # the other buffer: last step's average, intact
        want = []
        for i in range(len(seen)):
            want += [("wait", i & 1, i), ("compute", i & 1, i), ("exchange", i & 1, i)]
        want += [("drain", 0, len(seen)), ("drain", 1, len(seen))]
        assert trace == want
    assert res[0][1] == res[1][1]                                  # bit-identical averaged buckets on both ranks


def test_step_schedule_ablation_switches():
    """`exchange=False` / `alternate=False` (bench.py --ablate) on one process without a process group: no collective, one bucket."""
    from depthg_amd.parallel import DoubleBufferedExchange
    buckets = [GradBucket(4, "cpu", None) for _ in range(2)]
    trace = []
    s = DoubleBufferedExchange(buckets, lambda k: buckets[k].flat.add_(1.0), exchange=False, alternate=False, trace=trace)
    for _ in range(3):
        s.step()
    s.drain()
    assert float(buckets[0].flat[0]) == 3.0 and float(buckets[1].flat[0]) == 0.0
    assert [t[0] for t in trace] == ["wait", "compute"] * 3 + ["drain", "drain"]


def _run_bench(args, env_extra, drop=()):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") + tuple(drop)}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_bench_gpus_n_spawns_n_ranks_by_itself():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must be a 2-rank run: the parent starts the ranks (a child
    torch.distributed.run, before any GPU call) and relays rank 0's line; DG_BENCH_DRYRUN keeps it on gloo and times nothing."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"DG_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                    # the JSON line is the only thing on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen"] == 2 and line["config"]["parallelism"] == "dp2"


@pytest.mark.timeout(300)
def test_bench_four_rank_dry_run_with_unequal_warmups():
    """World 4: the schedule class driven by bench.py's own rank code over gloo, every rank with another warm-up count (no collective
    in there), then the same number of steps - each step's bucket must be the mean over the FOUR ranks of that step's pattern."""
    import json
    r = _run_bench(["--gpus", "4", "--steps", "5", "--warmup", "2"], {"DG_BENCH_DRYRUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert line["n_gpus"] == 4 and line["config"]["ranks_seen"] == 4
    assert len(set(line["warm_calls_by_rank"])) == 4 and line["steps_paired_on_every_rank"] is True


def test_device_count_comes_from_sysfs_not_from_the_runtime(tmp_path):
    """spawn_ranks' parent must not touch HIP: GPUs are the KFD topology nodes with SIMDs; a visible-devices list caps the count;
    an unreadable topology gives None; a profiler preload is recognised from the environment."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for i, simd in enumerate([0, 0, 1024, 1024, 1024]):          # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n")
    assert bench.count_gpus_sysfs(str(tmp_path), {}) == 3
    assert bench.count_gpus_sysfs(str(tmp_path), {"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert bench.count_gpus_sysfs(str(tmp_path / "missing"), {}) is None
    assert bench.profiler_preloaded({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})
    assert bench.profiler_preloaded({"ROCPROFILER_LIBRARY_CTOR": "1"}) and not bench.profiler_preloaded({"PATH": "/usr/bin"})
    src = open(os.path.join(root, "bench.py")).read()
    body = src[src.index("def spawn_ranks"):src.index("def dryrun_rank")]
    assert "torch.cuda" not in body.split('"""')[2]            # (the code behind the docstring)


def test_bench_refuses_a_line_for_more_gpus_than_ranks():
    # a launcher that made ONE rank for --gpus 2: no line, non-zero exit (round 4's bench printed n_gpus: 1 and exited 0)
    r = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "DG_BENCH_DRYRUN": "1"})
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()
    # and without the dry-run switch in this GPU-less container: fewer devices than ranks asked for
    if not torch.cuda.is_available():
        r = _run_bench(["--gpus", "2"], {}, drop=("DG_BENCH_DRYRUN",))
        assert r.returncode == 2 and "GPU(s)" in r.stderr and not r.stdout.strip()
