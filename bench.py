#!/usr/bin/env python3
"""Headline benchmark: correlation-loss steps/sec at B=32, C=384, 28x28 (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W
(for N > 1 launched by the driver through torch.distributed.run, one rank per GPU, RCCL).

A "step" = one forward + backward of the DepthG correlation loss, starting from the fp32 feature /
code maps as the featurizer hands them over, resident in HBM:
    sample() of feats/code (dense 28x28 identity grid) -> norm() -> 7 pair-sets of helper()
    (1 intra, 1 inter, 5 negatives) + depth_feature_correlation() -> 4 loss means + 4 cd means ->
    d/d orig_code, d/d orig_code_pos (through norm() and sample()),
    and, for N > 1, one RCCL all-reduce of a head-gradient-sized fp32 buffer (201,740 elements =
    DinoFeaturizer cluster1+cluster2 at C=384, dim=70).
Batch is sharded data-parallel: every rank runs B=32 of its own synthetic images (weak scaling).
"""
import argparse
import subprocess
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HEADLINE = dict(B=32, C=384, D=70, h=28, w=28, S=28, n_neg=5, depth_hw=224)
# --config: the workloads of BASELINE.json `configs` 2-5 with the reference's recipe scalars
# (paper_reproduction.sh:5-14); the headline stays the default and is the line the driver records.
CONFIGS = {
    "headline": dict(H=HEADLINE, sampling="none", dense=True, pointwise=True,
                     scal=dict(pos_intra_shift=0.07, pos_inter_shift=0.025, neg_inter_shift=0.761, depth_feat_shift=0.03,
                               pos_intra_weight=0.58, pos_inter_weight=0.36, neg_inter_weight=0.7, depth_feat_weight=0.19),
                     what="headline: B=32/GPU, C=384, D=70, 28x28 dense grid (S=28, P=784), 5 negatives, depth term on, pointwise, "
                          "zero_clamp; fwd + bwd to orig_code/orig_code_pos", cpu_B=32),
    "headline+head": dict(H=HEADLINE, sampling="none", dense=True, pointwise=True, head=True,
                          scal=dict(pos_intra_shift=0.07, pos_inter_shift=0.025, neg_inter_shift=0.761, depth_feat_shift=0.03,
                                    pos_intra_weight=0.58, pos_inter_weight=0.36, neg_inter_weight=0.7, depth_feat_weight=0.19),
                          what="headline + the segmentation head: backbone features (B,384,28,28) x 2 -> cluster1 / cluster2 with their "
                               "three Dropout2d draws (HIP head, bf16 MFMA) -> the headline loss -> backward into the 201,740 head "
                               "parameters (the all-reduced bucket holds these real gradients)", cpu_B=32),
    "C2": dict(H=dict(B=16, C=384, D=90, h=28, w=28, S=11, n_neg=5, depth_hw=224), sampling="fps", dense=False, pointwise=True,
               scal=dict(pos_intra_shift=0.2, pos_inter_shift=0.09, neg_inter_shift=0.63, depth_feat_shift=0.14,
                         pos_intra_weight=0.61, pos_inter_weight=0.34, neg_inter_weight=0.72, depth_feat_weight=0.13),
               what="config 2: Potsdam ViT-S recipe, B=16, C=384, dim=90, 28x28 maps, feature_samples=11 (P=121), depth_sampling=fps "
                    "(sampler inside the step), depth term on; fwd + bwd", cpu_B=16),
    "C3": dict(H=dict(B=32, C=768, D=100, h=28, w=28, S=11, n_neg=5, depth_hw=224), sampling="none", dense=False, pointwise=False,
               scal=dict(pos_intra_shift=0.39, pos_inter_shift=0.25, neg_inter_shift=0.26, depth_feat_shift=0.03,
                         pos_intra_weight=0.95, pos_inter_weight=1.02, neg_inter_weight=0.57, depth_feat_weight=0.09),
               what="config 3: Cityscapes ViT-B recipe, B=32, C=768, dim=100, 28x28 maps, feature_samples=11, random coords, "
                    "pointwise=False; fwd + bwd", cpu_B=32),
    "C4shard": dict(H=dict(B=8, C=768, D=90, h=28, w=28, S=12, n_neg=5, depth_hw=224), sampling="fps", dense=False, pointwise=True,
                    scal=dict(pos_intra_shift=0.123, pos_inter_shift=0.21, neg_inter_shift=0.975, depth_feat_shift=0.0359,
                              pos_intra_weight=0.2305, pos_inter_weight=1.05, neg_inter_weight=0.2485, depth_feat_weight=0.16),
                    what="config 4, one rank's shard: COCO-Stuff ViT-B recipe, global batch 64 over 8 GPUs -> B=8/GPU, C=768, dim=90, "
                         "feature_samples=12 (P=144), depth_sampling=fps; fwd + bwd", cpu_B=8),
    "C5": dict(H=dict(B=32, C=384, D=70, h=56, w=56, S=56, n_neg=5, depth_hw=448), sampling="none", dense=True, pointwise=True,
               scal=dict(pos_intra_shift=0.07, pos_inter_shift=0.025, neg_inter_shift=0.761, depth_feat_shift=0.03,
                         pos_intra_weight=0.58, pos_inter_weight=0.36, neg_inter_weight=0.7, depth_feat_weight=0.19),
               what="config 5: 56x56 maps (ViT-S/8 at 448 input), B=32/GPU, C=384, D=70, dense grid (P=3136), positives = a "
                    "different random tensor (kNN positive), 5 negatives, depth term on; fwd + bwd", cpu_B=1),
    # not a BASELINE.json configuration: the headline's dense 28x28 grid at the ViT-B width (VERDICT r05 item 6 asks for the line) -
    # 768-channel vectors do not fit k_corr2's accumulation registers, the step runs on the round-1 kernel k_corr_main
    "denseB": dict(H=dict(B=32, C=768, D=70, h=28, w=28, S=28, n_neg=5, depth_hw=224), sampling="none", dense=True, pointwise=True,
                   scal=dict(pos_intra_shift=0.07, pos_inter_shift=0.025, neg_inter_shift=0.761, depth_feat_shift=0.03,
                             pos_intra_weight=0.58, pos_inter_weight=0.36, neg_inter_weight=0.7, depth_feat_weight=0.19),
                   what="dense 28x28 grid at the ViT-B width: B=32/GPU, C=768, D=70, P=784, 5 negatives, depth term on, pointwise, "
                        "zero_clamp; fwd + bwd (not a BASELINE configuration)", cpu_B=8),
}
HEAD_GRAD_ELEMS = 201_740          # cluster1 (26,950) + cluster2 (174,790) parameters, reference src/modules.py:75-88
PEAK_BF16_TFLOPS = 2500.0          # dense bf16/f16 MFMA peak of MI355X (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBPS = 8000.0             # HBM3E peak of MI355X (same guide)
PEAK_CLOCK_GHZ = 2.4               # ... which is 256 CUs x 4 SIMDs x 1024 flop/cycle at this shader clock


def make_cfg(conf, **over):
    from types import SimpleNamespace
    # recipe scalars of the reference (paper_reproduction.sh) on top of src/configs/local_config.yml
    H = conf["H"]
    cfg = SimpleNamespace(
        feature_samples=H["S"], use_salience=False, depth_sampling=conf["sampling"], fps_gpu=False, pointwise=conf["pointwise"],
        zero_clamp=True, stabalize=False, neg_samples=H["n_neg"], depth_feat_correlation_loss=True,
        correspondence_weight=1.0, dg_outputs="reduced", dg_dense_grid=conf["dense"], **conf["scal"])
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def synth_inputs(B, seed, device, H=HEADLINE):
    g = torch.Generator().manual_seed(seed)
    f = torch.randn(B, H["C"], H["h"], H["w"], generator=g)
    fp = torch.randn(B, H["C"], H["h"], H["w"], generator=g)
    c = torch.randn(B, H["D"], H["h"], H["w"], generator=g)
    cp = torch.randn(B, H["D"], H["h"], H["w"], generator=g)
    d = torch.randint(0, 256, (B, 1, H["depth_hw"], H["depth_hw"]), generator=g).float()
    dp = torch.randint(0, 256, (B, 1, H["depth_hw"], H["depth_hw"]), generator=g).float()
    return [t.to(device) for t in (f, fp, c, cp, d, dp)]


def algorithmic_gflop(B, P, C, D, n_neg):
    """one correlation = 2*B*P^2*K flop (real K, no padding; nothing recomputed is counted).
    Returns (step total, share done by the fused kernel, share done by the k_gs launch):
      forward   (2+n) feature correlations (K=C), (3+n) code correlations (K=D: the pair-sets + the depth term's cd),
                1 rank-1 depth product (K=1)                                                   -> k_corr_main
      backward  d/d(stationary code) for the (2+n) pair-sets + the depth term: (3+n) * K=D     -> k_corr_main
                d/d(streamed code) for the (2+n) pair-sets: (2+n) * K=D                        -> k_gs
                (the depth term is symmetric: its streamed side is the stationary side, counted once, doubled as a factor)"""
    corr = lambda k: 2.0 * B * P * P * k / 1e9
    fwd = (2 + n_neg) * corr(C) + (3 + n_neg) * corr(D) + corr(1)
    main = fwd + (3 + n_neg) * corr(D)
    gs = (2 + n_neg) * corr(D)
    # the depth term (its cd correlation, the rank-1 dd, its stationary-side gradient) runs as blocks of the k_gs launch on a
    # gradient pass (dg_corr.hip gs_depth_block), not in the fused kernel
    dep = 2 * corr(D) + corr(1)
    return main + gs, main - dep, gs + dep


def _cpu_baseline_at(conf, ncores, seconds_budget, Bs):
    from oracle import depthg_oracle as O
    H = conf["H"]
    torch.set_num_threads(ncores)
    cfg = O.default_cfg(feature_samples=H["S"], neg_samples=H["n_neg"], dim=H["D"], pointwise=conf["pointwise"],
                        depth_sampling=conf["sampling"], **conf["scal"])
    f, fp, c, cp, d, dp = synth_inputs(Bs, 4321, "cpu", H)
    g = torch.Generator().manual_seed(7)
    perms = [O.super_perm(Bs, g) for _ in range(H["n_neg"])]
    if conf.get("head"):
        C_, D_ = H["C"], H["D"]
        head_prm = [(torch.randn(sh, generator=g) * 0.05).requires_grad_(True)
                    for sh in ((D_, C_), (D_,), (C_, C_), (C_,), (D_, C_), (D_,))]
    times = []
    t_start = time.time()
    for rep in range(40):
        c1 = c.clone().requires_grad_(True)
        cp1 = cp.clone().requires_grad_(True)
        t0 = time.time()
        if conf["dense"]:
            coords1 = coords2 = O.identity_coords(Bs, H["S"])
        elif conf["sampling"] == "fps":        # the sampler is part of the step (two calls, src/modules.py:1304-1308)
            coords1 = O.farthest_point_sampling_depth((H["h"], H["w"]), d, H["S"]) * 2 - 1
            coords2 = O.farthest_point_sampling_depth((H["h"], H["w"]), dp, H["S"]) * 2 - 1
        else:
            coords1 = torch.rand(Bs, H["S"], H["S"], 2, generator=g) * 2 - 1
            coords2 = torch.rand(Bs, H["S"], H["S"], 2, generator=g) * 2 - 1
        if conf.get("head"):               # the head in front of the loss: c / cp play no role, the code comes from the features
            from oracle import head_oracle as HO
            keeps = [tuple((torch.rand(Bs, H["C"], generator=g) > 0.1).float() for _ in range(3)) for _ in range(2)]
            c1, f_in = HO.head_forward(f, *head_prm, keeps=keeps[0])
            cp1, fp_in = HO.head_forward(fp, *head_prm, keeps=keeps[1])
        else:
            f_in, fp_in = f, fp
        out = O.forward(cfg, f_in, fp_in, c1, cp1, d, dp, coords1=coords1, coords2=coords2, perms=perms)
        O.total_loss(cfg, out).backward()
        times.append(time.time() - t0)
        if time.time() - t_start > seconds_budget and len(times) >= 2:
            break
    timed = times[1:] if len(times) > 1 else times      # first repetition is the warm-up
    return min(timed), len(timed)


def cpu_baseline(conf, seconds_budget=20.0):
    """The CPU restatement (oracle/, kind "port") timed on the host cores on a bounded sample of the SAME workload: same C, D,
    S, pair-sets, sampler and backward, `cpu_B` images of the batch; scaled to steps/s of the full batch.  Timed at 16 threads
    in this process and at 32 / 64 / 128 threads and every host CPU in child processes with a wall-clock limit each (torch's
    intra-op pool thrashes on the pool's 256-CPU hosts: 600 x slower at 256 threads than at 16); the fastest leg is the reported
    value, the others ride along in `also`."""
    H = conf["H"]
    host = os.cpu_count() or 1
    runs, skipped = [], []
    n16 = min(host, 16)
    t16, reps16 = _cpu_baseline_at(conf, n16, seconds_budget * 0.5, conf["cpu_B"])
    runs.append({"cores": n16, "value": (conf["cpu_B"] / H["B"]) / t16, "seconds": t16, "reps": reps16, "Bs": conf["cpu_B"]})
    name = [k for k, v in CONFIGS.items() if v is conf][0]
    for n in [n for n in (32, 64, 128) if n < host] + ([host] if host > n16 else []):
        # a child process per thread count, ended by subprocess.run when its limit passes.  The sample shrinks to what would
        # take ~1 s per repetition at the 16-thread rate for the all-CPU leg (known to thrash); the others keep the full sample
        # unless one repetition at the 16-thread rate would not fit the limit twice
        limit = 14.0
        Bs = conf["cpu_B"]
        if n == host:
            Bs = min(Bs, int(conf["cpu_B"] * 0.15 / t16))
        elif t16 * 2.5 > limit:
            Bs = int(conf["cpu_B"] * limit / (t16 * 2.5))
        if Bs < 1:
            skipped.append({"cores": n, "skipped": f"one image takes {t16 / conf['cpu_B']:.1f} s on {n16} threads; this leg is not waited for"})
            continue
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-leg", f"{name},{n},{Bs},{limit * 0.4}"],
                               capture_output=True, text=True, timeout=limit)
            t, reps = json.loads(r.stdout.strip().splitlines()[-1])
            runs.append({"cores": n, "value": (Bs / H["B"]) / t, "seconds": t, "reps": reps, "Bs": Bs})
        except subprocess.TimeoutExpired:
            skipped.append({"cores": n, "skipped": f"two repetitions at B={Bs} did not finish in {limit:.0f} s on {n} threads "
                                                   f"(16 threads: {t16 * Bs / conf['cpu_B']:.2f} s each)"})
        except Exception as e:      # the legs that ran are reported; say what happened to this one
            skipped.append({"cores": n, "skipped": f"child failed: {type(e).__name__}"})
    best = max(runs, key=lambda r: r["value"])
    other = [r for r in runs if r is not best]
    return {"value": best["value"], "unit": "steps/s", "cores": best["cores"], "host_cpus": host, "kind": "port",
            "sample": f"oracle forward+backward at B={best['Bs']} (of {H['B']}), C={H['C']}, D={H['D']}, S={H['S']}, "
                      f"{H['n_neg']} negatives, sampling={conf['sampling']}, {best['cores']} threads of {host} host CPUs, "
                      f"min of {best['reps']} timed reps = {best['seconds']:.2f} s; value = ({best['Bs']}/{H['B']}) / t; "
                      f"thread counts tried: {sorted(r['cores'] for r in runs)}",
            "also": [{"cores": r["cores"], "value": r["value"], "unit": "steps/s", "sample_B": r["Bs"]} for r in other] + skipped}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def count_gpus_sysfs(root=KFD_NODES, environ=None):
    """GPUs of this node WITHOUT touching the HIP / HSA runtime: the KFD topology nodes with SIMDs (CPU nodes have `simd_count 0`),
    capped by a HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES list.  None when the topology cannot be read (no amdgpu driver, a
    container without /sys/class/kfd): the caller then lets the ranks themselves fail on a missing device."""
    environ = os.environ if environ is None else environ
    try:
        nodes = sorted(os.listdir(root))
    except OSError:
        return None
    have = 0
    for nd in nodes:
        try:
            with open(os.path.join(root, nd, "properties")) as fh:
                props = dict(ln.split()[:2] for ln in fh if len(ln.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            have += 1
    for key in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = environ.get(key)
        if v is not None:
            have = min(have, len([x for x in v.split(",") if x.strip() != ""]))
    return have


def profiler_preloaded(environ=None):
    """True under rocprofv3 / rocprof: its preloaded tool library has initialised the GPU before this program started, so this
    process must not act as a launcher of further GPU processes (the ranks have to be profiled directly)."""
    environ = os.environ if environ is None else environ
    if "rocprof" in environ.get("LD_PRELOAD", "").lower():
        return True
    return any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_TOOL")) for k in environ)


def spawn_ranks(n, argv):
    """`--gpus N` (N > 1) with WORLD_SIZE unset: start the N ranks as a CHILD `python -m torch.distributed.run` (one process per
    GPU, rendezvous on 127.0.0.1) from this parent, which makes no GPU call at all: the devices are counted from the KFD topology in
    sysfs (count_gpus_sysfs), not through the runtime, and nothing here re-execs an initialised process.  Refused (exit 6) under a
    profiler's preload, which has initialised the GPU already: profile the ranks, not the launcher.  Rank 0's JSON line is relayed
    as the last line of stdout.  Non-zero exit when the node has fewer than N GPUs, when a rank fails, or when the line was not
    produced by N ranks."""
    dry = bool(os.environ.get("DG_BENCH_DRYRUN"))
    if profiler_preloaded() and not dry:
        print(f"[bench] --gpus {n} under a profiler preload (rocprofv3): the GPU is already initialised in this process, which "
              f"must not start further GPU processes.  Profile a rank instead: rocprofv3 ... -- python3 -m torch.distributed.run "
              f"--nproc-per-node {n} ... bench.py --gpus {n} is also a launcher hop; run one rank with --force-dist", file=sys.stderr)
        return 6
    have = count_gpus_sysfs()
    if have is None and not os.path.exists("/dev/kfd"):
        have = 0                               # no KFD topology and no compute device node: there is no GPU to run on
    if have is not None and have < n and not dry:
        print(f"[bench] --gpus {n}: this node shows {have} GPU(s) in {KFD_NODES}; not running a {n}-GPU line on fewer devices",
              file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    print("[bench] starting the ranks: " + " ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)         # banners of the launcher / RCCL: kept, but not on stdout
    if r.returncode != 0:
        print(f"[bench] the {n}-rank run exited {r.returncode}", file=sys.stderr)
        return r.returncode
    if line is None:
        print("[bench] rank 0 printed no JSON line", file=sys.stderr)
        return 4
    got = json.loads(line)
    seen = got.get("config", {}).get("ranks_seen")
    if got.get("n_gpus") != n or seen != n:
        print(f"[bench] asked for {n} ranks, the line reports n_gpus={got.get('n_gpus')} ranks_seen={seen}", file=sys.stderr)
        return 5
    print(line, flush=True)
    return 0


def dryrun_rank(args, world, rank):
    """DG_BENCH_DRYRUN=1 (tests/test_dp_gloo.py, no GPU): the ranks rendezvous over gloo and drive the N > 1 step schedule itself
    (depthg_amd.parallel.DoubleBufferedExchange, CPU buckets, a stub in place of the step's kernels) the way main() does: a
    warm-up of a DIFFERENT length on every rank with no collective in it, then exactly --steps steps with one all-reduce each.
    Rank 0 prints a line of the usual shape with no measurement in it - what is checked is the launch path and that the
    collectives of the timed steps pair up, not the step."""
    import torch.distributed as dist
    from depthg_amd.parallel import DoubleBufferedExchange, GradBucket
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    one = torch.ones(1)
    dist.all_reduce(one)
    buckets = [GradBucket(64, "cpu", dist) for _ in range(2)]
    sched = None

    def stub(k):
        buckets[k].flat.fill_(float(1000 * rank + sched.count))      # "this step's gradients": names (rank, step)
        return sched.count

    sched = DoubleBufferedExchange(buckets, stub, comm_stream=None)
    warm_calls = 3 + 2 * rank + args.warmup                          # unequal on purpose (main(): time-based, so it differs too)
    for _ in range(warm_calls):
        sched.warm()
    dist.barrier()
    ok = True
    for i in range(args.steps):
        sched.step()
        want = 1000.0 * (world - 1) / 2.0 + i                        # mean over the ranks of 1000 r + i
        ok = ok and abs(float(buckets[i & 1].flat[0]) - want) < 1e-3
    sched.drain()
    flags = [None] * world
    dist.all_gather_object(flags, (warm_calls, bool(ok)))
    if rank == 0:
        print(json.dumps({"metric": "dry run: no step was timed", "value": None, "unit": "steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dryrun": True,
                          "warm_calls_by_rank": [f[0] for f in flags], "steps_paired_on_every_rank": all(f[1] for f in flags),
                          "config": {"name": args.config, "ranks_seen": int(one.item()), "parallelism": f"dp{world}"}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if all(f[1] for f in flags) else 7


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="headline",
                    help="workload: the headline (default, BASELINE.json metric) or one of BASELINE.json's configs 2-5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-allreduce", action="store_true",
                    help="N > 1: fill + all-reduce on the compute stream, which waits for the collective right away.  Default: the "
                         "exchange runs on a side stream behind the step's backward (GradBucket.exchange_on) and overlaps the next "
                         "step's kernels, as it overlaps the frozen ViT forward in training; exchanges are ordered among themselves "
                         "and every collective of the timed steps completes inside the timed region either way")
    ap.add_argument("--graph", action="store_true",
                    help="(default since round 3, kept for old command lines) the step is recorded once in a hipGraph "
                         "(torch.cuda.graph) and replayed; the negatives' permutations advance on the device (cfg.dg_graph_safe)")
    ap.add_argument("--eager", action="store_true",
                    help="launch the step's kernels from Python every step instead of replaying them from a hipGraph")
    ap.add_argument("--strict-graph", action="store_true",
                    help="exit 3 when the hipGraph capture of the step fails (default: say so on stderr and in config.schedule, and time "
                         "the host-launched step on that rank)")
    ap.add_argument("--clock-warmup-s", type=float, default=1.0,
                    help="untimed steps are run for this many seconds BEFORE the --warmup steps, so that the GPU has left its idle "
                         "power state when the timed region starts (an idle MI355X needs tens of ms of load to reach its "
                         "sustained clock; reported as `clock_warmup_steps`)")
    ap.add_argument("--exact-masks", action="store_true",
                    help="cfg.dg_exact_masks: the clamp mask 1[cd >= 0] from fp32 dot products instead of the fp16 cd of the MFMA "
                         "chain (gradient error 1.4e-2 -> below 2e-3 relative L2; dense ViT-S grids; the default is the fast path)")
    ap.add_argument("--ablate", choices=["", "noexchange", "onegraph", "streamwait", "twocalls", "torchmasks", "writefeats"], default="", help=argparse.SUPPRESS)
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) and run the collective path even with one rank (self-test)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: THIS process never touches the GPU, it starts the N ranks itself
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # a line quoted for N GPUs must have been produced by N ranks
        print(f"[bench] --gpus {args.gpus} but the launcher created WORLD_SIZE={world} ranks: refusing to run", file=sys.stderr)
        raise SystemExit(2)
    if os.environ.get("DG_BENCH_DRYRUN"):
        raise SystemExit(dryrun_rank(args, world, rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)                  # one process per GPU: bind before the process group is created
    dev = torch.device("cuda", local_rank)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    from depthg_amd import ContrastiveCorrelationLoss, ops
    from depthg_amd.parallel import GradBucket

    conf = CONFIGS[args.config]
    H = conf["H"]
    # the fused correlation kernel's workgroups stamp their entry / exit times (dg_prof_main_span) - also inside the replayed hipGraph,
    # so the roofline leg below reads the kernel's execution span INSIDE steps, what a kernel trace shows, not a re-launch beside them
    # (the timed steps run WITHOUT the stamps: the timer is armed only for a second recording of the step, made behind the timed
    #  region, that the roofline leg replays - the pointer in force at capture time is baked into a hipGraph)
    ktimer = ops.MainKernelTimer(dev)
    # ONE schedule for every N: the compute part of the step is replayed from a hipGraph (the eager step's Python side, 0.2 ms,
    # plus the collective's, 0.05-0.14 ms, would make the host the limit of a 0.3-ms step); N > 1 adds the collective on a side
    # stream and nothing else.  --eager / --sync-allreduce opt out (the JSON line says which schedule ran).
    graph_mode = not args.eager and not args.sync_allreduce
    cfg = make_cfg(conf, dg_graph_safe=graph_mode, dg_exact_masks=args.exact_masks)
    loss_fn = ContrastiveCorrelationLoss(cfg)
    f, fp, c, cp, d, dp = synth_inputs(H["B"], 1234 + rank, dev, H)
    c.requires_grad_(True)
    cp.requires_grad_(True)
    # loss-only configurations: a stand-in for the head gradients - a buffer of the head's size filled from this step's d/d code,
    # averaged over the ranks (RCCL over xGMI); --config headline+head: the real head in front of the loss and its real
    # gradients in the bucket.  Two buffers: step i + 1 fills the other one while step i's is in flight
    head, head_params = None, []
    if conf.get("head"):
        from depthg_amd.head import ProjectionHead
        torch.manual_seed(1234 + rank)
        head = ProjectionHead(H["C"], H["D"], "nonlinear").to(dev).train()
        head_params = list(head.parameters())
        c.requires_grad_(False)
        cp.requires_grad_(False)
        buckets = [GradBucket.for_parameters(head_params, dist if use_dist else None) for _ in range(2)]
        assert buckets[0].flat.numel() == HEAD_GRAD_ELEMS
    else:
        buckets = [GradBucket(HEAD_GRAD_ELEMS, dev, dist if use_dist else None) for _ in range(2)]
    comm = torch.cuda.Stream() if use_dist else None     # the gradient exchange runs here
    seed_grad = torch.ones((), device=dev)   # d(total)/d(total): the upstream of the step, resident like the other inputs
    counter = [0]

    def compute_with_head(bucket=None):
        for prm in head_params:
            prm.grad = None
        # f / fp are the frozen backbone's outputs here; three Dropout2d draws per pass, both passes in one set of launches
        if args.ablate == "twocalls":      # (the round-3 form: one call per pass, autograd adds the two passes' weight gradients)
            code, feats = head(f)
            code_pos, feats_pos = head(fp)
        else:
            # (graph-recorded step: the six Dropout2d masks from the device-resident generator, one launch)
            keeps = ops.keep_masks_state(loss_fn._graph_state(dev), 2 * H["B"], H["C"], head.p) if (graph_mode and args.ablate != "torchmasks") else None
            # (on the identity grid the Dropout2d of the returned feats is applied by the loss's operand preparation:
            #  --ablate writefeats is the form that writes the dropped tensors and reads them back)
            defer = args.ablate != "writefeats" and loss_fn.takes_deferred_dropout(f.shape[-2:])
            (code, feats), (code_pos, feats_pos) = head.forward_pair(f, fp, True, keeps, defer)
        loss_fn(feats, feats_pos, None, None, code, code_pos, d, dp)
        total = loss_fn.total
        total.backward(gradient=seed_grad)
        if bucket is not None:
            bucket.pack()                  # the real head gradients
        return total

    def compute(bucket=None):
        if head is not None:
            return compute_with_head(bucket)
        c.grad = None
        cp.grad = None
        loss_fn(f, fp, None, None, c, cp, d, dp)
        total = loss_fn.total          # weighted total of the four loss means (training_step's term), formed by the library
        total.backward(gradient=seed_grad)
        if bucket is not None:
            bucket.fill_from(c.grad)
        return total

    def capture(fn):
        if os.environ.get("DG_BENCH_FAIL_CAPTURE"):      # (test hook: tests/test_gpu_configs.py drives the fall-back below)
            raise RuntimeError("capture failure injected by DG_BENCH_FAIL_CAPTURE")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fn()
        return g, out

    exchange_mode = "none"
    capture_note = [None]

    def capture_or_die(fn):
        """The step as a hipGraph.  A failed capture never changes the schedule SILENTLY: with --strict-graph the run exits 3; without
        it this rank times the host-launched step (same kernels, same collective pattern: ranks may fall back independently), says so
        on stderr, and the JSON line carries the reason under config.schedule."""
        try:
            return capture(fn)
        except RuntimeError as e:
            msg = str(e).splitlines()[0] if str(e) else type(e).__name__
            print(f"[bench] hipGraph capture failed on rank {rank}: {e}", file=sys.stderr)
            if args.strict_graph:
                print("[bench] --strict-graph: re-run with --eager to time the host-launched step instead", file=sys.stderr)
                raise SystemExit(3)
            print("[bench] timing the host-launched (eager) step on this rank instead", file=sys.stderr)
            capture_note[0] = f"eager on rank {rank} (hipGraph capture failed: {msg})"
            torch.cuda.synchronize()
            return None

    if not use_dist:
        captured = capture_or_die(compute) if graph_mode else None
        if captured is not None:
            graph, total_static = captured

            def step():
                graph.replay()
                return total_static
        else:
            step = compute
    elif args.sync_allreduce:
        exchange_mode = "every step, on the compute stream (which waits for it)"

        def step():
            total = compute(buckets[0])
            buckets[0].allreduce_mean_(even_if_alone=args.force_dist)
            return total
    else:
        # The N > 1 schedule proper (depthg_amd.parallel.DoubleBufferedExchange; tests/test_dp_gloo.py drives the same class over
        # gloo with a stub compute): wait for the collective that read bucket k two steps ago -> the step's kernels, which end by
        # filling bucket k -> the all-reduce of bucket k on the side stream.
        from depthg_amd.parallel import DoubleBufferedExchange
        if graph_mode:
            exchange_mode = ("every step, on a side stream behind the step's graph: overlaps the next step's kernels, completes inside "
                             "the timed region; the step (with the fill of its bucket) is replayed from one of two hipGraphs")
            graphs = [capture_or_die(lambda k=k: compute(buckets[k])) for k in range(2)]

            # (an external event node at the end of each graph would keep the compute stream's queue free of event records -
            #  torch refuses them on ROCm: "External events are disallowed in rocm")
            def run_kernels(k):
                if graphs[k] is None:          # (capture failed on this rank, reported: the host-launched step)
                    return compute(buckets[k])
                graphs[k][0].replay()
                return graphs[k][1]
        else:
            exchange_mode = ("every step, on a side stream: overlaps the next step's kernels, completes inside the timed region "
                             "(eager step)")

            def run_kernels(k):
                return compute(buckets[k])
        # (--ablate: timing experiments only - where the N > 1 schedule's extra time goes)
        sched = DoubleBufferedExchange(buckets, run_kernels, comm, even_if_alone=args.force_dist,
                                       exchange=args.ablate != "noexchange", alternate=args.ablate != "onegraph",
                                       host_wait=args.ablate != "streamwait")
        step = sched.step

    def sync():
        if use_dist:
            for b in buckets:
                b.wait_exchange()          # every all-reduce of the timed steps completes inside the timed region
            # (the host must not enter the barrier before its own collectives are done being ENQUEUED-and-complete: the barrier
            #  is a collective on the same communicator)
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    # the GPU out of its idle power state first (untimed, reported; the step's kernels WITHOUT the collective, so that ranks
    # running different counts cannot mis-order RCCL calls), then the W warm-up steps, then EXACTLY K timed steps
    if not use_dist:
        warm = step
    elif args.sync_allreduce:
        warm = compute
    else:
        warm = sched.warm
    clock_warmup_steps = 0
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < args.clock_warmup_s:
        for _ in range(10):
            warm()
        clock_warmup_steps += 10
        torch.cuda.synchronize()
    sync()
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        total = step()
    host_elapsed = time.perf_counter() - t0     # the Python loop alone (enqueue side); equals `elapsed` when the host is the limit
    sync()
    elapsed = time.perf_counter() - t0
    dist_diag = None
    if use_dist:
        # what a first real N > 1 run needs to be read: every rank's own time (a slow rank, a straggling collective), how long the
        # compute stream actually stalls on the exchange, and the world size RCCL reports
        mine = torch.tensor([elapsed, host_elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t[0]) / args.steps * 1e3 for t in every]
        elapsed = max(float(t[0]) for t in every)
        dist_diag = {"ranks_seen": dist.get_world_size(), "backend": dist.get_backend(),
                     "ms_per_step_min_rank": round(min(per_rank), 4), "ms_per_step_max_rank": round(max(per_rank), 4),
                     "ms_per_step_by_rank": [round(v, 4) for v in per_rank],
                     "host_ms_per_step_max_rank": round(max(float(t[1]) for t in every) / args.steps * 1e3, 4)}
        if not args.sync_allreduce:
            # device-side stall of the compute stream on `wait_exchange` (the collective of two steps ago), measured on extra,
            # untimed steps: events around the wait itself (every rank runs the same count: the collectives stay paired)
            nprobe, stall = 20, 0.0
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nprobe)]
            for i in range(nprobe):
                k = sched.count & 1 if sched.alternate else 0
                evs[i][0].record()
                buckets[k].wait_exchange()
                evs[i][1].record()
                step()
            t_d = time.perf_counter()
            sync()
            dist_diag["drain_ms"] = round((time.perf_counter() - t_d) * 1e3, 4)      # host wait for the tail: last collectives + barrier
            stall = sum(a.elapsed_time(b) for a, b in evs) / nprobe
            dist_diag["exchange_wait_ms"] = round(stall, 5)
            # the SAME K steps with the collective on the compute stream, which waits for it right away (what --sync-allreduce
            # times, here with the replayed step): the overlapped `value` above hides the collective under the next step's kernels,
            # this figure has all of it on the critical path - a real training step lies between the two
            t_s = time.perf_counter()
            for _ in range(args.steps):
                run_kernels(0)
                buckets[0].allreduce_mean_(even_if_alone=args.force_dist)
            sync()
            t_sync = torch.tensor([time.perf_counter() - t_s], dtype=torch.float64, device=dev)
            dist.all_reduce(t_sync, op=dist.ReduceOp.MAX)
            dist_diag["ms_per_step_sync_allreduce"] = round(float(t_sync) / args.steps * 1e3, 4)
            dist_diag["value_sync_allreduce"] = round(world * args.steps / float(t_sync), 2)
    if use_dist:
        # a rank that fell back to the host-launched step is named on the line whichever rank it was (rank 0 prints)
        notes = [None] * world
        dist.all_gather_object(notes, capture_note[0])
        notes = [n for n in notes if n]
        capture_note[0] = "; ".join(notes) if notes else None
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps / elapsed   # every rank completes `steps` steps of its own B=32 shard

    # ---- roofline of the dominant kernel (the fused correlation launch), measured live with HIP events on the launch stream
    desc, perms_t, ws = loss_fn.last_call
    # The kernel's launch duration UNDER THE STEP'S CONDITIONS: [step + one extra launch of the fused kernel] minus [step], both
    # timed with HIP events over `reps` iterations on the launch stream, five interleaved rounds, median.  (A bare loop of
    # back-to-back re-launches keeps the matrix cores saturated, the chip lowers its clock, and the same kernel measures 5-6 %
    # slower than rocprofv3's kernel trace shows it inside the step: 160 against 151 us, profiles/r03_SUMMARY.md.  `kernel_ms_loop`
    # keeps that figure.)
    reps = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def timed(fn):
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(reps):
            fn()
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

    def step_plus_kernel():
        warm()
        ops.corr_relaunch_main(desc, perms_t, ws)

    for _ in range(3):
        ops.corr_relaunch_main(desc, perms_t, ws)
    diffs = []
    for _ in range(5):
        t_step = timed(warm)
        t_both = timed(step_plus_kernel)
        diffs.append(t_both - t_step)
    kern_ms_diff = sorted(diffs)[len(diffs) // 2]
    kern_ms_loop = timed(lambda: ops.corr_relaunch_main(desc, perms_t, ws))
    # the kernel inside the step: `reps` back-to-back steps, the span stamps reset in front of the LAST one and read behind it; 16 such
    # samples, mean and spread.  This is the figure the committed rocprofv3 kernel trace of the same command reports as the kernel's
    # average (minus the dispatch ramp) - `frac` is quoted on it.
    ktimer.arm(True)
    probe_graph = capture_or_die(compute) if graph_mode else None       # (kept alive to the end of main(): it writes to ktimer.span)
    probe = (lambda: probe_graph[0].replay()) if probe_graph is not None else compute
    in_step, held = [], []
    for _ in range(16):
        for _ in range(reps - 1):
            probe()
        ktimer.reset()
        probe()
        ms, ghz = ktimer.last()
        in_step.append(ms)
        held.append(ghz)
    # the same 20 steps with and without the stamps (what the instrumentation costs the step; the timed region above ran without)
    t_probe = timed(probe)
    ktimer.arm(False)
    t_plain = timed(warm)
    held = [v for v in held if v == v and v > 0.0]
    held_ghz = sum(held) / len(held) if held else None
    in_step = [v for v in in_step if v == v and v > 0.0]
    kern_ms = sum(in_step) / len(in_step) if in_step else kern_ms_diff
    kern_method = ("span of the kernel's workgroups (entry / exit stamps of the GPU's 100-MHz clock, atomic min / max) inside the "
                   + ("replayed" if graph_mode else "host-launched") + " step, last of 20 back-to-back steps, mean of 16 samples"
                   if in_step else "HIP events: [step + 1 extra launch] - [step], 5 x 20 iterations, median (no span stamps were read)")
    step_gf, main_gf, gs_gf = algorithmic_gflop(H["B"], H["S"] ** 2, H["C"], H["D"], H["n_neg"])
    # k_corr2's FOLD: the fused launch also forms the intra pair-set's streamed-side gradient (one correlation with K = D of the
    # reference's work, by symmetry and one extra MFMA per chain) - that share moves from the k_gs launch to this one
    intra_folded = ops.corr_intra_folded(desc)
    if intra_folded:
        share = 2.0 * H["B"] * (H["S"] ** 2) ** 2 * H["D"] / 1e9
        main_gf += share
        gs_gf -= share
    # which kernel that launch is: the library's own predicate (dg_corr_main_kernel_name), not a copy of it
    kname = ops.corr_main_kernel_name(desc)
    # exact clamp masks on the dense grid: k_corr2's exact-mask form has NO cd chain (round 6) - the mask words come from k_cd_mask3,
    # which forms every cd in split fp16 (three times the MFMAs of the chain) - so the (2 + n) code correlations of the forward are
    # not the fused launch's work any more: they leave its algorithmic flops (nothing is credited for k_cd_mask3's recomputation)
    cd_outside = bool(args.exact_masks) and kname == "k_corr2" and H["S"] ** 2 > 160
    if cd_outside:
        main_gf -= (2 + H["n_neg"]) * 2.0 * H["B"] * (H["S"] ** 2) ** 2 * H["D"] / 1e9
    achieved = main_gf / 1e3 / (kern_ms / 1e3)   # TFLOP/s of the fused kernel alone
    # HBM bytes per launch of that kernel: counters cannot be read from inside a run, so this is the figure of the committed PMC
    # passes of THIS round (scripts/profile_round.sh: FETCH_SIZE x2 on gfx950 + WRITE_SIZE) - null when there is none
    traffic, traffic_source = None, None
    if args.config == "headline":
        src = next((c for c in (os.path.join("profiles", f"r0{r}_pmc_per_launch.json") for r in (6, 5, 4))
                    if os.path.exists(os.path.join(ROOT, c))), os.path.join("profiles", "r04_pmc_per_launch.json"))
        try:
            pmc = json.load(open(os.path.join(ROOT, src)))
            for name, vals in pmc.items():
                if kname in name and "hbm_traffic_bytes_per_launch" in vals:
                    traffic, traffic_source = float(vals["hbm_traffic_bytes_per_launch"]), src
        except (OSError, ValueError):
            pass
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                "kernel": kname, "kernel_ms": round(kern_ms, 4),
                "kernel_ms_min": round(min(in_step), 4) if in_step else None, "kernel_ms_max": round(max(in_step), 4) if in_step else None,
                "kernel_ms_diff": round(kern_ms_diff, 4), "kernel_ms_loop": round(kern_ms_loop, 4),
                "kernel_ms_method": kern_method + "; kernel_ms_diff: [step + 1 extra launch] - [step], 5 x 20 iterations, median; "
                                    "kernel_ms_loop: 20 back-to-back re-launches",
                "algorithmic_gflop_per_launch": round(main_gf, 2), "algorithmic_gflop_per_step": round(step_gf, 2),
                "intra_folded": bool(intra_folded), "cd_in_mask_kernel": cd_outside,
                # the shader clock the kernel's CUs held while they ran it (sum of the workgroups' s_memtime cycles / sum of their
                # 100-MHz wall ticks, same 16 in-step samples as kernel_ms) and the fraction of the peak AT THAT CLOCK: the
                # 2.5-PFLOP/s peak is 1024 SIMDs x 1024 flop/cycle at 2.4 GHz.  Box-to-box differences of `frac` with equal
                # cycles per launch are differences of this clock (power management), not of the kernel
                "held_clock_ghz": round(held_ghz, 4) if held_ghz else None,
                "frac_at_held_clock": round(achieved / PEAK_BF16_TFLOPS * PEAK_CLOCK_GHZ / held_ghz, 4) if held_ghz else None,
                "kernel_mcycles": round(kern_ms * 1e-3 * held_ghz * 1e9 / 1e6, 3) if held_ghz else None,
                "stamps_cost_ms_per_step": round(t_probe - t_plain, 5)}

    if H["S"] ** 2 <= 160:
        # Sample grids of <= 160 positions (configs 2-4: every recipe the reference ships): the step is a few MB of MFMA work behind
        # samplers that read the whole maps - bytes and launch latencies, not flops.  The yardstick is HBM: the bytes that MUST move
        # (the four maps and the depth maps read once, the two code gradients written once) over the step time, against 8 TB/s.  The
        # fused kernel's MFMA figures stay on the line as `mfma_*` (they say how little of the step the contraction is).
        nmaps = H["B"] * H["h"] * H["w"] * 4.0
        depth_b = H["B"] * float(H["depth_hw"]) ** 2 * 4.0
        compulsory = 2 * nmaps * H["C"] + 2 * nmaps * H["D"] + 2 * nmaps * H["D"] + (2 if conf["sampling"] == "fps" else 1) * depth_b
        hbm = compulsory / (ms_per_step * 1e-3) / 1e9
        roofline = {"bound": "hbm", "achieved": round(hbm, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(hbm / PEAK_HBM_GBPS, 4),
                    "traffic": None, "compulsory_bytes_per_step": int(compulsory),
                    "what": "compulsory bytes of one step (feats + code maps and the depth maps read once, d/d code and d/d code_pos written "
                            "once) / ms_per_step; the step's kernels also write and re-read the sampled rows (a design choice, not counted)",
                    "mfma": roofline}
    if rank == 0:
        line = {
            "metric": "correlation-loss steps/sec at B=32,C=384,28x28 (fwd+bwd, per-GPU batch 32)" if args.config == "headline"
                      else f"correlation-loss steps/sec, {args.config} (fwd+bwd, per-GPU batch {H['B']})",
            "value": round(value, 2), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "host_ms_per_step": round(host_elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 (feats) / f16 (code) MFMA inputs, f32 accumulate", "data": "synthetic",
            "config": {"workload": conf["what"] + (" [step replayed from a hipGraph]" if (graph_mode and not capture_note[0]) else " [eager step]"),
                       "name": args.config, "schedule": capture_note[0] or ("hipGraph replay" if graph_mode else "eager"),
                       "exact_masks": bool(args.exact_masks),
                       "random_draws": ("device-resident generator (coordinates, batch maps, Dropout2d masks: one launch each)" if graph_mode
                                        else "torch generator for coordinates and masks, library seed for the batch maps"),
                       "global_batch": H["B"] * world, "parallelism": f"dp{world}",
                       "ranks_seen": dist_diag["ranks_seen"] if dist_diag else 1,
                       "clock_warmup_steps": clock_warmup_steps,
                       "allreduce_elems": HEAD_GRAD_ELEMS if use_dist else 0,
                       "allreduce": exchange_mode},
            "loss_total": float(total.detach()),
            "roofline": roofline,
        }
        if dist_diag:
            line["dist"] = dist_diag
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(conf)
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its banner to the C-level stdout: flush that first, so that the JSON line is the LAST thing on stdout
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-leg":      # child of cpu_baseline(): one timing leg, never touches the GPU
        _name, _n, _Bs, _budget = sys.argv[2].split(",")
        print(json.dumps(_cpu_baseline_at(CONFIGS[_name], int(_n), float(_budget), int(_Bs))))
        sys.exit(0)
    main()
