#!/usr/bin/env python3
"""Headline benchmark: particle-edge updates / s of one EncodeProcessDecode forward.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A "step" is one forward of the hot path (encoder + L message-passing rounds + decoders,
reference graph_network.py:154-164) over a prebuilt periodic k-NN graph of a synthetic
uniform particle box resident in HBM.  Default workload = BASELINE.json configs[2]:
1,000,000 particles, k=16, latent=128, 10 rounds, bf16 edge MLP with f32 accumulation
(the node path stays at f32 accuracy so the outputs keep the 1e-5 gate).
metric value = E * L * (ranks) / wall time per step, max over ranks.

With N > 1 every rank owns one spatial tile of the box and ghost-node latents are exchanged over
RCCL each round (cosmology_gnn_simulation_amd/dist.py).  ``--scaling weak`` (default): --particles
per GPU, i.e. an N-times larger box; ``--scaling strong``: --particles in total, split over the N
tiles (BASELINE.json: cfg3 = 1M on 1/2/4/8 GPUs, ``--config cfg4`` = 4M / 8, ``--config cfg5`` = 1M,
k=32, latent 256, 15 rounds / 8).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0
MFMA_F32_PEAK_TFLOPS = 157.3


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--particles", type=int, default=None,
                   help="particles per GPU (--scaling weak) or in total (--scaling strong); default 1,000,000")
    p.add_argument("--scaling", default=None, choices=["weak", "strong"],
                   help="N > 1: weak = --particles per GPU (default), strong = --particles in total")
    p.add_argument("--config", default=None, choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"],
                   help="a BASELINE.json configuration: sets particles / neighbours / latent / rounds / precisions "
                        "(cfg4, cfg5: strong scaling of their total particle count)")
    p.add_argument("--neighbors", type=int, default=None)
    p.add_argument("--latent", type=int, default=None)
    p.add_argument("--hidden", type=int, default=None)
    p.add_argument("--mp-steps", type=int, default=None)
    p.add_argument("--hidden-layers", type=int, default=2)
    p.add_argument("--edge-precision", default=None, choices=["bf16", "fp32", "fp16x2"],
                   help="fp16x2 = f32 accuracy from two fp16 terms on the matrix cores (latent = hidden = 128), else exact f32")
    p.add_argument("--node-precision", default=None, choices=["bf16", "fp32", "fp32x3", "fp16x2"],
                   help="f32 emulated on the matrix cores, both hold the 1e-5 gate: fp32x3 = three bf16 terms, six products; "
                        "fp16x2 = two fp16 terms, three products (half the matrix work; |activation| < 65504)")
    p.add_argument("--message-source", default="x_j", choices=["x_j", "edge"])
    p.add_argument("--no-fuse-rounds", action="store_true",
                   help="x_j mode: one edge-kernel launch per round instead of cgnn_edge_stream (all rounds in one launch)")
    p.add_argument("--edge-stream-kernel", default="tile32w", choices=["tile32w", "tile32", "tile16"],
                   help="one-launch edge stream: tile32w = two waves per SIMD (latent 128, fixed-k graphs; other shapes take "
                        "tile32), tile32 = one wave per SIMD with two tiles, tile16 = the first generation")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-particles", type=int, default=16384, help="bounded CPU-baseline sample size")
    p.add_argument("--cpu-threads", type=int, default=None, help="host threads for the CPU baseline (default: all)")
    p.add_argument("--seed", type=int, default=None)   # 1234 + configuration number
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="process-group backend for --gpus > 1 (gloo = single-GPU rehearsal, staged through the host)")
    p.add_argument("--hip-graph", action="store_true",
                   help="replay the forward from a captured HIP graph (launch-bound sizes; single GPU)")
    p.add_argument("--no-strong-leg", action="store_true",
                   help="N > 1, weak scaling: skip the second timed leg (the same --particles box split over the N ranks)")
    p.add_argument("--no-cpu-cfg2", action="store_true",
                   help="do not run BASELINE cfg2's complete CPU forward (about a minute of host time); report the extrapolation")
    p.add_argument("--check", action="store_true",
                   help="N > 1: also run the unsharded forward on rank 0 and compare (small sizes only)")
    a = p.parse_args()
    # BASELINE.json configs[0..4] (N, k, latent, rounds, edge / node arithmetic, seed 1234 + cfg)
    presets = {"cfg1": (4096, 8, 64, 5, "fp32", "fp32", 1235, "weak"), "cfg2": (262144, 16, 128, 10, "fp16x2", "fp16x2", 1236, "weak"),
               "cfg3": (1_000_000, 16, 128, 10, "bf16", "fp16x2", 1237, "weak"),
               "cfg4": (4_000_000, 16, 128, 10, "bf16", "fp16x2", 1238, "strong"),
               "cfg5": (1_000_000, 32, 256, 15, "bf16", "fp16x2", 1239, "strong")}
    pre = presets[a.config or "cfg3"]
    for name, val in zip(("particles", "neighbors", "latent", "mp_steps", "edge_precision", "node_precision", "seed"), pre):
        if getattr(a, name) is None:
            setattr(a, name, val)
    if a.scaling is None:
        a.scaling = pre[7] if a.config in ("cfg4", "cfg5") else "weak"
    return a


def _usable_cpus():
    """Host threads this process may really use: os.cpu_count() capped by the cgroup CPU quota (the GPU box gives a
    share of its host)."""
    n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return n


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(args, dev_outputs=None):
    """The oracle (pure-torch op-for-op restatement of the reference forward: real PyG cannot exist on the box) timed on
    this host's cores, as BASELINE.md section 5 plans it, on bounded samples (about 15 s in all):
      * the headline workload's shape (k / latent / rounds) on fewer particles -> `value`;
      * cfg1 (4,096 particles, k=8, latent 64, 5 rounds) complete, on its real periodic k-NN graph;
      * the k-NN build the reference runs on the CPU (torch_cluster over the 27x ghost-extended set), stood in for
        by scipy's cKDTree on the same extended set, one thread and all threads."""
    import time
    import numpy as np
    from cosmology_gnn_simulation_amd import synthetic
    from oracle import cpu_ref
    n, k, d, L = args.cpu_particles, args.neighbors, args.latent, args.mp_steps
    h = args.hidden or d
    cores = max(1, min(args.cpu_threads or _usable_cpus(), os.cpu_count() or 1))
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(args.seed)
    x = torch.randn(n, 17, generator=g)
    # graph structure does not change the CPU cost model: random senders, fixed in-degree, self loop first
    snd = torch.randint(0, n, (n, k), generator=g)
    snd[:, 0] = torch.arange(n)
    ei = torch.stack([snd.reshape(-1), torch.arange(n).repeat_interleave(k)])
    ea = torch.randn(n * k, 4, generator=g) * 0.05
    sd = synthetic.make_state_dict(d, h, args.hidden_layers, L, 3)
    sec = cpu_ref.time_forward(sd, x, ei, ea, args.hidden_layers, L, repeats=1)
    out = {"value": n * k * L / sec, "unit": "edge-updates/s", "cores": torch.get_num_threads(), "kind": "port",
           "cpu_model": _cpu_model(), "host_cpus": os.cpu_count(),
           "sample": f"oracle/cpu_ref.encode_process_decode, 1 forward, N={n} k={k} latent={d} L={L} fp32 "
                     f"({sec:.2f} s on {cores} host threads)"}
    # cfg2 (262,144 particles) has the sample's k / latent / rounds: its CPU forward is the sample's rate on 16x the edges
    # (BASELINE.md section 5 allows the flagged extrapolation; the whole forward would take about a minute of host time)
    if (k, d, L, h, args.hidden_layers) == (16, 128, 10, 128, 2):
        out["cfg2"] = cpu_cfg2_forward(args, cores, out["value"], n)
    # cfg1, complete: window -> 27-image k-NN graph (cpu_ref.preprocess) -> forward
    snap = synthetic.make_snapshot(4096, seed=1235)
    meta = synthetic.make_metadata()
    t0 = time.perf_counter()
    g1 = cpu_ref.preprocess(snap["Coordinates"][:5], snap["InternalEnergy"][:5], meta, None, None, 0.0, 8, meta["dt"],
                            meta["box_size"])
    t_graph = time.perf_counter() - t0
    sd1 = synthetic.make_state_dict(64, 64, 2, 5, 3)
    sec1 = cpu_ref.time_forward(sd1, g1["x"], g1["edge_index"], g1["edge_attr"], 2, 5, repeats=3)
    out["cfg1"] = {"edge_updates_per_s": 4096 * 8 * 5 / sec1, "forward_ms": round(sec1 * 1e3, 2),
                   "graph_build_ms": round(t_graph * 1e3, 1),
                   "sample": "4096 particles, k=8, latent 64, 5 rounds, fp32, real periodic k-NN graph, best of 3"}
    # k-NN on the host: cKDTree over the 27 periodic images (reference data_utils.py:9-33,149 semantics)
    try:
        from scipy.spatial import cKDTree
        nk = min(args.particles, 65536)
        pos = np.random.default_rng(args.seed).random((nk, 3), dtype=np.float32)
        shifts = np.array([[i, j, l] for i in (-1.0, 0.0, 1.0) for j in (-1.0, 0.0, 1.0) for l in (-1.0, 0.0, 1.0)], np.float32)
        t0 = time.perf_counter()
        ext = (pos[None, :, :] + shifts[:, None, :]).reshape(-1, 3)
        tree = cKDTree(ext)
        t_build = time.perf_counter() - t0
        t0 = time.perf_counter()
        tree.query(pos, k=k, workers=1)
        t_q1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        tree.query(pos, k=k, workers=-1)
        t_qa = time.perf_counter() - t0
        out["knn"] = {"particles": nk, "k": k, "tree_build_s": round(t_build, 3), "query_1_thread_s": round(t_q1, 3),
                      "query_all_threads_s": round(t_qa, 3),
                      "edges_per_s_all_threads": nk * k / (t_build + t_qa),
                      "sample": "scipy cKDTree over the 27x extended set (stand-in for torch_cluster's nanoflann)"}
    except Exception as exc:      # scipy missing or out of memory: report, do not fail the bench
        out["knn"] = {"error": repr(exc)}
    return out


def cpu_cfg2_forward(args, cores, sample_rate, sample_n):
    """the oracle's forward at cfg2 (262,144 particles, k=16, latent 128, 10 rounds, fp32), complete,
    once, on this host's cores -- about a minute.  The torch-op restatement materialises [E, 3D] and friends (about 12 GB at
    this size): with less than 24 GB of host memory available, or with --no-cpu-cfg2, the flagged extrapolation from the
    bounded sample is reported instead."""
    import time
    from cosmology_gnn_simulation_amd import synthetic
    from oracle import cpu_ref
    n2, k2, d2, L2 = 262144, 16, 128, 10
    extrap = {"edge_updates_per_s": sample_rate, "extrapolated": True, "forward_s": round(n2 * k2 * L2 / sample_rate, 1),
              "sample": f"extrapolated from the N={sample_n} forward above by the edge-count ratio (same k, latent, rounds)"}
    if args.no_cpu_cfg2:
        return extrap
    try:
        avail = 0
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) * 1024
        for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            try:
                txt = open(path).read().strip()
                if txt != "max":
                    avail = min(avail, int(txt))
            except (OSError, ValueError):
                pass
    except OSError:
        avail = 0
    if avail < 24 * 2**30:
        extrap["sample"] += f" (host memory available {avail / 2**30:.0f} GiB < 24 GiB: the complete forward was not run)"
        return extrap
    g = torch.Generator().manual_seed(1236)
    x = torch.randn(n2, 17, generator=g)
    snd = torch.randint(0, n2, (n2, k2), generator=g)
    snd[:, 0] = torch.arange(n2)
    ei = torch.stack([snd.reshape(-1), torch.arange(n2).repeat_interleave(k2)])
    ea = torch.randn(n2 * k2, 4, generator=g) * 0.05
    sd = synthetic.make_state_dict(d2, d2, 2, L2, 3)
    t0 = time.perf_counter()
    sec = cpu_ref.time_forward(sd, x, ei, ea, 2, L2, repeats=1)
    wall = time.perf_counter() - t0
    return {"edge_updates_per_s": n2 * k2 * L2 / sec, "extrapolated": False, "forward_s": round(sec, 1),
            "sample": f"oracle/cpu_ref.encode_process_decode, 1 complete forward, N={n2} k={k2} latent={d2} L={L2} fp32, "
                      f"fixed in-degree random senders ({sec:.1f} s on {cores} host threads; {wall:.1f} s incl. set-up)"}


def _traffic(key, kernel):
    """HBM bytes per launch from the PMC passes committed under profiles/ (profiles/traffic.json).  An entry is only
    reported while the kernel source it was measured on is unchanged (sha256 of the .hip file and the headers it
    includes): a stale number is worse than none."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        db = json.load(open(path))
    except Exception:
        return None
    ent = db.get(key)
    if not isinstance(ent, dict):
        return None
    try:
        sha = kernel_source_sha16(ent.get("source", ""))
    except OSError:
        return None
    if sha != ent.get("source_sha16"):
        return None
    return ent.get("hbm_bytes_per_launch")


def _traffic_extra(key):
    """Clock and matrix-pipe occupancy recorded with a traffic.json entry (same validity rule as `_traffic`)."""
    if _traffic(key, "") is None:
        return None
    try:
        ent = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[key]
    except Exception:
        return None
    return {k: ent[k] for k in ("clock_ghz_under_profiler", "mfma_busy_frac", "kernel_ms_under_profiler") if k in ent} or None


def kernel_source_sha16(name):
    """sha256 (first 16 hex digits) of a kernel source file together with every csrc header it includes, directly or
    not: the key under which profiles/traffic.json remembers what a PMC pass was measured on.  (include/cgnn.h, the
    public declarations, is left out: it changes with every new entry point and carries no kernel code.)"""
    import hashlib
    import re
    csrc = os.path.join(ROOT, "cosmology_gnn_simulation_amd", "csrc")
    seen, todo = [], [name]
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.append(f)
        text = open(os.path.join(csrc, f), "r").read()
        todo += [h for h in re.findall(r'#include "([^"]+)"', text) if os.path.exists(os.path.join(csrc, h))]
    h = hashlib.sha256()
    for f in sorted(seen):
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def _check_against_unsharded(args, model, sharded, runner, dev, rank, world, meta, k):
    """Every rank compares its owned rows with the unsharded forward of the same global box (bit-exact: each
    receiver sums its senders in the same order)."""
    import torch.distributed as dist
    from cosmology_gnn_simulation_amd import data_utils, synthetic
    per_rank = args.particles if args.scaling == "weak" else args.particles // world
    snap = synthetic.make_snapshot(per_rank * world, seed=args.seed)
    g = data_utils.preprocess(snap["Coordinates"][:5], snap["InternalEnergy"][:5], meta, None, None, 0.0, k,
                              meta["dt"], meta["box_size"], device=dev)
    with torch.no_grad():
        want = model(g)
        got = runner()
    ok = all(torch.equal(got[key], want[key][sharded.owned_global]) for key in ("acceleration", "temp_rate"))
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN) if dist.get_backend() == "gloo" else None
    print(f"[rank {rank}] sharded == unsharded on owned rows: {ok}", file=sys.stderr, flush=True)
    if not ok:
        raise SystemExit("sharded forward differs from the unsharded one")


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if args.backend == "gloo":          # rehearsal: every rank shares the visible GPU(s)
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from cosmology_gnn_simulation_amd import data_utils, graph_network, ops, synthetic

    d = args.latent
    h = args.hidden or d
    k, L = args.neighbors, args.mp_steps
    # particles per rank: --particles each (weak) or --particles split over the ranks (strong)
    per_rank = args.particles if (world == 1 or args.scaling == "weak") else args.particles // world
    model = graph_network.EncodeProcessDecode(d, h, args.hidden_layers, L, 3)
    model.load_state_dict(synthetic.make_state_dict(d, h, args.hidden_layers, L, 3))
    model = model.to(dev).eval()
    model.edge_precision, model.node_precision = args.edge_precision, args.node_precision
    model.message_source = args.message_source
    model.fuse_rounds = not args.no_fuse_rounds
    model.edge_stream_kernel = args.edge_stream_kernel

    dist_ctx = None
    if world > 1:
        import torch.distributed as dist
        from cosmology_gnn_simulation_amd import dist as cdist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        dist_ctx = cdist

    # ---- synthetic input, resident in HBM before the timed region -----------------------------------
    meta = synthetic.make_metadata()
    t0 = time.perf_counter()
    if world == 1:
        snap = synthetic.make_snapshot(per_rank, seed=args.seed)
        graph = data_utils.preprocess(snap["Coordinates"][:5], snap["InternalEnergy"][:5], meta, None, None, 0.0, k,
                                      meta["dt"], meta["box_size"], device=dev)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        # graph build alone (k-NN + edge features), device resident positions
        with ops.OpTimer() as tm:
            data_utils.knn_graph_periodic(graph.pos, meta["box_size"], k)
        knn_ms = tm.summary()["knn_periodic"][1]
        n_local, e_local = graph.x.shape[0], graph.edge_index.shape[1]
        run = lambda: model(graph)  # noqa: E731
        # end to end from HOST buffers (never `value`): window H2D + graph build + forward + outputs D2H
        win_p, win_t = snap["Coordinates"][:5].contiguous(), snap["InternalEnergy"][:5].contiguous()
        with torch.no_grad():
            model(graph)                                                   # pack weights, warm caches
            torch.cuda.synchronize()
            e2e = []
            for _ in range(3):
                t1 = time.perf_counter()
                g2 = data_utils.preprocess(win_p, win_t, meta, None, None, 0.0, k, meta["dt"], meta["box_size"],
                                           device=dev, reference_rng=False)
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                o2 = model(g2)
                host = (o2["acceleration"].cpu(), o2["temp_rate"].cpu())   # noqa: F841
                t3 = time.perf_counter()
                e2e.append((t2 - t1, t3 - t2))
            e2e_build, e2e_fwd = min(a for a, _ in e2e), min(b for _, b in e2e)
        if args.hip_graph:
            from cosmology_gnn_simulation_amd.graphed import GraphedForward
            run = GraphedForward(model, graph)
    else:
        sharded = dist_ctx.build_synthetic_shard(per_rank, world, rank, k, args.seed, dev, meta)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        knn_ms = sharded.knn_ms
        n_local, e_local = sharded.n_owned, sharded.n_owned * k
        runner = dist_ctx.ShardedForward(model, sharded)
        run = runner  # noqa: E731

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            run()
        barrier()
        with ops.OpTimer() as tm:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run()
            barrier()
            elapsed = time.perf_counter() - t0
        per_op = tm.summary()
        packed_now = model._packed[1] if getattr(model, "_packed", None) else {}
        stream_kernel = ("cgnn::edge_stream32w_kernel" if packed_now.get("image_w8") is not None else
                         "cgnn::edge_stream32_kernel" if packed_now.get("image") is not None else
                         "cgnn::edge_stream_n16_kernel")
        # BASELINE.md section 2 asks for the median of >= 10 synchronised iterations: measured next to the contract's
        # K back-to-back steps (which `value` comes from), one device synchronisation per iteration
        sync_ms = []
        for _ in range(max(10, args.steps)):
            barrier()
            t1 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            sync_ms.append((time.perf_counter() - t1) * 1e3)
        sync_ms.sort()
        median_ms = sync_ms[len(sync_ms) // 2]

    # Reference point inside the same run (single GPU, outside the timed region): the same forward with one
    # HBM-bound edge-kernel launch per round instead of cgnn_edge_stream.
    per_round = None
    if world == 1 and model.fuse_rounds and not args.hip_graph and "edge_stream" in per_op:
        model.fuse_rounds = False
        with torch.no_grad():
            run()
            torch.cuda.synchronize()
            with ops.OpTimer() as tm2:
                t1 = time.perf_counter()
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                dt_pr = (time.perf_counter() - t1) / 3
            eb_calls, eb_ms = tm2.summary()["edge_block"]
        model.fuse_rounds = True
        eb_bytes = 2 * e_local * d * 4 + 2 * e_local * 4 + 2 * n_local * h * (2 if args.edge_precision == "bf16" else 4)
        per_round = {"ms_per_step": round(dt_pr * 1e3, 3), "edge_updates_per_s": e_local * L / dt_pr,
                     "edge_kernel_avg_ms": round(eb_ms / eb_calls, 4),
                     "edge_kernel_hbm_GBps": round(eb_bytes / (eb_ms / eb_calls * 1e-3) / 1e9, 1),
                     "edge_kernel_hbm_frac": round(eb_bytes / (eb_ms / eb_calls * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "note": "same forward, one edge-kernel launch per round (--no-fuse-rounds), 3 steps, untimed region"}

    strong_leg = None
    if world > 1:
        import torch.distributed as dist
        cdev = dev if args.backend == "nccl" else torch.device("cpu")

        def over_ranks(seconds, edges):      # the contract's reduction: MAX of the ranks' times, SUM of their edges
            t = torch.tensor([seconds], device=cdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cnt = torch.tensor([float(edges)], device=cdev, dtype=torch.float64)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            return float(t.item()), int(cnt.item())

        elapsed, e_total = over_ranks(elapsed, e_local)
        if args.check:
            _check_against_unsharded(args, model, sharded, runner, dev, rank, world, meta, k)
        if args.scaling == "weak" and not args.no_strong_leg:
            # BASELINE.json quotes the metric "at 1M particles ... 1/2/4/8 GPU", i.e. STRONG scaling of the headline box.
            # `value` above is the weak-scaling job the contract's `scaling` field describes (--particles per GPU); this
            # second, equally timed leg splits the SAME --particles box over the N ranks, so that one run of the driver's
            # N = 1, 2, 4, 8 sequence yields both curves (N = 1 is the same job in both).
            del runner, sharded
            torch.cuda.empty_cache()
            sh2 = dist_ctx.build_synthetic_shard(args.particles // world, world, rank, k, args.seed, dev, meta)
            run2 = dist_ctx.ShardedForward(model, sh2)
            with torch.no_grad():
                for _ in range(args.warmup):
                    run2()
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    run2()
                barrier()
                el2 = time.perf_counter() - t0
            el2, e2_total = over_ranks(el2, sh2.n_owned * k)
            strong_leg = {"scaling": "strong", "particles_total": args.particles, "particles_per_gpu": args.particles // world,
                          "owned_particles_rank0": sh2.n_owned, "ghost_particles_rank0": int(sh2.n_local - sh2.n_owned),
                          "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": el2 / args.steps * 1e3, "value": e2_total * L / (el2 / args.steps),
                          "unit": "edge-updates/s",
                          "note": "the same --particles box split over the N ranks (BASELINE.json: 'at 1M particles ... "
                                  "1/2/4/8 GPU'); timed like `value`: barrier + synchronize on both sides, max over ranks"}
    else:
        e_total = e_local
    ms_per_step = elapsed / args.steps * 1e3
    value = e_total * L / (elapsed / args.steps)

    if rank == 0:
        # ---- roofline of the dominant kernel, from HIP events on the launch stream inside the timed region ----
        sz_p = 2 if args.edge_precision == "bf16" else 4
        mfma_peak = MFMA_BF16_PEAK_TFLOPS if args.edge_precision == "bf16" else MFMA_F32_PEAK_TFLOPS
        # per round: MFMA flops of the edge model as EXECUTED (first Linear split by columns: only the We block runs per
        # edge; the sender / receiver thirds are per-node work in the node kernel) and as the reference formulates it
        # (3D-wide first Linear per edge, SURVEY 8(d): 10 D^2 = 163,840 flop per edge update at D = 128)
        flops_exec = 2.0 * e_local * (d * h + (args.hidden_layers - 1) * h * h + h * d)
        flops_alg = 2.0 * e_local * (3 * d * h + (args.hidden_layers - 1) * h * h + h * d)
        if "edge_stream" in per_op:
            # all L rounds in one launch (reference data flow): the edge latents cross HBM once, the kernel is bound by
            # the matrix pipe.  `achieved` / `frac` = flops the kernel really issues over the hardware peak.
            calls, total_ms = per_op["edge_stream"]
            edge_ms = total_ms / calls
            # the edge encoder runs inside the same launch when it has the rounds' shape (then only the node encoder
            # and the two decoders are left as mlp_rows calls): its output is never written, its input is E x 4 floats
            enc_fused = per_op.get("mlp_rows", (0, 0.0))[0] <= 3 * calls
            enc_flops = 2.0 * e_local * (4 * h + (args.hidden_layers - 1) * h * h + h * d) if enc_fused else 0.0
            alg_bytes = ((e_local * d * 4 + e_local * 16) if enc_fused else 2 * e_local * d * 4) + 2 * e_local * 4 + \
                L * 2 * n_local * h * sz_p
            tf_exec = (L * flops_exec + enc_flops) / (edge_ms * 1e-3) / 1e12
            tf_ref = (L * flops_alg + enc_flops) / (edge_ms * 1e-3) / 1e12
            kname = stream_kernel
            roofline = {"kernel": f"{kname}<{d // 32}>", "bound": "mfma",
                        "achieved": round(tf_exec, 1), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(tf_exec / mfma_peak, 4),
                        "traffic": _traffic(f"edge_stream{'+enc' if enc_fused else ''}:{n_local}:{k}:{d}:{L}", kname),
                        # from the same PMC passes (profiles/traffic.json): the clock the chip held under this kernel
                        # (GRBM_GUI_ACTIVE / 8 / kernel time; the peak is quoted at 2.4 GHz) and the share of cycles the
                        # matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES / SIMDs / cycles, selector MFMAs included)
                        "pmc": _traffic_extra(f"edge_stream{'+enc' if enc_fused else ''}:{n_local}:{k}:{d}:{L}"),
                        "avg_launch_ms": round(edge_ms, 4), "launches": calls,
                        "executed_flops_per_launch": L * flops_exec + enc_flops,
                        "edge_encoder_in_launch": bool(enc_fused), "encoder_flops_per_launch": enc_flops,
                        "reference_equivalent": {
                            "flops_per_launch": L * flops_alg + enc_flops, "tflops": round(tf_ref, 1),
                            "flops_per_edge_update": flops_alg / e_local,
                            "note": "the reference's formulation (3D-wide first Linear per edge, SURVEY 8(d)); 10/6 of the "
                                    "executed flops -- NOT a utilisation figure"},
                        "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                                "achieved_GBps": round(alg_bytes / (edge_ms * 1e-3) / 1e9, 1), "peak_GBps": HBM_PEAK_GBS,
                                "frac": round(alg_bytes / (edge_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
        else:
            if "edge_block" not in per_op:      # HIP-graph replay: launches are inside the graph, no per-op events
                per_op = dict(per_op, edge_block=(1, float("nan")))
            calls, total_ms = per_op["edge_block"]
            edge_ms = total_ms / calls
            # algorithmic HBM bytes per launch: f32 edge latents read once + written once, src/dst indices, and the
            # per-node Ps/Pd tables read once (gather re-reads are cache traffic, not algorithmic)
            alg_bytes = 2 * e_local * d * 4 + 2 * e_local * 4 + 2 * n_local * h * sz_p
            achieved = alg_bytes / (edge_ms * 1e-3) / 1e9
            n16 = args.edge_precision == "bf16" and d <= 128 and h <= 128
            ring256 = args.edge_precision == "bf16" and d == 256 and h == 256 and args.hidden_layers <= 3
            f2 = args.edge_precision == "fp16x2" and d == 128 and h == 128 and args.hidden_layers <= 3
            edge_kernel_name = (f"cgnn::edge_block_n16_kernel<{h // 32},{d // 32}>" if n16 else
                                f"cgnn::edge_block_ring256_kernel<{args.hidden_layers}, false>" if ring256 else
                                f"cgnn::edge_block_f2ring_kernel<{args.hidden_layers}, false>" if f2 else
                                f"cgnn::edge_block_kernel<{'fp32' if args.edge_precision == 'fp16x2' else args.edge_precision},"
                                f"{h // 32},{d // 32}>")
            if f2:      # f32 emulated by three fp16 products per element on the fp16 matrix cores
                flops_exec, mfma_peak = 3.0 * flops_exec, MFMA_BF16_PEAK_TFLOPS
            roofline = {"kernel": edge_kernel_name, "bound": "hbm", "achieved": round(achieved, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": _traffic(f"edge_block:{n_local}:{k}:{d}:{args.edge_precision}", "cgnn::edge_block"),
                        "avg_launch_ms": round(edge_ms, 4), "launches": calls,
                        "algorithmic_bytes_per_launch": alg_bytes,
                        "mfma": {"executed_tflops": round(flops_exec / (edge_ms * 1e-3) / 1e12, 1),
                                 "reference_equivalent_tflops": round(flops_alg / (edge_ms * 1e-3) / 1e12, 1),
                                 "peak_tflops": mfma_peak,
                                 "frac_executed": round(flops_exec / (edge_ms * 1e-3) / 1e12 / mfma_peak, 4),
                                 "note": ("fp16x2: three fp16 MFMA products per f32 product (executed flops = 3 x the f32 "
                                          "count), priced against the dense 16-bit peak") if f2 else ""}}
        kernels = {name: {"calls": c, "avg_ms": round(ms / c, 4)} for name, (c, ms) in sorted(per_op.items())}
        if "aggregate" in per_op:
            c, ms = per_op["aggregate"]
            # rows gathered (k per receiver, mostly served by LDS / L2) against the compulsory traffic (every table row
            # and index once, the sums once): the first is a cache-side rate, only the second is priced against HBM
            agg_bytes = e_local * d * 4 + e_local * 4 + n_local * d * 4
            compulsory = 2 * n_local * d * 4 + e_local * 4
            kernels["aggregate"]["gathered_GBps"] = round(agg_bytes / (ms / c * 1e-3) / 1e9, 1)
            kernels["aggregate"]["compulsory_bytes"] = compulsory
            kernels["aggregate"]["compulsory_frac_of_hbm_peak"] = round(compulsory / (ms / c * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            # measured HBM bytes per launch (PMC pass, profiles/): what really crosses the memory interface
            tr = _traffic(f"aggregate:{n_local}:{k}:{d}", "cgnn::aggregate_fixedk_kernel")
            kernels["aggregate"]["traffic"] = tr
            if tr:
                kernels["aggregate"]["hbm_measured_frac"] = round(tr / (ms / c * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if "node_block" in per_op:
            c, ms = per_op["node_block"]
            tr = _traffic(f"node_block:{n_local}:{d}", "cgnn::node_block_f2ring_kernel")
            kernels["node_block"]["traffic"] = tr
            kernels["node_block"]["algorithmic_bytes"] = 3 * n_local * d * 4 + 2 * n_local * d * 2    # x, agg in; x out; Ps, Pd
            if tr:
                kernels["node_block"]["hbm_measured_frac"] = round(tr / (ms / c * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args)
        line = {
            "metric": "particle-edge updates/sec (E x MP-steps)", "value": value, "unit": "edge-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "median_ms_synchronised": round(median_ms, 4),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp16x2": "f32 (two fp16 terms per operand, f32 accumulate)"}.get(args.edge_precision, "f32"),
            "data": "synthetic",
            "config": {"workload": f"{'BASELINE ' + args.config + ': ' if args.config else ''}"
                                   f"{per_rank} particles/GPU ({per_rank * world} in all, {args.scaling} scaling) "
                                   f"uniform periodic box, k={k}, latent={d}, "
                                   f"hidden={h}, {L} MP rounds, edge MLP {args.edge_precision} (f32 accumulate, f32 "
                                   f"latents/LayerNorm/residual), node path {args.node_precision}, "
                                   f"message_source={args.message_source}",
                       "particles_per_gpu": per_rank, "edges_per_gpu": e_local, "k": k, "latent": d,
                       "mp_steps": L, "parallelism": "single GPU" if world == 1 else f"{world} spatial tiles + halo"},
            "roofline": roofline, "cpu_baseline": cpu, "kernels": kernels, "one_launch_per_round": per_round,
            "strong_scaling": strong_leg,
            "graph_build": {"knn_ms": round(knn_ms, 3), "snapshot_plus_preprocess_s": round(t_build, 3)},
            "end_to_end_from_host": None if world > 1 else {
                "preprocess_ms_incl_h2d": round(e2e_build * 1e3, 2), "forward_ms_incl_d2h": round(e2e_fwd * 1e3, 2),
                "edge_updates_per_s": e_local * L / (e2e_build + e2e_fwd),
                "note": "host window -> H2D -> features + k-NN graph -> forward -> outputs D2H (PCIe inclusive)"},
        }
        def _no_nan(o):     # HIP-graph replay has no per-op events: report null, not NaN (strict JSON)
            if isinstance(o, float) and o != o:
                return None
            if isinstance(o, dict):
                return {k_: _no_nan(v_) for k_, v_ in o.items()}
            return o
        print(json.dumps(_no_nan(line)))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
