#!/usr/bin/env python3
"""bench.py — views/sec of the GVCNN hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--shapes S] [--exchange allgather|scores]

Workload (BASELINE.json configs[1]): ModelNet-shaped synthetic views, 12 views x 224x224 x 3 per
shape, Inception-v3 backbone, num_groups = 7, fp32, forward only (inference BatchNorm).  One "step"
is one pass of the whole hot path — folded backbone over all views, scorer, device-side group
assignment, view pooling + group fusion, classifier — over one batch that is already resident in
HBM.  For N > 1 the driver launches one rank per GPU (torch.distributed.run); per-GPU work is fixed
(weak scaling): the batch is cut on shape boundaries, the ranks all-gather the scorer responses over
RCCL (the batch-mean score of nets/model.py:146 is the only coupling), and each rank pools the shapes
it owns; --exchange allgather additionally all-gathers the final view descriptors.

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

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3                    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0                  # MI355X_MICROARCH.md: bf16 MFMA dense (spec)
# fp32 products evaluated as N bf16 MFMA passes (GV_MATH_*): the ceiling in ALGORITHMIC fp32 FLOPs
MATH = {"f32": ("conv_igemm_f32", PEAK_F32_MFMA_TFLOPS, "v_mfma_f32_32x32x2_f32 (exact fp32 chain)"),
        "bf16x3": ("conv_igemm_bf16s<NP=3>", PEAK_BF16_MFMA_TFLOPS / 6,
                   "fp32 operands split into 3 bf16 planes, 6 v_mfma_f32_32x32x16_bf16 per product block, "
                   "fp32 accumulate: fp32-level accuracy (dropped terms <= 2^-24 relative); peak = bf16 dense / 6"),
        "bf16x2": ("conv_igemm_bf16s<NP=2>", PEAK_BF16_MFMA_TFLOPS / 3, "2 bf16 planes, 3 MFMAs (~2^-16 relative)"),
        "bf16x1": ("conv_igemm_bf16s<NP=1>", PEAK_BF16_MFMA_TFLOPS, "plain bf16 products, fp32 accumulate"),
        # 16-bit STORAGE (configs c3-c5; not the bench line, which is fp32): --storage bf16 | f16
        "bf16": ("conv_igemm_lp<bf16>", PEAK_BF16_MFMA_TFLOPS, "bf16 activations/filters in HBM, v_mfma_f32_32x32x16_bf16, fp32 accumulate + epilogue"),
        "f16": ("conv_igemm_lp<f16>", PEAK_BF16_MFMA_TFLOPS, "fp16 activations/filters in HBM, v_mfma_f32_32x32x16_f16, fp32 accumulate + epilogue")}
V, H, W, G, C = 12, 224, 224, 7, 10             # configs[1]; ModelNet10 -> 10 classes
BACKBONE = "inception_v3"
# other BASELINE.json configs, forward pass in the dtype the config names (parity-test cases; not the bench line)
PRESETS = {"c2": ("inception_v3", 12, 224, 7, 10), "c3": ("inception_v3", 12, 224, 7, 40),
           "c4": ("resnet_v2_50", 12, 224, 10, 40), "c5": ("inception_v3", 20, 299, 10, 40)}
PRESET_STORAGE = {"c2": "f32", "c3": "bf16", "c4": "bf16", "c5": "f16"}      # the dtype each config names


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--shapes", type=int, default=32, help="shapes per GPU per step (x12 views)")
    ap.add_argument("--exchange", default="scores", choices=["allgather", "scores"],
                    help="multi-GPU: 'scores' exchanges only the scorer responses (each rank pools the shapes it "
                         "owns); 'allgather' also all-gathers the final view descriptors (north_star form)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--tile-cache", default=None, help="JSON file with measured per-launch tile choices")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-fp32-MFMA reference run")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--preset", default="c2", choices=sorted(PRESETS), help="c2 = BASELINE.json configs[1] (the bench line)")
    ap.add_argument("--no-lanes", action="store_true", help="single-stream launch order (no branch concurrency)")
    ap.add_argument("--math", default="bf16x3", choices=["f32", "bf16x3", "bf16x2", "bf16x1"],
                    help="how fp32 convolutions are evaluated on the matrix cores (GV_MATH_*)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (the real multi-GPU path); gloo + --same-device: control-flow check of the N>1 path "
                         "on a single-GPU box (every rank on cuda:0, collectives through host memory)")
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--dp", default="views", choices=["views", "shapes"],
                    help="--train with N > 1: shard the views of every shape (BN statistics stay local) or the shapes "
                         "(every BN layer all-reduces its per-view sums)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one hipGraph (GVCNN.capture): for small, launch-bound view batches; N = 1 only")
    ap.add_argument("--train", action="store_true",
                    help="time the TRAINING step instead (SURVEY a12 / configs[2]: train-mode forward + loss + backward + "
                         "BN moving averages + Momentum; fp32 storage, bf16x3 math).  N > 1: view-sharded data "
                         "parallelism (any world size up to the number of views)")
    ap.add_argument("--storage", default=None, choices=["f32", "bf16", "f16"],
                    help="activation/filter storage type (default: the preset's own: c2 f32 = the bench line, "
                         "c3/c4 bf16, c5 f16; forward only)")
    a = ap.parse_args()
    if a.storage is None:
        a.storage = PRESET_STORAGE[a.preset]
    if a.storage != "f32":
        a.math = a.storage
    return a


def roofline(eng, x, math, iters=5):
    """Per-launch hipEvent timing (on the launch stream) of every implicit-GEMM conv launch of one
    step; achieved = algorithmic conv FLOPs of the step / summed conv kernel time."""
    plan = eng.plan
    flops = t_ms = 0.0
    n = 0
    worst = None
    for i, op in enumerate(plan.ops):
        if op["kind"] != "conv":
            continue
        ms = plan.time_range(x, i, 1, iters)
        flops += op["flops"]
        t_ms += ms
        n += 1
        tf = op["flops"] / (ms * 1e-3) / 1e12
        if worst is None or ms > worst[1]:
            worst = (op["name"], ms, tf)
    achieved = flops / (t_ms * 1e-3) / 1e12
    kname, peak, how = MATH[math]
    # HBM bytes per conv launch from the committed rocprofv3 PMC passes of this same command (FETCH_SIZE /
    # WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes; tools/profile_round.sh) — fp32 storage only
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r1_h_traffic_pmc.json")
    if math in ("f32", "bf16x3", "bf16x2", "bf16x1") and os.path.exists(tpath):
        try:
            traffic = round(json.load(open(tpath))["conv_hbm_bytes_per_launch"])
        except Exception:
            traffic = None
    alg_bytes = sum(op["bytes"] for op in plan.ops if op["kind"] == "conv") / n
    return {"bound": "mfma", "kernel": "%s<*> (%d launches/step)" % (kname, n), "math": how,
            "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic,
            "traffic_note": "HBM bytes per conv launch, rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE (profiles/r1_h_traffic_pmc.json)",
            "algorithmic_bytes_per_launch": round(alg_bytes),
            "flops_per_step": flops, "avg_launch_us": round(t_ms * 1e3 / n, 2),
            "conv_ms_per_step": round(t_ms, 3),
            "longest_launch": {"name": worst[0], "ms": round(worst[1], 4), "tflops": round(worst[2], 2)}}


def cpu_baseline(P, Hd, seconds):
    """The CPU oracle (a port of the reference graph; TensorFlow itself cannot run here) timed on
    this box's host cores on a bounded sample of the same workload: 1 shape x 12 views per pass,
    reference-shaped (V sequential backbone calls, nets/model.py:129-141)."""
    from oracle import model as OM
    # 16 threads: on a 256-thread host the oneDNN convs of a batch-1 view stop scaling (and
    # collapse from oversubscription) well before that; `cores` reports what was actually used.
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    x = torch.rand(1, V, H, W, 3, generator=torch.Generator().manual_seed(0)) - 0.5
    t0 = time.time()
    OM.gvcnn_scores(x[:, :2].contiguous(), P, Hd, BACKBONE)               # page in, size the sample
    per_pass = (time.time() - t0) * V / 2
    max_passes = 1 if per_pass > seconds else 50     # a cold estimate; the loop below is time-bound
    t0 = time.time()
    passes = 0
    while passes < max_passes:
        OM.gvcnn(x, C, P, Hd, G, BACKBONE, num_bins=G)
        passes += 1
        if time.time() - t0 >= seconds:
            break
    dt = time.time() - t0
    return {"value": round(passes * V / dt, 2), "unit": "views/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d passes of 1 shape x %d views %dx%d %s, torch-CPU fp32 oracle, %.1f s"
                      % (passes, V, H, W, BACKBONE, dt)}


def train_main(a, world, rank, dev):
    """One JSON line for the training step (not the driver's bench line)."""
    from gvcnn_tf_amd.training import TrainGVCNN
    from gvcnn_tf_amd.sharding import ShardedTrainGVCNN, view_shard_range
    N = a.shapes
    if a.dp == "views":
        lo, hi = view_shard_range(V, world, rank)
        eng = TrainGVCNN(BACKBONE, N, hi - lo, H, W, C, G, device=dev, num_bins=G, head_views=V, view_offset=lo,
                         storage=a.storage)
        sh = ShardedTrainGVCNN(eng)
        x = (torch.rand(N, V, H, W, 3, generator=torch.Generator().manual_seed(0)) - 0.5)[:, lo:hi].contiguous().to(dev)
        labels = torch.randint(0, C, (N,), generator=torch.Generator().manual_seed(1))
        views_per_step, views_local = N * V, N * (hi - lo)
    else:                                            # N shapes PER RANK (weak scaling), all their views
        lo, hi = 0, V
        eng = TrainGVCNN(BACKBONE, N, V, H, W, C, G, device=dev, num_bins=G, storage=a.storage)
        sh = ShardedTrainGVCNN(eng, mode="shapes")
        x = (torch.rand(N, V, H, W, 3, generator=torch.Generator().manual_seed(rank)) - 0.5).to(dev)
        labels = torch.randint(0, C, (N,), generator=torch.Generator().manual_seed(100 + rank))
        views_per_step, views_local = N * V * world, N * V
    sh.train_step(x, labels, lr=1e-6)
    eng.autotune()                                   # untimed: per-launch tile choice
    for _ in range(max(a.warmup, 1)):
        sh.train_step(x, labels, lr=1e-6)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        sh.train_step(x, labels, lr=1e-6)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        ms = dt / a.steps * 1e3
        flops = 3.0 * sum(op.get("flops", 0) for op in eng.plan.ops) * world
        print(json.dumps({
            "metric": "views/sec (training step)", "value": round(views_per_step / (ms * 1e-3), 1), "unit": "views/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "strong" if a.dp == "views" else "weak", "vs_baseline": None,
            "dtype": a.storage, "data": "synthetic",
            "config": {"workload": "training step (SURVEY a12): %s, %d shapes x %d views x %dx%d, train-mode BN per view, "
                                   "CE loss, backward, BN moving averages, Momentum; %s; %s sharded "
                                   "over the ranks" % (BACKBONE, N, V, H, W,
                                                       "fp32 storage, bf16x3 math" if a.storage == "f32" else
                                                       a.storage + " activations and gradients on the 16-bit MFMA, fp32 "
                                                       "master weights", a.dp), "views_per_gpu": views_local},
            "step_tflops": round(flops / (ms * 1e-3) / 1e12, 2)}), flush=True)


def main():
    global BACKBONE, V, H, W, G, C
    a = parse()
    BACKBONE, V, H, G, C = PRESETS[a.preset]
    W = H
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run "
                     "--nproc-per-node %d" % (a.gpus, a.gpus))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if a.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import gvcnn_tf_amd as gv
    from gvcnn_tf_amd.sharding import ShardedGVCNN
    if a.train:
        return train_main(a, world, rank, dev)

    N = a.shapes
    eng = gv.GVCNN(BACKBONE, N, V, H, W, C, G, device=dev, num_bins=G, math=a.math if a.storage == "f32" else "f32",
                   lanes=not a.no_lanes, storage=a.storage)
    P = gv.params.init_backbone_params(eng.plan.param_shapes(), seed=2, perturb_bn=True)
    Hd = gv.params.init_head_params(V, eng.raw.c, eng.final.c, C, seed=3, spread_scores=True)
    eng.plan.bind(P)
    eng.set_head(Hd)
    sh = ShardedGVCNN(eng, exchange=a.exchange)
    x = (torch.rand(N, V, H, W, 3, generator=torch.Generator().manual_seed(rank)) - 0.5).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed: per-launch tile choice, measured on this device (value-neutral); --tile-cache reuses a
    # table written by an earlier run so that a profiled run contains no tuning launches
    if a.tile_cache and os.path.exists(a.tile_cache):
        eng.plan.apply_tiles(json.load(open(a.tile_cache)))
    else:
        chosen = eng.plan.autotune(x.view(N * V, H, W, 3))
        if a.tile_cache and rank == 0:
            json.dump({k: v[0] for k, v in chosen.items()}, open(a.tile_cache, "w"))
    step = (lambda: sh.forward(x, check=False))
    if a.graph:
        assert world == 1, "--graph is a single-GPU option"
        step = eng.capture(x)
    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    eng.check_status()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        views_per_step = N * V * world
        ms = dt / a.steps * 1e3
        out = {
            "metric": "views/sec", "value": round(views_per_step / (ms * 1e-3), 1), "unit": "views/s",
            "shapes_per_sec": round(N * world / (ms * 1e-3), 2),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.storage,
            "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1]: " if a.preset == "c2" and a.storage == "f32"
                                    else "preset %s (%s forward variant): " % (a.preset, a.storage))
                                   + "ModelNet-shaped, %d views x %dx%dx3, %s backbone (raw tap %s, final tap %s), "
                                     "num_groups=%d, %s, forward only" % (V, H, W, BACKBONE, eng.plan.raw_tap,
                                                                          eng.plan.final_tap, G, a.storage),
                       "shapes_per_gpu": N, "views_per_gpu": N * V, "global_views": views_per_step,
                       "num_groups": G, "num_classes": C,
                       "exchange": a.exchange if world > 1 else "none",
                       "gflop_per_view": round(eng.plan.total_flops / (N * V) / 1e9, 3)},
        }
        step_tflops = eng.plan.total_flops / (ms * 1e-3) / 1e12
        out["step_tflops_per_gpu"] = round(step_tflops, 2)
        if not a.no_roofline:
            out["roofline"] = roofline(eng, x.view(N * V, H, W, 3), a.math)
        out["config"]["math"] = a.math + ": " + MATH[a.math][2]
        out["config"]["branch_lanes"] = eng.plan.lanes_used if not a.no_lanes else 1
        out["config"]["launch"] = "hipGraph replay" if a.graph else "eager"

        if world == 1 and a.math != "f32" and not a.no_exact:
            # the same step on the exact fp32 MFMA path, for reference (short run, same inputs)
            e32 = gv.GVCNN(BACKBONE, N, V, H, W, C, G, device=dev, num_bins=G, math="f32", lanes=not a.no_lanes)
            e32.plan.bind(P)
            e32.set_head(Hd)
            e32.plan.autotune(x.view(N * V, H, W, 3))
            for _ in range(2):
                e32.forward(x, check=False)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                e32.forward(x, check=False)
            torch.cuda.synchronize()
            ms32 = (time.perf_counter() - t1) / 5 * 1e3
            r32 = roofline(e32, x.view(N * V, H, W, 3), "f32", iters=3)
            dS = float((e32.shape_descriptor - eng.shape_descriptor.float()).abs().max() /
                       e32.shape_descriptor.abs().max())
            out["exact_f32_mfma"] = {"views_per_sec": round(N * V / (ms32 * 1e-3), 1), "ms_per_step": round(ms32, 3),
                                     "roofline_frac": r32["frac"], "achieved_tflops": r32["achieved"],
                                     "max_rel_diff_shape_descriptor_vs_default_math": dS}
            del e32
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P, Hd, a.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
