#!/usr/bin/env python3
"""bench.py — views/sec of the GVCNN hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--shapes S] [--exchange allgather|scores]

Workload (BASELINE.json configs[1]): ModelNet-shaped synthetic views, 12 views x 224x224 x 3 per
shape, Inception-v3 backbone, num_groups = 7, fp32, forward only (inference BatchNorm).  One "step"
is one pass of the whole hot path — folded backbone over all views, scorer, device-side group
assignment, view pooling + group fusion, classifier — over one batch that is already resident in
HBM.  N > 1: one rank per GPU.  Launched by torch.distributed.run the process IS a rank (RANK /
LOCAL_RANK / WORLD_SIZE in the environment); typed as plain `python bench.py --gpus N` the process is
only a launcher — it makes no GPU call, starts `python -m torch.distributed.run --nproc-per-node N
bench.py <same flags>` as a child, relays rank 0's JSON line and exits with the child's code.
Per-GPU work is fixed (weak scaling): the batch is cut on shape boundaries and the ranks all-gather
the scorer responses over RCCL (the batch-mean score of nets/model.py:146 couples the shapes of a
batch).  --exchange allgather (default; the form BASELINE.json north_star names) also all-gathers the
final view descriptors so every rank pools all shapes; --exchange scores stops at the scorer
responses and every rank pools the shapes it owns.  With --other-exchange (always over gloo) an N > 1
line times BOTH and reports the other one under "other_exchange".

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3                    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0                  # MI355X_MICROARCH.md: bf16 MFMA dense (spec)
HBM_PEAK = 8.0e12                               # MI355X_MICROARCH.md: HBM3E bytes/s (spec)
# fp32 products evaluated as N bf16 MFMA passes (GV_MATH_*): the ceiling in ALGORITHMIC fp32 FLOPs
MATH = {"f32": ("conv_igemm_f32", PEAK_F32_MFMA_TFLOPS, "v_mfma_f32_32x32x2_f32 (exact fp32 chain)"),
        "bf16x3": ("the bf16x3 convolution family: conv_ws_x3 (loader waves + MFMA consumer waves: strip mode on three-plane input, GEMM mode on fp32 input), conv_dma<NP=3> (LDS-DMA, three-plane input), conv3x3_halo_x3_k32 (stem 3x3), "
                   "conv_igemm_bf16s<NP=3> (fp32 input), conv_stem_patch_x3", PEAK_BF16_MFMA_TFLOPS / 6,
                   "fp32 operands split into 3 bf16 planes, 6 bf16 MFMA products (v_mfma_f32_32x32x16_bf16 / 16x16x32) per product block, "
                   "fp32 accumulate: fp32-level accuracy (dropped terms <= 2^-24 relative); peak = bf16 dense / 6"),
        "bf16x2": ("conv_igemm_bf16s<NP=2>", PEAK_BF16_MFMA_TFLOPS / 3, "2 bf16 planes, 3 MFMAs (~2^-16 relative)"),
        "bf16x1": ("conv_igemm_bf16s<NP=1>", PEAK_BF16_MFMA_TFLOPS, "plain bf16 products, fp32 accumulate"),
        # 16-bit STORAGE (configs c3-c5; not the bench line, which is fp32): --storage bf16 | f16
        "bf16": ("the 16-bit convolution family: conv_ws (loader waves + MFMA consumer waves), conv_dma<NP=1> (LDS-DMA), conv_igemm_lp<bf16>, conv_chain_lp (ResNet bottleneck launches), stem strip / halo kernels", PEAK_BF16_MFMA_TFLOPS, "bf16 activations/filters in HBM, v_mfma_f32_32x32x16_bf16, fp32 accumulate + epilogue"),
        "f16": ("the 16-bit convolution family: conv_ws (loader waves + MFMA consumer waves), conv_dma<NP=1> (LDS-DMA), conv_igemm_lp<f16>, conv_chain_lp (ResNet bottleneck launches), stem strip / halo kernels", PEAK_BF16_MFMA_TFLOPS, "fp16 activations/filters in HBM, v_mfma_f32_32x32x16_f16, fp32 accumulate + epilogue")}
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
    ap.add_argument("--exchange", default="allgather", choices=["allgather", "scores"],
                    help="multi-GPU: 'allgather' (north_star form) all-gathers the scorer responses AND the final view "
                         "descriptors, every rank pools all shapes; 'scores' exchanges only the scorer responses (each "
                         "rank pools the shapes it owns).  The N > 1 line reports the other one too (other_exchange)")
    ap.add_argument("--no-other-exchange", action="store_true", help="N > 1: time only --exchange")
    ap.add_argument("--other-exchange", action="store_true",
                    help="N > 1 over RCCL: also time the other exchange form and the other gather mode (extra timed loops after "
                         "the one the line's `value` comes from; always on over gloo, where the control flow is what is tested)")
    ap.add_argument("--gather", default="collective", choices=["collective", "direct"],
                    help="N > 1: how a gather travels — 'collective' = RCCL all_gather_into_tensor (the library picks ring / "
                         "tree), 'direct' = one point-to-point send to and receive from EVERY peer (each xGMI link carries "
                         "one shard: SURVEY 8e's one-shot all-gather).  The N > 1 line times the other one too (other_gather)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1, --exchange allgather: gather and pool inside the step instead of overlapping the "
                         "all-gather of step k with the backbone of step k+1 (ShardedGVCNN(overlap=True))")
    ap.add_argument("--no-traffic", action="store_true",
                    help="skip the in-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE children (roofline.traffic = null)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--tile-cache", default=None, help="JSON file with measured per-launch tile choices")
    ap.add_argument("--no-exact", action="store_true", help="skip the exact-fp32-MFMA reference run")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--preset", default="c2", choices=sorted(PRESETS), help="c2 = BASELINE.json configs[1] (the bench line)")
    ap.add_argument("--no-lanes", action="store_true", help="single-stream launch order (no branch concurrency)")
    ap.add_argument("--no-tune", action="store_true",
                    help="skip the per-launch tile measurement (default tiles): control-flow checks of the N > 1 path, never a "
                         "line to quote")
    ap.add_argument("--p3", default="default", choices=["default", "all", "none"],
                    help="bf16x3 math: conv -> conv intermediates stored as three bf16 planes (value neutral; A/B switch)")
    ap.add_argument("--math", default="bf16x3", choices=["f32", "bf16x3", "bf16x2", "bf16x1"],
                    help="how fp32 convolutions are evaluated on the matrix cores (GV_MATH_*)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (the real multi-GPU path); gloo + --same-device: control-flow check of the N>1 path "
                         "on a single-GPU box (every rank on cuda:0, collectives through host memory)")
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--dp", default="hybrid", choices=["views", "shapes", "hybrid"],
                    help="--train with N > 1: hybrid (default: view groups x shape shards, equal work on every rank, BN "
                         "sums only inside a shape group; pure view sharding whenever N divides the 12 views), views "
                         "(shard the views of every shape, BN statistics stay local; 8 ranks get 2,2,2,2,1,1,1,1) or "
                         "shapes (every BN layer all-reduces its per-view sums)")
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


def roofline(eng, x, math, iters=10, traffic=None):
    """Every implicit-GEMM conv launch of one step timed IN SEQUENCE (gv_plan_time_each: `iters` whole passes of the
    plan in launch order, an event behind every op — a launch is timed behind its real predecessor, not as a warm
    repeat of itself); achieved = algorithmic conv FLOPs of the step / summed conv launch time."""
    plan = eng.plan
    flops = t_ms = 0.0
    n = 0
    worst = None
    each = plan.time_each(x, iters)
    for i, op in enumerate(plan.ops):
        if op["kind"] != "conv":
            continue
        ms = each[i]
        flops += op["flops"]
        t_ms += ms
        n += 1
        tf = op["flops"] / (ms * 1e-3) / 1e12
        if worst is None or ms > worst[1]:
            worst = (op["name"], ms, tf)
    achieved = flops / (t_ms * 1e-3) / 1e12
    kname, peak, how = MATH[math]
    # the same ratio per STAGE of the backbone (north_star asks for the conv stages; SURVEY hard part (i): the Cin = 3 /
    # 32-channel stem is HBM-bound by construction, the 5x5 maps of Mixed_7 are tiny-M GEMMs): algorithmic FLOPs of the
    # stage's conv launches over their in-sequence time, and how much of that time the stage's HBM-bound launches take
    stages = {}
    for i, op in enumerate(plan.ops):
        if op["kind"] != "conv":
            continue
        sname = stage_of(op["name"])
        d = stages.setdefault(sname, {"flops": 0.0, "ms": 0.0, "launches": 0, "hbm_ms": 0.0})
        d["flops"] += op["flops"]
        d["ms"] += each[i]
        d["launches"] += 1
        if op["bytes"] / HBM_PEAK > op["flops"] / (peak * 1e12):
            d["hbm_ms"] += each[i]
    per_stage = {k: {"frac": round(d["flops"] / (d["ms"] * 1e-3) / 1e12 / peak, 4),
                     "achieved": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 1), "ms": round(d["ms"], 3),
                     "launches": d["launches"], "hbm_bound_ms": round(d["hbm_ms"], 3)} for k, d in stages.items()}
    # per-launch ceiling: a launch can finish no sooner than its FLOPs at the MFMA peak or its algorithmic bytes at the
    # HBM peak, whichever is later; `attainable` is the step's FLOPs over the sum of those times (== peak when no launch
    # is HBM-bound: ResNet's 64-channel 1x1 layers are, and cap the whole step well below the MFMA peak)
    t_min = sum(max(op["flops"] / (peak * 1e12), op["bytes"] / HBM_PEAK) for op in plan.ops if op["kind"] == "conv")
    n_hbm = sum(1 for op in plan.ops if op["kind"] == "conv" and op["bytes"] / HBM_PEAK > op["flops"] / (peak * 1e12))
    attainable = flops / t_min / 1e12
    alg_bytes = sum(op["bytes"] for op in plan.ops if op["kind"] == "conv") / n
    return {"bound": "mfma", "kernel": "%s; %d launches/step, every one timed in sequence" % (kname, n), "math": how,
            "timing": "hipEvents behind every op over %d whole passes of the plan in launch order (single lane)" % iters,
            "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4),
            "attainable": round(attainable, 1), "frac_of_attainable": round(achieved / attainable, 4),
            "hbm_bound_launches": n_hbm,
            "traffic": round(traffic["conv_hbm_bytes_per_launch"]) if traffic and "conv_hbm_bytes_per_launch" in traffic else None,
            "traffic_note": (traffic or {}).get("note", "not measured (--no-traffic)"),
            "algorithmic_bytes_per_launch": round(alg_bytes),
            "flops_per_step": flops, "avg_launch_us": round(t_ms * 1e3 / n, 2),
            "conv_ms_per_step": round(t_ms, 3),
            "stages": per_stage,
            "longest_launch": {"name": worst[0], "ms": round(worst[1], 4), "tflops": round(worst[2], 2)}}


def stage_of(op_name):
    """Backbone stage of a conv launch: Inception-v3's stem (Conv2d_1a .. 4a), Mixed_5 (35x35-equivalent maps), Mixed_6
    (17x17-equivalent, Mixed_6a's reduction included), Mixed_7 (8x8-equivalent); ResNet-v2-50's conv1 and blocks 1 - 4."""
    first = op_name.split("+")[0]                            # (a fused sibling GEMM: its first member's name)
    parts = first.split("/")
    if len(parts) > 1 and parts[0] == "InceptionV3":
        return parts[1][:7] if parts[1].startswith("Mixed_") else "stem"
    if len(parts) > 1 and parts[0] == "resnet_v2_50":
        return parts[1] if parts[1].startswith("block") else "stem"
    return parts[0]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(P, Hd, seconds):
    """The CPU oracle (a port of the reference graph; TensorFlow itself cannot run here) timed on this
    box's host cores on a bounded sample of the same workload: configs[0] as written
    (2 shapes x 6 views) and configs[1] reduced to 2 shapes, each REFERENCE-SHAPED (V sequential
    backbone calls at batch N, nets/model.py:129-141) and FOLDED (one call at batch N*V).  `value` is the
    reference-shaped rate on the bench line's own config; `cores` = the torch threads that gave it."""
    from oracle import model as OM
    host = os.cpu_count() or 1
    budget = max(seconds, 4.0)
    t_all = time.time()

    def rate(fn, views, max_s):
        t0 = time.time()
        fn()                                            # page in (counts when it alone exceeds the share)
        first = time.time() - t0
        if first >= max_s:
            return views / first
        t0 = time.time()
        n = 0
        while True:
            fn()
            n += 1
            if time.time() - t0 >= max_s or n >= 20:
                break
        return n * views / (time.time() - t0)

    # thread count: oneDNN convolutions at these batch sizes stop scaling well below a 2-socket host's thread
    # count (measured on the 256-thread GPU box, folded c2 pass: 16 threads 93 views/s, 32: 73, 64: 34, all 256: 0.2 —
    # BASELINE.md's "all threads" would have taken minutes per pass): pick the best of a few, report what was used
    x2 = torch.rand(2, V, H, W, 3, generator=torch.Generator().manual_seed(0)) - 0.5
    cands = sorted({min(host, t) for t in (8, 16, 32)})
    best_t, best_r, tried = cands[0], 0.0, {}
    for t in cands:
        torch.set_num_threads(t)
        r = rate(lambda: OM.folded_backbone(x2, P, BACKBONE), 2 * V, budget * 0.08)
        tried[str(t)] = round(r, 1)
        if r > best_r:
            best_t, best_r = t, r
    torch.set_num_threads(best_t)
    share = budget * 0.17
    res = {}
    res["c2_reduced_reference_shaped"] = rate(lambda: OM.gvcnn(x2, C, P, Hd, G, BACKBONE, num_bins=G), 2 * V, share)
    res["c2_reduced_folded"] = rate(lambda: OM.folded_backbone(x2, P, BACKBONE), 2 * V, share)
    if BACKBONE == "inception_v3" and H == 224:         # configs[0]: 2 shapes x 6 views, G = 5 (same weights)
        x1 = x2[:, :6].contiguous()
        Hd1 = OM.init_head_params(6, Hd["dense/kernel"].shape[0], Hd["dense_%d/kernel" % V].shape[0], C, seed=3)
        res["c1_reference_shaped"] = rate(lambda: OM.gvcnn(x1, C, P, Hd1, 5, BACKBONE, num_bins=5), 12, share)
        res["c1_folded"] = rate(lambda: OM.folded_backbone(x1, P, BACKBONE), 12, share)
    return {"value": round(res["c2_reduced_reference_shaped"], 2), "unit": "views/s", "cores": best_t,
            "host_cores": host, "cpu_model": cpu_model(), "kind": "port",
            "views_per_sec": {k: round(v, 2) for k, v in res.items()},
            "threads_tried_folded_views_per_sec": tried,
            "sample": "torch-CPU fp32 oracle (oracle/model.py), %s %dx%d: configs[1] reduced to 2 shapes x %d views and "
                      "configs[0] as written (2 shapes x 6 views), each reference-shaped (V sequential backbone calls at "
                      "batch 2 + grouping head) and folded (one backbone call); %.1f s of CPU time in all"
                      % (BACKBONE, H, W, V, time.time() - t_all)}


# ------------------------------------------------------------------------------------------------
# launcher / profiler children (the parent makes no GPU call before they have exited)
# ------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        return sck.getsockname()[1]


def launch_ranks(a):
    """`python bench.py --gpus N` typed without a launcher: start one rank per GPU as children
    (python -m torch.distributed.run, rendezvous on 127.0.0.1), relay rank 0's JSON line, exit with
    the children's code.  This process never initialises the GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env["GVBENCH_SELF"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    for line in p.stdout:
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return p.wait()


def _is_conv_kernel(n):
    return "conv_igemm" in n or "conv3x3_halo" in n or "conv_stem_patch" in n or "conv_dma" in n or "conv_ws" in n or "conv_chain" in n


def _pmc_rows(path, counter, steps, marker="dense_f32", total_steps=None):
    """The dispatches of the last `steps` steps of a child run.  A step ends with its last `marker` kernel (the classifier
    of a forward pass, the optimizer of a training step); a step may launch the marker SEVERAL times (apply_momentum:
    one launch for the decayed range, one for the rest), so the launches per step are counted — marker dispatches over
    the steps the child ran (total_steps = warmup + steps) — instead of assumed to be one."""
    import csv
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    per_step = 1
    if total_steps:
        if not ends or len(ends) % total_steps:
            raise RuntimeError("%d '%s' dispatches over %d steps: cannot delimit the steps" % (len(ends), marker, total_steps))
        per_step = len(ends) // total_steps
    if not ends:
        return rows
    last = ends[-1]
    start = ends[-per_step * steps - 1] + 1 if len(ends) > per_step * steps else 0
    window = rows[start:last + 1]
    if len(window) % steps:
        raise RuntimeError("%d dispatches in a window of %d steps: not a whole number per step" % (len(window), steps))
    return window


def measure_traffic(a, tiles_path):
    """HBM bytes per conv launch of THIS run's configuration, from two rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE need separate passes: 3 + 2 of the 4 TCC slots) over a short child run of this same file with the
    same per-launch tiles; FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), KiB units
    (MI355X_MICROARCH.md, HBM).  The program sits directly behind `--`; the children run before this process
    touches the GPU.  Returns a dict (or a dict with only a note when the profiler is unavailable)."""
    import glob
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return {"note": "rocprofv3 not found: traffic not measured"}
    steps = 2
    tmp = tempfile.mkdtemp(prefix="gvbench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    base = [sys.executable, os.path.abspath(__file__), "--pmc-child", "1", "--preset", a.preset, "--storage", a.storage,
            "--math", a.math if a.storage == "f32" else "bf16x3", "--shapes", str(a.shapes), "--steps", str(steps),
            "--warmup", "1", "--no-lanes", "--tile-cache", tiles_path]
    out = {}
    try:
        for tag, counter in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
            cmd = [rp, "--kernel-trace", "--pmc", counter, "-d", os.path.join(tmp, tag), "-o", tag,
                   "--output-format", "csv", "--"] + base
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                               timeout=420)
            files = glob.glob(os.path.join(tmp, tag, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"note": "rocprofv3 --pmc %s pass failed (rc %d): traffic not measured: %s"
                                % (counter, r.returncode, r.stderr[-200:].replace("\n", " "))}
            out[counter] = _pmc_rows(files[0], counter, steps)
        fetch, write = out["FETCH_SIZE"], out["WRITE_SIZE"]
        n_conv = sum(1 for r in fetch if _is_conv_kernel(r["Kernel_Name"]))
        f_conv = sum(float(r["Counter_Value"]) for r in fetch if _is_conv_kernel(r["Kernel_Name"])) * 1024 * 2
        w_conv = sum(float(r["Counter_Value"]) for r in write if _is_conv_kernel(r["Kernel_Name"])) * 1024
        if n_conv == 0:
            return {"note": "no conv dispatches in the PMC passes: traffic not measured"}
        return {"conv_hbm_bytes_per_launch": (f_conv + w_conv) / n_conv, "conv_launches": n_conv, "steps": steps,
                "conv_fetch_bytes_per_step": f_conv / steps, "conv_write_bytes_per_step": w_conv / steps,
                "note": "measured in this run: HBM bytes per conv launch from two rocprofv3 --kernel-trace --pmc child "
                        "passes (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE; KiB units) over %d steps of this same "
                        "command with the same per-launch tiles, single launch lane" % steps}
    except Exception as e:                                  # the profiler must never take the bench line down
        return {"note": "traffic not measured: %s" % (str(e)[:200],)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


TRAIN_FAMILIES = (("wgrad", ("conv_wgrad", "dw_reduce_slices")), ("conv", ("conv_igemm", "conv3x3_halo", "conv_stem_patch", "conv_dma", "conv_ws", "conv_chain")),
                  ("bn", ("grouped_sums", "bn_stream", "scale_shift_act_grouped", "bn_bwd_apply")),
                  ("pool", ("pool2d", "maxpool", "avgpool3x3")))


def _train_family(kernel_name):
    for fam, keys in TRAIN_FAMILIES:
        if any(k in kernel_name for k in keys):
            return fam
    return "other"


def measure_traffic_train(a):
    """HBM bytes per step of the training step, by kernel family, from two rocprofv3 --pmc child passes (FETCH_SIZE x2
    gfx950 correction, WRITE_SIZE; KiB units: MI355X_MICROARCH.md, HBM) of this same command; run BEFORE this process
    touches the GPU, program directly behind `--`."""
    import glob
    import shutil
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return {"note": "rocprofv3 not found: traffic not measured"}
    steps = 2
    tmp = tempfile.mkdtemp(prefix="gvbench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    base = [sys.executable, os.path.abspath(__file__), "--train", "--pmc-child", "1", "--preset", a.preset, "--storage",
            a.storage, "--shapes", str(a.shapes), "--steps", str(steps), "--warmup", "1"]
    fam_bytes = {}
    try:
        for tag, counter, mult in (("f", "FETCH_SIZE", 2048.0), ("w", "WRITE_SIZE", 1024.0)):
            cmd = [rp, "--kernel-trace", "--pmc", counter, "-d", os.path.join(tmp, tag), "-o", tag,
                   "--output-format", "csv", "--"] + base
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                               timeout=600)
            files = glob.glob(os.path.join(tmp, tag, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"note": "rocprofv3 --pmc %s pass failed (rc %d): traffic not measured: %s"
                                % (counter, r.returncode, r.stderr[-200:].replace("\n", " "))}
            for row in _pmc_rows(files[0], counter, steps, marker="sgd_momentum", total_steps=steps + 2):   # (the child: one step before autotune, one warm-up, `steps` timed)
                fam = _train_family(row["Kernel_Name"])
                d = fam_bytes.setdefault(fam, {"fetch": 0.0, "write": 0.0, "launches": 0})
                d["fetch" if tag == "f" else "write"] += float(row["Counter_Value"]) * mult / steps
                if tag == "f":
                    d["launches"] += 1
        for d in fam_bytes.values():
            d["launches"] = d["launches"] / steps
        return {"families": fam_bytes, "steps": steps,
                "note": "measured in this run: HBM bytes per step and kernel family from two rocprofv3 --kernel-trace --pmc "
                        "child passes (FETCH_SIZE x2 gfx950 correction, WRITE_SIZE; KiB units) over %d steps of this same "
                        "command" % steps}
    except Exception as e:                                  # the profiler must never take the bench line down
        return {"note": "traffic not measured: %s" % (str(e)[:200],)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def train_roofline(eng, x, labels, storage, traffic, steps=3):
    """`roofline` of the training line.  Every launch of `steps` whole steps is timed IN SEQUENCE (gvcnn_tf_amd.steptime:
    an event pair around every entry-point call while a real step runs).  MFMA side: the algorithmic FLOPs of the
    convolution-shaped launches — forward, data gradient (every convolution but the first), filter gradient, 2*M*N*K
    each — over their summed launch time, against the dense 16-bit MFMA peak.  HBM side: BatchNorm and pooling launches,
    algorithmic bytes over their summed time against 8 TB/s; a train-mode BatchNorm is priced at what it cannot avoid
    once its sums ride on the launches that produce z and dy: read z + write y forward, read dy + read z + write dz
    backward = 5 elements moved per activation element (the separate sums passes move 3 more)."""
    from gvcnn_tf_amd import steptime
    es = 4 if storage == "f32" else 2
    recs = steptime.timed_step(eng, x, labels, steps=steps)
    fam = {}
    for st in recs:
        for k, (ms, n) in steptime.by_family(st).items():
            t, c = fam.get(k, (0.0, 0))
            fam[k] = (t + ms / steps, c + n / steps)
    ops = eng.plan.ops
    f_fwd = sum(op["flops"] for op in ops if op["kind"] == "conv")
    f_dg = sum(op["flops"] for op in ops if op["kind"] == "conv" and op["x"].vbuf >= 0)
    flops = {"conv": f_fwd, "dgrad": f_dg, "wgrad": f_fwd}
    peak = PEAK_BF16_MFMA_TFLOPS if storage != "f32" else PEAK_BF16_MFMA_TFLOPS / 6
    tr = (traffic or {}).get("families", {})

    def mfma(keys):
        t = sum(fam.get(k, (0.0, 0))[0] for k in keys)
        n = sum(fam.get(k, (0.0, 0))[1] for k in keys)
        fl = sum(flops[k] for k in keys)
        ach = fl / (t * 1e-3) / 1e12 if t > 0 else 0.0
        return {"bound": "mfma", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "flops_per_step": fl, "ms_per_step": round(t, 3), "launches_per_step": round(n)}
    bn_el = sum(op["x"].npix * op["x"].c for op in ops if op["kind"] == "bn")
    pool_b = 0.0
    for op in ops:
        if op["kind"] == "pool":
            i, o = op["x"].npix * op["x"].c, op["y"].npix * op["y"].c
            arg = o if op["mode"] == 0 else 0                        # the recorded argmax byte of a max pool
            pool_b += (i + o) * es + arg + (o * es + arg + i * es)

    def hbm(key, nbytes, traffic_keys):
        t, n = fam.get(key, (0.0, 0))
        ach = nbytes / (t * 1e-3) / 1e9 if t > 0 else 0.0
        got = [tr[k] for k in traffic_keys if k in tr]
        return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(ach / (HBM_PEAK / 1e9), 4), "algorithmic_bytes_per_step": nbytes, "ms_per_step": round(t, 3),
                "launches_per_step": round(n),
                "traffic": round(sum(g["fetch"] + g["write"] for g in got)) if got else None}
    allm = mfma(("conv", "dgrad", "wgrad"))
    got = [tr[k] for k in ("conv", "wgrad") if k in tr]
    out = dict(allm)
    out["kernel"] = ("all convolution-shaped launches of the step: forward + data gradient (conv_dma<NP=1> / conv_igemm_lp / "
                     "halo / strip kernels) and filter gradient (conv_wgrad_dma / conv_wgrad_lp)")
    out["traffic"] = round(sum(g["fetch"] + g["write"] for g in got)) if got else None
    out["traffic_note"] = (traffic or {}).get("note", "not measured")
    out["timing"] = "hipEvent pair around every launch of %d whole steps, in sequence (gvcnn_tf_amd/steptime.py)" % steps
    out["conv"] = mfma(("conv", "dgrad"))
    out["wgrad"] = mfma(("wgrad",))
    out["bn"] = hbm("bn", 5.0 * bn_el * es, ("bn",))
    out["bn"]["folded_sums"] = {"forward": sum(1 for op in ops if op.get("_st_f_done")),
                                "backward": sum(1 for op in ops if op.get("_st_b_done")),
                                "batchnorm_layers": sum(1 for op in ops if op["kind"] == "bn")}
    out["pool"] = hbm("pool", pool_b, ("pool",))
    out["in_sequence_ms_per_step"] = {k: round(v[0], 3) for k, v in sorted(fam.items())}
    return out


def train_main(a, world, rank, dev, traffic=None):
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
    elif a.dp == "hybrid":                           # view groups x shape shards, per-rank work fixed (weak scaling)
        from gvcnn_tf_amd.sharding import hybrid_coords, hybrid_grid
        vg, shs = hybrid_grid(V, world)
        gi, si = hybrid_coords(V, world, rank)
        v_l, n_l = V // vg, N * vg                   # N*vg shapes per shape shard: N*V view images per rank, as at 1 GPU
        lo, hi = gi * v_l, (gi + 1) * v_l
        eng = TrainGVCNN(BACKBONE, n_l, v_l, H, W, C, G, device=dev, num_bins=G, head_views=V, view_offset=lo,
                         storage=a.storage)
        sh = ShardedTrainGVCNN(eng, mode="hybrid")
        x = (torch.rand(n_l, V, H, W, 3, generator=torch.Generator().manual_seed(si)) - 0.5)[:, lo:hi].contiguous().to(dev)
        labels = torch.randint(0, C, (n_l,), generator=torch.Generator().manual_seed(100 + si))
        views_per_step, views_local = n_l * shs * V, n_l * v_l
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
    if a.pmc_child:
        return
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        ms = dt / a.steps * 1e3
        flops = 3.0 * sum(op.get("flops", 0) for op in eng.plan.ops) * world
        roof = {}
        if world == 1 and not a.no_roofline:
            roof = {"roofline": train_roofline(eng, x, labels, a.storage, traffic)}
        print(json.dumps({**{
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
            "step_tflops": round(flops / (ms * 1e-3) / 1e12, 2)}, **roof}), flush=True)


def main():
    global BACKBONE, V, H, W, G, C
    a = parse()
    BACKBONE, V, H, G, C = PRESETS[a.preset]
    W = H
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1")) if under_launcher else 1
    rank = int(os.environ.get("RANK", "0")) if under_launcher else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if under_launcher else 0
    if not under_launcher and a.gpus > 1:
        sys.exit(launch_ranks(a))                       # no GPU call was made in this process
    if world != a.gpus:
        sys.exit("bench.py --gpus %d but the launcher started %d ranks" % (a.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # N = 1: tile tuning and the HBM-traffic PMC passes run as CHILDREN before this process touches the GPU
    traffic = None
    tmp_tiles = None
    want_traffic = (world == 1 and not a.pmc_child and not a.train and not a.no_roofline and not a.no_traffic
                    and not a.graph)
    if want_traffic:
        if not (a.tile_cache and os.path.exists(a.tile_cache)):
            if not a.tile_cache:
                import tempfile
                fd, tmp_tiles = tempfile.mkstemp(prefix="gvbench_tiles_", suffix=".json", dir="/tmp")
                os.close(fd)
                os.unlink(tmp_tiles)
                a.tile_cache = tmp_tiles
            tune = [sys.executable, os.path.abspath(__file__), "--pmc-child", "tune", "--preset", a.preset, "--storage",
                    a.storage, "--math", a.math if a.storage == "f32" else "bf16x3", "--shapes", str(a.shapes),
                    "--tile-cache", a.tile_cache] + (["--no-lanes"] if a.no_lanes else [])
            r = subprocess.run(tune, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=600)
            if r.returncode != 0 or not os.path.exists(a.tile_cache):
                sys.stderr.write("bench.py: tuning child failed (rc %d), tuning in-process instead\n%s\n"
                                 % (r.returncode, r.stderr[-500:]))
        if os.path.exists(a.tile_cache):
            traffic = measure_traffic(a, a.tile_cache)
        else:
            traffic = {"note": "traffic not measured: no tile table from the tuning child"}

    if world == 1 and a.train and not a.pmc_child and not a.no_roofline and not a.no_traffic:
        traffic = measure_traffic_train(a)                  # (children; this process has not touched the GPU yet)

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
        return train_main(a, world, rank, dev, traffic)

    N = a.shapes
    eng = gv.GVCNN(BACKBONE, N, V, H, W, C, G, device=dev, num_bins=G, math=a.math if a.storage == "f32" else "f32",
                   lanes=not a.no_lanes, storage=a.storage, p3={"default": True, "all": "all", "none": False}[a.p3])
    P = gv.params.init_backbone_params(eng.plan.param_shapes(), seed=2, perturb_bn=True)
    Hd = gv.params.init_head_params(V, eng.raw.c, eng.final.c, C, seed=3, spread_scores=True)
    eng.plan.bind(P)
    eng.set_head(Hd)
    sh = ShardedGVCNN(eng, exchange=a.exchange, overlap=not a.no_overlap, gather_mode=a.gather)
    x = (torch.rand(N, V, H, W, 3, generator=torch.Generator().manual_seed(rank)) - 0.5).to(dev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tuned_by = None
    # untimed: per-launch tile choice, measured on this device (value-neutral); --tile-cache reuses a
    # table written by an earlier run so that a profiled run contains no tuning launches
    if a.tile_cache and os.path.exists(a.tile_cache):
        eng.plan.apply_tiles(json.load(open(a.tile_cache)))
    elif a.no_tune:
        pass
    elif world > 1:
        # N > 1: rank 0 measures, every rank installs rank 0's table — ONE tuning instead of N concurrent ones on one
        # host (67 launches x 25+ candidates each), and identical kernels on every rank (tiles of different kernel families
        # sum k in different orders — fp32-rounding-level differences — so one table also keeps the ranks' values identical
        # where they must be; and ranks that pick different tiles add noise to the MAX-over-ranks step time)
        table = None
        if rank == 0:
            table = {k: v[0] for k, v in eng.plan.autotune(x.view(N * V, H, W, 3)).items()}
            if a.tile_cache:
                json.dump(table, open(a.tile_cache, "w"))
        box = [table]
        dist.broadcast_object_list(box, src=0)
        if rank != 0:
            eng.plan.apply_tiles(box[0])
        tuned_by = "rank 0 (table broadcast to the other %d ranks)" % (world - 1)
    else:
        chosen = eng.plan.autotune(x.view(N * V, H, W, 3))
        if a.tile_cache and rank == 0:
            json.dump({k: v[0] for k, v in chosen.items()}, open(a.tile_cache, "w"))
    if tmp_tiles and os.path.exists(tmp_tiles):
        os.unlink(tmp_tiles)
    if a.pmc_child == "tune":
        return

    def timed(step):
        """W untimed steps, then exactly K steps between two barrier + synchronize pairs; MAX over the ranks."""
        for _ in range(a.warmup):
            step()
        if getattr(step, "flush", None):
            step.flush()                                    # (the warm-up's last head is not part of the timed steps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        if getattr(step, "flush", None):
            step.flush()                                    # the overlapped exchange's last head belongs to the timed steps
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def step():
        return sh.forward(x, check=False)
    if sh.overlap:
        step.flush = sh.flush
    if a.graph:
        assert world == 1, "--graph is a single-GPU option"
        step = eng.capture(x)
    dt = timed(step)
    eng.check_status()
    if a.pmc_child:
        return
    other = other_g = None
    # (the driver's scaling run gets the ONE measurement it asks for: nothing after it can cost it the line)
    if world > 1 and not a.no_other_exchange and (a.other_exchange or a.backend == "gloo"):
        name = "scores" if a.exchange == "allgather" else "allgather"
        sh2 = ShardedGVCNN(eng, exchange=name, gather_mode=a.gather)
        dt2 = timed(lambda: sh2.forward(x, check=False))
        other = (name, dt2)
        if a.exchange == "allgather":                       # the same exchange, the other way of moving it
            gname = "direct" if a.gather == "collective" else "collective"
            sh3 = ShardedGVCNN(eng, exchange="allgather", overlap=not a.no_overlap, gather_mode=gname)
            step3 = lambda: sh3.forward(x, check=False)     # noqa: E731
            if sh3.overlap:
                step3.flush = sh3.flush
            other_g = (gname, timed(step3))

    # what every rank ran on / with (gathered outside the timed region)
    rank_facts = None
    other_ov = None
    if world > 1:
        import hashlib
        tiles = sorted((op["name"], int(op.get("tile", 0))) for op in eng.plan.ops
                       if op["kind"] == "conv" and not op.get("maxpool"))     # (one kernel serves the pooled form: no tile)
        mine = {"rank": rank, "device": torch.cuda.current_device(), "world_size": dist.get_world_size(),
                "tile_table": hashlib.sha1(json.dumps(tiles).encode()).hexdigest()[:12]}
        rank_facts = [None] * world
        dist.all_gather_object(rank_facts, mine)
        if sh.overlap and not a.no_other_exchange:
            # the same exchange with call-by-call semantics (every forward returns its own result): the 1 -> N ratio can be
            # read on either; a failure here must not cost the line above
            try:
                sh_cc = ShardedGVCNN(eng, exchange=a.exchange, overlap=False, gather_mode=a.gather)
                other_ov = timed(lambda: sh_cc.forward(x, check=False))
            except Exception as e:                          # noqa: BLE001
                other_ov = None
                sys.stderr.write("bench.py: call-by-call re-run failed: %r\n" % (e,))

    out = None
    if rank == 0:
        views_per_step = N * V * world
        ms = dt / a.steps * 1e3
        f = eng.final
        esz = 4 if a.storage == "f32" else 2
        desc_bytes = N * V * f.h * f.w * f.c * esz
        xbytes = {"scores": {"sent_per_rank": N * V * 4, "received_per_rank": N * V * 4 * (world - 1)},
                  "allgather": {"sent_per_rank": N * V * 4 + desc_bytes,
                                "received_per_rank": (N * V * 4 + desc_bytes) * (world - 1)}}
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
        if world > 1:
            out["config"]["world_size"] = dist.get_world_size()          # what the ranks saw, not the flag
            out["config"]["backend"] = "rccl (torch.distributed nccl)" if a.backend == "nccl" else "gloo (control-flow check)"
            out["config"]["exchange_bytes_per_step"] = xbytes[a.exchange]
            out["config"]["gather"] = a.gather + (": one send to / receive from every peer (batch_isend_irecv)"
                                                  if a.gather == "direct" else ": RCCL all_gather_into_tensor")
            out["config"]["exchange_overlap"] = ("all-gather of step k in flight under the backbone of step k+1 (results one "
                                                 "call later; the last head is inside the timed region)") if sh.overlap else "none"
            out["config"]["launched_by"] = "bench.py (self-launched ranks)" if os.environ.get("GVBENCH_SELF") else "external torch.distributed.run"
            out["config"]["ranks"] = rank_facts
            out["config"]["tiles_tuned_by"] = tuned_by or ("--tile-cache" if a.tile_cache else "--no-tune")
            try:
                out["config"]["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if a.backend == "nccl" else None
            except Exception:                               # noqa: BLE001
                out["config"]["rccl_version"] = None
            if other_ov is not None:
                ms4 = other_ov / a.steps * 1e3
                out["other_overlap"] = {"exchange_overlap": "none (call-by-call: every forward returns its own result)",
                                        "value": round(views_per_step / (ms4 * 1e-3), 1), "unit": "views/s",
                                        "ms_per_step": round(ms4, 3)}
            if other is not None:
                ms2 = other[1] / a.steps * 1e3
                out["other_exchange"] = {"exchange": other[0], "value": round(views_per_step / (ms2 * 1e-3), 1),
                                         "unit": "views/s", "ms_per_step": round(ms2, 3),
                                         "exchange_bytes_per_step": xbytes[other[0]]}
            if other_g is not None:
                ms3 = other_g[1] / a.steps * 1e3
                out["other_gather"] = {"gather": other_g[0], "value": round(views_per_step / (ms3 * 1e-3), 1),
                                       "unit": "views/s", "ms_per_step": round(ms3, 3)}
        step_tflops = eng.plan.total_flops / (ms * 1e-3) / 1e12
        out["step_tflops_per_gpu"] = round(step_tflops, 2)
        if not a.no_roofline:
            out["roofline"] = roofline(eng, x.view(N * V, H, W, 3), a.math, traffic=traffic)
        out["config"]["math"] = a.math + ": " + MATH[a.math][2]
        out["config"]["branch_lanes"] = eng.plan.lanes_used if not a.no_lanes else 1
        out["config"]["launch"] = "hipGraph replay" if a.graph else "eager"
        if a.no_tune:
            out["config"]["tiles"] = "default (--no-tune: a control-flow check, not a line to quote)"
        elif getattr(eng.plan, "autotune_moved", None) is not None:
            out["config"]["tiles"] = ("measured per launch: warm repeats shortlist four, timed in sequence the launch keeps "
                                      "the fastest (%d launches left their warm-repeat choice)" % eng.plan.autotune_moved)

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
