#!/usr/bin/env python
"""Headline benchmark: self-supervised monodepth TRAINING samples/s at 192x640, ResNet-18 depth+pose
(BASELINE.json metric / configs[1]: bf16, batch 12 per GPU), full optimisation step
(forward, photometric loss, backward, global-norm clip, Adam) on synthetic 3-frame triplets.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job samples/s, plus
`roofline` (dominant kernel family = implicit-GEMM conv on MFMA, timed live with HIP events) and, at
N=1, `cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def synthetic_device_batches(B, H, W, dev, rank, n=4, fisheye=False):
    """SURVEY §8(d) synthetic triplets from the package's own SyntheticTripletDataset through its collate_fn,
    generated once and kept resident in HBM (the timed region contains no host-to-device transfer)."""
    from fsnet_amd.vision_base.data.datasets.dataset_utils import collate_fn
    from fsnet_amd.vision_base.data.datasets.synthetic import SyntheticTripletDataset
    ds = SyntheticTripletDataset(size=n * B, height=H, width=W, seed=1000 + rank, fisheye=fisheye)
    out = []
    for i in range(n):
        d = collate_fn([ds[i * B + j] for j in range(B)])
        out.append({k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in d.items()})
    return out


def pmc_traffic(kind):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in
    separate runs of this same command; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  The passes are
    collected offline with tools/pmc_traffic.sh and committed as profiles/pmc_traffic.json; None if absent."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        return json.load(open(path)).get(kind, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def pmc_valu_insts(kind):
    """wave-level vector-ALU instructions per launch (SQ_INSTS_VALU pass of tools/pmc_traffic.sh); None if absent"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        return json.load(open(path)).get(kind, {}).get("valu_insts_per_launch")
    except Exception:
        return None


N_SIMD = 256 * 4            # MI355X: 256 CUs x 4 SIMDs; a wave64 vector instruction occupies its SIMD's issue port for 4 cycles
CLOCK_HZ = 2.4e9            # peak engine clock, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(max_seconds=25.0):
    """Oracle (CPU restatement of the reference path, parity-pinned to the reference) on this box's host
    cores: BASELINE config[0] (B=2, 192x640, depth+pose, fp32), full step incl. clip + Adam."""
    from oracle import fsnet_oracle as O
    ncores = os.cpu_count() or 1
    try:
        import psutil
        ncores = psutil.cpu_count(logical=False) or ncores
    except Exception:
        pass
    # B = 2 at 192x640 does not feed 128 cores (round 1: 0.46 samples/s with 128 threads against 1.28 on 8): oneDNN's
    # conv parallelism over such a small batch stops paying beyond a few dozen threads
    ncores = min(ncores, 32)
    torch.set_num_threads(ncores)
    B, H, W = 2, 192, 640
    tr = O.OracleTrainer(O.init_state(seed=0, with_pose=True), with_pose=True)
    data = O.synthetic_batch(B, H, W, seed=0)
    tr.step(data)                      # warm-up
    t0 = time.time()
    n = 0
    while n < 3 or (time.time() - t0 < max_seconds and n < 12):
        tr.step(data)
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 4), "unit": "samples/s", "cores": ncores, "kind": "port",
            "sample": "%d full training steps, B=2, 192x640, R18 depth+pose, fp32 torch-CPU oracle (%.2f s/step)" % (n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--depth", type=int, default=18)
    ap.add_argument("--workload", default="kitti", choices=["kitti", "fisheye"],
                    help="kitti: BASELINE configs[1] (default, the headline metric); fisheye: BASELINE configs[3] at the "
                         "reference's own size — KITTI-360 fisheye, ResNet-18 + FishEyeDecoder (Mei camera model), 64 depth "
                         "bins, 384x384, batch 16, max depth 150, weight decay 1e-5 (configs/kitti360_fisheye_example:72,"
                         "83-86,198-207), dataset poses")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    args = ap.parse_args()
    fisheye = args.workload == "fisheye"
    if fisheye:
        defaults = ap.parse_args([])
        if (args.batch, args.height, args.width) == (defaults.batch, defaults.height, defaults.width):
            args.batch, args.height, args.width = 16, 384, 384

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # FSNET_AMD_BENCH_SHARED_DEVICE=1 (test rigs with ONE GPU): every rank uses device 0 and the ranks talk over gloo,
    # so that the multi-rank code path of this script can be exercised without a multi-GPU node.  Never set by the
    # driver: its N-GPU runs use one device per rank and RCCL.
    shared = os.environ.get("FSNET_AMD_BENCH_SHARED_DEVICE", "0") != "0"
    dev_index = 0 if shared else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        torch.distributed.init_process_group(backend="gloo" if shared else "nccl", init_method="env://")

    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.hip.conv import LaunchProfile
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    from fsnet_amd.vision_base.utils.utils import set_random_seed

    set_random_seed(123)
    RT.set_compute_dtype(args.dtype)
    B, H, W = args.batch, args.height, args.width
    if fisheye:
        model = build(**meta_arch_cfg(H, W, with_pose=False, depth=args.depth, num_output_channels=64, max_depth=150.0,
                                      fisheye=True)).to(dev).train()
        tc = training_cfg(clip_gradients=35.0, lr=1e-4, weight_decay=1e-5)
    else:
        model = build(**meta_arch_cfg(H, W, with_pose=True, depth=args.depth)).to(dev).train()
        tc = training_cfg(clip_gradients=35.0, lr=1e-4)
    optimizer = build_optimizer(model, **tc.optimizer)
    hook = build(**tc.training_hook)
    batches = synthetic_device_batches(B, H, W, dev, rank, fisheye=fisheye)

    def run_steps(n, start):
        for i in range(n):
            hook(dict(batches[(start + i) % len(batches)]), model, optimizer, global_step=start + i)

    # data parallel: the training hook first times the step with the encoders as two chains and as two lanes on the ranks
    # present and keeps the faster (BaseTrainingHook, encoder-pass autotune) — set-up, before the contract's warm-up steps
    pre = 0
    if world > 1:
        while not hook.tune_done and pre < 200:
            run_steps(1, pre)
            pre += 1
    run_steps(args.warmup, pre)
    args_start = pre + args.warmup
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps, args_start)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    loss = float(model.head._pl.out[-1])
    assert loss == loss and abs(loss) < 1e3, "training diverged (loss=%r)" % loss

    # ---- a longer window of the same replayed step, outside the contract's timed region: the driver's default 20 steps are
    # 0.12 s, which does not resolve a per-cent effect ----
    ms_long = None
    if world == 1 and args.steps < 200 and hook.graph_replays > 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(200, args_start + args.steps)
        torch.cuda.synchronize()
        ms_long = (time.perf_counter() - t1) / 200 * 1e3

    # ---- live kernel timing for the roofline object (extra steps, outside the timed region) ----
    roofline, extra = None, {}
    if not args.no_kernel_profile:
        # per-launch HIP events need the launches to come from the host: these extra steps run the same kernels
        # eagerly (the timed region above replays them from a hipGraph)
        eager_hook = build(use_graph=False, **tc.training_hook)
        LaunchProfile.begin()
        nprof = 3
        for i in range(nprof):
            eager_hook(dict(batches[i % len(batches)]), model, optimizer, global_step=args_start + args.steps + i)
        rec = LaunchProfile.end()
        tagged = list(LaunchProfile.tagged)
        agg = {}
        for kind, work, dt in rec:
            a = agg.setdefault(kind, [0, 0.0, 0.0])
            a[0] += 1; a[1] += work; a[2] += dt
        # ... and the same steps once more with everything on ONE stream (no pose chain beside the depth chain, no companion
        # streams), so that every launch also has samples taken while nothing else was running.  "Alone" figures = every
        # launch priced at the MINIMUM duration seen for its grid (kind + shape tag) over all 2 x nprof samples — what
        # tools/rocprof_bygrid_csv.py + tools/sum_alone.py compute from a rocprofv3 trace (profiles/r06*_sum_alone_by_family.txt)
        ov, ws = RT.overlap, RT.wgrad_streams
        RT.overlap, RT.wgrad_streams = False, 0
        try:
            LaunchProfile.begin()
            for i in range(nprof):
                eager_hook(dict(batches[i % len(batches)]), model, optimizer, global_step=args_start + args.steps + nprof + i)
            LaunchProfile.end()
            serial = list(LaunchProfile.tagged)
        finally:
            RT.overlap, RT.wgrad_streams = ov, ws
        best = {}
        for kind, work, dt, tag in tagged + serial:
            key = (kind, tag if tag is not None else work)
            best[key] = min(best.get(key, dt), dt)
        alone = {}
        for kind, work, dt, tag in tagged:                       # the in-situ passes' launch list, re-priced
            a = alone.setdefault(kind, [0, 0.0, 0.0])
            a[0] += 1; a[1] += work; a[2] += best[(kind, tag if tag is not None else work)]
        # sum of the per-launch durations of one step (HIP events around every launch of the eager steps; their streams still
        # overlap, so this is the in-situ sum — the alone-time sum comes from the rocprof per-grid table, profiles/r05*_by_grid)
        extra["sum_launch_ms"] = round(sum(a[2] for a in agg.values()) / nprof * 1e3, 3)
        extra["launches_per_step"] = sum(a[0] for a in agg.values()) // nprof
        # the same launches at their grids' minimum durations: what the replayed step's duration tracks
        extra["sum_alone_ms"] = round(sum(a[2] for a in alone.values()) / nprof * 1e3, 3)
        extra["timed_launches_note"] = ("HIP-event pairs around the convolution, BatchNorm, pooling / upsample, bias-sum and "
                                        "photometric launches (the loss's small kernels, heads, clip + Adam, re-pack and "
                                        "zeroing launches are not bracketed: see profiles/ for the rocprofv3 totals)")
        def tf(a):
            return a[1] / a[2] / 1e12
        # (the committed PMC passes were taken on the default workload: no figure for any other)
        default_workload = (args.depth, args.height, args.width, args.batch, args.dtype, fisheye) == (18, 192, 640, 12, "bf16", False)
        # dominant kernel family by time: 3x3/s1 fwd+dgrad on the LDS-halo kernels (fs_conv3x3_halo picks per launch — the
        # 32x32-tile kernel, the 16x16-tile kernel, the persistent one-chunk kernel of the decoder's 16-channel layers; the
        # profile kind says which one ran: fs_conv3x3_halo_plan)
        parts = {k: agg[k] for k in ("conv3x3_t32", "conv3x3_halo", "conv3x3_p1") if k in agg}
        ch = [sum(a[i] for a in parts.values()) for i in range(3)]
        ach = tf(ch)
        roofline = {"kernel": "%s (3x3/s1 fwd+dgrad LDS-halo family, %d launches/step)" % (" + ".join(k + "_kernel" for k in parts), ch[0] // nprof),
                    "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_TFLOPS[args.dtype], 4), "traffic": pmc_traffic("conv3x3_halo") if default_workload else None,
                    "avg_launch_us": round(ch[2] / ch[0] * 1e6, 2),
                    "algorithmic_flop_per_launch": round(ch[1] / ch[0]),
                    "kernel_split": {k + "_kernel": {"launches_per_step": a[0] // nprof, "avg_launch_us": round(a[2] / a[0] * 1e6, 2),
                                                     "achieved_tflops": round(tf(a), 2)} for k, a in parts.items()}}
        pa_ = [a for k, a in alone.items() if k in parts]
        if pa_:
            ca = [sum(a[i] for a in pa_) for i in range(3)]
            roofline["frac_alone"] = round(tf(ca) / PEAK_TFLOPS[args.dtype], 4)
            roofline["achieved_alone"] = round(tf(ca), 2)
            roofline["avg_launch_alone_us"] = round(ca[2] / ca[0] * 1e6, 2)
        allc = [ch]
        # (a family may be absent: with the 1x1 GEMM kernel the fisheye configuration has no implicit-GEMM launch left)
        for key, name in (("conv_igemm", "conv_igemm_generic"), ("conv_wgrad", "conv_wgrad")):
            if key in agg and agg[key][0]:
                a = agg[key]
                extra[name] = {"achieved_tflops": round(tf(a), 2), "launches_per_step": a[0] // nprof,
                               "avg_launch_us": round(a[2] / a[0] * 1e6, 2)}
                allc.append(a)
        if "conv3x3_s2" in agg:     # 3x3 / stride-2 forward of the ResNet stage entries (LDS-halo kernel, stride-2 variant)
            c2 = agg["conv3x3_s2"]
            extra["conv3x3_s2"] = {"achieved_tflops": round(tf(c2), 2), "launches_per_step": c2[0] // nprof,
                                   "avg_launch_us": round(c2[2] / c2[0] * 1e6, 2)}
            allc.append(c2)
        if "conv3x3_s2d" in agg:    # 3x3 / stride-2 data gradient, four parity classes from one dY halo (+ the 1x1 downsample's)
            c2 = agg["conv3x3_s2d"]
            extra["conv3x3_s2d"] = {"achieved_tflops": round(tf(c2), 2), "launches_per_step": c2[0] // nprof,
                                    "avg_launch_us": round(c2[2] / c2[0] * 1e6, 2)}
            allc.append(c2)
        if "conv1x1" in agg:        # 1x1 forward / stride-1 data gradient on the GEMM kernels of fs_conv1x1
            c1 = agg["conv1x1"]
            extra["conv1x1"] = {"achieved_tflops": round(tf(c1), 2), "launches_per_step": c1[0] // nprof,
                                "avg_launch_us": round(c1[2] / c1[0] * 1e6, 2)}
            allc.append(c1)
        if "conv_stem" in agg:      # 7x7 stems (forward) have their own kernel
            cs = agg["conv_stem"]
            extra["conv_stem"] = {"achieved_tflops": round(tf(cs), 2), "launches_per_step": cs[0] // nprof,
                                  "avg_launch_us": round(cs[2] / cs[0] * 1e6, 2)}
            allc.append(cs)
        extra["conv_all"] = {"achieved_tflops": round(sum(a[1] for a in allc) / sum(a[2] for a in allc) / 1e12, 2),
                             "frac_mfma_peak": round(sum(a[1] for a in allc) / sum(a[2] for a in allc) / 1e12 / PEAK_TFLOPS[args.dtype], 4)}
        for k in ("photo_fused_fwd", "photo_fused_bwd"):
            if k not in agg:
                continue
            a = agg[k]
            extra[k] = {"algorithmic_GBps": round(a[1] / a[2] / 1e9, 1), "frac_hbm_peak": round(a[1] / a[2] / 1e9 / PEAK_HBM_GBS, 4),
                        "avg_launch_us": round(a[2] / a[0] * 1e6, 2)}
            if k in alone:
                extra[k]["avg_launch_alone_us"] = round(alone[k][2] / alone[k][0] * 1e6, 2)
            # the roof these kernels are against is vector-instruction issue, not HBM (their real traffic is 0.4x the
            # algorithmic bytes): wave instructions x 4 cycles / (SIMDs x cycles of the launch), instruction count from the
            # committed SQ_INSTS_VALU pass
            vi = pmc_valu_insts(k) if default_workload else None
            if vi:
                dt_alone = (alone[k][2] / alone[k][0]) if k in alone else (a[2] / a[0])
                extra[k]["valu_insts_per_launch"] = int(vi)
                extra[k]["frac_valu_issue"] = round(vi * 4.0 / (N_SIMD * dt_alone * CLOCK_HZ), 4)
        conv_time = sum(a[2] for a in allc) / nprof
        extra["conv_time_ms_per_step"] = round(conv_time * 1e3, 3)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        line = {
            "metric": ("training samples/sec (3-frame triplets) at %dx%d, ResNet-%d depth+pose" % (H, W, args.depth) if not fisheye
                       else "training samples/sec (3-frame fisheye triplets) at %dx%d, ResNet-%d + FishEyeDecoder" % (H, W, args.depth)),
            "value": round(B * world * args.steps / elapsed, 2), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "ms_per_step_200": (round(ms_long, 3) if ms_long is not None else (round(ms, 3) if args.steps >= 200 else None)),
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("KITTI Eigen-Zhou-shaped synthetic triplets, ResNet-%d depth+pose, %dx%d, %s, "
                                    "batch %d/GPU, full step (fwd+loss+bwd+clip35+Adam); inputs HBM-resident, no H2D in "
                                    "the timed region" % (args.depth, H, W, args.dtype, B)) if not fisheye else
                                   ("KITTI-360-fisheye-shaped synthetic triplets (Mei camera model, two calibrations per batch), "
                                    "ResNet-%d + FishEyeDecoder, 64 bins, max depth 150, dataset poses, %dx%d, %s, batch %d/GPU, "
                                    "full step (fwd+loss+bwd+clip35+Adam, weight decay 1e-5); inputs HBM-resident"
                                    % (args.depth, H, W, args.dtype, B)),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "final_loss": round(loss, 6),
                       "encoder_pass": ((("two lanes (depth + stacked pose encoder share every launch)" if RT.lanes
                                          else "two chains (depth and pose encoder on two streams)")
                                         + ("" if RT.dp is None else "; weight gradients " + RT.dp.wgrad_mode)) if not fisheye else "one network"),
                       "encoder_pass_ms": RT.encoder_pass_ms, "autotune_steps": pre,
                       "hipgraph_replays": hook.graph_replays, "frames_per_s": round(3 * B * world * args.steps / elapsed, 1),
                       "dp_collectives": (None if RT.dp is None else ("rccl-direct" + ("+hipgraph" if RT.dp.capturable else "")
                                                                        if RT.dp.direct else "torch.distributed")),
                       "dp_world": (None if RT.dp is None else RT.dp.world),
                       "dp_capture_selftest": (None if RT.dp is None or RT.dp._direct is None else RT.dp._direct.capture_test),
                       "syncbn_exchanges_per_step": (None if RT.dp is None else RT.dp.n_small),
                       "gradient_buckets_per_step": (None if RT.dp is None else RT.dp.n_bucket)},
            "roofline": roofline, "kernels": extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        if RT.dp is not None:
            RT.dp.close()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
