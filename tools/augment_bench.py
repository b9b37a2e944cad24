"""Throughput of the device input pipeline at the reference's training shape: batches of 12
samples x 3 raw 375x1242 uint8 frames -> ('image', i), ('original_image', i) at 192x640 + patched mask.
Prints one JSON line: kernel time, samples/s with the frames already in HBM and with the pinned H2D upload inside
the timed region, and the numpy oracle's time per sample on the host for scale.
    python tools/augment_bench.py [--batch 12] [--iters 50]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from fsnet_amd.vision_base.data.augmentations.augmentations import DeviceAugment, PLAN  # noqa: E402
from fsnet_amd.vision_base.utils.builder import build  # noqa: E402
from tests import helpers_augment as HA  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = dict(HA.golden())
    g["out_h"], g["out_w"] = np.int64(192), np.int64(640)
    transform = build(**HA.pipeline_cfg(g))
    rs = np.random.RandomState(0)
    frames = [rs.randint(0, 256, size=(375, 1242, 3)).astype(np.uint8) for _ in HA.FRAME_IDXS]
    P2 = np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.0027]])
    poses = [np.eye(4, dtype=np.float32)] * 2
    t0 = time.perf_counter()
    samples = [transform(HA.sample_dict(frames, P2, poses)) for _ in range(args.batch)]
    host_plan_ms = (time.perf_counter() - t0) * 1e3 / args.batch
    aug = DeviceAugment(HA.FRAME_IDXS)
    t0 = time.perf_counter()
    batch = aug.collate(samples)
    collate_ms = (time.perf_counter() - t0) * 1e3
    plan = batch[PLAN]

    def run(resident):
        b = {PLAN: dict(plan)}
        if resident:
            b[PLAN]["src"] = src_dev
        return aug.materialize(b, dev)

    src_dev = plan["src"].to(dev)
    out = {}
    for resident in (True, False):
        for _ in range(3):
            run(resident)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.iters):
            run(resident)
        torch.cuda.synchronize()
        out[resident] = (time.perf_counter() - t0) / args.iters
    # kernel alone (events around the launch, tensors allocated once)
    import ctypes as C
    from fsnet_amd.hip.binding import lib, check, stream_ptr, FsAugArgs
    B, F = args.batch, 3
    image = torch.empty(F, B, 3, 192, 640, device=dev); orig = torch.empty_like(image)
    mask = torch.empty(B, 192, 640, dtype=torch.float64, device=dev)
    minv, iplan, fplan = plan["minv"].to(dev), plan["iplan"].to(dev), plan["fplan"].to(dev)
    a = FsAugArgs()
    a.src, a.minv, a.iplan, a.fplan = src_dev.data_ptr(), minv.data_ptr(), iplan.data_ptr(), fplan.data_ptr()
    a.image, a.original, a.mask = image.data_ptr(), orig.data_ptr(), mask.data_ptr()
    for k in range(3):
        a.mean[k], a.std[k] = float(plan["mean"][k]), float(plan["std"][k])
    a.B, a.F, a.Hs, a.Ws, a.H, a.W = B, F, 375, 1242, 192, 640
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        check(lib.fs_augment_frames(C.byref(a), stream_ptr()), "aug")
    e0.record()
    for _ in range(args.iters):
        check(lib.fs_augment_frames(C.byref(a), stream_ptr()), "aug")
    e1.record()
    torch.cuda.synchronize()
    kernel_us = e0.elapsed_time(e1) / args.iters * 1e3
    # bytes the kernel must move: outputs 2 x F x 3 x H x W fp32 + mask f64; inputs: the sampled source pixels
    out_bytes = B * (2 * F * 3 * 192 * 640 * 4 + 192 * 640 * 8)
    from oracle import augment_oracle as A
    p = samples[0][PLAN]
    oplan = dict(M=p["warp"]["M"], mirror=p["mirror"], order=[o for o, _ in p["ops"]],
                 brightness=dict(p["ops"]).get(0), contrast=dict(p["ops"]).get(1), saturation=dict(p["ops"]).get(2))
    t0 = time.perf_counter()
    A.run_sample(frames, oplan, 640, 192, g["mean"], g["std"])
    oracle_ms = (time.perf_counter() - t0) * 1e3
    print(json.dumps({
        "workload": "B=%d x 3 frames 375x1242 u8 -> 192x640 (warp, mirror, colour chain, normalise) + mask" % B,
        "kernel_us": round(kernel_us, 1), "kernel_out_GBps": round(out_bytes / kernel_us / 1e3, 1),
        "samples_per_s_resident": round(B / out[True], 1), "samples_per_s_with_h2d": round(B / out[False], 1),
        "host_plan_ms_per_sample": round(host_plan_ms, 3), "host_collate_ms_per_batch": round(collate_ms, 2),
        "numpy_oracle_ms_per_sample": round(oracle_ms, 1)}))


if __name__ == "__main__":
    main()
