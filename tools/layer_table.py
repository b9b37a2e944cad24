"""Per-layer conv timing table (HIP events around each launch, eager execution, default bench workload).
    python tools/layer_table.py [--batch 12] > profiles/<name>.md"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--depth", type=int, default=18)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    args = ap.parse_args()
    import bench
    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.hip.conv import LaunchProfile
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    dev = torch.device("cuda", 0)
    RT.set_compute_dtype("bf16")
    RT.overlap = False
    model = build(**meta_arch_cfg(args.height, args.width, with_pose=True, depth=args.depth)).to(dev).train()
    tc = training_cfg(clip_gradients=35.0, lr=1e-4)
    opt = build_optimizer(model, **tc.optimizer)
    hook = build(use_graph=False, **tc.training_hook)
    batches = bench.synthetic_device_batches(args.batch, args.height, args.width, dev, 0)
    for i in range(3):
        hook(dict(batches[i % len(batches)]), model, opt)
    LaunchProfile.begin()
    for i in range(args.steps):
        hook(dict(batches[i % len(batches)]), model, opt)
    LaunchProfile.end()
    agg = {}
    for kind, work, dt, tag in LaunchProfile.tagged:
        a = agg.setdefault((kind, tag), [0, 0.0, 0.0])
        a[0] += 1; a[1] += work; a[2] += dt
    tot = sum(a[2] for a in agg.values())
    print("| kernel family | layer | launches/step | us/launch | TFLOP/s (or GB/s) | ms/step | % |")
    print("|---|---|---|---|---|---|---|")
    for (kind, tag), a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        rate = a[1] / a[2] / (1e12 if kind.startswith("conv") else 1e9)
        print("| %s | %s | %.1f | %.1f | %.1f | %.3f | %.1f |" % (kind, tag, a[0] / args.steps, a[2] / a[0] * 1e6, rate,
                                                               a[2] / args.steps * 1e3, 100 * a[2] / tot))
    print("\ntotal timed %.3f ms/step" % (tot / args.steps * 1e3))


if __name__ == "__main__":
    main()
