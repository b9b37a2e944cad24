"""Fisheye path (BASELINE configs[3]) on the GPU: the Mei ray table, cam2image inside the fused loss
chain (forward + backward) and a FishEyeDecoder training step, against the golden vectors of the REAL reference
(tests/golden/fisheye.npz) and the CPU oracle (oracle/fisheye_oracle.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import fisheye_oracle as FO
from oracle import fsnet_oracle as O
from tests.test_oracle_golden import fisheye_chain_inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def to_dev(data, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}


def test_mei_ray_table_matches_reference_golden(dev):
    from fsnet_amd.monodepth.networks.utils.mei_fisheye_utils import MeiCameraProjection
    g = np.load(os.path.join(GOLD, "fisheye.npz"))
    proj = MeiCameraProjection()
    for v in range(2):
        P = torch.from_numpy(g["lut%d_P" % v])
        k1, k2, xi = (float(x) for x in g["lut%d_calib" % v])
        calib = {"distortion_parameters": {"k1": k1, "k2": k2}, "mirror_parameters": {"xi": xi}}
        norm = torch.ones(1, 1, 48, 48, device=dev)
        pts, mask = proj.image2cam(norm, P[None], [calib])
        torch.cuda.synchronize()
        ref = g["lut%d" % v]
        got = np.concatenate([pts[0, 0].permute(2, 0, 1).cpu().numpy(), mask[0].cpu().numpy()], 0)
        assert (got[3] == ref[3]).mean() == 1.0                       # validity mask: exact
        assert np.abs(got[:3] - ref[:3]).max() < 2e-6                 # bisection tolerance of the reference is 1e-6
        assert len(proj.cache) == v + 1
    proj.image2cam(norm, P[None], [calib])
    assert len(proj.cache) == 2                                       # second use of a calibration: cached table


def test_fisheye_chain_vs_reference_golden(dev):
    """fused loss chain in ray-table mode: loss_dict, warped images, overlap masks, d/d norm_s and d/dT against the
    REAL FishEyeDecoder.loss"""
    from fsnet_amd.hip import ops
    from fsnet_amd.monodepth.networks.utils.mei_fisheye_utils import MeiCameraProjection
    g = np.load(os.path.join(GOLD, "fisheye.npz"))
    data = fisheye_chain_inputs(g)
    B, H, W = data["P2"].shape[0], int(g["H"]), int(g["W"])
    depths = [torch.from_numpy(g["depth_%d" % s]) for s in range(4)]
    disps = [O.depth_to_disp(d, 0.5, 150.0) for d in depths]
    Ts, leaves = [], []
    for f, tag in ((1, "p"), (-1, "m")):
        aa = torch.from_numpy(g["aa_" + tag]).requires_grad_(True)
        tr = torch.from_numpy(g["tr_" + tag]).requires_grad_(True)
        leaves.append((aa, tr))
        Ts.append(O.transformation_from_parameters(aa, tr, invert=(f < 0)))
    pl = ops.PhotometricLoss(B, H, W, [0, 1, 2, 3], dev, 0.5, 150.0)
    tabs, rows = MeiCameraProjection().tables(H, W, data["P2"], data["calib_meta"], dev)
    pl.stage_fisheye(tabs, rows)
    out = pl.forward(data[("original_image", 0)].to(dev), [data[("original_image", 1)].to(dev), data[("original_image", -1)].to(dev)],
                     data["P2"].to(dev), [t.detach().to(dev).contiguous() for t in Ts], data["patched_mask"].to(dev),
                     [d.to(dev) for d in depths], [d.to(dev) for d in disps], noise_seed=-1)
    d_depth, d_disp, dT = pl.backward()
    torch.cuda.synchronize()
    out = out.cpu()
    assert abs(float(out[8]) - float(g["total_loss"])) < 5e-7
    for s in range(4):
        assert abs(float(out[s]) - float(g["ld_loss_%d" % s])) < 1e-6
        assert abs(float(out[4 + s]) - float(g["ld_smooth_loss_%d" % s])) < 1e-9
    for f, tag in ((0, "p"), (1, "m")):
        assert (pl.pred[0, f].cpu()[:, :, ::2, ::2] - torch.from_numpy(g["warp0_" + tag])).abs().max() < 3e-5
        assert (pl.ov[0, f].cpu().bool().numpy() == g["ovmask0_" + tag]).mean() > 0.9995
    for s in range(4):
        d = depths[s].clone().requires_grad_(True)
        O.depth_to_disp(d, 0.5, 150.0).backward(d_disp[s].cpu())
        got = d_depth[s].cpu() + d.grad
        ref = torch.from_numpy(g["gdepth_%d" % s])
        assert float((got - ref).norm() / ref.norm()) < 3e-3, s
    for i, tag in enumerate(("p", "m")):
        aa, tr = leaves[i]
        (Ts[i] * dT[i].cpu()).sum().backward()
        for got, key in ((aa.grad, "gaa_" + tag), (tr.grad, "gtr_" + tag)):
            ref = torch.from_numpy(g[key])
            assert (got - ref).abs().max() < 1e-2 * ref.abs().max() + 1e-9, key


def _fisheye_model(H, W, dev, sd0, dtype=torch.float32):
    from fsnet_amd.configs import meta_arch_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.utils.builder import build
    RT.set_compute_dtype(dtype)
    RT.tie_noise = False
    m = build(**meta_arch_cfg(H, W, with_pose=False, num_output_channels=64, max_depth=150.0, fisheye=True))
    m.load_state_dict({k: v.clone() for k, v in sd0.items()}, strict=True)
    return m.to(dev).train()


def _fisheye_batch(B, H, W, seed):
    data = O.synthetic_batch(B, H, W, seed=seed)
    Ps, calibs = zip(*[FO.synthetic_calib(H, W, b % 2) for b in range(B)])
    data["P2"] = torch.stack(Ps, 0)
    data["calib_meta"] = list(calibs)
    for f, tx in ((1, 0.6), (-1, -0.6)):
        T = torch.eye(4).repeat(B, 1, 1)
        T[:, 0, 3] = tx
        T[:, 2, 3] = 0.03
        data[("relative_pose", f)] = T
    return data


def _oracle_step(sd0, data):
    sd = {k: v.clone().requires_grad_(O.is_param(k) and v.dtype.is_floating_point) for k, v in sd0.items()}
    feats = O.resnet_forward(sd, "depth_backbone.", data[("image", 0)])
    outputs = O.depth_decoder_forward(sd, "head.depth_decoder.", feats, 0.5, 150.0)
    for f in (1, -1):
        outputs[("cam_T_cam", f)] = data[("relative_pose", f)]
    total, ld = FO.photometric_loss(outputs, data)
    names = [k for k in sd if O.is_param(k)]
    grads = torch.autograd.grad(total, [sd[k] for k in names], allow_unused=True)
    return total.detach(), ld, outputs, dict(zip(names, grads))


@pytest.mark.parametrize("graph_warmup", [99, 2], ids=["eager", "hipgraph"])
def test_fisheye_training_step_matches_oracle(dev, graph_warmup):
    """configs[3] wiring (MonoDepthWPose + FishEyeDecoder, ResNet-18, 64 bins, max depth 150, 384x384 scaled down):
    loss and parameter gradients of a step against the oracle; three hook steps eager and replayed from a hipGraph
    (calibrations change between steps: the staged pointer table must follow)."""
    from fsnet_amd.configs import training_cfg
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    B, H, W = 2, 64, 64
    sd0 = O.init_state(seed=5, with_pose=False, num_out=64, max_depth=150.0)
    m = _fisheye_model(H, W, dev, sd0)
    data = _fisheye_batch(B, H, W, seed=61)
    out = m(to_dev(data, dev), dict(is_training=True))
    out["loss"].backward()
    torch.cuda.synchronize()
    total, ld, o_out, raw = _oracle_step(sd0, data)
    assert abs(float(out["loss"].detach()) - float(total)) < 2e-5 * abs(float(total))
    gmax = max(float(r.norm()) for r in raw.values() if r is not None)
    for k, p in m.named_parameters():
        ref = raw[k]
        if ref is None or float(ref.norm()) < 1e-3 * gmax:
            continue
        rel = float((p.grad.cpu() - ref).norm() / ref.norm())
        assert rel < 3e-2, (k, rel)
    # prediction: z = Z_table x norm
    m.eval()
    with torch.no_grad():
        pred = m(to_dev(data, dev), dict(is_training=False))
    m.train()
    sde = {k: v.clone() for k, v in m.state_dict().items()}
    fe = O.resnet_forward({k: v.cpu() for k, v in sde.items()}, "depth_backbone.", data[("image", 0)], train=False)
    oe = O.depth_decoder_forward({k: v.cpu() for k, v in sde.items()}, "head.depth_decoder.", fe, 0.5, 150.0, train=False)
    zref = FO.get_prediction(oe[("depth", 0, 0)], data["P2"], data["calib_meta"])
    assert float((pred["depth"].cpu() - zref).abs().max()) < 2e-3 * float(zref.abs().max())
    # hook steps: eager vs replayed
    m2 = _fisheye_model(H, W, dev, sd0)
    tc = training_cfg()
    opt = build_optimizer(m2, name="adam", lr=1e-4, weight_decay=1e-5)
    hook = build(**dict(tc.training_hook, graph_warmup=graph_warmup))
    losses = []
    for it in range(4):
        d = _fisheye_batch(B, H, W, seed=70 + it)
        if it % 2:
            d["calib_meta"] = d["calib_meta"][::-1]
            d["P2"] = d["P2"].flip(0)
        o = hook(to_dev(d, dev), m2, opt)
        torch.cuda.synchronize()
        losses.append(float(o["loss"]))
    if graph_warmup == 2:
        assert hook.graph_replays >= 1
    # the first step is the oracle's step on the same weights
    d0 = _fisheye_batch(B, H, W, seed=70)
    t0, _, _, _ = _oracle_step(sd0, d0)
    assert abs(losses[0] - float(t0)) < 2e-5 * abs(float(t0))
    assert all(np.isfinite(losses))
    test_fisheye_training_step_matches_oracle.losses = getattr(test_fisheye_training_step_matches_oracle, "losses", {})
    test_fisheye_training_step_matches_oracle.losses[graph_warmup] = losses
    both = test_fisheye_training_step_matches_oracle.losses
    if len(both) == 2:
        for a, b in zip(both[99], both[2]):
            assert abs(a - b) < 1e-4 * abs(a)
