"""Pins oracle/fsnet_oracle.py (the CPU restatement) to golden vectors produced by the REAL
reference (tools/gen_golden.py, run in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import fsnet_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def T(a):
    return torch.from_numpy(np.asarray(a))


def maxdev(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


def test_ops_golden():
    g = np.load(os.path.join(GOLD, "ops.npz"))
    depth, P2 = T(g["geo_depth"]), T(g["geo_P2"])
    aa, tr = T(g["geo_aa"]), T(g["geo_tr"])
    assert maxdev(O.transformation_from_parameters(aa, tr, False), g["geo_T"]) < 1e-6
    assert maxdev(O.transformation_from_parameters(aa, tr, True), g["geo_Tinv"]) < 1e-6
    K, invK = O.intrinsics(P2)
    B, _, H, W = depth.shape
    pix = O.project(O.backproject(depth, invK), K, T(g["geo_T"]), H, W)
    assert maxdev(pix, g["geo_pix"]) < 1e-5
    x, y = T(g["ssim_x"]), T(g["ssim_y"])
    assert maxdev(O.ssim(x, y), g["ssim_out"]) < 1e-6
    assert maxdev(O.reprojection_loss(x, y), g["reproj_out"]) < 1e-6
    assert maxdev(O.smooth_loss(T(g["smooth_disp"]), x), g["smooth_out"]) < 1e-7
    assert maxdev(O.depth_bins(0.5, 100.0, 16), g["head_bins"]) < 1e-5
    d = O.gather_activation(T(g["head_logits"]), T(g["head_bins"]))
    assert maxdev(d, g["head_depth"]) < 1e-4
    assert maxdev(O.depth_to_disp(d, 0.5, 100.0), g["head_disp"]) < 1e-6


def _chain_inputs(g):
    data = {("original_image", 0): T(g["img_0"]), ("original_image", 1): T(g["img_p"]),
            ("original_image", -1): T(g["img_m"]), "P2": T(g["P2"]), "patched_mask": T(g["patched_mask"])}
    return data


def test_loss_chain_golden():
    g = np.load(os.path.join(GOLD, "loss_chain.npz"))
    data = _chain_inputs(g)
    outputs, leaves = {}, {}
    for s in range(4):
        d = T(g["depth_%d" % s]).clone().requires_grad_(True)
        leaves[s] = d
        outputs[("depth", s, s)] = d
        outputs[("disp", s)] = O.depth_to_disp(d, 0.5, 100.0)
    pose = {}
    for f, tag in ((1, "p"), (-1, "m")):
        aa = T(g["aa_" + tag]).clone().requires_grad_(True)
        tr = T(g["tr_" + tag]).clone().requires_grad_(True)
        pose[tag] = (aa, tr)
        outputs[("cam_T_cam", f)] = O.transformation_from_parameters(aa, tr, invert=(f < 0))
    total, ld = O.photometric_loss(outputs, data)
    assert total.dtype == torch.float64  # patched_mask promotes the loss to float64
    total.backward()
    # the reference adds randn*1e-5 tie-break noise: bounded effect on the loss
    assert abs(float(total.detach()) - float(g["total_loss"])) < 2e-7
    for s in range(4):
        assert abs(float(ld["loss/%d" % s]) - float(g["ld_loss_%d" % s])) < 5e-7
        assert abs(float(ld["smooth_loss/%d" % s]) - float(g["ld_smooth_loss_%d" % s])) < 1e-10
        ref = T(g["gdepth_%d" % s])
        # argmin flips under the tie-break noise touch isolated pixels only: compare in L2
        rel = float((leaves[s].grad - ref).norm() / ref.norm())
        assert rel < 2e-3, (s, rel)
    for tag in ("p", "m"):
        for got, key in ((pose[tag][0].grad, "gaa_" + tag), (pose[tag][1].grad, "gtr_" + tag)):
            ref = T(g[key])
            assert maxdev(got, ref) < 2e-3 * float(ref.abs().max()) + 1e-9
        assert maxdev(outputs[("original_image", 1 if tag == "p" else -1, 0)][:, :, ::4, ::4], g["warp0_" + tag]) < 1e-5
        assert (outputs[("overlapped_mask", 1 if tag == "p" else -1, 0)].numpy() == g["ovmask0_" + tag]).mean() > 0.9999


@pytest.mark.parametrize("tag,with_pose", [("depthpose", True), ("wpose", False)])
def test_model_golden(tag, with_pose):
    g = np.load(os.path.join(GOLD, "model_%s.npz" % tag))
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    sd0 = O.init_state(seed=int(g["init_seed"]), with_pose=with_pose)
    # forward tensors at the initial state
    data = O.synthetic_batch(B, H, W, seed=100)
    sd = {k: v.clone() for k, v in sd0.items()}
    feats = O.resnet_forward(sd, "depth_backbone.", data[("image", 0)])
    outs = O.depth_decoder_forward(sd, "head.depth_decoder.", feats, 0.5, 100.0)
    assert maxdev(feats[4], g["feat4"]) < 1e-4
    for s in range(4):
        ref = T(g["disp_%d" % s])
        assert maxdev(outs[("disp", s)], ref) <= 1e-5 * float(ref.abs().max())
    if with_pose:
        pf = O.resnet_forward(sd, "pose_backbone.", torch.cat([data[("image", 0)], data[("image", 1)]], 1))
        aa, tr = O.pose_decoder_forward(sd, "head.pose_decoder.", pf[-1])
        assert maxdev(aa, g["axisangle_p"]) < 1e-7 and maxdev(tr, g["translation_p"]) < 1e-7
    # three optimisation steps (clip 35, Adam 1e-4) in lock-step with the reference hook
    trn = O.OracleTrainer(sd0, with_pose=with_pose)
    for it in range(3):
        total, ld, _, raw, norm = trn.step(O.synthetic_batch(B, H, W, seed=100 + it))
        # tolerances = the reference's own run-to-run spread under its tie-break randn
        assert abs(float(total) - float(g["loss_%d" % it])) < 5e-5 * abs(float(g["loss_%d" % it]))
        assert abs(float(norm) - float(g["totalnorm_%d" % it])) < 2e-2 * float(g["totalnorm_%d" % it])
        if it == 0:
            gn = torch.stack([raw[k].norm() for k in trn.names])
            ref = T(g["gradnorm_0"])
            big = ref > 1e-4 * ref.max()
            assert float(((gn - ref).abs() / ref)[big].max()) < 2e-2


def test_eval_oracle_matches_reference_compute_errors():
    """oracle/eval_oracle.compute_errors against the outputs of the real monodepth_utils.compute_errors"""
    from oracle import eval_oracle as EO
    g = np.load(os.path.join(GOLD, "eval.npz"))
    for k in range(3):
        mine = np.array(EO.compute_errors(g["gt_%d" % k], g["pred_%d" % k]), dtype=np.float64)
        assert np.abs(mine - g["err_%d" % k]).max() <= 1e-7 * max(1.0, np.abs(g["err_%d" % k]).max())


def test_eval_oracle_resize_and_single_loss_properties():
    """the cv2.resize restatement (parity unpinned: cv2 is not in the image): identity at equal size, exact on
    affine images away from the clamped borders, constant-preserving; single_loss: perfect prediction -> zero error,
    a globally scaled prediction -> the same scaled errors and ratio = 1/scale, empty mask -> ValueError"""
    from oracle import eval_oracle as EO
    rng = np.random.RandomState(0)
    img = rng.rand(24, 40).astype(np.float32)
    assert np.array_equal(EO.cv2_resize_linear(img, 40, 24), img)
    yy, xx = np.mgrid[0:24, 0:40].astype(np.float32)
    ramp = 2 * xx + 3 * yy + 1
    up = EO.cv2_resize_linear(ramp, 80, 48)
    Y, X = np.mgrid[0:48, 0:80].astype(np.float32)
    want = 2 * ((X + 0.5) / 2 - 0.5) + 3 * ((Y + 0.5) / 2 - 0.5) + 1
    assert np.abs(up[2:-2, 2:-2] - want[2:-2, 2:-2]).max() < 1e-4
    assert np.abs(EO.cv2_resize_linear(np.full((7, 9), 3.5, np.float32), 31, 17) - 3.5).max() < 1e-6
    gt = np.zeros((100, 300), np.float32)
    gt[45:95, 20:280] = rng.rand(50, 260).astype(np.float32) * 60 + 2
    r = EO.single_loss(gt.copy() + (gt == 0), gt)
    assert abs(r["ratio"] - 1) < 1e-6 and max(r["error"][:4]) < 1e-6 and min(r["error"][4:]) == 1.0
    r2 = EO.single_loss((gt + (gt == 0)) * 0.5, gt)
    assert abs(r2["ratio"] - 2) < 1e-5 and max(r2["error"][:4]) < 1e-5 and r2["abs_error"][0] > 0.4
    import pytest
    with pytest.raises(ValueError):
        EO.single_loss(np.ones((100, 300), np.float32), np.zeros((100, 300), np.float32))


def test_distill_oracle_matches_reference_golden():
    """oracle/distill_oracle.py against losses / gradient norms of the REAL DistillWPoseMeta (tests/golden/distill.npz)"""
    from oracle import distill_oracle as D
    g = np.load(os.path.join(GOLD, "distill.npz"))
    sd = D.init_states(seed=int(g["seed"]), teacher_seed=int(g["teacher_seed"]))
    names = [k for k in sd if D.is_student_param(k)]
    for k in names:
        sd[k].requires_grad_(True)
    tot, losses, _ = D.forward_train(sd, O.synthetic_batch(int(g["B"]), int(g["H"]), int(g["W"]), seed=int(g["batch_seed"])))
    tot.backward()
    assert abs(float(tot.detach()) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
    for s in range(4):
        assert abs(float(losses["distilation/%d" % s]) - float(g["ld_distilation_%d" % s])) < 1e-5 * float(g["ld_distilation_%d" % s])
    assert float((sd["head.depth_decoder.decoder.14.weight"].grad - torch.from_numpy(g["unc_w_grad"])).abs().max()) \
        < 1e-5 * float(np.abs(g["unc_w_grad"]).max())


def fisheye_chain_inputs(g):
    """batch + synthetic network outputs of tests/golden/fisheye.npz (shared with the GPU parity test)"""
    data = _chain_inputs(g)
    data["calib_meta"] = [{"distortion_parameters": {"k1": float(c[0]), "k2": float(c[1])},
                           "mirror_parameters": {"xi": float(c[2])}} for c in g["calib"]]
    return data


def test_fisheye_oracle_matches_reference_golden():
    """oracle/fisheye_oracle.py against the REAL MeiCameraProjection / FishEyeDecoder (tests/golden/fisheye.npz):
    LUT (X, Y, Z, mask), cam2image, loss chain with gradients, get_prediction"""
    from oracle import fisheye_oracle as FO
    g = np.load(os.path.join(GOLD, "fisheye.npz"))
    for v in range(2):
        P, (k1, k2, xi) = g["lut%d_P" % v], g["lut%d_calib" % v]
        lut = np.stack(FO.mei_lut(48, 48, float(P[0, 0]), float(P[1, 1]), float(P[0, 2]), float(P[1, 2]), k1, k2, xi), 0)
        assert (lut[3] == g["lut%d" % v][3]).all()
        assert np.abs(lut[:3] - g["lut%d" % v][:3]).max() < 1e-6
        assert 0.5 < lut[3].mean() < 0.8          # the mirror equation has no solution in the image corners
    P, calib = FO.synthetic_calib(48, 48, 0)
    u, v_ = FO.cam2image(T(g["c2i_points"]), P, calib)
    assert maxdev(torch.stack([u, v_], -1), g["c2i_uvz"][..., :2]) < 1e-4
    data = fisheye_chain_inputs(g)
    outputs, leaves, pose = {}, {}, {}
    for s in range(4):
        d = T(g["depth_%d" % s]).clone().requires_grad_(True)
        leaves[s] = d
        outputs[("depth", s, s)] = d
        outputs[("disp", s)] = O.depth_to_disp(d, 0.5, 150.0)
    for f, tag in ((1, "p"), (-1, "m")):
        aa = T(g["aa_" + tag]).clone().requires_grad_(True)
        tr = T(g["tr_" + tag]).clone().requires_grad_(True)
        pose[tag] = (aa, tr)
        outputs[("cam_T_cam", f)] = O.transformation_from_parameters(aa, tr, invert=(f < 0))
    total, ld = FO.photometric_loss(outputs, data)
    total.backward()
    assert total.dtype == torch.float64
    assert abs(float(total.detach()) - float(g["total_loss"])) < 2e-7
    for s in range(4):
        assert abs(float(ld["loss/%d" % s]) - float(g["ld_loss_%d" % s])) < 5e-7
        ref = T(g["gdepth_%d" % s])
        assert float((leaves[s].grad - ref).norm() / ref.norm()) < 2e-3
    for tag in ("p", "m"):
        f = 1 if tag == "p" else -1
        for got, key in ((pose[tag][0].grad, "gaa_" + tag), (pose[tag][1].grad, "gtr_" + tag)):
            ref = T(g[key])
            # (the reference run carries its randn*1e-5 tie-break noise; the rotation gradient is a small sum of
            # large cancelling terms)
            assert maxdev(got, ref) < 5e-3 * float(ref.abs().max()) + 1e-9
        assert maxdev(outputs[("original_image", f, 0)][:, :, ::2, ::2], g["warp0_" + tag]) < 1e-5
        assert (outputs[("overlapped_mask", f, 0)].numpy() == g["ovmask0_" + tag]).mean() > 0.9999
    assert maxdev(FO.get_prediction(leaves[0].detach(), data["P2"], data["calib_meta"]), g["pred_depth"]) < 1e-5


def r50fx_batch(g):
    data = O.synthetic_batch(int(g["B"]), int(g["H"]), int(g["W"]), seed=int(g["batch_seed"]))
    for b, mul in enumerate(g["fx_mul"]):
        data["P2"][b, 0, 0] *= float(mul)
    return data


def test_r50_basefx_oracle_matches_reference_golden():
    """BASELINE configs[4] wiring (ResNet-50 Bottleneck, 64 bins, base_fx = 492, per-sample focal lengths) against
    the REAL MonoDepthWPose (tests/golden/model_r50fx.npz): forward tensors, loss, gradient norms"""
    g = np.load(os.path.join(GOLD, "model_r50fx.npz"))
    sd0 = O.init_state(seed=int(g["init_seed"]), depth=50, with_pose=False, num_out=64)
    data = r50fx_batch(g)
    sd = {k: v.clone() for k, v in sd0.items()}
    feats = O.resnet_forward(sd, "depth_backbone.", data[("image", 0)], depth=50)
    outs = O.depth_decoder_forward(sd, "head.depth_decoder.", feats, 0.5, 100.0, P2=data["P2"], base_fx=float(g["base_fx"]))
    assert maxdev(feats[4][:, ::16], g["feat4"]) < 1e-4
    for s in range(4):
        assert maxdev(outs[("depth", s, s)], g["depth_%d" % s]) <= 1e-5 * float(np.abs(g["depth_%d" % s]).max())
        assert maxdev(outs[("disp", s)], g["disp_%d" % s]) <= 1e-5
    # the focal-length scale really is per sample: fx = 0.58 W x (1.4, 1.0) over base_fx 492
    ratio = float(outs[("depth", 0, 0)][0].mean() / outs[("depth", 0, 0)][1].mean())
    assert 1.2 < ratio < 1.6
    trn = O.OracleTrainer(sd0, depth=50, with_pose=False, base_fx=float(g["base_fx"]))
    total, ld, _, raw, norm = trn.step(data)
    assert abs(float(total) - float(g["loss"])) < 5e-5 * abs(float(g["loss"]))
    gn = torch.stack([raw[k].norm() for k in trn.names])
    ref = T(g["gradnorm"])
    big = ref > 1e-3 * ref.max()
    assert float(((gn - ref).abs() / ref)[big].max()) < 3e-2


def loss_options_case(g):
    """inputs of tests/golden/loss_options.npz (shared with the GPU test)"""
    data = O.synthetic_batch(2, int(g["H"]), int(g["W"]), seed=int(g["seed"]))
    data["motion_mask"] = T(g["motion_mask"])
    return data


def test_motion_mask_and_pose_term_oracle_matches_reference_golden():
    """precomputed motion mask (monodepth2_decoder.py:243-246) + pose L1 term (:176-183) against the REAL decoder"""
    g = np.load(os.path.join(GOLD, "loss_options.npz"))
    data = loss_options_case(g)
    outputs, leaves, pose = {}, {}, {}
    for s in range(4):
        d = T(g["depth_%d" % s]).clone().requires_grad_(True)
        leaves[s] = d
        outputs[("depth", s, s)] = d
        outputs[("disp", s)] = O.depth_to_disp(d, 0.5, 100.0)
    for f, tag in ((1, "p"), (-1, "m")):
        aa = T(g["aa_" + tag]).clone().requires_grad_(True)
        tr = T(g["tr_" + tag]).clone().requires_grad_(True)
        pose[tag] = (aa, tr)
        outputs[("cam_T_cam", f)] = O.transformation_from_parameters(aa, tr, invert=(f < 0))
    total, ld = O.photometric_loss(outputs, data)
    pl = sum((data[("relative_pose", f)] - outputs[("cam_T_cam", f)]).abs().mean() for f in (1, -1))
    total = total + float(g["pose_loss_weight"]) * pl
    total.backward()
    assert abs(float(total.detach()) - float(g["total_loss"])) < 1e-6 * float(g["total_loss"])
    assert abs(float(pl.detach()) - float(g["ld_pose_loss"])) < 1e-7
    for s in range(4):
        ref = T(g["gdepth_%d" % s])
        assert float((leaves[s].grad - ref).norm() / ref.norm()) < 1e-4
        assert int((outputs[("min_idx", s)] < 2).sum()) == 0          # no identity candidates with a motion mask
    for tag in ("p", "m"):
        for got, key in ((pose[tag][0].grad, "gaa_" + tag), (pose[tag][1].grad, "gtr_" + tag)):
            ref = T(g[key])
            assert maxdev(got, ref) < 1e-4 * float(ref.abs().max()) + 1e-9


def test_sigmoid_depth_decoder_matches_reference():
    """the oracle's restatement of the base-class DepthDecoder (sigmoid disparity, disp_to_depth, base_fx scale)
    against the reference's class: outputs, feature gradients, parameter-gradient norms (sigmoid_decoder.npz)"""
    from tests.helpers_sigmoid import oracle_run, thin
    G = np.load(os.path.join(GOLD, "sigmoid_decoder.npz"))
    for tag, base_fx in (("plain", None), ("fx", 600.0)):
        o, fl, params, loss = oracle_run(base_fx, torch.float32)
        assert float(loss) == pytest.approx(float(G[tag + "_loss"]), rel=1e-5)
        for s in range(4):
            np.testing.assert_allclose(thin(o[("depth", s, s)]), G["%s_depth%d" % (tag, s)], rtol=2e-5, atol=1e-6)
            np.testing.assert_allclose(thin(o[("disp", s)]), G["%s_disp%d" % (tag, s)], rtol=2e-5, atol=1e-7)
        for k, f in enumerate(fl):
            ref = G["%s_gfeat%d" % (tag, k)]
            np.testing.assert_allclose(thin(f.grad), ref, rtol=2e-3, atol=2e-4 * float(np.abs(ref).max()))
        gn = np.array([float(v.grad.norm()) for v in params.values()])
        np.testing.assert_allclose(gn, G[tag + "_gnorm"], rtol=2e-3)


def frozen_state(seed=3):
    """initial state of the frozen-stage goldens: running statistics away from (0, 1)"""
    sd0 = O.init_state(seed=seed, with_pose=False)
    g = torch.Generator().manual_seed(5)
    for k in sd0:
        if k.endswith("running_mean"):
            sd0[k] = 0.1 * torch.randn(sd0[k].shape, generator=g)
        elif k.endswith("running_var"):
            sd0[k] = 0.5 + torch.rand(sd0[k].shape, generator=g)
    return sd0


@pytest.mark.parametrize("tag,fs,ne", [("fs1", 1, False), ("ne", -1, True)])
def test_frozen_stages_and_norm_eval_match_reference(tag, fs, ne):
    """ResNet.train() with frozen_stages / norm_eval (resnet.py:169-197): eval-mode BatchNorms and frozen parameters
    inside three training steps of the reference (frozen.npz)"""
    g = np.load(os.path.join(GOLD, "frozen.npz"))
    sd0 = frozen_state()
    trn = O.OracleTrainer(sd0, with_pose=False, frozen_stages=fs, norm_eval=ne)
    names = [k for k in sd0 if O.is_param(k)]
    for it in range(3):
        total, ld, _, raw, norm = trn.step(O.synthetic_batch(2, 64, 128, seed=300 + it))
        assert abs(float(total) - float(g["%s_loss_%d" % (tag, it)])) < 1e-4 * abs(float(g["%s_loss_%d" % (tag, it)]))
        ps = torch.stack([trn.sd[k].double().sum() for k in names]).numpy()
        ref = g["%s_psum_%d" % (tag, it)]
        # a weight moves by <= lr per step: sums agree to a few lr x numel; frozen ones exactly
        numel = np.array([trn.sd[k].numel() for k in names], dtype=np.float64)
        assert np.all(np.abs(ps - ref) <= 3e-4 * (it + 1) * np.sqrt(numel) + 1e-6 * np.abs(ref))
        for k, a, b in zip(names, ps, ref):
            if k.startswith(trn.frozen):
                assert abs(a - float(sd0[k].double().sum())) < 1e-9 and abs(b - a) < 1e-4 * max(1.0, abs(a)), k
    rm = torch.cat([trn.sd[k].flatten() for k in trn.sd if k.endswith("running_mean")]).numpy()
    np.testing.assert_allclose(rm, g[tag + "_rm_final"], atol=5e-3 if ne else 1e-5)
    nbt = np.array([int(trn.sd[k]) for k in trn.sd if k.endswith("num_batches_tracked")])
    assert np.array_equal(nbt, g[tag + "_nbt_final"])
    assert (nbt == 0).sum() == {"fs1": 5, "ne": 20}[tag]        # the eval-mode BatchNorms never count a batch


def no_overlap_case(H=64, W=96, B=2, seed=43):
    """inputs of tests/golden/no_overlap_mask.npz (as tools/gen_golden.py::no_overlap_case builds them)"""
    data = O.synthetic_batch(B, H, W, seed=seed)
    g = torch.Generator().manual_seed(18)
    depths = []
    for s in range(4):
        h, w = H >> s, W >> s
        ys = torch.linspace(0, 1, h).view(1, 1, h, 1)
        depths.append(4 + 25 * (1 - ys) + 3 * torch.rand(B, 1, h, w, generator=g))
    poses = {}
    for f in (1, -1):
        aa = 0.03 * torch.randn(B, 1, 3, generator=g)
        tr = torch.tensor([[[0.9 if f > 0 else -0.9, -0.05, -0.8 if f > 0 else 0.8]]]).repeat(B, 1, 1) + 0.02 * torch.randn(B, 1, 3, generator=g)
        poses[f] = (aa, tr)
    return data, depths, poses


def test_overlapped_mask_off_oracle_matches_reference_golden():
    """overlapped_mask=False (multi_dataset / nusc configs): every reprojection sample counts, border-clamped, also
    where it left the source frame (14.6 % of this case's samples) — against the REAL decoder (no_overlap_mask.npz)"""
    g = np.load(os.path.join(GOLD, "no_overlap_mask.npz"))
    data, depths, poses = no_overlap_case(int(g["H"]), int(g["W"]))
    outputs, leaves, pl = {}, {}, {}
    for s in range(4):
        d = depths[s].clone().requires_grad_(True)
        leaves[s] = d
        outputs[("depth", s, s)] = d
        outputs[("disp", s)] = O.depth_to_disp(d, 0.5, 100.0)
    for f, tag in ((1, "p"), (-1, "m")):
        aa, tr = poses[f][0].clone().requires_grad_(True), poses[f][1].clone().requires_grad_(True)
        pl[tag] = (aa, tr)
        outputs[("cam_T_cam", f)] = O.transformation_from_parameters(aa, tr, invert=(f < 0))
    total, ld = O.photometric_loss(outputs, data, overlapped_mask=False)
    total.backward()
    assert float(g["outside_frac"]) > 0.1 and abs(float(g["total_loss"]) - float(g["total_loss_masked"])) > 1e-3
    assert abs(float(total.detach()) - float(g["total_loss"])) < 2e-6 * float(g["total_loss"])
    for s in range(4):
        ref = T(g["gdepth_%d" % s])
        assert float((leaves[s].grad - ref).norm() / ref.norm()) < 1e-2, s       # (the reference's tie-break randn)
    for tag in ("p", "m"):
        for got, key in ((pl[tag][0].grad, "gaa_" + tag), (pl[tag][1].grad, "gtr_" + tag)):
            ref = T(g[key])
            assert maxdev(got, ref) < 1e-2 * float(ref.abs().max()) + 1e-9, key
