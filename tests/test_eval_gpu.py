"""Device-side depth evaluation (fs_resize_linear, fs_depth_eval, KittiEigenEvaluator, KittiEvaluationHook) against
the numpy oracle of the reference's evaluation (oracle/eval_oracle.py; compute_errors pinned to the reference)."""
import numpy as np
import pytest
import torch

from oracle import eval_oracle as EO

pytestmark = pytest.mark.gpu


def _pair(rng, H, W, h, w, density=0.3):
    gt = np.zeros((H, W), np.float32)
    m = rng.rand(H, W) < density                       # sparse like projected lidar
    gt[m] = (rng.rand(int(m.sum())) * 85).astype(np.float32)      # some beyond 80 m: masked out
    base = rng.rand(h, w).astype(np.float32) * 40 + 1
    return base, gt


@pytest.mark.parametrize("shape", [(375, 1242, 192, 640), (96, 320, 96, 320), (64, 200, 33, 77)])
def test_resize_linear_matches_cv2_restatement(dev, shape):
    H, W, h, w = shape
    rng = np.random.RandomState(H + w)
    src = rng.rand(h, w).astype(np.float32) * 30 + 0.5
    from fsnet_amd.hip import ops
    got = ops.resize_linear(torch.from_numpy(src).to(dev), H, W).cpu().numpy()
    want = EO.cv2_resize_linear(src, W, H)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
    got_inv = ops.resize_linear(torch.from_numpy(src).to(dev), H, W, invert=True).cpu().numpy()
    want_inv = 1 / EO.cv2_resize_linear(1 / src, W, H)
    assert np.abs(got_inv - want_inv).max() <= 5e-6 * np.abs(want_inv).max()


@pytest.mark.parametrize("shape", [(375, 1242, 192, 640), (370, 1226, 370, 1226), (60, 200, 31, 99)])
def test_depth_eval_matches_oracle(dev, shape):
    H, W, h, w = shape
    rng = np.random.RandomState(H * 3 + w)
    from fsnet_amd.monodepth.evaluation.kitti_unsupervised_eval import KittiEigenEvaluator
    preds, gts = zip(*[_pair(rng, H, W, h, w) for _ in range(3)])
    ev = KittiEigenEvaluator(gt_depths=list(gts), device=dev)
    for i in range(3):
        want = EO.single_loss(preds[i].copy(), gts[i].copy())
        got = ev.single_call(torch.from_numpy(preds[i]).to(dev), i)
        assert abs(float(got["ratio"]) - float(want["ratio"])) <= 1e-5 * float(want["ratio"])
        for key in ("error", "abs_error"):
            a, b = np.array(got[key], np.float64), np.array(want[key], np.float64)
            # sums here are f64, numpy's are float32 pairwise; threshold counts may differ by a borderline pixel
            assert np.abs(a[:4] - b[:4]).max() <= 2e-5 * max(1.0, np.abs(b[:4]).max()), (key, a, b)
            assert np.abs(a[4:] - b[4:]).max() <= 3.0 / max(1, (gts[i] > 1e-3).sum() // 4), (key, a, b)


def test_depth_eval_edge_cases(dev):
    from fsnet_amd.monodepth.evaluation.kitti_unsupervised_eval import KittiEigenEvaluator
    H, W = 100, 300
    ev = KittiEigenEvaluator(gt_depths=[np.zeros((H, W), np.float32)], device=dev)
    with pytest.raises(ValueError):                       # no valid ground truth (kitti_unsupervised_eval.py:62-63)
        ev.single_call(torch.ones(H, W, device=dev), 0)
    gt = np.zeros((H, W), np.float32)
    gt[60, 150] = 10.0                                    # a single valid pixel: median of one element
    ev = KittiEigenEvaluator(gt_depths=[gt], device=dev)
    r = ev.single_call(torch.full((H, W), 5.0, device=dev), 0)
    assert abs(float(r["ratio"]) - 2.0) < 1e-6 and r["error"][0] < 1e-6 and abs(r["abs_error"][0] - 0.5) < 1e-6
    with pytest.raises(FileNotFoundError):      # no cache: the export starts and finds no split file (like the reference)
        KittiEigenEvaluator(data_path="/nonexistent", split_file="x", gt_saved_file="/nonexistent/gt.npz")


def test_evaluation_hook_end_to_end(dev):
    """eval-mode forward of the meta-arch + device-side metrics through the reference's hook surface, against the
    same pipeline assembled from the oracle on the host"""
    from torch.utils.data import Dataset
    from fsnet_amd.configs import meta_arch_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.utils.builder import build
    from oracle import fsnet_oracle as O
    RT.set_compute_dtype(torch.float32)
    h, w, H, W = 64, 128, 90, 250
    m = build(**meta_arch_cfg(h, w, with_pose=False))
    m.load_state_dict(O.init_state(seed=2, with_pose=False), strict=True)
    m = m.to(dev)
    rng = np.random.RandomState(5)
    gts = [_pair(rng, H, W, h, w)[1] for _ in range(4)]

    class Val(Dataset):
        def __len__(self):
            return 4

        def __getitem__(self, i):
            d = {k: (v[0] if isinstance(v, torch.Tensor) else v) for k, v in O.synthetic_batch(1, h, w, seed=70 + i).items()}
            d[('image_resize', 'effective_size')] = np.array([h - 4, w - 8])
            d[('original_image', 0)] = np.zeros((H, W, 3), np.float32)     # the hook only reads its shape
            return d

    hook = build(name="fsnet_amd.monodepth.pipeline_hooks.evaluation_hooks.base_evaluation_hooks.KittiEvaluationHook",
                 test_run_hook_cfg=dict(name="fsnet_amd.vision_base.pipeline_hooks.train_val_hooks.base_validation_hooks.BaseValidationHook"),
                 dataset_eval_cfg=dict(name="fsnet_amd.monodepth.evaluation.kitti_unsupervised_eval.KittiEigenEvaluator",
                                       gt_depths=gts, device=dev),
                 batch_size=2, num_workers=0)
    res = hook(m, Val())
    # host pipeline on the same network outputs
    m.eval()
    want = []
    with torch.no_grad():
        for i in range(4):
            d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in O.synthetic_batch(1, h, w, seed=70 + i).items()}
            depth = m(d, dict(is_training=False))["depth"][0, 0, :h - 4, :w - 8].float().cpu().numpy()
            depth_0 = 1 / EO.cv2_resize_linear(1 / depth, W, H)
            want.append(EO.single_loss(depth_0, gts[i].copy())["error"])
    want = np.array(want, np.float64).mean(0)
    assert np.abs(res["mean_errors"][:4] - want[:4]).max() <= 1e-4 * max(1.0, np.abs(want[:4]).max())
    assert np.abs(res["mean_errors"][4:] - want[4:]).max() <= 2e-3
