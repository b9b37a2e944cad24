"""The corridor scene of tests/helpers_scene.py against the oracle's loss chain (monodepth2_decoder.py:61-116, 205-304): the
generating depth and camera motion must BE the photometric minimum — otherwise a training test on it shows nothing."""
import torch
import torch.nn.functional as F

from oracle import fsnet_oracle as O
from tests.helpers_scene import corridor_batch, log_depth_correlation

H, W = 64, 208


def _loss(batch, depth, Ts, scales=(0, 1, 2, 3)):
    outputs = {}
    for s in scales:
        d = F.interpolate(depth, [H >> s, W >> s], mode="bilinear", align_corners=True)
        outputs[("depth", s, s)] = d
        outputs[("disp", s)] = 1.0 / d
    for f, T in Ts.items():
        outputs[("cam_T_cam", f)] = T
    total, losses = O.photometric_loss(outputs, batch, scales=scales)
    return float(total), outputs


def test_generating_geometry_is_the_photometric_minimum():
    batch, truth = corridor_batch(3, H, W, seed=5)
    Ts = {f: truth[("T", f)] for f in (1, -1)}
    at_truth, out = _loss(batch, truth["depth"], Ts)
    # the reconstruction from the true geometry is the target (bilinear resampling + occlusion at the frame border apart)
    for f in (1, -1):
        err = (out[("original_image", f, 0)] - batch[("original_image", 0)]).abs().mean(1)
        inside = out[("overlapped_mask", f, 0)]
        assert float(err[inside].mean()) < 0.02, float(err[inside].mean())
    ident = {f: torch.eye(4).repeat(3, 1, 1) for f in (1, -1)}
    flat = torch.full_like(truth["depth"], float(truth["depth"].median()))
    wrong = {
        "no motion": _loss(batch, truth["depth"], ident)[0],
        "flat depth": _loss(batch, flat, Ts)[0],
        "depth x 2": _loss(batch, truth["depth"] * 2.0, Ts)[0],
        "depth / 2": _loss(batch, truth["depth"] * 0.5, Ts)[0],
        "swapped frames": _loss(batch, truth["depth"], {1: Ts[-1], -1: Ts[1]})[0],
    }
    for name, v in wrong.items():
        assert at_truth < 0.6 * v, (name, at_truth, v)
    # scale ambiguity of a learned pose: (k depth, k translation) is the same minimum
    k = 1.7
    Tk = {f: T.clone() for f, T in Ts.items()}
    for T in Tk.values():
        T[:, :3, 3] *= k
    assert abs(_loss(batch, truth["depth"] * k, Tk)[0] - at_truth) < 0.05 * at_truth


def test_batch_contract_and_determinism():
    a, ta = corridor_batch(2, H, W, seed=9)
    b, tb = corridor_batch(2, H, W, seed=9)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    ref = O.synthetic_batch(2, H, W)
    assert set(a.keys()) == set(ref.keys())
    for k in ref:
        assert a[k].shape == ref[k].shape and a[k].dtype == ref[k].dtype, k
    assert torch.equal(a["P2"], ref["P2"])
    assert float(ta["depth"].min()) > 2.0 and float(ta["depth"].max()) <= 60.0
    assert 0.999 < log_depth_correlation(ta["depth"], ta["depth"] * 3.0) < 1.0 + 1e-9
    img = a[("original_image", 0)]
    assert 0.0 <= float(img.min()) and float(img.max()) <= 1.0 and float(img.std()) > 0.08
