"""Input pipeline without a GPU: the oracle against the golden vectors made by the reference's own
augmentation classes, the mirrored host classes (draw order, P2 / pose bookkeeping, plans) against the same vectors,
and known-answer checks of the two OpenCV restatements the golden cannot pin."""
import numpy as np
import torch

from oracle import augment_oracle as A
from tests import helpers_augment as HA


def test_oracle_matches_reference_pipeline_vectors():
    g = HA.golden()
    H, W, oh, ow = int(g["H"]), int(g["W"]), int(g["out_h"]), int(g["out_w"])
    rngs = {k: np.random.default_rng(int(g["seed_" + k])) for k in ("warp", "bright", "contrast", "sat")}
    np.random.seed(int(g["global_seed"]))
    mirrored = 0
    for n in range(int(g["n"])):
        frames, P2, poses = HA.sample_inputs(g, n)
        plan = A.draw_plan(rngs["warp"], rngs["bright"], rngs["contrast"], rngs["sat"], H, W, ow, oh)
        assert plan["mirror"] == bool(g["s%d_mirror" % n]) and list(plan["order"]) == list(g["s%d_order" % n])
        imgs, origs, mask = A.run_sample(frames, plan, ow, oh, g["mean"], g["std"])
        for j in range(3):
            assert np.array_equal(imgs[j], g["s%d_image_%d" % (n, j)])          # bit-exact
            assert np.array_equal(origs[j], g["s%d_orig_%d" % (n, j)])
        assert np.array_equal(mask, g["s%d_mask" % n])
        P = A.warp_P2(P2, plan["final_scale"], plan["shift_w"], plan["shift_h"])
        if plan["mirror"]:
            P = A.mirror_P2(P, ow)
            mirrored += 1
        assert np.array_equal(P.astype(np.float32), g["s%d_P2" % n])
        for j in range(2):
            want = A.flip_relative_pose(poses[j].copy(), 0) if plan["mirror"] else poses[j]
            assert np.array_equal(want.astype(np.float32), g["s%d_pose_%d" % (n, j)])
    assert 0 < mirrored < int(g["n"])          # both branches are in the fixture


def test_host_classes_draw_and_book_keep_like_the_reference():
    from fsnet_amd.vision_base.utils.builder import build
    from fsnet_amd.vision_base.data.augmentations.augmentations import PLAN, OP_SATURATION
    g = HA.golden()
    transform = build(**HA.pipeline_cfg(g))
    np.random.seed(int(g["global_seed"]))
    for n in range(int(g["n"])):
        frames, P2, poses = HA.sample_inputs(g, n)
        out = transform(HA.sample_dict(frames, P2, poses))
        plan = out[PLAN]
        assert plan["mirror"] == bool(g["s%d_mirror" % n])
        assert [o for o, _ in plan["ops"]] == list(g["s%d_order" % n])
        assert all(v is not None for _, v in plan["ops"])                 # distort_prob = 1
        assert isinstance(out["P2"], torch.Tensor) and out["P2"].dtype == torch.float32
        assert np.array_equal(out["P2"].numpy(), g["s%d_P2" % n])
        for j, i in enumerate(HA.FRAME_IDXS[1:]):
            assert np.array_equal(np.asarray(out[("relative_pose", i)], dtype=np.float32), g["s%d_pose_%d" % (n, j)])
        assert out[("image", 0)].dtype == np.uint8                        # pixels untouched on the host
        assert (OP_SATURATION in [o for o, _ in plan["ops"]]) and not plan["hsv"]


def test_collate_builds_the_device_plan():
    from fsnet_amd.vision_base.utils.builder import build
    from fsnet_amd.vision_base.data.augmentations.augmentations import DeviceAugment, PLAN
    g = HA.golden()
    transform = build(**HA.pipeline_cfg(g))
    np.random.seed(int(g["global_seed"]))
    samples = [transform(HA.sample_dict(*HA.sample_inputs(g, n))) for n in range(int(g["n"]))]
    batch = DeviceAugment(HA.FRAME_IDXS).collate(samples)
    p = batch[PLAN]
    B = int(g["n"])
    assert p["src"].shape == (B, 3, int(g["H"]), int(g["W"]), 3) and p["src"].dtype == torch.uint8
    assert p["iplan"].shape == (B, 8) and p["fplan"].shape == (B, 4) and p["minv"].shape == (B, 6)
    for n in range(B):
        assert list(p["iplan"][n, :3]) == list(g["s%d_order" % n])
        assert int(p["iplan"][n, 3]) == 1 + 2 + 4 + 8 and int(p["iplan"][n, 4]) == int(g["s%d_mirror" % n])
        assert tuple(p["iplan"][n, 5:7].tolist()) == (int(g["H"]), int(g["W"]))
        M = samples[n][PLAN]["warp"]["M"]
        assert np.allclose(p["minv"][n].numpy(), A.invert_affine(M), rtol=0, atol=0)
    assert batch["P2"].shape == (B, 3, 4) and batch[("relative_pose", 1)].shape == (B, 4, 4)


def test_transforms_refuse_float_images():
    import pytest
    from fsnet_amd.vision_base.data.augmentations.augmentations import ConvertToFloat
    with pytest.raises(TypeError):
        ConvertToFloat(image_keys=["image"])({"image": np.zeros((4, 4, 3), dtype=np.float32)})


# ---- known answers for the OpenCV restatements (unpinned: OpenCV is not in the image) ----
def test_warp_affine_known_answers():
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, size=(20, 30, 3)).astype(np.float32)
    ident = np.array([[1, 0, 0], [0, 1, 0]], dtype=np.float32)
    assert np.array_equal(A.warp_affine_linear(img, ident, 30, 20), img)
    shift = np.array([[1, 0, 3], [0, 1, -2]], dtype=np.float32)          # dst(x, y) = src(x - 3, y + 2)
    out = A.warp_affine_linear(img, shift, 30, 20)
    assert np.array_equal(out[:18, 3:], img[2:, :27]) and not out[18:].any() and not out[:, :3].any()
    half = np.array([[1, 0, 0.5], [0, 1, 0]], dtype=np.float32)          # half-pixel shift = mean of neighbours
    out = A.warp_affine_linear(img, half, 30, 20)
    assert np.array_equal(out[:, 1:], img[:, :-1] * np.float32(0.5) + img[:, 1:] * np.float32(0.5))
    assert np.array_equal(out[:, 0], img[:, 0] * np.float32(0.5))        # BORDER_CONSTANT 0 on the left
    up = np.array([[2, 0, 0], [0, 2, 0]], dtype=np.float32)
    m = A.warp_affine_nearest(np.ones((20, 30)), up, 80, 50)
    assert m[:39, :59].all() and not m[41:].any() and not m[:, 61:].any() and m.dtype == np.float64


def test_hsv_known_answers():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [128, 128, 128], [0, 0, 0]]], dtype=np.float32)
    hsv = A.rgb2hsv(px)[0]
    assert np.allclose(hsv[:, 0], [0, 120, 240, 60, 0, 0], atol=1e-4)
    assert np.allclose(hsv[:4, 1], 1.0, atol=1e-6) and np.allclose(hsv[4:, 1], 0.0, atol=1e-6)
    assert np.array_equal(hsv[:, 2], [255, 255, 255, 255, 128, 0])
    rs = np.random.RandomState(5)
    img = rs.uniform(-20, 300, size=(16, 16, 3)).astype(np.float32)       # brightness / contrast leave [0, 255]
    back = A.hsv2rgb(A.rgb2hsv(img))
    assert np.abs(back - img).max() < 2e-3 * 300


def test_validation_resize_oracle_and_host_class_match_reference_vectors():
    g = HA.golden()
    for tag, kw in HA.VAL_CASES:
        frame = HA.val_frame(g, tag)
        assert np.array_equal(A.run_val_sample(frame, (48, 160), g["mean"], g["std"], **kw), g["val_%s_image" % tag]), tag
        out = HA.val_pipeline(g, kw)({('image', 0): frame.copy(), 'P2': HA.VAL_P2.copy()})
        assert np.array_equal(out['P2'].numpy(), g["val_%s_P2" % tag])
        assert list(out[('image_resize', 'effective_size')]) == list(g["val_%s_effective" % tag])
        assert list(out[('image_resize', 'original_shape')]) == list(g["val_%s_original" % tag])
        assert out[('image', 0)].dtype == np.uint8


def test_resize_known_answers():
    rs = np.random.RandomState(2)
    img = rs.randint(0, 256, size=(12, 20, 3)).astype(np.float32)
    assert np.array_equal(A.resize_linear(img, 20, 12), img)                       # same size: identity
    up = A.resize_linear(img, 40, 24)                                              # 2x: taps at +-0.25
    assert np.array_equal(up[0, 0], img[0, 0]) and np.array_equal(up[-1, -1], img[-1, -1])   # clamped borders
    want = img[0, 0] * np.float32(0.75) + img[0, 1] * np.float32(0.25)
    assert np.array_equal(up[0, 1], want)
    ramp = np.tile(np.arange(20, dtype=np.float32)[None, :, None], (12, 1, 3))
    out = A.resize_linear(ramp, 10, 12)                                            # 2x down: mean of the pair
    assert np.array_equal(out[:, :, 0], np.tile(np.arange(10, dtype=np.float32) * 2 + np.float32(0.5), (12, 1)))


def test_resize_training_chain_oracle_and_host_classes_match_reference_vectors():
    """the training input of the Resize-based shipped configs (multi_dataset / nusc / kitti360_fisheye examples):
    oracle pixels bit for bit, host classes' draws and P2 / pose / mirror bookkeeping (augment_resize.npz)"""
    from fsnet_amd.vision_base.data.augmentations.augmentations import PLAN
    from fsnet_amd.vision_base.utils.builder import build
    g = np.load(HA.GOLD_RESIZE)
    size = tuple(int(v) for v in g["size"])
    rngs = {k: np.random.default_rng(int(g["seed_" + k])) for k in ("bright", "contrast", "sat")}
    transform = build(**HA.resize_pipeline_cfg(g))
    np.random.seed(int(g["global_seed"]))
    state = np.random.get_state()
    mirrored = 0
    for n in range(int(g["n"])):
        frames, P2, poses = HA.resize_sample_inputs(g, n)
        np.random.set_state(state)
        plan = A.draw_resize_plan(rngs["bright"], rngs["contrast"], rngs["sat"])
        imgs, origs, mask, syx = A.run_resize_train_sample(frames, plan, size, g["mean"], g["std"])
        for j in range(3):
            assert np.array_equal(imgs[j], g["s%d_image_%d" % (n, j)]) and np.array_equal(origs[j], g["s%d_orig_%d" % (n, j)])
        assert np.array_equal(mask, g["s%d_mask" % n])
        assert bool(g["s%d_mirror" % n]) == plan["mirror"] and list(g["s%d_order" % n]) == list(plan["order"])
        # host classes from the same point of the global stream
        np.random.set_state(state)
        out = transform(HA.sample_dict(frames, P2, poses))
        state = np.random.get_state()
        hp = out[PLAN]
        assert hp["mirror"] == plan["mirror"] and [o for o, _ in hp["ops"]] == list(plan["order"])
        vals = dict(hp["ops"])
        assert vals[0] == plan["brightness"] and vals[1] == plan["contrast"] and vals[2] == plan["saturation"]
        assert np.allclose(out["P2"].numpy(), g["s%d_P2" % n], rtol=0, atol=1e-4)
        for j, i in enumerate(HA.FRAME_IDXS[1:]):
            assert np.allclose(np.asarray(out[("relative_pose", i)]), g["s%d_pose_%d" % (n, j)], atol=1e-6)
        r = hp["resize"]
        assert (r["out_h"], r["out_w"]) == size and r["gt_keys"] == ["patched_mask"]
        mirrored += plan["mirror"]
    assert 0 < mirrored < int(g["n"])


def test_resize_nearest_known_answers():
    src = np.arange(12, dtype=np.float64).reshape(3, 4)
    assert np.array_equal(A.resize_nearest(src, 4, 3), src)
    assert np.array_equal(A.resize_nearest(src, 2, 3), src[:, [0, 2]])            # floor(d * 2)
    assert np.array_equal(A.resize_nearest(src, 8, 3)[0], np.repeat(src[0], 2))   # floor(d / 2)
    assert np.array_equal(A.resize_nearest(np.ones((7, 5)), 3, 9), np.ones((9, 3)))
