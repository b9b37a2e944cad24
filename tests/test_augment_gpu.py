"""Input pipeline on the GPU (fs_augment_frames through DeviceAugment): bit-exact against the golden vectors made by
the reference's own augmentation classes and, at KITTI size with ragged frame sizes, against the oracle."""
import numpy as np
import pytest
import torch

from oracle import augment_oracle as A
from tests import helpers_augment as HA

pytestmark = pytest.mark.gpu


def _pipeline(g):
    from fsnet_amd.vision_base.utils.builder import build
    return build(**HA.pipeline_cfg(g))


def test_device_pipeline_matches_reference_vectors(dev):
    from fsnet_amd.vision_base.data.augmentations.augmentations import DeviceAugment
    g = HA.golden()
    transform = _pipeline(g)
    np.random.seed(int(g["global_seed"]))
    samples = [transform(HA.sample_dict(*HA.sample_inputs(g, n))) for n in range(int(g["n"]))]
    batch = DeviceAugment(HA.FRAME_IDXS)(samples, dev)
    torch.cuda.synchronize()
    for n in range(int(g["n"])):
        for j, i in enumerate(HA.FRAME_IDXS):
            assert np.array_equal(batch[("image", i)][n].cpu().numpy(), g["s%d_image_%d" % (n, j)]), (n, i)
            assert np.array_equal(batch[("original_image", i)][n].cpu().numpy(), g["s%d_orig_%d" % (n, j)]), (n, i)
        assert np.array_equal(batch["patched_mask"][n].cpu().numpy(), g["s%d_mask" % n])
        assert np.array_equal(batch["P2"][n].cpu().numpy(), g["s%d_P2" % n])
    assert batch[("image", 0)].is_contiguous() and batch[("image", 0)].shape == (int(g["n"]), 3, int(g["out_h"]), int(g["out_w"]))
    assert batch["patched_mask"].dtype == torch.float64 and batch["P2"].is_cuda


def test_kitti_size_ragged_batch_matches_oracle(dev):
    """192x640 crops of 370..376 x 1224..1242 frames (the sizes KITTI raw mixes), zoom-out borders included"""
    from fsnet_amd.vision_base.data.augmentations.augmentations import DeviceAugment, PLAN
    g = dict(HA.golden())
    g["out_h"], g["out_w"] = np.int64(192), np.int64(640)
    transform = _pipeline(g)
    sizes = [(375, 1242), (370, 1224), (376, 1241), (374, 1238)]
    rs = np.random.RandomState(17)
    np.random.seed(5)
    samples, raw = [], []
    for h, w in sizes:
        frames = [rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8) for _ in HA.FRAME_IDXS]
        P2 = np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.0027]])
        poses = [np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)]
        samples.append(transform(HA.sample_dict(frames, P2, poses)))
        raw.append(frames)
    plans = [s[PLAN] for s in samples]
    batch = DeviceAugment(HA.FRAME_IDXS)(samples, dev)
    torch.cuda.synchronize()
    borders = 0
    for n, (frames, p) in enumerate(zip(raw, plans)):
        oplan = dict(M=p["warp"]["M"], mirror=p["mirror"], order=[o for o, _ in p["ops"]],
                     brightness=dict(p["ops"]).get(0), contrast=dict(p["ops"]).get(1), saturation=dict(p["ops"]).get(2))
        imgs, origs, mask = A.run_sample(frames, oplan, 640, 192, g["mean"], g["std"])
        for j, i in enumerate(HA.FRAME_IDXS):
            assert np.array_equal(batch[("image", i)][n].cpu().numpy(), imgs[j]), (n, i)
            assert np.array_equal(batch[("original_image", i)][n].cpu().numpy(), origs[j]), (n, i)
        assert np.array_equal(batch["patched_mask"][n].cpu().numpy(), mask)
        borders += int((mask == 0).any())
    assert borders > 0            # at least one sample was zoomed out past the frame


def test_identity_plan_returns_the_frame(dev):
    """no warp scale, no mirror, no colour op: original_image = frame / 255 exactly, image = normalised frame"""
    import ctypes as C
    from fsnet_amd.hip.binding import lib, check, stream_ptr, FsAugArgs
    rs = np.random.RandomState(1)
    B, F, H, W = 2, 2, 37, 53
    src = torch.from_numpy(rs.randint(0, 256, size=(B, F, H, W, 3)).astype(np.uint8)).to(dev)
    minv = torch.tensor([[1, 0, 0, 0, 1, 0]] * B, dtype=torch.float64, device=dev)
    iplan = torch.tensor([[3, 3, 3, 0, 0, H, W, 0]] * B, dtype=torch.int32, device=dev)
    fplan = torch.zeros(B, 4, device=dev)
    image = torch.empty(F, B, 3, H, W, device=dev)
    orig = torch.empty(F, B, 3, H, W, device=dev)
    mask = torch.empty(B, H, W, dtype=torch.float64, device=dev)
    a = FsAugArgs()
    a.src, a.minv, a.iplan, a.fplan = src.data_ptr(), minv.data_ptr(), iplan.data_ptr(), fplan.data_ptr()
    a.image, a.original, a.mask = image.data_ptr(), orig.data_ptr(), mask.data_ptr()
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    for k in range(3):
        a.mean[k], a.std[k] = mean[k], std[k]
    a.B, a.F, a.Hs, a.Ws, a.H, a.W = B, F, H, W, H, W
    check(lib.fs_augment_frames(C.byref(a), stream_ptr()), "augment")
    torch.cuda.synchronize()
    want = src.permute(1, 0, 4, 2, 3).cpu().numpy().astype(np.float32) / np.float32(255.0)   # true division
    assert np.array_equal(orig.cpu().numpy(), want)
    m = np.array(mean, dtype=np.float32).reshape(1, 1, 3, 1, 1)
    s = np.array(std, dtype=np.float32).reshape(1, 1, 3, 1, 1)
    assert np.array_equal(image.cpu().numpy(), (want - m) / s)
    assert bool((mask == 1).all())
    a.std[1] = 0.0
    assert lib.fs_augment_frames(C.byref(a), stream_ptr()) == 1


def test_pipeline_feeds_a_training_step(dev):
    """DataLoader(collate_fn=DeviceAugment.collate) -> materialize -> BaseTrainingHook: the batch dict the device
    pipeline produces is the one the reference's collate + .cuda() hands to the hook (keys, dtypes, layouts)."""
    from torch.utils.data import DataLoader, Dataset
    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.data.augmentations.augmentations import DeviceAugment
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    H, W = 64, 96
    g = dict(HA.golden())
    g["out_h"], g["out_w"] = np.int64(H), np.int64(W)

    class Frames(Dataset):                      # stands in for KittiDepthMonoDataset: decoded uint8 frames + calib
        def __init__(self):
            self.transform = _pipeline(g)

        def __len__(self):
            return 8

        def __getitem__(self, n):
            rs = np.random.RandomState(n)
            base = rs.randint(0, 256, size=(300 // 20, 420 // 20, 3)).astype(np.uint8)
            frame = np.kron(base, np.ones((20, 20, 1), dtype=np.uint8))              # blocky texture
            frames = [np.roll(frame, 3 * i, axis=1) for i in HA.FRAME_IDXS]
            P2 = np.array([[240.0, 0, 210.0, 10.0], [0, 240.0, 150.0, 0.1], [0, 0, 1, 0.003]])
            poses = []
            for i in HA.FRAME_IDXS[1:]:
                T = np.eye(4, dtype=np.float32)
                T[0, 3], T[2, 3] = 0.01, (-0.8 if i > 0 else 0.8)
                poses.append(T)
            return self.transform(HA.sample_dict(frames, P2, poses))

    aug = DeviceAugment(HA.FRAME_IDXS)
    loader = DataLoader(Frames(), batch_size=4, collate_fn=aug.collate, num_workers=0)
    RT.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    m = build(**meta_arch_cfg(H, W, with_pose=False)).to(dev).train()
    tc = training_cfg(clip_gradients=35.0, lr=1e-4)
    opt = build_optimizer(m, **tc.optimizer)
    hook = build(use_graph=False, **tc.training_hook)
    losses = []
    for batch in loader:
        batch = aug.materialize(batch, dev)
        assert batch[("image", 0)].shape == (4, 3, H, W) and batch[("image", 0)].dtype == torch.float32
        assert batch["patched_mask"].dtype == torch.float64 and batch["P2"].shape == (4, 3, 4)
        assert 0.0 <= float(batch[("original_image", 1)].min()) and float(batch[("original_image", 1)].max()) <= 1.0
        out = hook(batch, m, opt)
        losses.append(float(out["loss"].detach()))
    torch.cuda.synchronize()
    assert len(losses) == 2 and all(l == l and 0 < l < 10 for l in losses), losses
    RT.set_compute_dtype(torch.bfloat16)


def test_validation_resize_matches_reference_vectors(dev):
    """Resize + Normalize of the validation path (stretch, both pads, crop) through fs_resize_frames"""
    from fsnet_amd.vision_base.data.augmentations.augmentations import DeviceAugment
    g = HA.golden()
    aug = DeviceAugment([0], original_family=None, mask_key=None)
    for tag, kw in HA.VAL_CASES:
        frame = HA.val_frame(g, tag)
        vt = HA.val_pipeline(g, kw)
        samples = [vt({('image', 0): frame.copy(), 'P2': HA.VAL_P2.copy()}),
                   vt({('image', 0): frame[:-3, :-5].copy(), 'P2': HA.VAL_P2.copy()})]      # ragged second sample
        if kw.get("preserve_aspect_ratio"):
            samples = samples[:1]       # a different aspect ratio changes the effective size, not a batch-mate
        batch = aug(samples, dev)
        torch.cuda.synchronize()
        assert np.array_equal(batch[('image', 0)][0].cpu().numpy(), g["val_%s_image" % tag]), tag
        assert np.array_equal(batch['P2'][0].cpu().numpy(), g["val_%s_P2" % tag])
        if len(samples) == 2:
            want = A.run_val_sample(frame[:-3, :-5], (48, 160), g["mean"], g["std"], **kw)
            assert np.array_equal(batch[('image', 0)][1].cpu().numpy(), want)


def test_kitti_dataset_through_device_pipeline_into_a_training_step(dev, tmp_path):
    """KittiDepthMonoDataset (PNG decode, calibration, poses) -> DataLoader with DeviceAugment.collate ->
    fs_augment_frames -> one optimisation step: the whole data side of configs/kitti_wpose_example on the mirror"""
    from torch.utils.data import DataLoader
    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.monodepth.data.datasets.mono_dataset import KittiDepthMonoDataset
    from fsnet_amd.vision_base.data.augmentations.augmentations import DeviceAugment
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    from tests import helpers_kitti as HK
    H, W = 64, 128
    g = dict(HA.golden())
    g["out_h"], g["out_w"] = np.int64(H), np.int64(W)
    raw, split = HK.make_tree(str(tmp_path), seed=5, H=300, W=420)      # room for the warp's crop centre
    cfg = HK.dataset_cfg(raw, split, prefix='fsnet_amd.')
    cfg["augmentation"] = HA.pipeline_cfg(g)
    ds = KittiDepthMonoDataset(**cfg)
    aug = DeviceAugment(HA.FRAME_IDXS)
    np.random.seed(3)
    loader = DataLoader(ds, batch_size=3, collate_fn=aug.collate, num_workers=0)
    RT.set_compute_dtype(torch.float32)
    torch.manual_seed(0)
    m = build(**meta_arch_cfg(H, W, with_pose=False)).to(dev).train()
    tc = training_cfg()
    opt = build_optimizer(m, **tc.optimizer)
    hook = build(use_graph=False, **tc.training_hook)
    n = 0
    for batch in loader:
        batch = aug.materialize(batch, dev)
        assert batch[("image", 0)].shape == (3, 3, H, W) and batch[("relative_pose", 1)].shape == (3, 4, 4)
        assert batch["P2"].shape == (3, 3, 4) and batch["patched_mask"].dtype == torch.float64
        out = hook(batch, m, opt)
        loss = float(out["loss"].detach())
        assert loss == loss and 0 < loss < 10
        n += 1
    assert n == 1
    RT.set_compute_dtype(torch.bfloat16)


def test_resize_training_chain_matches_reference_vectors(dev):
    """Resize -> colour Shuffle -> RandomMirror -> Normalize x2 (configs/multi_dataset_example:178-205) on the device:
    one fs_resize_frames launch per batch, bit-exact against the vectors of the reference's own classes; the four
    samples have different frame sizes (ragged batch) and pad on different sides"""
    from fsnet_amd.vision_base.data.augmentations.augmentations import DeviceAugment
    from fsnet_amd.vision_base.utils.builder import build
    g = np.load(HA.GOLD_RESIZE)
    transform = build(**HA.resize_pipeline_cfg(g))
    np.random.seed(int(g["global_seed"]))
    N = int(g["n"])
    samples = [transform(HA.sample_dict(*HA.resize_sample_inputs(g, n))) for n in range(N)]
    batch = DeviceAugment(HA.FRAME_IDXS)(samples, dev)
    torch.cuda.synchronize()
    for n in range(N):
        for j, i in enumerate(HA.FRAME_IDXS):
            assert np.array_equal(batch[("image", i)][n].cpu().numpy(), g["s%d_image_%d" % (n, j)]), (n, i)
            assert np.array_equal(batch[("original_image", i)][n].cpu().numpy(), g["s%d_orig_%d" % (n, j)]), (n, i)
        assert np.array_equal(batch["patched_mask"][n].cpu().numpy(), g["s%d_mask" % n])
        assert np.allclose(batch["P2"][n].cpu().numpy(), g["s%d_P2" % n], rtol=0, atol=1e-4)
    H, W = (int(v) for v in g["size"])
    assert batch[("image", 0)].shape == (N, 3, H, W) and batch["patched_mask"].dtype == torch.float64
    assert float(batch["patched_mask"].min()) == 0.0            # some sample was padded
