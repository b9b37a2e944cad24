"""KITTI ground-truth export from raw velodyne scans (generate_depth_map + KittiEigenEvaluator._precompute mirrors)
against the reference's depth maps over the same seeded on-disk tree (tests/golden/velo_gt.npz)."""
import os

import numpy as np

from tests import helpers_kitti as HK

GOLD = os.path.join(os.path.dirname(__file__), "golden", "velo_gt.npz")


def test_generate_depth_map_matches_reference_bit_for_bit(tmp_path):
    from fsnet_amd.monodepth.networks.utils.monodepth_utils import generate_depth_map
    raw, split = HK.make_tree(str(tmp_path))
    HK.add_velodyne(raw)
    gt = np.load(GOLD)["gt"]
    lines = [l.split() for l in open(split)]
    assert len(lines) == gt.shape[0]
    dup = 0
    for k, (folder, frame_id, _) in enumerate(lines):
        got = generate_depth_map(os.path.join(raw, folder.split("/")[0]),
                                 os.path.join(raw, folder, "velodyne_points/data", "%010d.bin" % int(frame_id)), 2, True)
        assert got.dtype == np.float64 and np.array_equal(got, gt[k]), k
        dup += int((got > 0).sum())
    assert 0.5 < dup / gt.size < 0.95            # dense enough that duplicate handling (and its quirk) mattered


def test_evaluator_exports_and_caches_ground_truth(tmp_path):
    from fsnet_amd.monodepth.evaluation.kitti_unsupervised_eval import KittiEigenEvaluator
    raw, split = HK.make_tree(str(tmp_path))
    HK.add_velodyne(raw)
    cache = str(tmp_path / "gt_depths.npz")
    ev = KittiEigenEvaluator(data_path=raw, split_file=split, gt_saved_file=cache)
    gt = np.load(GOLD)["gt"].astype(np.float32)
    assert len(ev.gt_depths) == gt.shape[0] and all(np.array_equal(a, b) for a, b in zip(ev.gt_depths, gt))
    again = KittiEigenEvaluator(data_path="/nonexistent", split_file="/nonexistent", gt_saved_file=cache)   # from the cache
    assert np.array_equal(np.asarray(again.gt_depths), gt)


def test_export_with_ragged_image_sizes_and_without_cache_file(tmp_path, monkeypatch):
    """recording dates of the Eigen split have different rectified sizes: the cache must hold a ragged list; and
    gt_saved_file=None keeps the export in memory instead of crashing at the save"""
    from fsnet_amd.monodepth.evaluation.kitti_unsupervised_eval import KittiEigenEvaluator
    from fsnet_amd.monodepth.networks.utils import monodepth_utils as MU
    raw, split = HK.make_tree(str(tmp_path))
    sizes = [(7, 11), (6, 12), (7, 11), (8, 10)]
    calls = []

    def fake(calib_dir, scan, cam, vel_depth):
        h, w = sizes[len(calls) % len(sizes)]
        calls.append(scan)
        return np.full((h, w), float(len(calls)))
    monkeypatch.setattr(MU, "generate_depth_map", fake)
    cache = str(tmp_path / "ragged.npz")
    ev = KittiEigenEvaluator(data_path=raw, split_file=split, gt_saved_file=cache)
    n = len(ev.gt_depths)
    assert n == len(calls) and n >= 2
    again = KittiEigenEvaluator(gt_saved_file=cache)
    assert len(again.gt_depths) == n
    for a, b in zip(ev.gt_depths, again.gt_depths):
        assert a.shape == b.shape and np.array_equal(a, b)
    calls.clear()
    mem = KittiEigenEvaluator(data_path=raw, split_file=split, gt_saved_file=None)
    assert len(mem.gt_depths) == n
    try:
        KittiEigenEvaluator(gt_saved_file=None)
    except ValueError:
        pass
    else:
        raise AssertionError("expected a ValueError without a cache file and without a data path")
