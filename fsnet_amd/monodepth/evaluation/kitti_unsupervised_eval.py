"""KittiEigenEvaluator with the reference's constructor and methods (monodepth/evaluation/
kitti_unsupervised_eval.py:11-127).  `_single_loss` runs on the device: resize to the ground truth's size, valid mask
+ Garg crop, median scaling, clamp and the seven depth errors are one HIP launch per image (fs_depth_eval), so a
validation pass does not copy depth maps to the host.  Ground-truth export from raw velodyne scans (`_precompute`,
:27-45) is the data layer's job and is not part of this package: the evaluator loads the reference's `gt_saved_file`."""
import os

import numpy as np
import torch

from fsnet_amd.hip import ops


class KittiEigenEvaluator(object):
    def __init__(self, data_path=None, split_file=None, gt_saved_file=None, is_evaluate_absolute=False, gt_depths=None,
                 device=None):
        self.is_evaluate_absolute = is_evaluate_absolute
        self.device = device
        if gt_depths is not None:
            self.gt_depths = gt_depths
        elif gt_saved_file is not None and os.path.isfile(gt_saved_file):
            self.gt_depths = np.load(gt_saved_file, fix_imports=True, encoding='latin1', allow_pickle=True)["data"]
        else:
            if data_path is None or split_file is None:
                raise ValueError("KittiEigenEvaluator: no cached ground truth (gt_saved_file=%r) and no data_path / "
                                 "split_file to export it from" % (gt_saved_file,))
            print("Start exporting ground truth depths specified by %s to %s" % (split_file, gt_saved_file))
            self._precompute(data_path, split_file, gt_saved_file)
        self._gt_dev = {}

    def _precompute(self, data_path, split_file, gt_saved_file):
        """ground truth of a split from the raw velodyne scans, cached as <gt_saved_file> (reference :27-46)"""
        from fsnet_amd.monodepth.networks.utils.monodepth_utils import generate_depth_map
        gts = []
        with open(split_file, "r") as f:
            for line in f:
                if not line.strip():
                    continue
                folder, frame_id, _ = line.split()
                scan = os.path.join(data_path, folder, "velodyne_points/data", "{:010d}.bin".format(int(frame_id)))
                gts.append(generate_depth_map(os.path.join(data_path, folder.split("/")[0]), scan, 2, True).astype(np.float32))
        # the Eigen split mixes recording dates whose rectified image sizes differ (375x1242, 370x1224, 376x1241 ...):
        # a ragged list only becomes an array as dtype=object (the reference's np.array(gts) relied on an older NumPy
        # doing that implicitly; NumPy >= 1.24 raises).  The loader above passes allow_pickle=True.
        if gt_saved_file is not None:
            if len({g.shape for g in gts}) <= 1:
                arr = np.array(gts)
            else:
                arr = np.empty(len(gts), dtype=object)
                for k, g in enumerate(gts):
                    arr[k] = g
            np.savez_compressed(gt_saved_file, data=arr)
        self.gt_depths = gts

    def _gt(self, index, device):
        g = self._gt_dev.get(index)
        if g is None or g.device != device:
            g = torch.as_tensor(np.asarray(self.gt_depths[index], dtype=np.float32)).to(device)
            if len(self._gt_dev) < 4096:
                self._gt_dev[index] = g
        return g

    def _single_loss(self, depth_0, gt_depth):
        """depth_0: predicted depth [h, w] (device tensor, or numpy as in the reference); gt_depth: [H, W]."""
        dev = self.device
        if isinstance(depth_0, torch.Tensor) and depth_0.is_cuda:
            dev = depth_0.device
        if dev is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        pred = torch.as_tensor(depth_0, dtype=torch.float32).to(dev)
        gt = torch.as_tensor(gt_depth, dtype=torch.float32).to(dev)
        out = ops.depth_eval(pred[None], gt[None])[0].cpu().numpy()
        if out[15] == 0:
            raise ValueError
        return dict(ratio=np.float32(out[0]), error=tuple(out[1:8]), abs_error=tuple(out[8:15]))

    def single_call(self, depth_0, index):
        dev = depth_0.device if isinstance(depth_0, torch.Tensor) and depth_0.is_cuda else (
            self.device or torch.device("cuda", torch.cuda.current_device()))
        return self._single_loss(depth_0, self._gt(index, dev))

    def log(self, writer, mean_errors, mean_abs_errors, global_step=0, epoch_num=0, is_print=True):
        log_str = f"Epoch {epoch_num}"
        log_str += "\n  " + ("{:>8} | " * 7).format("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")
        log_str += "\n" + ("&{: 8.3f}  " * 7).format(*np.asarray(mean_errors).tolist()) + "\\\\"
        log_str += f"\nEpoch {epoch_num}| Abs Error without Scaled"
        log_str += "\n  " + ("{:>8} | " * 7).format("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")
        log_str += "\n" + ("&{: 8.3f}  " * 7).format(*np.asarray(mean_abs_errors).tolist()) + "\\\\"
        if writer is not None:
            writer.add_text("evaluation logs", log_str.replace(' ', '&nbsp;').replace('\n', '  \n'), global_step=epoch_num)
        if is_print:
            print(log_str)
        return log_str
