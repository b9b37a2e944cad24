"""Mei unified fisheye camera model with the reference's class name and cache key
(monodepth/networks/utils/mei_fisheye_utils.py:122-187).  The reference fills the per-calibration table
(X, Y, Z, mask) with numba-JIT CPU loops and re-uploads it every step; here fs_mei_lut builds it on the device once
per calibration (same Newton / bisection iterations in f64) and the loss kernels read it in place through a
per-sample pointer table."""
import numpy as np
import torch

from fsnet_amd.hip import lib, check, stream_ptr


def calib_key(H, W, P_row, calib):
    """the reference's cache key (:147): (H, W, gamma1, gamma2, u0, v0, k1, k2, xi); P_row = 12 floats of one P2"""
    return (int(H), int(W), float(P_row[0]), float(P_row[5]), float(P_row[2]), float(P_row[6]),
            float(calib["distortion_parameters"]["k1"]), float(calib["distortion_parameters"]["k2"]),
            float(calib["mirror_parameters"]["xi"]))


class MeiCameraProjection(object):
    def __init__(self):
        self.cache = {}          # (key, device) -> ray table [4, H, W] fp32 on the device

    def table(self, key, device):
        t = self.cache.get((key, device))
        if t is None:
            H, W, g1, g2, u0, v0, k1, k2, xi = key
            t = torch.empty(4, H, W, dtype=torch.float32, device=device)
            check(lib.fs_mei_lut(t.data_ptr(), H, W, g1, g2, u0, v0, k1, k2, xi, stream_ptr()), "mei_lut")
            self.cache[(key, device)] = t
        return t

    def tables(self, H, W, P, calib, device):
        """per-sample (table, parameter row) for a batch: P [B,3,4] (any device; read once on the host), calib = list
        of the dataset's calibration dicts"""
        Ph = P.detach().to("cpu", torch.float32).reshape(P.shape[0], -1).numpy()
        tabs, rows = [], np.zeros((P.shape[0], 8), dtype=np.float32)
        for b in range(P.shape[0]):
            key = calib_key(H, W, Ph[b], calib[b])
            tabs.append(self.table(key, device))
            rows[b, :7] = (key[6], key[7], key[8], key[2], key[3], key[4], key[5])
        return tabs, rows

    def image2cam(self, norm, P, calib):
        """points [B,1,H,W,3] = ray table x norm and the table mask [B,1,H,W] (:131-187)"""
        B, _, H, W = norm.shape
        tabs, _ = self.tables(H, W, P, calib, norm.device)
        ptrs = torch.tensor([t.data_ptr() for t in tabs], dtype=torch.int64).to(norm.device)
        pts = torch.empty(B, 1, H, W, 3, dtype=torch.float32, device=norm.device)
        check(lib.fs_mei_points(ptrs.data_ptr(), norm.contiguous().float().data_ptr(), pts.data_ptr(), B, H, W,
                                stream_ptr()), "mei_points")
        mask = torch.stack([t[3] for t in tabs], 0).unsqueeze(1)
        return pts, mask
