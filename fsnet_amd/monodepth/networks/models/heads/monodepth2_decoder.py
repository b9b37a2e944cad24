"""MonoDepth2Decoder with the reference's constructor and method names
(monodepth/networks/models/heads/monodepth2_decoder.py:19-347).  loss() runs the fused HIP
photometric chain (backproject -> project -> grid_sample -> SSIM+L1 -> per-pixel min -> masked mean,
plus edge-aware smoothness), forward and backward, for the option set the shipped configs use;
every other option raises instead of silently taking a different path."""
import torch
import torch.nn as nn

from fsnet_amd.engine.runtime import RT, require_gpu
from fsnet_amd.hip import ops
from fsnet_amd.vision_base.utils.builder import build

_UNSUPPORTED_FLAGS = ("is_residual_flow", "is_light_compensate", "learnable_photometric_uncertain", "is_ssim_weight")


class _PhotoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pl, S, img0, src_a, src_b, P2, patched_mask, motion_mask, T_a, T_b, *dd):
        ctx.set_materialize_grads(False)
        depths = [d.contiguous().float() for d in dd[:S]]
        disps = [d.contiguous().float() for d in dd[S:]]
        seed = None if RT.tie_noise else -1     # None: device-resident seed, bumped in-stream every step
        RT.mark("photo.fwd.start")
        out = pl.forward(img0.contiguous().float(), [src_a.contiguous().float(), src_b.contiguous().float()],
                         P2.contiguous().float(), [T_a.contiguous().float(), T_b.contiguous().float()], patched_mask,
                         depths, disps, noise_seed=seed,
                         motion_mask=None if motion_mask is None else motion_mask.contiguous().float())
        ctx.pl, ctx.S = pl, S
        vec = out                       # fresh tensors of this call (ops.PhotometricLoss.forward): no copies
        ctx.mark_non_differentiable(vec)
        return pl.total, vec

    @staticmethod
    def backward(ctx, g_total, _g_vec):
        pl, S = ctx.pl, ctx.S
        gout = None
        if g_total is not None:
            gout = g_total.detach().double().contiguous()
        RT.mark("photo.bwd.start")
        d_depth, d_disp, dT = pl.backward(gout)
        RT.mark("photo.bwd.end")
        return (None, None, None, None, None, None, None, None, dT[0], dT[1]) + tuple(d_depth) + tuple(d_disp)


class _DistillFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, teacher, unc):
        pred, teacher = pred.contiguous().float(), teacher.contiguous().float()
        unc = None if unc is None else unc.contiguous().float()
        ctx.save_for_backward(pred, teacher, unc)
        return ops.distill_fwd(pred, teacher, unc)

    @staticmethod
    def backward(ctx, g):
        pred, teacher, unc = ctx.saved_tensors
        d_pred, d_unc = ops.distill_bwd(pred, teacher, unc, g.detach().double().contiguous())
        return d_pred, None, d_unc


class MonoDepth2Decoder(nn.Module):
    def __init__(self, scales, height, width, frame_ids, depth_decoder_cfg, pose_decoder_cfg=None,
                 multiscale_head_cfg=None, **kwargs):
        super().__init__()
        self.scales = list(scales)
        self.num_scales = len(self.scales)
        self.height, self.width = height, width
        self.frame_ids = list(frame_ids)
        self.depth_decoder = build(**depth_decoder_cfg)
        if pose_decoder_cfg is not None:
            self.pose_decoder = build(**pose_decoder_cfg)
        if multiscale_head_cfg is not None:
            raise NotImplementedError("multiscale_head_cfg (residual flow head) is outside the monodepth hot path")
        for key in kwargs:           # min_depth, max_depth, overlapped_mask, is_log_image, ... (reference :48-49)
            setattr(self, key, kwargs[key])
        self._pl = None

    # ---- network heads -------------------------------------------------------------------
    def forward_pose(self, *args, **kwargs):
        return self.pose_decoder(*args, **kwargs)

    def forward_pose_transform(self, features, invert):
        """fused (axisangle, translation, cam_T_cam): pose_decoder.py:26-45 + monodepth2_model.py:42-43."""
        return self.pose_decoder.forward_with_transform(features, invert)

    def forward_pose_pairs(self, features, inverts):
        """all image pairs of the step in one pass over the stacked pose-encoder feature"""
        return self.pose_decoder.forward_pairs_with_transform(features, inverts)

    def forward_depth(self, features, *args, **kwargs):
        return self.depth_decoder(features, *args, **kwargs)

    def get_prediction(self, input_dict, output_dict):
        return dict(depth=output_dict[("depth", 0, 0)])

    # ---- loss ---------------------------------------------------------------------------------
    def _check_options(self, input_dict):
        for flag in _UNSUPPORTED_FLAGS:
            if getattr(self, flag, False):
                raise NotImplementedError("MonoDepth2Decoder option %s is not implemented in the HIP loss chain" % flag)
        if getattr(self, "residualflow_weight", 0) > 0:
            raise NotImplementedError("MonoDepth2Decoder term residualflow_weight > 0 is not implemented in the HIP loss chain")
        if getattr(self, "distillation_loss_weight", 0) > 0 and getattr(self, "is_unscaled_distill", False):
            raise NotImplementedError("is_unscaled_distill=True is not implemented (no shipped config enables it)")
        if len(self.frame_ids) != 3 or "s" in self.frame_ids:
            raise NotImplementedError("the HIP loss chain handles frame_ids=[0, a, b] (two temporal source frames)")

    def _loss_engine(self, img0):
        B, _, H, W = img0.shape
        return self._loss_engine_for(B, H, W, img0.device)

    def _loss_engine_for(self, B, H, W, device):
        key = (B, H, W, tuple(self.scales), device)
        if self._pl is None or self._pl_key != key:
            # the warped images only reach HBM when somebody looks at them (logging; output_dict entries below)
            self._pl = ops.PhotometricLoss(B, H, W, self.scales, device, self.min_depth, self.max_depth,
                                           want_pred=bool(getattr(self, "is_log_image", True)
                                                          or getattr(self, "keep_warped_images", False)),
                                           overlapped_mask=bool(getattr(self, "overlapped_mask", False)))
            self._pl_key = key
        return self._pl

    def _stage_geometry(self, input_dict):
        """camera-model inputs of the loss engine beyond P2 (none for the pinhole model)"""
        self._pl.fisheye = False

    def stage_step_inputs(self, input_dict):
        """host-side per-step staging the training hook runs before replaying a captured step (non-tensor inputs)"""
        return

    @staticmethod
    def _mask64(input_dict):
        pm = input_dict.get("patched_mask", None)
        if pm is not None and pm.dtype != torch.float64:
            pm = pm.double()
        return pm.contiguous() if pm is not None else None

    def prefetch_loss_inputs(self, input_dict):
        """launch the input-only part of the loss chain (identity reprojection, colour pyramid) on the current
        stream — the meta-arch calls this on the pose stream, off the depth chain's critical path"""
        if len(self.frame_ids) != 3 or "s" in self.frame_ids:
            return
        img0 = input_dict[("original_image", 0)]
        if not (img0.is_cuda and img0.dtype == torch.float32 and img0.is_contiguous()):
            return
        fa, fb = self.frame_ids[1], self.frame_ids[2]
        srcs = [input_dict[("original_image", fa)], input_dict[("original_image", fb)]]
        if not all(s.dtype == torch.float32 and s.is_contiguous() for s in srcs):
            return
        pm = input_dict.get("patched_mask", None)
        if pm is not None and not (pm.dtype == torch.float64 and pm.is_contiguous()):
            return                    # a converted copy would not be the tensor loss() sees
        self._loss_engine(img0).prefetch(img0, srcs, pm)

    def compute_total_reprojection_loss(self, output_dict, input_dict):
        self._check_options(input_dict)
        img0 = input_dict[("original_image", 0)]
        require_gpu(img0, "MonoDepth2Decoder.loss")
        B, _, H, W = img0.shape
        S = self.num_scales
        for s in self.scales:
            d = output_dict[("depth", s, s)]
            if d.shape[2] != (H >> s) or d.shape[3] != (W >> s):
                raise NotImplementedError("depth at scale %d must be %dx%d" % (s, H >> s, W >> s))
        self._loss_engine(img0)
        self._stage_geometry(input_dict)
        fa, fb = self.frame_ids[1], self.frame_ids[2]
        pm = self._mask64(input_dict)
        depths = [output_dict[("depth", s, s)] for s in self.scales]
        disps = [output_dict[("disp", s)] for s in self.scales]
        # (what the loss differentiates: the training hook's per-chain capture cuts the backward here)
        self._loss_inputs = ([output_dict[("cam_T_cam", fa)], output_dict[("cam_T_cam", fb)]], depths + disps)
        total, vec = _PhotoLossFn.apply(self._pl, S, img0, input_dict[("original_image", fa)],
                                        input_dict[("original_image", fb)], input_dict["P2"], pm,
                                        input_dict.get("motion_mask", None),
                                        output_dict[("cam_T_cam", fa)], output_dict[("cam_T_cam", fb)],
                                        *depths, *disps)
        losses = {}
        for k, s in enumerate(self.scales):
            losses["loss/%d" % s] = vec[k]
            losses["smooth_loss/%d" % s] = vec[S + k]
        # warped images / masks the reference leaves in output_dict (_generate_images_pred :98-116).  The fused loss
        # kernels keep them in registers: they are written out when is_log_image (the reference default) or
        # keep_warped_images=True asks for them
        if self._pl.pred is not None:
            for k, s in enumerate(self.scales):
                for j, f in enumerate((fa, fb)):
                    output_dict[("original_image", f, s)] = self._pl.pred[k, j]
                    if self._pl.overlapped_mask:      # (the reference only produces it when the option is on)
                        output_dict[("overlapped_mask", f, s)] = self._pl.ov[k, j].view(torch.bool)   # 0/1 bytes
        hm = {}
        if getattr(self, "is_log_image", True) and self._pl.pred is not None:
            hm["original_image"] = img0[0:1]
            for j, f in enumerate((fa, fb)):
                hm["predicted_image_%s" % f] = self._pl.pred[0, j, 0:1]
            hm["loss_mask_%d" % self.scales[0]] = dict(data=(self._pl.sel[0, 0:1] >= 2).unsqueeze(1))
        return losses, hm, total

    def compute_pose_loss(self, output_dict, input_dict):
        """sum over the source frames of mean |relative_pose - cam_T_cam| (reference :176-183): the mean-absolute-difference
        kernel of the distillation term on the 4x4 matrices; its gradient reaches the pose networks through cam_T_cam"""
        pose_loss = 0
        for f in self.frame_ids[1:]:
            target = input_dict[('relative_pose', f)].detach()
            pose_loss = pose_loss + _DistillFn.apply(output_dict[("cam_T_cam", f)], target, None)
        return pose_loss

    def compute_distill_loss(self, output_dict, input_dict, scale):
        """monodepth2_decoder.py:185-203: mean |teacher - pred| (/ uncertain_z + log(uncertain_z + 1e-5))"""
        pred = output_dict[('depth', scale, scale)]
        teacher = output_dict[('teacher_depth', scale, scale)].detach()
        unc = output_dict[('uncertain_z', scale)] if getattr(self, 'is_uncertain_distill', False) else None
        return _DistillFn.apply(pred, teacher, unc)

    def loss(self, output_dict, input_dict):
        losses, hm, total = self.compute_total_reprojection_loss(output_dict, input_dict)
        pose_weight = getattr(self, 'pose_loss_weight', 0)
        if pose_weight > 0:                                           # reference :176-183, 322-326
            pose_loss = self.compute_pose_loss(output_dict, input_dict)
            losses['pose_loss'] = pose_loss.detach()
            total = total + pose_weight * pose_loss
        distillation_weight = getattr(self, 'distillation_loss_weight', 0)
        if distillation_weight > 0:                                   # reference :328-334
            for scale in self.scales:
                dl = self.compute_distill_loss(output_dict, input_dict, scale)
                losses["distilation/{}".format(scale)] = dl.detach()
                total = total + dl * distillation_weight
        losses["total_loss"] = total.detach()
        if not getattr(self, "is_log_image", True):
            hm = {}
        return {"loss": total, "loss_dict": losses, "hm": hm}


class FishEyeDecoder(MonoDepth2Decoder):
    """MonoDepth2Decoder for the Mei unified fisheye model (monodepth2_decoder.py:350-420): the network output is the
    ray norm, the 3-D point is the per-calibration ray table x norm, the relative pose acts on it directly and
    cam2image (mirror + radial distortion) maps it into the source frame; the overlap mask samples
    patched_mask x table mask.  Same fused HIP loss chain, ray-table variant (FsPhotoArgs.lut_ptrs)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        from fsnet_amd.monodepth.networks.utils.mei_fisheye_utils import MeiCameraProjection
        self.mei_projection = MeiCameraProjection()
        self._staged = None          # (P2 tensor, its version, calib list) of the last staging: strong references,
        self._hook_staged = False    # so an equal identity really is the same batch object

    def stage_step_inputs(self, input_dict, from_hook=True):
        """ray tables of this batch's calibrations (built once each, cached by the reference's key) -> the loss
        engine's persistent pointer table.  Host side; the training hook runs it before every step (before the H2D
        move, so P2 is normally still a host tensor) and before every hipGraph replay."""
        P, calib = input_dict["P2"], input_dict["calib_meta"]
        img0 = input_dict[("original_image", 0)]
        B, _, H, W = img0.shape
        dev = img0.device if img0.is_cuda else next(self.parameters()).device
        pl = self._loss_engine_for(B, H, W, dev)
        self._hook_staged = from_hook
        st = self._staged
        if st is not None and st[0] is P and st[1] == P._version and st[2] is calib and st[3] is pl and pl.fisheye:
            return
        tabs, rows = self.mei_projection.tables(H, W, P, calib, dev)
        pl.stage_fisheye(tabs, rows)
        self._staged = (P, P._version, calib, pl)

    def _stage_geometry(self, input_dict):
        if "calib_meta" not in input_dict:
            raise KeyError("FishEyeDecoder.loss needs input_dict['calib_meta'] (list of Mei calibration dicts)")
        if torch.cuda.is_current_stream_capturing():
            if not self._pl.fisheye:
                raise RuntimeError("FishEyeDecoder: stage_step_inputs() must run before the step is captured")
        elif self._hook_staged and self._pl.fisheye:
            pass                      # the hook staged this step's calibrations already
        else:
            self.stage_step_inputs(input_dict, from_hook=False)
        self._hook_staged = False

    def prefetch_loss_inputs(self, input_dict):
        return        # MonoDepthWPose has no side stream; the identity terms run inside loss()

    def get_prediction(self, input_dict, output_dict):
        norm = output_dict[("depth", 0, 0)]
        points, _ = self.mei_projection.image2cam(norm, input_dict["P2"], input_dict["calib_meta"])
        return dict(depth=points[..., 2], norm=norm)
