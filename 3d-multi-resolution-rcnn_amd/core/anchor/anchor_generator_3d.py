"""AnchorGenerator3D (mmdet/core/anchor/anchor_generator_3d.py:6-92): base anchors on the host (a handful of
numbers), the per-level grid on the device in one kernel, in the reference's flat order
((h*W + w)*D + d)*A + a (np.meshgrid(x, y, z) with 'xy' indexing)."""
import numpy as np
import torch

from ... import _hip


class AnchorGenerator3D(object):

    def __init__(self, base_size, scales, depth_scales, ratios, anchor_depth_base, scale_major=True, ctr=None):
        self.base_size = base_size
        self.anchor_depth_base = anchor_depth_base
        self.scales = torch.Tensor(scales)
        self.anchor_depth_scales = torch.Tensor(depth_scales)
        self.ratios = torch.Tensor(ratios)
        self.scale_major = scale_major
        self.ctr = ctr
        self.base_anchors = self.gen_base_anchors()
        self._dev = {}

    @property
    def num_base_anchors(self):
        return self.base_anchors.size(0)

    def gen_base_anchors(self):
        w = h = self.base_size
        z = self.anchor_depth_base
        if self.ctr is None:
            x_ctr, y_ctr, z_ctr = 0.5 * (w - 1), 0.5 * (h - 1), 0.5 * (z - 1)
        else:
            x_ctr, y_ctr, z_ctr = self.ctr
        h_ratios = torch.sqrt(self.ratios)
        w_ratios = 1 / h_ratios
        z_ratios = h_ratios
        if self.scale_major:
            ws = (w * w_ratios[:, None] * self.scales[None, :]).view(-1)
            hs = (h * h_ratios[:, None] * self.scales[None, :]).view(-1)
            zs = (z * z_ratios[:, None] * self.anchor_depth_scales[None, :]).view(-1)
        else:
            ws = (w * self.scales[:, None] * w_ratios[None, :]).view(-1)
            hs = (h * self.scales[:, None] * h_ratios[None, :]).view(-1)
            zs = (z * self.anchor_depth_scales[:, None] * z_ratios[None, :]).view(-1)
        return torch.stack([x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1), x_ctr + 0.5 * (ws - 1), y_ctr + 0.5 * (hs - 1),
                            z_ctr - 0.5 * (zs - 1), z_ctr + 0.5 * (zs - 1)], dim=-1).round()

    def base_anchors_on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = self.base_anchors.to(device).contiguous()
        return self._dev[key]

    def grid_anchors(self, featmap_size, stride=16, depth_stride=2, device='cuda'):
        """featmap_size = (feat_z, feat_h, feat_w) -> [z*h*w*A, 6]"""
        return _hip.grid_anchors3d(self.base_anchors_on(torch.device(device)), featmap_size, stride, depth_stride)

    def valid_flags(self, featmap_size, valid_size, device='cuda'):
        feat_z, feat_h, feat_w = featmap_size
        valid_d, valid_h, valid_w = valid_size
        assert valid_h <= feat_h and valid_w <= feat_w and valid_d <= feat_z
        dev = torch.device(device)
        vx = torch.arange(feat_w, device=dev) < valid_w
        vy = torch.arange(feat_h, device=dev) < valid_h
        vz = torch.arange(feat_z, device=dev) < valid_d
        valid = (vy[:, None, None] & vx[None, :, None] & vz[None, None, :]).reshape(-1)      # (h, w, d) order
        return valid[:, None].expand(valid.size(0), self.num_base_anchors).contiguous().view(-1).to(torch.uint8)
