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

    def _extents(self, base, ratio_factor, scales):
        """flattened [ratio x scale] (scale_major) or [scale x ratio] table of box extents along one axis; the product is
        taken in the reference's order ((base * first) * second) so that the rounded anchors are identical"""
        first, second = (ratio_factor, scales) if self.scale_major else (scales, ratio_factor)
        return ((base * first).unsqueeze(1) * second.unsqueeze(0)).reshape(-1)

    def gen_base_anchors(self):
        """anchor_generator_3d.py:22-53: one box per (ratio, scale) pair centred on the stride cell; xy use `scales`, z
        uses `depth_scales`; h and z grow with sqrt(ratio), w shrinks with it; corners = centre -/+ (extent - 1) / 2,
        rounded.  Columns: x1, y1, x2, y2, z1, z2."""
        sizes = (self.base_size, self.base_size, self.anchor_depth_base)               # w, h, z of the cell
        centres = self.ctr if self.ctr is not None else tuple(0.5 * (v - 1) for v in sizes)
        root = torch.sqrt(self.ratios)
        ext = (self._extents(sizes[0], 1 / root, self.scales), self._extents(sizes[1], root, self.scales),
               self._extents(sizes[2], root, self.anchor_depth_scales))
        lo = [c - 0.5 * (e - 1) for c, e in zip(centres, ext)]
        hi = [c + 0.5 * (e - 1) for c, e in zip(centres, ext)]
        return torch.stack([lo[0], lo[1], hi[0], hi[1], lo[2], hi[2]], dim=-1).round()

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
