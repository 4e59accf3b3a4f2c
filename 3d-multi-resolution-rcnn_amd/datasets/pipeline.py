"""Training-sample preparation of the shipped dataset on the MI355X.

Mirrors `Coco3D2ScalesDataset.prepare_train_img` (mmdet/datasets/coco_3d_2scales.py:169-326) and its `RandomCrop3D`
augmentation (mmdet/datasets/extra_aug.py:146-211):
    .npy grey volume [H,W,D] -> RandomCrop3D (patch = H/4 x W/4 x full depth that keeps >= 1 gt box strictly inside and
    meets the drawn min-IoU) -> pathway 2 = the SAME patch resampled by the upscale factor (1.5) -> per-slice grey->RGB,
    normalise, pad to a multiple of 32 -> [3,D,H,W]; gt boxes shifted (x upscale for pathway 2); instance masks cropped,
    padded and transposed to [G,D,H,W] uint8.
What differs from the reference, by design:
  * the host only DECIDES (a few boxes, a few random draws); every voxel stays on the device: the volume is uploaded once as
    one channel (uint8 or fp32), cropped by slicing, and `mr3d_volume_prepare` (csrc/input_ops.hip) does replicate +
    normalise + pad + the 1.5x trilinear resample in one launch per pathway (the reference: ~80 per-slice numpy / mmcv calls
    plus a whole-volume skimage.resize on the CPU, then a 3x larger H2D copy);
  * the random stream is an explicit `numpy.random.RandomState` (same draw order as the reference's module-level numpy
    calls: mode choice, then left, then top), so a recorded seed reproduces the reference's patch;
  * gt masks travel as device tensors (the reference keeps them on the CPU for its skimage mask_target)."""
import numpy as np
import torch

from .volume import prepare_volume


def iou3d_one_to_many(patch, boxes):
    """IoU of one box against [n,6] boxes, inclusive +1 convention, float32 (mmdet/core/evaluation/bbox_overlaps.py:51-99)"""
    p = np.asarray(patch, np.float32)
    b = np.asarray(boxes, np.float32).reshape(-1, 6)
    lo = np.maximum(p[[0, 1, 4]], b[:, [0, 1, 4]])
    hi = np.minimum(p[[2, 3, 5]], b[:, [2, 3, 5]])
    inter = np.prod(np.maximum(hi - lo + 1, 0), axis=1)
    vol_p = np.prod(p[[2, 3, 5]] - p[[0, 1, 4]] + 1)
    vol_b = np.prod(b[:, [2, 3, 5]] - b[:, [0, 1, 4]] + 1, axis=1)
    return (inter / (vol_p + vol_b - inter)).astype(np.float32)


class RandomCrop3D(object):
    """extra_aug.py:146-211.  Patch size is fixed to (W/4, H/4, D); a candidate origin along an axis is redrawn until at least
    one box lies inside the patch along that axis; the patch is accepted when the smallest patch-vs-box IoU reaches the drawn
    mode (1, one of min_ious, or 0) and at least one box lies strictly inside in all three axes."""

    def __init__(self, min_ious=(0.1, 0.3, 0.5, 0.7, 0.9), min_crop_size=0.3, rng=None):
        self.sample_mode = (1, *min_ious, 0)
        self.min_crop_size = min_crop_size                   # kept for signature compatibility; unused, as in the reference
        self.rng = rng if rng is not None else np.random

    def _origin(self, extent, size, box_lo, box_hi):
        while True:
            start = int(self.rng.randint(extent - size))
            if np.any((box_lo >= start) & (box_hi < start + size)):
                return start

    def choose_patch(self, hwd, boxes):
        """-> (patch [x1,y1,x2,y2,z1,z2] with exclusive upper corners, boolean keep mask over the boxes)"""
        h, w, d = hwd
        pw, ph = int(w / 4), int(h / 4)
        boxes = np.asarray(boxes)
        while True:
            min_iou = self.sample_mode[int(self.rng.randint(len(self.sample_mode)))]         # random.choice draws one randint
            left = self._origin(w, pw, boxes[:, 0], boxes[:, 2])
            top = self._origin(h, ph, boxes[:, 1], boxes[:, 3])
            patch = np.array([left, top, left + pw, top + ph, 0, d])
            if iou3d_one_to_many(patch, boxes).min() < min_iou:
                continue
            inside = ((boxes[:, [0, 1, 4]] >= patch[[0, 1, 4]]) & (boxes[:, [2, 3, 5]] < patch[[2, 3, 5]])).all(axis=1)
            if inside.any():
                return patch, inside

    def __call__(self, img, boxes, labels, masks=None):
        """img [H,W,D] (numpy or device tensor), boxes [n,6], labels [n], masks: sequence of [H,W,D] or None"""
        patch, keep = self.choose_patch(tuple(img.shape), boxes)
        x1, y1, x2, y2, z1, z2 = (int(v) for v in patch)
        crop = img[y1:y2, x1:x2, z1:z2]
        new_boxes = np.asarray(boxes)[keep] - np.array([x1, y1, x1, y1, z1, z1], dtype=np.asarray(boxes).dtype)
        new_masks = None
        if masks is not None:
            new_masks = [m[y1:y2, x1:x2, z1:z2] for m, k in zip(masks, keep) if k]
        return crop, new_boxes, np.asarray(labels)[keep], new_masks


def load_npy_volume(path, device='cuda'):
    """the dataset's per-scan `.npy` ([H,W,D] grey; coco_3d_2scales.py:181) -> device tensor, ONE channel, through pinned memory"""
    a = np.load(path, allow_pickle=True)
    if a.dtype not in (np.uint8, np.float32):
        a = a.astype(np.float32)
    t = torch.from_numpy(np.ascontiguousarray(a))
    if torch.cuda.is_available():
        t = t.pin_memory()
    return t.to(device, non_blocking=True)


def prepare_train_sample(grey_hwd, gt_bboxes, gt_labels, gt_masks=None, upscale_factor=1.5, extra_aug=None, size_divisor=32,
                         device='cuda'):
    """coco_3d_2scales.py:169-326 for one scan.  grey_hwd: [H,W,D] volume (numpy or tensor; uploaded once); gt_bboxes [n,6]
    numpy in voxel units; gt_masks: sequence of [H,W,D] 0/1 arrays / tensors or None.  Returns the reference's `data` dict with
    device tensors: imgs / imgs_2 [1,3,D,Hp,Wp], img_meta / img_meta_2, gt_bboxes(_2), gt_labels(_2), gt_masks [G,D,Hp,Wp] uint8
    (gt_masks_2 is None: turned off in the reference, :222-228); None when no gt box is left (the reference skips the scan)."""
    vol = torch.as_tensor(grey_hwd)
    if vol.dtype not in (torch.uint8, torch.float32):
        vol = vol.float()
    vol = vol.to(device)
    boxes = np.asarray(gt_bboxes, np.float32).reshape(-1, 6)
    labels = np.asarray(gt_labels)
    if len(boxes) == 0:
        return None
    masks = None if gt_masks is None else [torch.as_tensor(m).to(device) for m in gt_masks]
    if extra_aug is not None:
        vol, boxes, labels, masks = extra_aug(vol, boxes, labels, masks)
    vol = vol.contiguous()
    img, meta = prepare_volume(vol, size_divisor=size_divisor, layout='HWD', device=device)
    img_2, meta_2 = prepare_volume(vol, size_divisor=size_divisor, upscale=upscale_factor, layout='HWD', device=device)
    dev = img.device
    data = dict(imgs=img, img_meta=meta, imgs_2=img_2, img_meta_2=meta_2,
                gt_bboxes=torch.from_numpy(np.ascontiguousarray(boxes)).to(dev),
                gt_bboxes_2=torch.from_numpy(np.ascontiguousarray(boxes * np.float32(upscale_factor))).to(dev),
                gt_labels=torch.from_numpy(np.ascontiguousarray(labels)).long().to(dev),
                gt_labels_2=torch.from_numpy(np.ascontiguousarray(labels)).long().to(dev), gt_masks=None, gt_masks_2=None)
    if masks is not None:
        hp, wp, _, d = meta['pad_shape']
        g = torch.zeros((len(masks), d, hp, wp), dtype=torch.uint8, device=dev)            # impad to pad_shape, then [G,D,H,W]
        for k, m in enumerate(masks):
            h, w, dd = m.shape
            g[k, :dd, :h, :w] = (m != 0).permute(2, 0, 1)
        data['gt_masks'] = g
    return data
