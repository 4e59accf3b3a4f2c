from .mask_target import mask_target, mask_target_single, paste_masks_3d

__all__ = ['mask_target', 'mask_target_single', 'paste_masks_3d']
