from .anchor_generator_3d import AnchorGenerator3D
from .anchor_target import anchor_target, anchor_target_single, anchor_inside_flags, images_to_levels

__all__ = ['AnchorGenerator3D', 'anchor_target', 'anchor_target_single', 'anchor_inside_flags', 'images_to_levels']
