from .anchor_generator_3d import AnchorGenerator3D

__all__ = ['AnchorGenerator3D']
