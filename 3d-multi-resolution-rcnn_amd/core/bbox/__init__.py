from .transforms import delta2bbox3D, bbox2delta3d, bbox2roi3D
from .geometry import bbox_overlaps

__all__ = ['delta2bbox3D', 'bbox2delta3d', 'bbox2roi3D', 'bbox_overlaps']
