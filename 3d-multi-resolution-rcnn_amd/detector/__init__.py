"""Host-side mirror of the reference's model pieces on the hot path (mmdet/models/...)."""
from .extract_feat import ExtractFeat3D

__all__ = ['ExtractFeat3D']
