"""Host-side mirror of the reference's model pieces on the hot path (mmdet/models/...)."""
from .extract_feat import ExtractFeat3D
from .rpn_proposals import get_bboxes_single
from .heads import RPNHead3D, SharedFCBBoxHead3D, FCNMaskHead3D
from .roi_extractor import SingleRoIExtractor
from .two_stage import TwoStageDetector3D, DEFAULT_TRAIN_CFG, DEFAULT_TEST_CFG, bbox2result3D
from .two_stage_2scales import TwoStageDetector3D2Scales
from . import layers

__all__ = ['ExtractFeat3D', 'get_bboxes_single', 'RPNHead3D', 'SharedFCBBoxHead3D', 'FCNMaskHead3D', 'SingleRoIExtractor', 'layers',
           'TwoStageDetector3D', 'DEFAULT_TRAIN_CFG', 'DEFAULT_TEST_CFG', 'bbox2result3D', 'TwoStageDetector3D2Scales']
