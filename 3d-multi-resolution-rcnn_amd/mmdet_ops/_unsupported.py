"""Names of mmdet.ops that are outside the 3D hot path:
importable, but using one raises NotImplementedError."""
import torch.nn as nn


def _raiser(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(f'mmdet.ops.{name} is a 2D operator outside the MI355X 3D hot path')
    fn.__name__ = name
    return fn


def _module(name):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(f'mmdet.ops.{name} is a 2D operator outside the MI355X 3D hot path')
    return type(name, (nn.Module,), {'__init__': __init__})


DeformConv = _module('DeformConv')
DeformConvPack = _module('DeformConvPack')
ModulatedDeformConv = _module('ModulatedDeformConv')
ModulatedDeformConvPack = _module('ModulatedDeformConvPack')
DeformRoIPooling = _module('DeformRoIPooling')
DeformRoIPoolingPack = _module('DeformRoIPoolingPack')
ModulatedDeformRoIPoolingPack = _module('ModulatedDeformRoIPoolingPack')
RoIPool = _module('RoIPool')
SigmoidFocalLoss = _module('SigmoidFocalLoss')
deform_conv = _raiser('deform_conv')
modulated_deform_conv = _raiser('modulated_deform_conv')
deform_roi_pooling = _raiser('deform_roi_pooling')
roi_pool = _raiser('roi_pool')
sigmoid_focal_loss = _raiser('sigmoid_focal_loss')
