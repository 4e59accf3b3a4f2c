from .transforms import delta2bbox3D, bbox2delta3d, bbox2roi3D
from .geometry import bbox_overlaps
from .assigners import MaxIoUAssigner, AssignResult
from .samplers import RandomSampler, SamplingResult
from .bbox_target import bbox_target_3d, bbox_target_single_3d


def build_assigner(cfg, **kwargs):
    """assign_sampling.py:6-14 (mmcv.runner.obj_from_dict replaced by a two-entry table)"""
    if isinstance(cfg, MaxIoUAssigner):
        return cfg
    cfg = dict(cfg)
    assert cfg.pop('type') == 'MaxIoUAssigner'
    return MaxIoUAssigner(**cfg, **kwargs)


def build_sampler(cfg, **kwargs):
    if isinstance(cfg, RandomSampler):
        return cfg
    cfg = dict(cfg)
    assert cfg.pop('type') == 'RandomSampler'
    return RandomSampler(**cfg, **kwargs)


def assign_and_sample(bboxes, gt_bboxes, gt_bboxes_ignore, gt_labels, cfg, draws=None, inside_flags=None):
    """assign_sampling.py:28-35"""
    get = (lambda k: cfg[k]) if isinstance(cfg, dict) else (lambda k: getattr(cfg, k))
    bbox_assigner, bbox_sampler = build_assigner(get('assigner')), build_sampler(get('sampler'))
    assign_result = bbox_assigner.assign(bboxes, gt_bboxes, gt_bboxes_ignore, gt_labels, inside_flags=inside_flags)
    sampling_result = bbox_sampler.sample(assign_result, bboxes, gt_bboxes, gt_labels, draws=draws)
    return assign_result, sampling_result


__all__ = ['delta2bbox3D', 'bbox2delta3d', 'bbox2roi3D', 'bbox_overlaps', 'MaxIoUAssigner', 'AssignResult', 'RandomSampler',
           'SamplingResult', 'bbox_target_3d', 'bbox_target_single_3d', 'build_assigner', 'build_sampler', 'assign_and_sample']
