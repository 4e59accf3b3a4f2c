"""a checkpoint of the reference's single-pathway model loads into the detector and
the detector saves under the same names (tests/golden/state_dict_keys.json = key -> shape of the reference modules as
configs/3d-multi-resolution-rcnn.py builds them, written by tests/golden/make_golden.py)."""
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _ref_keys():
    return json.load(open(os.path.join(HERE, 'golden', 'state_dict_keys.json')))


def test_state_dict_keys_and_shapes_equal_reference():
    import mr3d_amd  # noqa: F401
    from mr3d_amd.detector import TwoStageDetector3D
    det = TwoStageDetector3D(device='cpu')
    ref = _ref_keys()
    sd = det.state_dict()
    assert sorted(sd) == sorted(ref)
    for k, shape in ref.items():
        assert list(sd[k].shape) == shape, k


def test_reference_checkpoint_round_trip_with_module_prefix_and_extra_heads():
    import mr3d_amd  # noqa: F401
    from mr3d_amd.detector import TwoStageDetector3D
    det = TwoStageDetector3D(device='cpu')
    g = torch.Generator().manual_seed(0)
    ref = _ref_keys()
    ckpt = {'module.' + k: (torch.zeros(shape, dtype=torch.long) if k.endswith('num_batches_tracked')
                            else torch.randn(shape, generator=g)) for k, shape in ref.items()}
    ckpt['module.rpn_head_2.rpn_conv.weight'] = torch.zeros(64, 64, 3, 3, 3)       # second pathway: not in this detector
    res = det.load_checkpoint({'state_dict': ckpt, 'meta': {}})
    assert res.missing_keys == [] and res.unexpected_keys == ['rpn_head_2.rpn_conv.weight']
    out = det.state_dict()
    for k in ref:
        assert torch.equal(out[k].cpu(), ckpt['module.' + k]), k
    # the flat parameter arena really holds the loaded values (parameters are views of it)
    w = det.extract.get_parameter('backbone.layer2.0.conv2.weight')
    assert torch.equal(w.detach(), ckpt['module.backbone.layer2.0.conv2.weight'])
    assert w.data_ptr() >= det.extract.flat_params.data_ptr()


def test_shipped_dual_pathway_model_state_dict_equals_reference():
    """MaskRCNN3D2Scales as configs/3d-multi-resolution-rcnn.py builds it (rpn_head_2, refinement_head without a
    classifier, refinement_mask_head): same checkpoint keys and shapes"""
    import mr3d_amd  # noqa: F401
    from mr3d_amd.detector import TwoStageDetector3D2Scales
    det = TwoStageDetector3D2Scales(device='cpu')
    ref = json.load(open(os.path.join(HERE, 'golden', 'state_dict_keys_2scales.json')))
    sd = det.state_dict()
    assert sorted(sd) == sorted(ref)
    for k, shape in ref.items():
        assert list(sd[k].shape) == shape, k
