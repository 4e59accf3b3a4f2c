"""BASELINE configs[1] size, end to end: the engine's ResNet3D-50+FPN3D forward and backward on a 1x3x80x512x512 volume
against the plain-PyTorch restatement (oracle/torch_ref_model.py, pinned to the reference by the goldens) run on the
SAME GPU through PyTorch-ROCm/MIOpen, with identical random weights.  It catches what small shapes cannot (32-bit
index overflow, tile-edge handling at production extents).  MIOpen needs minutes to pick its 3D kernels, so the test
only runs when MR3D_FULLSIZE=1 (run once per round; result recorded in DESIGN.md section 6)."""
import os

import pytest
import torch

from oracle.torch_ref_model import RefExtractFeat

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(os.environ.get('MR3D_FULLSIZE') != '1', reason='set MR3D_FULLSIZE=1 (takes minutes: MIOpen kernel search)')
def test_engine_full_size_vs_torch_on_gpu():
    import mr3d_amd  # noqa: F401
    from mr3d_amd.detector import ExtractFeat3D
    dev = torch.device('cuda:0')
    torch.manual_seed(7)
    ref = RefExtractFeat()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.dim() == 1:
                p.copy_(torch.rand_like(p) * 0.5 + 0.5 if n.endswith('weight') else torch.randn_like(p) * 0.1)
        for n, b in ref.named_buffers():
            if n.endswith('running_mean'):
                b.copy_(torch.randn_like(b) * 0.1)
            elif n.endswith('running_var'):
                b.copy_(torch.rand_like(b) + 0.5)
    eng = ExtractFeat3D(device=dev)
    eng.load_state_dict(ref.state_dict())
    ref = ref.to(dev).eval()
    x = torch.randn(1, 3, 80, 512, 512, device=dev)
    outs = [o.clone() for o in eng.forward_features(x)]
    cots = [torch.randn_like(o) for o in outs]
    eng.zero_grad()
    eng.backward_features(cots)
    outs_ref = ref(x)
    sum((o * c).sum() for o, c in zip(outs_ref, cots)).backward()
    rel = lambda a, b: ((a - b).abs().max() / b.abs().max().clamp(min=1e-12)).item()
    worst_out = max(rel(o, r.detach()) for o, r in zip(outs, outs_ref))
    assert worst_out <= 2e-4, worst_out          # two fp32 implementations with different summation orders
    worst = ('', 0.0)
    for k, p in ref.named_parameters():
        got, exp = eng.get_parameter(k).grad, p.grad
        # norm-wise: element-wise maxima of cancelling sums over 10^6-10^7 positions carry fp32 noise in BOTH runs, and
        # ReLU / max-pool ties route gradient differently between two summation orders (see test_extract_feat.py)
        e = ((got - exp).norm() / exp.norm().clamp(min=1e-12)).item()
        if e > worst[1]:
            worst = (k, e)
    print(f'full-size parity: worst output rel err {worst_out:.2e}; worst parameter-gradient L2 rel err {worst[1]:.2e} ({worst[0]})')
    assert worst[1] <= 3e-2, worst
