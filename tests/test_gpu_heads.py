"""GPU parity of the HIP-backed layers / heads against plain torch CPU modules with the
same parameters: forward outputs and every parameter / input gradient."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def _rel(a, b):
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def _clone_to(module, dev):
    import copy
    return copy.deepcopy(module).to(dev)


def test_conv3d_linear_deconv_layers_fwd_bwd():
    from mr3d_amd.detector import layers
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    # conv3d 3x3x3 + bias + relu, Cout not a multiple of 16
    x = torch.randn(2, 32, 5, 9, 8, generator=g, requires_grad=True)
    w = (torch.randn(7, 32, 3, 3, 3, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(7, generator=g).requires_grad_(True)
    ref = F.relu(F.conv3d(x, w, b, padding=1))
    cot = torch.randn(ref.shape, generator=g)
    ref.backward(cot)
    xd, wd, bd = (t.detach().to(dev).requires_grad_(True) for t in (x, w, b))
    out = layers.conv3d(xd, wd, bd, padding=1, relu=True)
    assert out.shape == ref.shape and _rel(out.detach().cpu(), ref.detach()) <= RTOL
    out.backward(cot.to(dev))
    assert _rel(xd.grad.cpu(), x.grad) <= RTOL and _rel(wd.grad.cpu(), w.grad) <= RTOL and _rel(bd.grad.cpu(), b.grad) <= RTOL
    # linear with rows not a multiple of 64
    xl = torch.randn(70, 128, generator=g, requires_grad=True)
    wl = (torch.randn(12, 128, generator=g) * 0.1).requires_grad_(True)
    bl = torch.randn(12, generator=g).requires_grad_(True)
    refl = F.linear(xl, wl, bl)
    cl = torch.randn(refl.shape, generator=g)
    refl.backward(cl)
    xld, wld, bld = (t.detach().to(dev).requires_grad_(True) for t in (xl, wl, bl))
    outl = layers.linear(xld, wld, bld, impl='conv')           # the pointwise-conv form of the first round
    assert _rel(outl.detach().cpu(), refl.detach()) <= RTOL
    outl.backward(cl.to(dev))
    assert _rel(xld.grad.cpu(), xl.grad) <= RTOL and _rel(wld.grad.cpu(), wl.grad) <= RTOL and _rel(bld.grad.cpu(), bl.grad) <= RTOL
    # transposed conv k2 s2 (fcn_mask_head_3d.py:60-66: 64 -> 64): generic and roi-shaped (5x7x7 / 2x7x7 tile) volumes
    for shape in ((3, 64, 3, 5, 4), (9, 64, 10, 14, 14)):
        xt = torch.randn(shape, generator=g, requires_grad=True)
        wt = (torch.randn(64, 64, 2, 2, 2, generator=g) * 0.1).requires_grad_(True)
        bt = torch.randn(64, generator=g).requires_grad_(True)
        reft = F.relu(F.conv_transpose3d(xt, wt, bt, stride=2))
        ct = torch.randn(reft.shape, generator=g)
        reft.backward(ct)
        xtd, wtd, btd = (t.detach().to(dev).requires_grad_(True) for t in (xt, wt, bt))
        outt = layers.conv_transpose3d_k2s2(xtd, wtd, btd, relu=True)
        assert outt.shape == reft.shape and _rel(outt.detach().cpu(), reft.detach()) <= RTOL
        outt.backward(ct.to(dev))
        assert _rel(xtd.grad.cpu(), xt.grad) <= RTOL and _rel(wtd.grad.cpu(), wt.grad) <= RTOL and _rel(btd.grad.cpu(), bt.grad) <= RTOL


@pytest.mark.parametrize('rows,k,o,relu,bias', [
    (512, 9408, 1024, True, True),       # shared_fcs[0] of the bbox head: split-K forward, unsplit dgrad / wgrad
    (469, 1024, 1024, True, True),       # rows not a multiple of anything
    (512, 1024, 2, False, True),         # fc_cls
    (512, 1024, 12, False, True),        # fc_reg
    (37, 100, 6, False, False),          # K, N, ld not multiples of 4 / 16: scalar edge loads
    (1, 16, 1, True, True),
])
def test_linear_gemm_kernels(rows, k, o, relu, bias):
    """csrc/linear.hip (the default of layers.linear): forward / dgrad / wgrad against float64 torch"""
    from mr3d_amd.detector import layers
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(rows + k + o)
    x = torch.randn(rows, k, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(o, k, generator=g, dtype=torch.float64) / k ** 0.5).requires_grad_(True)
    b = torch.randn(o, generator=g, dtype=torch.float64).requires_grad_(True) if bias else None
    ref = F.linear(x, w, b)
    if relu:
        ref = F.relu(ref)
    cot = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(cot)
    xd, wd = (t.detach().float().to(dev).requires_grad_(True) for t in (x, w))
    bd = b.detach().float().to(dev).requires_grad_(True) if bias else None
    out = layers.linear(xd, wd, bd, relu=relu)
    assert out.shape == ref.shape and _rel(out.detach().cpu().double(), ref.detach()) <= 2e-5
    out.backward(cot.float().to(dev))
    assert _rel(xd.grad.cpu().double(), x.grad) <= 2e-5 and _rel(wd.grad.cpu().double(), w.grad) <= 2e-5
    if bias:
        assert _rel(bd.grad.cpu().double(), b.grad) <= 2e-5
    # no rows (an image without sampled rois): empty output, zero weight gradient
    e = layers.linear(torch.zeros(0, k, device=dev, requires_grad=True), wd, bd)
    assert e.shape == (0, o)


class _RefRPN(nn.Module):          # rpn_head_3d.py:18-35
    def __init__(self):
        super().__init__()
        self.rpn_conv = nn.Conv3d(64, 64, 3, padding=1)
        self.rpn_cls = nn.Conv3d(64, 1, 1)
        self.rpn_reg = nn.Conv3d(64, 6, 1)

    def forward(self, x):
        x = F.relu(self.rpn_conv(x))
        return self.rpn_cls(x), self.rpn_reg(x)


def test_rpn_head_forward_backward_and_proposals():
    from mr3d_amd.detector import RPNHead3D
    dev = torch.device('cuda:0')
    torch.manual_seed(1)
    ref = _RefRPN()
    for p in ref.parameters():
        nn.init.normal_(p, 0, 0.05)
    head = RPNHead3D().to(dev)
    head.load_state_dict(ref.state_dict())                       # same key names as the reference head
    feats = [torch.randn(1, 64, d, h, w) for (d, h, w) in ((6, 16, 16), (3, 8, 8))]
    cots = []
    ref_out = []
    for f in feats:
        c, r = ref(f)
        cots.append((torch.randn_like(c), torch.randn_like(r)))
        ref_out.append((c, r))
    sum((c * a).sum() + (r * b).sum() for (c, r), (a, b) in zip(ref_out, cots)).backward()
    cls, reg = head([f.to(dev) for f in feats])
    for i in range(2):
        assert cls[i].shape == ref_out[i][0].shape and reg[i].shape == ref_out[i][1].shape
        assert _rel(cls[i].detach().cpu(), ref_out[i][0].detach()) <= RTOL and _rel(reg[i].detach().cpu(), ref_out[i][1].detach()) <= RTOL
    sum((cls[i] * cots[i][0].to(dev)).sum() + (reg[i] * cots[i][1].to(dev)).sum() for i in range(2)).backward()
    for (k, p), (k2, q) in zip(sorted(head.named_parameters()), sorted(ref.named_parameters())):
        assert k == k2 and _rel(p.grad.cpu(), q.grad) <= RTOL, k
    # proposals straight from the head's channels-last outputs
    cfg = dict(nms_across_levels=False, nms_pre=2000, nms_post=2000, max_num=300, nms_thr=0.7, min_bbox_size=0)
    head.anchor_strides, head.anchor_strides_depth = head.anchor_strides[:2], head.anchor_strides_depth[:2]
    props = head.get_bboxes([c.detach() for c in cls], [r.detach() for r in reg], [dict(img_shape=(64, 64, 3, 12))], cfg)
    assert len(props) == 1 and props[0].shape == (300, 7) and bool((props[0][:-1, 6] >= props[0][1:, 6]).all())


def test_bbox_head_and_mask_head_vs_torch():
    from mr3d_amd.detector import SharedFCBBoxHead3D, FCNMaskHead3D
    dev = torch.device('cuda:0')
    torch.manual_seed(2)
    # ---- bbox head: 9408 -> 1024 -> 1024 -> {2, 12}
    head = SharedFCBBoxHead3D().to(dev)
    ref = nn.ModuleDict(dict(fc0=nn.Linear(9408, 1024), fc1=nn.Linear(1024, 1024), cls=nn.Linear(1024, 2), reg=nn.Linear(1024, 12)))
    ref['fc0'].load_state_dict(head.shared_fcs[0].state_dict()); ref['fc1'].load_state_dict(head.shared_fcs[1].state_dict())
    ref['cls'].load_state_dict(head.fc_cls.state_dict()); ref['reg'].load_state_dict(head.fc_reg.state_dict())
    with torch.no_grad():
        for m in (ref['cls'], ref['reg']):
            m.weight.mul_(30)          # the reference's std = 0.01 / 0.001 init would make the check vacuous
        head.fc_cls.weight.copy_(ref['cls'].weight); head.fc_reg.weight.copy_(ref['reg'].weight)
    x = torch.randn(37, 64, 3, 7, 7)
    h = F.relu(ref['fc1'](F.relu(ref['fc0'](x.view(37, -1)))))
    rc, rr = ref['cls'](h), ref['reg'](h)
    a, b = torch.randn_like(rc), torch.randn_like(rr)
    ((rc * a).sum() + (rr * b).sum()).backward()
    xd = x.to(dev).requires_grad_(True)
    c, r = head(xd)
    assert c.shape == (37, 2) and r.shape == (37, 12)
    assert _rel(c.detach().cpu(), rc.detach()) <= RTOL and _rel(r.detach().cpu(), rr.detach()) <= RTOL
    ((c * a.to(dev)).sum() + (r * b.to(dev)).sum()).backward()
    assert _rel(head.shared_fcs[0].weight.grad.cpu(), ref['fc0'].weight.grad) <= RTOL
    assert _rel(head.shared_fcs[1].bias.grad.cpu(), ref['fc1'].bias.grad) <= RTOL
    assert _rel(head.fc_reg.weight.grad.cpu(), ref['reg'].weight.grad) <= RTOL
    assert _rel(head.fc_cls.bias.grad.cpu(), ref['cls'].bias.grad) <= RTOL
    # ---- mask head: 4 x (conv 3^3 + relu) -> deconv k2 s2 + relu -> 1x1x1 logits
    mh = FCNMaskHead3D().to(dev)
    convs = nn.ModuleList([nn.Conv3d(64, 64, 3, padding=1) for _ in range(4)])
    up, logits = nn.ConvTranspose3d(64, 64, 2, stride=2), nn.Conv3d(64, 2, 1)
    for i in range(4):
        convs[i].load_state_dict(mh.convs[i].conv.state_dict())
    up.load_state_dict(mh.upsample.state_dict()); logits.load_state_dict(mh.conv_logits.state_dict())
    xm = torch.randn(5, 64, 10, 14, 14)

    def run(dt):
        cv = [nn.Conv3d(64, 64, 3, padding=1).to(dt) for _ in range(4)]
        u, lg = nn.ConvTranspose3d(64, 64, 2, stride=2).to(dt), nn.Conv3d(64, 2, 1).to(dt)
        for i in range(4):
            cv[i].load_state_dict(convs[i].state_dict())
        u.load_state_dict(up.state_dict()); lg.load_state_dict(logits.state_dict())
        xi = xm.detach().clone().to(dt).requires_grad_(True)
        yy = xi
        for c_ in cv:
            yy = F.relu(c_(yy))
        yy = lg(F.relu(u(yy)))
        (yy * cot.to(dt)).sum().backward()
        return yy.detach(), dict(c0=cv[0].weight.grad, b3=cv[3].bias.grad, up=u.weight.grad, lg=lg.weight.grad, x=xi.grad)

    with torch.no_grad():
        y_shape = logits(F.relu(up(convs[0](xm)))).shape
    cot = torch.randn(y_shape)
    y32, g32 = run(torch.float32)
    y64, g64 = run(torch.float64)          # ground truth: calibrates what fp32 can promise through 5 ReLU layers
    xmd = xm.detach().to(dev).requires_grad_(True)
    ym = mh(xmd)
    assert ym.shape == (5, 2, 20, 28, 28) and _rel(ym.detach().cpu().double(), y64) <= RTOL
    (ym * cot.to(dev)).sum().backward()
    got = dict(c0=mh.convs[0].conv.weight.grad, b3=mh.convs[3].conv.bias.grad, up=mh.upsample.weight.grad,
               lg=mh.conv_logits.weight.grad, x=xmd.grad)
    for k in got:
        e_gpu, e_cpu = _rel(got[k].cpu().double(), g64[k]), _rel(g32[k].double(), g64[k])
        # Every layer of this chain is held to ~1e-6 in isolation (test above: fwd, dx, dw, db).  Through 5 ReLUs,
        # though, a pre-activation within fp32 rounding of 0 takes its mask from the summation order, and ONE flipped
        # element moves a weight gradient by |dy*x| / max|dW| ~ 1/150 here (measured: 1.0e-3 with one flip, while the
        # single-layer errors are 4e-7).  The bound therefore only guards against structural errors.
        l2 = float((got[k].cpu().double() - g64[k]).norm() / g64[k].norm())
        assert l2 <= 2e-3 and e_gpu <= max(3e-2, 5 * e_cpu), (k, l2, e_gpu, e_cpu)
