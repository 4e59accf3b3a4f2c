"""ExtractFeat3D (ResNet3D-50 + FPN3D on the HIP conv engine): CPU structural tests + GPU parity against
the reference-generated golden vectors and a plain-PyTorch fp32 restatement."""
import numpy as np
import pytest
import torch

from weightfill import closed_form_array, volume_like, cotangent_like
from oracle.torch_ref_model import RefExtractFeat

RTOL = 1e-4


def _fill_like_golden(named):
    """same closed-form fill as tests/golden/make_golden.py: backbone names (salt 1) / neck names (salt 2),
    keyed WITHOUT the detector-level prefix"""
    with torch.no_grad():
        for name, t in named:
            if name.endswith('num_batches_tracked') or '.' not in name:     # skip the engine's private flat arenas
                continue
            pref, rest = name.split('.', 1)
            salt = {'backbone': 1, 'neck': 2}[pref]
            t.copy_(torch.from_numpy(closed_form_array(rest, tuple(t.shape), salt)).to(t.device))


def test_torch_restatement_matches_reference_golden(golden_dir):
    """pins tests/torch_ref_model.py to the real reference modules (outputs + gradients)"""
    g = np.load(f'{golden_dir}/backbone_fpn.npz')
    m = RefExtractFeat()
    _fill_like_golden(list(m.named_parameters()) + list(m.named_buffers()))
    x = volume_like((1, 3, 8, 32, 32))
    outs = m(x)
    loss = sum((o * cotangent_like(o.shape, i)).sum() for i, o in enumerate(outs))
    loss.backward()
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.detach().numpy(), g[f'a_p{i}'], rtol=1e-4, atol=1e-4 * np.abs(g[f'a_p{i}']).max())
    norms = dict(zip(g['a_grad_names'].tolist(), g['a_grad_norms'].tolist()))
    for k, p in m.named_parameters():
        assert abs(p.grad.double().norm().item() - norms[k]) <= 1e-3 * max(norms[k], 1e-6), k


@pytest.mark.gpu
def test_engine_state_dict_and_arena():
    from mr3d_amd.detector import ExtractFeat3D
    eng = ExtractFeat3D()
    ref = RefExtractFeat()
    assert set(eng.state_dict().keys()) == set(ref.state_dict().keys())
    assert len(eng.state_dict()) == 318 + 16                    # 318 backbone entries + 16 FPN
    assert sum(p.numel() for p in eng.parameters()) == 2909744 + 504320
    for p in eng.parameters():                                   # every .grad is a view of the flat gradient arena
        assert p.grad.data_ptr() >= eng.flat_grads.data_ptr()
    # zero-init of the last norm of every block (resnet3d.py:458-461)
    assert float(eng.get_parameter('backbone.layer3.2.bn3d3.weight').detach().abs().sum()) == 0.0
    eng.load_state_dict(ref.state_dict())                        # a reference-shaped checkpoint loads in place
    np.testing.assert_array_equal(eng.get_parameter('backbone.conv1.weight').detach().cpu().numpy(),
                                  ref.backbone.conv1.weight.detach().numpy())


def _rel_err(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize('tag,shape', [('a', (1, 3, 8, 32, 32)), ('b', (1, 3, 16, 64, 64))])
def test_engine_matches_reference_golden(golden_dir, tag, shape):
    from mr3d_amd.detector import ExtractFeat3D
    g = np.load(f'{golden_dir}/backbone_fpn.npz')
    eng = ExtractFeat3D()
    _fill_like_golden(list(eng.named_parameters()) + list(eng.named_buffers()))
    x = volume_like(shape).cuda()
    outs = eng.forward_features(x)
    for i, o in enumerate(outs):
        assert o.shape == g[f'{tag}_p{i}'].shape
        assert _rel_err(o.cpu().numpy(), g[f'{tag}_p{i}']) <= RTOL, f'P{i + 2}'
    eng.zero_grad()
    eng.backward_features([cotangent_like(o.shape, i).cuda() for i, o in enumerate(outs)])
    norms = dict(zip(g[f'{tag}_grad_names'].tolist(), g[f'{tag}_grad_norms'].tolist()))

    def tol(name):
        # Outputs are continuous in the arithmetic and are held to RTOL above.  Gradients are not: they pass
        # through MaxPool3d's argmax and ReLU's mask, and a window whose two largest values agree to fp32 rounding
        # (or a pre-activation within rounding of 0) routes a whole gradient contribution differently depending on
        # the summation order of the conv kernel -- the reference's own fp32 and fp64 runs disagree the same way.
        # One such flip moves a stem gradient by ~1/sqrt(#positions).  Per-op tests (test_gpu_conv3d.py,
        # test_gpu_aux_ops.py) hold every kernel to 1e-4 on data without ties; here the bound only has to catch
        # structural errors, which are O(1).
        # Measured: torch's own CPU fp32 run of this case on the GPU box's EPYC host differs from its fp64 run (and
        # from the golden generated on another host) by 1.1e-2 in backbone.layer2.0.bn3d2.weight and 2.3e-3 in
        # backbone.bn3d1.weight -- the same values this engine produces with its ci-major stem kernel.
        return 3e-2

    for k, p in eng.named_parameters():
        got = p.grad.double().norm().item()
        assert abs(got - norms[k]) <= tol(k) * max(norms[k], 1e-6) + 1e-7, (k, got, norms[k])
    for key in g.files:
        if key.startswith(f'{tag}_g_'):
            name = key[len(f'{tag}_g_'):]
            assert _rel_err(eng.get_parameter(name).grad.cpu().numpy(), g[key]) <= tol(name), name


@pytest.mark.gpu
def test_engine_vs_torch_restatement_random_weights_batch2():
    """random (not closed-form) weights, batch 2, odd sizes; autograd bridge; gradient accumulation"""
    from mr3d_amd.detector import ExtractFeat3D
    torch.manual_seed(3)
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
    eng = ExtractFeat3D()
    eng.load_state_dict(ref.state_dict())
    x = torch.randn(2, 3, 10, 40, 36)
    outs_ref = ref(x)
    cots = [torch.randn_like(o) for o in outs_ref]
    sum((o * c).sum() for o, c in zip(outs_ref, cots)).backward()
    # fp64 run of the same graph = ground truth; the fp32 CPU run's distance from it calibrates the tolerance
    # (bias / BN gradients are cancelling sums over ~10^5 positions: their fp32 noise is far above 1e-7)
    import copy
    ref64 = copy.deepcopy(ref).double()
    ref64.zero_grad()
    outs64 = ref64(x.double())
    sum((o * c.double()).sum() for o, c in zip(outs64, cots)).backward()
    eng.zero_grad()
    outs = eng(x.cuda())                                        # through torch.autograd
    for o, r in zip(outs, outs64):
        assert _rel_err(o.detach().cpu().double().numpy(), r.detach().numpy()) <= RTOL
    sum((o * c.cuda()).sum() for o, c in zip(outs, cots)).backward()
    p64 = dict(ref64.named_parameters())
    for (k, p), (k2, q) in zip(sorted(eng.named_parameters()), sorted(ref.named_parameters())):
        assert k == k2
        truth = p64[k].grad.numpy()
        err_cpu32 = _rel_err(q.grad.double().numpy(), truth)
        err_gpu = _rel_err(p.grad.cpu().double().numpy(), truth)
        # discontinuities (argmax / ReLU-mask flips at fp32 ties, see tol() above) bound what an end-to-end
        # gradient comparison can promise; kernels are held to 1e-4 in the per-op tests
        assert err_gpu <= max(5e-2, 5 * err_cpu32), (k, err_gpu, err_cpu32)
    # a second backward accumulates (autograd semantics of .grad)
    g1 = eng.flat_grads.clone()
    outs = eng(x.cuda())
    sum((o * c.cuda()).sum() for o, c in zip(outs, cots)).backward()
    assert _rel_err(eng.flat_grads.cpu().numpy(), 2 * g1.cpu().numpy()) <= 1e-5


@pytest.mark.gpu
def test_plan_graph_replay_equals_stream_launch():
    """mr3d_plan_capture / mr3d_graph_launch: the captured hipGraph (side-stream forks and joins included) reproduces
    the stream-launched forward outputs bit for bit and the parameter gradients up to atomic summation order"""
    import mr3d_amd  # noqa: F401
    from mr3d_amd.detector import ExtractFeat3D
    dev = torch.device('cuda:0')
    torch.manual_seed(11)
    eng = ExtractFeat3D(device=dev)
    eng.init_weights()
    x = torch.randn(1, 3, 16, 64, 64, device=dev)
    outs = [o.clone() for o in eng.forward_features(x)]
    cots = [torch.randn_like(o) for o in outs]
    eng.zero_grad()
    eng.backward_features(cots)
    g_ref = eng.flat_grads.clone()
    P = eng._plan_for(x.shape)
    eng.capture(P)
    outs2 = eng.forward_features(x)
    for a, b in zip(outs, outs2):
        assert torch.equal(a, b)
    eng.zero_grad()
    eng.backward_features(cots)
    torch.cuda.synchronize()
    scale = g_ref.abs().max().item()
    assert (eng.flat_grads - g_ref).abs().max().item() <= 1e-5 * scale


@pytest.mark.gpu
def test_captured_graphs_keep_their_own_transformed_weight_scratch():
    """The two-dimensional Winograd kernels read pre-transformed weights from a scratch buffer.  It is plan-owned (a torch
    buffer the Plan keeps alive, filled by its WINO2D_PREP record): capturing a small plan, then a larger one (more
    qualifying convolutions, a bigger scratch) must leave the first graph's addresses valid -- replaying the first graph
    after the second was captured and run reproduces its stream-launched outputs bit for bit."""
    import mr3d_amd  # noqa: F401
    from mr3d_amd import _hip
    from mr3d_amd.detector import ExtractFeat3D
    dev = torch.device('cuda:0')
    torch.manual_seed(12)
    eng = ExtractFeat3D(device=dev)
    eng.init_weights()
    # P2 of a 48x192x192 volume (24x48x48: 432 tiles of 2x8x8) is the smallest map here that takes the two-dimensional form
    # (>= 256 tiles); at 80x256x256 layer1's 16 -> 16 convolutions qualify too
    xs = torch.randn(1, 3, 48, 192, 192, device=dev)
    xl = torch.randn(1, 3, 80, 256, 256, device=dev)
    ref_s = [o.clone() for o in eng.forward_features(xs)]
    ref_l = [o.clone() for o in eng.forward_features(xl)]
    Ps, Pl = eng._plan_for(xs.shape), eng._plan_for(xl.shape)
    n_prep = lambda plan: sum(1 for k in range(plan.n) if plan.array[k].kind == _hip.OP_WINO2D_PREP)
    ws = lambda plan: sum(t.numel() for t in plan._u_ws)
    assert n_prep(Ps.fwd) == 1 and n_prep(Pl.fwd) == 1 and ws(Pl.fwd) > ws(Ps.fwd) > 0       # the larger volume takes more such convolutions
    assert not (set(t.data_ptr() for t in Ps.fwd._u_ws) & set(t.data_ptr() for t in Pl.fwd._u_ws))
    eng.capture(Ps)
    eng.capture(Pl)
    out_l = eng.forward_features(xl)
    for a, b in zip(ref_l, out_l):
        assert torch.equal(a, b)
    out_s = eng.forward_features(xs)                                             # replay of the FIRST graph
    for a, b in zip(ref_s, out_s):
        assert torch.equal(a, b)
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_conv3d_forward_without_scratch_takes_an_instance_that_needs_none():
    """mr3d_conv3d_forward never allocates: with no workspace in the descriptor a 3x3x3 64->64 convolution on a large map
    runs on the one-dimensional Winograd / direct kernels; with the scratch lent it takes the two-dimensional form; both
    agree with each other to the 1e-4 parity bound"""
    import ctypes
    import mr3d_amd  # noqa: F401
    from mr3d_amd import _hip
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    x = torch.randn(1, 20, 64, 64, 64, device=dev)
    w = _hip.pack_conv_weight(torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05)
    y2d = _hip.conv3d_forward(x, w, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    d = _hip.Conv3dDesc()
    y = torch.empty_like(y2d)
    d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    d.n, d.d, d.h, d.w_in, d.cin, d.cout = 1, 20, 64, 64, 64, 64
    d.kd = d.kh = d.kw = 3
    d.sd = d.sh = d.sw = 1
    d.pd = d.ph = d.pw = 1
    v = _hip.Conv3dVariant()
    _hip.check(_hip.lib().mr3d_conv3d_forward_variant(ctypes.byref(d), ctypes.byref(v)), 'variant')
    assert v.winograd in (2, 4)                                                  # no scratch: not the weight-transforming form
    d.workspace_bytes = -1
    _hip.check(_hip.lib().mr3d_conv3d_forward_variant(ctypes.byref(d), ctypes.byref(v)), 'variant')
    assert v.winograd == 8
    d.workspace_bytes = 0
    _hip.check(_hip.lib().mr3d_conv3d_forward(ctypes.byref(d), _hip.stream_ptr(dev)), 'mr3d_conv3d_forward')
    torch.cuda.synchronize()
    assert (y - y2d).abs().max().item() <= 1e-4 * y2d.abs().max().item()


@pytest.mark.gpu
def test_zero_copy_binding_equals_staged_copy():
    """bind_input / bind_grads: a dense NCDHW volume and channels-last P2..P4 gradients are read in place; any other
    layout goes through the plan's staging buffers -- same outputs and gradients either way; an in-place change of a
    bound volume before backward is refused"""
    import mr3d_amd  # noqa: F401
    from mr3d_amd.detector import ExtractFeat3D
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    eng = ExtractFeat3D(device=dev)
    eng.init_weights()
    x = torch.randn(1, 3, 16, 64, 64, device=dev)
    outs = [o.clone() for o in eng.forward_features(x)]
    P = eng._last
    assert P.bound_x is x                                                    # read in place
    cots_cl = [torch.randn(o.shape, device=dev).contiguous(memory_format=torch.channels_last_3d) for o in outs]
    eng.zero_grad()
    eng.backward_features(cots_cl)
    assert all(P.bound_g[i].data_ptr() == cots_cl[i].data_ptr() for i in range(3))
    g_direct = eng.flat_grads.clone()
    # the same pass through the staging copies: a strided (non-dense) view of the volume, NCDHW-contiguous gradients
    xs = torch.zeros(1, 3, 16, 64, 128, device=dev)[..., ::2]
    xs.copy_(x)
    outs2 = eng.forward_features(xs)
    assert eng._last.bound_x is eng._last.x_in
    for a, b in zip(outs, outs2):
        assert torch.equal(a, b)
    eng.zero_grad()
    eng.backward_features([c.contiguous() for c in cots_cl])
    assert all(P.bound_g[i] is P.gouts[i] for i in range(3))
    scale = g_direct.abs().max().item()
    assert (eng.flat_grads - g_direct).abs().max().item() <= 1e-5 * scale
    # modified in place after forward: refused
    eng.forward_features(x)
    x.add_(1.0)
    with pytest.raises(RuntimeError):
        eng.backward_features(cots_cl)


def engine_activation_pattern(P):
    """The ReLU masks and max-pool arg-max of the engine's last forward pass, read from its plan buffers, in the
    restatement's layout (oracle/torch_ref_model.RefExtractFeat.linearize_at)."""
    def ncdhw(t):
        return t.permute(0, 4, 1, 2, 3).contiguous().cpu()
    pat = {'stem': (ncdhw(P.stem_out) > 0)}
    N, Ds, Hs, Ws, _ = P.stem_out.shape
    words = P.pool_idx.cpu().numpy().view(np.uint8).reshape(P.pool_idx.shape[:4] + (16,))     # tap id per channel (4 x u8 per word)
    tap = torch.from_numpy(words.astype(np.int64)).permute(0, 4, 1, 2, 3)                       # [N,16,Do,Ho,Wo]
    Do, Ho, Wo = tap.shape[2:]
    od, oh, ow = torch.meshgrid(torch.arange(Do), torch.arange(Ho), torch.arange(Wo), indexing='ij')
    idd, ihh, iww = 2 * od + tap // 9 - 1, 2 * oh + (tap // 3) % 3 - 1, 2 * ow + tap % 3 - 1
    assert (idd >= 0).all() and (idd < Ds).all() and (ihh >= 0).all() and (ihh < Hs).all() and (iww >= 0).all() and (iww < Ws).all()
    pat['pool_index'] = (idd * Hs + ihh) * Ws + iww
    counters = {}
    for (blk, st) in zip(P.blocks_meta, P.blk):
        key = f"layer{blk['stage'] + 1}.{blk['block']}"
        pat[key + '.relu1'], pat[key + '.relu2'], pat[key + '.relu3'] = ncdhw(st['a1']) > 0, ncdhw(st['a2']) > 0, ncdhw(st['out']) > 0
    return pat


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 3, 10, 40, 36), (1, 3, 16, 64, 64)])
def test_engine_gradients_at_its_own_activation_pattern(shape):
    """VERDICT r1 #3 (tie-free gradient parity): the fp64 restatement is evaluated AT THE ENGINE'S OWN ReLU masks and
    max-pool arg-max (read from its plan buffers), which removes the only discontinuities of the graph -- what remains is
    pure arithmetic, and every parameter gradient is held to 1e-3 of its own scale element-wise and 2e-4 in L2 (the
    end-to-end test above needs 3e-2 because fp32 ties flip masks between two summation orders)."""
    import copy
    from mr3d_amd.detector import ExtractFeat3D
    torch.manual_seed(21)
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
    eng = ExtractFeat3D()
    eng.load_state_dict(ref.state_dict())
    x = torch.randn(*shape)
    outs = [o.clone() for o in eng.forward_features(x.cuda())]
    P = eng._last
    P.blocks_meta = eng.blocks
    cots = [torch.randn(o.shape) for o in outs]
    eng.zero_grad()
    eng.backward_features([c.cuda() for c in cots])
    pattern = engine_activation_pattern(P)
    ref64 = copy.deepcopy(ref).double().linearize_at(pattern)
    outs64 = ref64(x.double())
    for o, r in zip(outs, outs64):
        assert _rel_err(o.cpu().double().numpy(), r.detach().numpy()) <= RTOL        # the pattern IS the engine's: outputs agree
    sum((o * c.double()).sum() for o, c in zip(outs64, cots)).backward()
    worst_max, worst_l2 = ('', 0.0), ('', 0.0)
    for k, q in ref64.named_parameters():
        got, exp = eng.get_parameter(k).grad.cpu().double(), q.grad
        e_max = ((got - exp).abs().max() / exp.abs().max().clamp(min=1e-30)).item()
        e_l2 = ((got - exp).norm() / exp.norm().clamp(min=1e-30)).item()
        worst_max = max(worst_max, (k, e_max), key=lambda t: t[1])
        worst_l2 = max(worst_l2, (k, e_l2), key=lambda t: t[1])
    print(f'pattern-pinned gradient parity {shape}: worst max-norm rel {worst_max[1]:.2e} ({worst_max[0]}), '
          f'worst L2 rel {worst_l2[1]:.2e} ({worst_l2[0]})')
    assert worst_max[1] <= 1e-3, worst_max
    assert worst_l2[1] <= 2e-4, worst_l2


@pytest.mark.gpu
def test_overlapped_exchange_is_ordered_when_the_mark_was_not_recorded():
    """dist_utils.allreduce_overlapped starts its big collective behind the backward plan's MARK_GRADS_BUT_STEM point.  The
    checked wait (mr3d_plan_stream_wait_mark_checked) must say when that point does not exist for this step -- side stream
    off, plan replayed from a hipGraph -- so that the communication stream falls back to waiting for the caller's stream."""
    import mr3d_amd  # noqa: F401
    from mr3d_amd import _hip
    from mr3d_amd.detector import ExtractFeat3D
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    eng = ExtractFeat3D(device=dev)
    x = torch.randn(1, 3, 16, 64, 64, device=dev)
    cots = [torch.randn_like(o) for o in eng.forward_features(x)]
    comm = torch.cuda.Stream(device=dev)
    slot = eng.MARK_GRADS_BUT_STEM

    def fwd_bwd():
        eng.zero_grad()
        eng.forward_features(x)
        eng.backward_features(cots)

    fwd_bwd()
    assert _hip.plan_stream_wait_mark_checked(slot, comm) is True          # live run: recorded
    assert _hip.plan_stream_wait_mark_checked(slot, comm) is False         # consumed: one wait per recorded mark
    fwd_bwd()
    assert _hip.plan_stream_wait_mark_checked(slot, comm) is True
    g_ref = eng.flat_grads.clone()
    try:
        _hip.lib().mr3d_plan_set_side_stream(0)
        fwd_bwd()
        assert _hip.plan_stream_wait_mark_checked(slot, comm) is False     # SYNC records were skipped
    finally:
        _hip.lib().mr3d_plan_set_side_stream(1)
    fwd_bwd()
    assert _hip.plan_stream_wait_mark_checked(slot, comm) is True
    eng.capture(eng._plan_for(x.shape))                                    # capture "records" the mark as a graph node only
    assert _hip.plan_stream_wait_mark_checked(slot, comm) is False
    fwd_bwd()                                                              # graph replay: the event is not re-recorded
    assert _hip.plan_stream_wait_mark_checked(slot, comm) is False
    torch.cuda.synchronize()
    scale = g_ref.abs().max().item()
    assert (eng.flat_grads - g_ref).abs().max().item() <= 1e-5 * scale


@pytest.mark.gpu
def test_early_bucket_is_final_at_its_mark():
    """Round 5: the gradients of layer3, layer4 and the neck are finished in the MIDDLE of the backward plan (a FINISH_GRADS record of
    their own + the MARK_GRADS_EARLY slot); the overlapped exchange reduces flat_grads[early_elems:] from that point on.  A stream that
    waits for the mark must see the final values of that range -- and the front range must still be finished exactly once."""
    import mr3d_amd  # noqa: F401
    from mr3d_amd import _hip
    from mr3d_amd.detector import ExtractFeat3D
    dev = torch.device('cuda:0')
    torch.manual_seed(11)
    eng = ExtractFeat3D(device=dev)
    x = torch.randn(1, 3, 16, 64, 64, device=dev)
    cots = [torch.randn_like(o) for o in eng.forward_features(x)]
    e0 = eng.early_elems
    assert eng.stem_elems < e0 < eng.flat_grads.numel()
    names = [n for n, _ in eng.named_parameters()]
    assert eng.rel_offsets['backbone.layer3.0.conv1.weight'] == e0 and any(n.startswith('neck.') for n in names)
    comm = torch.cuda.Stream(device=dev)
    snaps = []
    for _ in range(3):
        eng.zero_grad()
        eng.forward_features(x)
        eng.backward_features(cots)
        assert _hip.plan_stream_wait_mark_checked(eng.MARK_GRADS_EARLY, comm) is True
        with torch.cuda.stream(comm):
            snaps.append(eng.flat_grads[e0:].clone())
        torch.cuda.synchronize()
        assert torch.equal(snaps[-1], eng.flat_grads[e0:]), 'the early bucket changed after its mark'
    # the same gradients as ONE finishing at the end gives (the two wide launches over all layers: MR3D_TUNE has no switch for the
    # plan's records, so compare against a run with the side stream off, where every record runs in order on one queue)
    g_split = eng.flat_grads.clone()
    try:
        _hip.lib().mr3d_plan_set_side_stream(0)
        eng.zero_grad()
        eng.forward_features(x)
        eng.backward_features(cots)
        torch.cuda.synchronize()
    finally:
        _hip.lib().mr3d_plan_set_side_stream(1)
    scale = g_split.abs().max().item()
    assert (eng.flat_grads - g_split).abs().max().item() <= 1e-5 * scale
    assert float(g_split[:e0].abs().sum()) > 0 and float(g_split[e0:].abs().sum()) > 0
