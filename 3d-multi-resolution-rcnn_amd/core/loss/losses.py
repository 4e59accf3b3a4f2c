"""Detection losses (mmdet/core/loss/losses.py) as autograd functions over the fused HIP kernels of
csrc/loss_ops.hip: the forward launch already produces d(loss)/d(prediction); backward scales it by the incoming
gradient.  Same names, argument order and normalisation as the reference; `avg_factor` may additionally be a
device scalar tensor (then nothing synchronises).  Losses keep the reference's shape `[1]`."""
import torch

from ... import _hip


class _FusedLoss(torch.autograd.Function):

    @staticmethod
    def forward(ctx, pred, run):
        loss, grad = run(pred.detach().contiguous(), pred.requires_grad)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


def _cuda_only(pred):
    if not pred.is_cuda:
        raise NotImplementedError('the loss kernels run on the MI355X only (no CPU fallback)')


def weighted_binary_cross_entropy(pred, label, weight, avg_factor=None):
    """losses.py:28-36"""
    _cuda_only(pred)
    if avg_factor is None:
        avg_factor = max(torch.sum(weight > 0).float().item(), 1.)
    elif torch.is_tensor(avg_factor):
        avg_factor = avg_factor.float().reshape(1)
    p2 = pred if pred.dim() == 2 else pred.reshape(-1, 1)
    return _FusedLoss.apply(p2, lambda p, need: _hip.weighted_bce_logits(p, label.reshape(-1), weight.reshape(-1).float(),
                                                                         avg_factor, need))


def smooth_l1_loss(pred, target, beta=1.0, reduction='mean'):
    """losses.py:77-99 ('none' is not offered: the reduction is fused)"""
    assert beta > 0 and pred.size() == target.size() and target.numel() > 0
    _cuda_only(pred)
    avg = float(pred.numel()) if reduction == 'mean' else 1.0
    assert reduction in ('mean', 'sum')
    return _FusedLoss.apply(pred, lambda p, need: _hip.weighted_smoothl1(p, target.contiguous(), torch.ones_like(p), beta, avg,
                                                                         need)).reshape(())


def weighted_smoothl1(pred, target, weight, beta=1.0, avg_factor=None):
    """losses.py:102-118"""
    _cuda_only(pred)
    if avg_factor is None:
        avg_factor = torch.sum(weight > 0).float().item() / 4 + 1e-6
    elif torch.is_tensor(avg_factor):
        avg_factor = avg_factor.float().reshape(1)
    return _FusedLoss.apply(pred, lambda p, need: _hip.weighted_smoothl1(p, target.contiguous().float(),
                                                                         weight.contiguous().float(), beta, avg_factor, need))


def weighted_cross_entropy(pred, label, weight, avg_factor=None, reduce=True, valid_rows=None):
    """losses.py:17-25 (reduce=True; avg_factor=None -> max(#(weight>0), 1), computed on the device)"""
    _cuda_only(pred)
    if not reduce or avg_factor is not None:
        raise NotImplementedError('only the shipped call form (reduce=True, avg_factor=None) is fused')
    return _FusedLoss.apply(pred, lambda p, need: _hip.weighted_softmax_ce(p, label, weight.float(), valid_rows, need)[::2])


def accuracy(pred, target, topk=1, valid_rows=None):
    """losses.py:121-137, top-1 only (the shipped call); returns a [1] tensor in percent"""
    assert topk == 1
    _cuda_only(pred)
    w = torch.ones(pred.size(0), dtype=torch.float32, device=pred.device)
    return _hip.weighted_softmax_ce(pred.detach().contiguous(), target, w, valid_rows, want_grad=False)[1]


def mask_cross_entropy(pred, target, label, valid_rows=None):
    """losses.py:69-74"""
    _cuda_only(pred)
    return _FusedLoss.apply(pred, lambda p, need: _hip.mask_bce(p, target.contiguous().float(), label, valid_rows, need))
