"""Losses of the reference's training loop on the device: one fused pass over the logits.

`bce_loss` is the `F.binary_cross_entropy_with_logits(logits, labels, pos_weight=pos_weight)` of train.py:144
(get_bce_loss_full), `symmetry_loss` has the signature and value of train.py:103-109.  Both are autograd
functions over gnnome_edge_loss_f32 (include/gnnome_hip.h): the per-edge terms, their deterministic mean and the
gradient with respect to the logits come out of ONE kernel, instead of the ~10 elementwise passes and the
[E]-sized temporaries of the torch expression.
"""
import torch

from . import ops


class _EdgeLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, org, rev, labels, pos_weight, alpha):
        need = org.requires_grad or (rev is not None and rev.requires_grad)
        loss, d_org, d_rev, _ = ops.edge_loss(org.detach().float().contiguous(), None if rev is None else rev.detach().float().contiguous(),
                                             labels.detach().float().contiguous(), pos_weight, alpha, need_grad=need)
        ctx.save_for_backward(d_org, d_rev)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        d_org, d_rev = ctx.saved_tensors
        return (None if d_org is None else g * d_org), (None if d_rev is None else g * d_rev), None, None, None


def bce_loss(logits, labels, pos_weight):
    """mean BCE-with-logits, positive class weighted by pos_weight (train.py:144); logits, labels: [E]."""
    return _EdgeLoss.apply(logits, None, labels, pos_weight, 0.0)


def symmetry_loss(org_scores, rev_scores, labels, pos_weight=1.0, alpha=1.0):
    """mean(bce(org) + bce(rev) + alpha * |org - rev|) - train.py:103-109, same argument order and defaults."""
    return _EdgeLoss.apply(org_scores, rev_scores, labels, pos_weight, float(alpha))
