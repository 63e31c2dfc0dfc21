"""Confusion counts and the scores derived from them (utils/metrics.py:6-46), counted on the device.

`calculate_tfpn` replaces four `torch.sum(torch.logical_and(...)).item()` round trips (each a device sync) plus the
rounded-sigmoid temporaries by one pass and ONE 32-byte copy; the arithmetic on the four integers is the reference's.
"""
from . import ops


def calculate_tfpn(edge_predictions, edge_labels):
    """(TP, TN, FP, FN) of round(sigmoid(edge_predictions)) against edge_labels - utils/metrics.py:6-12."""
    logits = edge_predictions.detach().float().reshape(-1).contiguous()
    labels = edge_labels.detach().float().reshape(-1).contiguous()
    _, _, _, counts = ops.edge_loss(logits, None, labels, 1.0, 0.0, need_grad=False, need_counts=True)
    tp, tn, fp, fn = counts.tolist()
    return tp, tn, fp, fn


def _scores(TP, TN, FP, FN):
    precision = TP / (TP + FP) if TP + FP else 0
    recall = TP / (TP + FN) if TP + FN else 0
    f1 = TP / (TP + 0.5 * (FP + FN)) if TP + 0.5 * (FP + FN) else 0
    accuracy = (TP + TN) / (TP + TN + FP + FN)
    return accuracy, precision, recall, f1


def calculate_metrics(TP, TN, FP, FN):
    """(accuracy, precision, recall, f1), zero where the reference catches ZeroDivisionError - utils/metrics.py:15-28."""
    return _scores(TP, TN, FP, FN)


def calculate_metrics_inverse(TP, TN, FP, FN):
    """the same with the negative class as the positive one - utils/metrics.py:31-46."""
    return _scores(TN, TP, FN, FP)
