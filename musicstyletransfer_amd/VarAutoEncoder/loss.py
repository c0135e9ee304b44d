"""The three loss blocks with the reference's constructor / call signatures AND call semantics
(VarAutoEncoder/loss.py:4-80), each one HIP kernel over device tensors; all return the per-sample loss [B].

As in the reference, `pred` is what `Model(...)` returns:
  SoftmaxCrossEntropy()(probs, labels)                     probabilities (model.py:256 applies the softmax); loss.py:21 log(pred)
  BinaryCrossEntropy(from_sigmoid=True)(probs, labels)     probabilities
  BinaryCrossEntropy(from_sigmoid=False)(pred, labels)     pre-sigmoid outputs; loss.py:40-42 applies the sigmoid
so `loss(Model(...)[0], labels)` means what it means there. The training step does not go through these classes:
engine.StepPlan fuses softmax / sigmoid, the loss and its gradient over the 16-bit output-layer result in one pass. That
fused form is reachable here as an explicit extra, `pre_activation=True` (pre-softmax input in the 16-bit activation type).
The extra is NOT spelled `from_logits`: in gluon's SoftmaxCrossEntropyLoss, which the reference subclasses (loss.py:15),
from_logits=True means the input already holds LOG-probabilities — a caller passing that flag would silently get another
function, so it is rejected."""
import torch

from .. import ops as o


def _rows(pred):
    """[B, T, V] -> contiguous [B*T, ld] view (a copy only if the input is not already laid out that way)"""
    B, T, V = pred.shape
    p2 = pred.reshape(B * T, V)
    return p2 if p2.is_contiguous() else p2.contiguous()


def _padded_logits(pred):
    """the fused kernels read 16-bit rows whose leading dimension is a multiple of 8 elements; an fp32 input would have to be
    rounded to 16 bits first, which changes the result — the caller decides that, not this helper"""
    B, T, V = pred.shape
    ld = o.roundup(V, 8)
    if pred.dtype not in (torch.bfloat16, torch.float16):
        raise TypeError(f"pre-activation inputs must be bfloat16 or float16 (what the output layer produces), got {pred.dtype}: "
                        "round them explicitly (pred.to(torch.bfloat16)) or pass probabilities")
    if ld == V and pred.is_contiguous():
        return pred.view(B * T, V)
    logits = torch.zeros(B * T, ld, dtype=pred.dtype, device=pred.device)
    logits[:, :V] = pred.reshape(B * T, V)
    return logits


class VariationalKLLoss:
    def __call__(self, z_means, z_vars):
        B, Z = z_means.shape
        kl = torch.zeros(B, dtype=torch.float32, device=z_means.device)
        z = torch.zeros_like(z_means)
        o.reparam_kl_fwd(z_means.float().contiguous(), z_vars.float().contiguous(), torch.zeros_like(z_means), z, kl)
        return kl


class SoftmaxCrossEntropy:
    def __init__(self, axis=-1, batch_axis=0, pre_activation=False, from_logits=False, **unused):
        assert axis in (-1, 2) and batch_axis == 0
        if from_logits:
            raise ValueError("from_logits=True (gluon: the input holds log-probabilities) is not what this path computes; the "
                             "reference calls the loss on probabilities (loss.py:21). For pre-softmax inputs use pre_activation=True")
        self.pre_activation = pre_activation

    def __call__(self, pred, label):
        B, T, V = pred.shape
        loss = torch.zeros(B, dtype=torch.float32, device=pred.device)
        labels = label.to(torch.int32).contiguous().view(-1)
        if self.pre_activation:
            o.softmax_ce(_padded_logits(pred), labels, loss, B, T, V)
        else:
            o.ce_from_probs(_rows(pred), labels, loss, B, T, V)
        return loss


class BinaryCrossEntropy:
    def __init__(self, from_sigmoid=False, label_smoothing=0.0, negative_label_downweighting=True):
        self.from_sigmoid = from_sigmoid
        self.label_smoothing = label_smoothing
        self.negative_label_downweighting = negative_label_downweighting

    def __call__(self, pred, label):
        B, T, P = pred.shape
        loss = torch.zeros(B, dtype=torch.float32, device=pred.device)
        labels = label.to(torch.uint8).contiguous().view(B * T, P)
        if self.from_sigmoid:
            o.bce_from_probs(pred.contiguous(), labels, loss, self.label_smoothing, self.negative_label_downweighting)
        else:
            npos = torch.zeros(B, dtype=torch.int32, device=pred.device)
            o.sigmoid_bce(_padded_logits(pred), labels, loss, B, T, P, label_smoothing=self.label_smoothing,
                          downweight=self.negative_label_downweighting, npos=npos)
        return loss
