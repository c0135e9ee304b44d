"""The three loss blocks with the reference's constructor / call signatures (VarAutoEncoder/loss.py:4-80),
each a single fused HIP kernel over device tensors; all return the per-sample loss [B].

`pred` for SoftmaxCrossEntropy and for BinaryCrossEntropy(from_sigmoid=False) is the PRE-activation
output of the decoder's output layer in 16-bit: the kernels apply softmax / sigmoid themselves (the reference
applies softmax inside the model, model.py:256, and takes log(pred) in the loss; the composition is the
same function). BinaryCrossEntropy(from_sigmoid=True) is not offered: the probabilities are produced, and
the loss computed, in the same pass."""
import torch

from .. import ops as o


class VariationalKLLoss:
    def __call__(self, z_means, z_vars):
        B, Z = z_means.shape
        kl = torch.zeros(B, dtype=torch.float32, device=z_means.device)
        z = torch.zeros_like(z_means)
        o.reparam_kl_fwd(z_means.float().contiguous(), z_vars.float().contiguous(), torch.zeros_like(z_means), z, kl)
        return kl


class SoftmaxCrossEntropy:
    def __init__(self, axis=-1, batch_axis=0):
        assert axis in (-1, 2) and batch_axis == 0

    def __call__(self, pred, label):
        B, T, V = pred.shape
        ld = o.roundup(V, 8)
        logits = torch.zeros(B * T, ld, dtype=pred.dtype, device=pred.device)
        logits[:, :V] = pred.reshape(B * T, V)
        loss = torch.zeros(B, dtype=torch.float32, device=pred.device)
        o.softmax_ce(logits, label.to(torch.int32).contiguous().view(-1), loss, B, T, V)
        return loss


class BinaryCrossEntropy:
    def __init__(self, from_sigmoid=False, label_smoothing=0.0, negative_label_downweighting=True):
        if from_sigmoid:
            raise NotImplementedError("pass pre-sigmoid outputs: the kernel fuses the sigmoid (see module docstring)")
        self.label_smoothing = label_smoothing
        self.negative_label_downweighting = negative_label_downweighting

    def __call__(self, pred, label):
        B, T, P = pred.shape
        ld = o.roundup(P, 8)
        logits = torch.zeros(B * T, ld, dtype=pred.dtype, device=pred.device)
        logits[:, :P] = pred.reshape(B * T, P)
        loss = torch.zeros(B, dtype=torch.float32, device=pred.device)
        npos = torch.zeros(B, dtype=torch.int32, device=pred.device)
        o.sigmoid_bce(logits, label.to(torch.uint8).contiguous().view(B * T, P), loss, B, T, P,
                      label_smoothing=self.label_smoothing, downweight=self.negative_label_downweighting, npos=npos)
        return loss
