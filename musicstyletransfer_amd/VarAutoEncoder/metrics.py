"""Masked token metrics (reference VarAutoEncoder/metrics.py:1-73), on numpy. They are reporting-only and
computed at log time from one batch's probabilities; the ELBO metrics (kl_loss, total_loss) are accumulated
on the device by the step itself (engine.StepPlan.metric_acc)."""
import numpy as np


class _Masked:
    def __init__(self, name, ignore_label=0):
        self.name, self.ignore_label = name, ignore_label
        self.reset()

    def reset(self):
        self.sum_metric, self.num_inst = 0.0, 0

    def get(self):
        return self.name, (self.sum_metric / self.num_inst if self.num_inst else float("nan"))


class Accuracy(_Masked):
    def __init__(self, name="acc", axis=2, ignore_label=0):
        super().__init__(name, ignore_label)
        self.axis = axis

    def update(self, labels, probs):
        pred = probs.argmax(axis=self.axis)
        keep = labels != self.ignore_label
        self.sum_metric += float(((pred == labels) & keep).sum())
        self.num_inst += int(keep.sum())


class TopKAccuracy(_Masked):
    def __init__(self, name="topk", top_k=5, ignore_label=0):
        super().__init__(name, ignore_label)
        self.top_k = top_k

    def update(self, labels, probs):
        k = min(self.top_k, probs.shape[-1])
        top = np.argpartition(-probs, k - 1, axis=-1)[..., :k]
        keep = labels != self.ignore_label
        hit = (top == labels[..., None]).any(axis=-1)
        self.sum_metric += float((hit & keep).sum())
        self.num_inst += int(keep.sum())


class Perplexity(_Masked):
    def __init__(self, name="ppl", ignore_label=0):
        super().__init__(name, ignore_label)

    def update(self, labels, probs):
        keep = labels != self.ignore_label
        p = np.take_along_axis(probs, labels[..., None].astype(np.int64), axis=-1)[..., 0]
        self.sum_metric += float(-np.log(np.maximum(p[keep], 1e-10)).sum())
        self.num_inst += int(keep.sum())

    def get(self):
        return self.name, (float(np.exp(self.sum_metric / self.num_inst)) if self.num_inst else float("nan"))
