"""Trainer with the reference's API (VarAutoEncoder/trainer.py:14-270): OptimizerConfig, TrainConfig,
TrainingState, Trainer(config, context, model, sampler).fit(dataset, model_folder, epochs, validation_dataset)
and ._step(batch, is_train).

_step is the hot path: the batch goes pinned-host -> HBM, then ONE hipGraph replay runs forward, CE/BCE +
kl_weight*KL, backward and the fused MXNet-rule Adam (engine.StepPlan); with several ranks the flat gradient
bucket is all-reduced over RCCL between the backward graph and the optimizer graph. Metric sums stay on the
device and are read at log time only (the reference syncs three times per step, trainer.py:181-186)."""
import os
from time import time

import numpy as np
import torch

from . import metrics, utils
from .. import parallel


class OptimizerConfig:
    def __init__(self, optimizer: str, optimizer_params: str, learning_rate: float):
        self.optimizer, self.optimizer_params, self.learning_rate = optimizer, optimizer_params, learning_rate

    def params_to_dict(self):
        """'k1:v1,k2:v2' -> {k1: float(v1), ...}; malformed pairs are skipped (trainer.py:23-35)"""
        out = {}
        for kv in self.optimizer_params.strip().split(","):
            kv = kv.split(":")
            if len(kv) == 2:
                out[str(kv[0])] = float(kv[1])
        return out


class TrainConfig:
    def __init__(self, batch_size: int, sampling_frequency: int, checkpoint_frequency: int, num_checkpoints_not_improved: int,
                 optimizer: OptimizerConfig, kl_loss: float, label_smoothing: float, negative_label_downscaling: bool,
                 verbose: bool, dtype: str = "bf16", max_steps: int = 0):
        self.batch_size = batch_size
        self.sampling_frequency = sampling_frequency
        self.checkpoint_frequency = checkpoint_frequency
        self.num_checkpoints_not_improved = num_checkpoints_not_improved
        self.optimizer = optimizer
        self.kl_loss_weight = kl_loss
        self.label_smoothing = label_smoothing
        self.negative_label_downscaling = negative_label_downscaling
        self.verbose = verbose
        self.dtype = dtype
        self.max_steps = max_steps


class TrainingState:
    def __init__(self):
        self.n_checkpoints = 0
        self.n_batches = 0
        self.num_checkpoints_not_improved = 0
        self.best_resconstruction_loss = np.inf


class Trainer:
    def __init__(self, config: TrainConfig, context, model, sampler=None):
        self.config, self.context, self.model, self.sampler = config, context, model, sampler
        if config.optimizer.optimizer != "adam":
            raise ValueError("only the 'adam' optimizer of scripts/train-vae.sh is implemented")
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.dist = parallel.init_process_group(self.world, self.rank) if self.world > 1 else None
        self.reduce_fn = parallel.make_grad_allreduce(self.dist) if self.dist is not None else None
        # data parallel: asynchronous two-range all-reduce overlapped with the tail of backward (engine.StepPlan.capture)
        self.reducer = parallel.GradReducer(self.dist) if self.dist is not None else None
        self._initialize_model()
        self._initialize_optimizers()
        # the step runs (and is captured) on its own stream: the legacy default stream cannot be captured
        self.stream = torch.cuda.Stream(device=self.model.store.device)
        torch.cuda.synchronize()  # parameter upload ran on the default stream; torch side streams do not wait for it
        self._captured = set()
        self._plans_used = []
        self.train_state = TrainingState()
        self.tokens_metrics = [metrics.Perplexity("ppl"), metrics.Accuracy("acc"), metrics.TopKAccuracy("topk", top_k=5)]

    def _initialize_model(self):
        """model.initialize(mx.init.Xavier(), ctx) (trainer.py:103-105); same seed on every rank"""
        adt = torch.bfloat16 if self.config.dtype == "bf16" else torch.float16
        self.model.initialize(self.context, act_dtype=adt)

    def _initialize_optimizers(self):
        """gluon.Trainer(params, 'adam', {learning_rate, **optimizer_params}) (trainer.py:94-101)"""
        extra = self.config.optimizer.params_to_dict()
        self.hyper = dict(lr=self.config.optimizer.learning_rate, clip_gradient=extra.get("clip_gradient", None),
                          kl_weight=self.config.kl_loss_weight, label_smoothing=self.config.label_smoothing,
                          negative_label_downscaling=self.config.negative_label_downscaling, internal_eps=True)
        self.opt_extra = {k: v for k, v in extra.items() if k in ("beta1", "beta2", "epsilon", "wd")}

    # ------------------------------------------------------------------ the hot loop
    def _plan(self, B, T, is_train):
        plan = self.model.plan(B, T, global_batch=B * self.world, want_probs=True, seed=1000 + self.rank, **self.hyper)
        if self.opt_extra and not getattr(plan, "_opt_set", False):
            for src, dst in (("beta1", "beta1"), ("beta2", "beta2"), ("epsilon", "eps"), ("wd", "wd")):
                if src in self.opt_extra:
                    plan.opt[dst] = self.opt_extra[src]
            plan._opt_set = True
        if plan not in self._plans_used:
            self._plans_used.append(plan)
        return plan

    def _step(self, batch, is_train=True):
        tokens, seq_lens, classes = batch.data
        labels = batch.label[0]
        if self.world > 1:  # contiguous shard of the global batch; padded length stays global (loss.py:23)
            lo, hi = parallel.shard_bounds(len(seq_lens), self.world, self.rank)
            tokens, seq_lens, classes, labels = tokens[lo:hi], seq_lens[lo:hi], classes[lo:hi], labels[lo:hi]
        B, T = tokens.shape[0], tokens.shape[1]
        if self.config.verbose:
            print("Step {}: tokens {}, classes {}, labels {}".format(self.train_state.n_batches, tokens.shape, classes.shape,
                                                                      labels.shape))
        with torch.cuda.stream(self.stream):
            plan = self._plan(B, T, is_train)
            plan.load_batch(tokens, seq_lens, classes, labels)
            key = (id(plan), is_train)
            if key not in self._captured:
                # first step of a shape runs eagerly (HIP modules load lazily), later ones replay its graph
                plan.step_kernels(is_train, reduce_fn=self.reduce_fn)
                if not hasattr(plan, "_graphs"):
                    plan._graphs = {}
                self.stream.synchronize()
                plan.capture(is_train, split_optimizer=self.world > 1, overlap=self.world > 1)
                plan._graphs[is_train] = (plan.graph, plan.graph_late, plan.graph_opt)
                self._captured.add(key)
            else:
                plan.graph, plan.graph_late, plan.graph_opt = plan._graphs[is_train]
                plan.run(reduce_fn=self.reduce_fn if is_train else None, reducer=self.reducer if is_train else None)
        self._last = (plan, labels)

    def fit(self, dataset, model_folder: str, epochs: int, validation_dataset=None):
        start_time = time()
        self.train_state = TrainingState()
        self._load_latest_checkpoint(model_folder)
        for epoch in range(epochs):
            for batch in dataset:
                self._step(batch)
                self.train_state.n_batches += 1
                if self.train_state.n_batches % 50 == 0:
                    self._periodic_log(epoch, start_time)
                if self.config.checkpoint_frequency > 0 and self.train_state.n_batches % self.config.checkpoint_frequency == 0:
                    self._checkpoint(model_folder, validation_dataset)
                    if self.train_state.num_checkpoints_not_improved == self.config.num_checkpoints_not_improved:
                        print("Maximum checkpoints not improved reached. Stopping training.")
                        return
                if (self.sampler is not None and self.config.sampling_frequency > 0
                        and self.train_state.n_batches % self.config.sampling_frequency == 0):
                    self.sampler.update_parameters(self.model)
                    self.sampler.process_batch(batch, os.path.join(model_folder, "samples/step-{}".format(self.train_state.n_batches)),
                                               dataset.num_classes())
                if self.config.max_steps and self.train_state.n_batches >= self.config.max_steps:
                    return

    # ------------------------------------------------------------------ metrics / logging
    def collect_metrics(self, reset=True):
        """kl_loss / total_loss batch means over every step since the last call (trainer.py:115-116,185-186)"""
        kl = tot = n = 0.0
        self.stream.synchronize()
        for plan in self._plans_used:
            acc = plan.metric_acc.cpu().tolist()
            kl, tot, n = kl + acc[0], tot + acc[1], n + acc[2]
            if reset:
                plan.metric_acc.zero_()
        if self.dist is not None:
            t = torch.tensor([kl, tot, n], dtype=torch.float64, device=self.model.store.device)
            self.dist.all_reduce(t)
            kl, tot, n = t.tolist()
        n = max(n, 1.0)
        return {"kl_loss": kl / n, "total_loss": tot / n}

    def _token_metrics_of_last_batch(self):
        plan, labels = self._last
        if plan.cfg.kind != "token" or plan.probs is None:
            return {}
        probs = plan.probs.float().cpu().numpy().reshape(plan.B, plan.T, -1)
        out = {}
        for m in self.tokens_metrics:
            m.reset()
            m.update(np.asarray(labels), probs)
            out[m.get()[0]] = m.get()[1]
        return out

    def _periodic_log(self, epoch, start_time):
        vals = dict(self._token_metrics_of_last_batch(), **self.collect_metrics())
        if self.rank == 0:
            print("Epoch [{}] Batch [{}] updates/sec: {:.2f} {}".format(
                epoch, self.train_state.n_batches, self.train_state.n_batches / (time() - start_time),
                " ".join("{}={:.3f}".format(k, v) for k, v in vals.items())), flush=True)

    # ------------------------------------------------------------------ checkpoint / resume (trainer.py:188-233)
    def _load_latest_checkpoint(self, model_folder):
        print("Looking into folder {} for a valid training.".format(model_folder))
        try:
            latest = utils.get_latest_checkpoint_index(model_folder)
        except (FileNotFoundError, OSError):
            print("No checkpoint was found. Starting training from scratch")
            return
        print("Checkpoint {} found. Resuming training.".format(latest))
        utils.load_model_parameters(self.model, os.path.join(model_folder, "params.{}".format(latest)), self.context)
        self.train_state = utils.load_object(os.path.join(model_folder, "train_state.pkl"))
        opt = os.path.join(model_folder, "optimizer.{}.npz".format(latest))
        if os.path.exists(opt):  # Adam moments + step count (the reference does not save optimizer state)
            st = self.model.store
            with np.load(opt) as z:
                st.m.copy_(torch.from_numpy(z["m"]))
                st.v.copy_(torch.from_numpy(z["v"]))
                st.step_state.copy_(torch.from_numpy(z["step_state"]))
        torch.cuda.synchronize()

    def _checkpoint(self, model_folder, validation_dataset):
        self.stream.synchronize()  # the state below is read on the default stream
        self.train_state.n_checkpoints += 1
        n = self.train_state.n_checkpoints
        print("\nCheckpoint {} reached.".format(n))
        if self.rank == 0:
            utils.create_directory_if_not_present(model_folder)
            utils.save_model(self.model, os.path.join(model_folder, "params.{}".format(n)))
            utils.save_object(self.train_state, os.path.join(model_folder, "train_state.pkl"))
            st = self.model.store
            np.savez(os.path.join(model_folder, "optimizer.{}.npz".format(n)), m=st.m.cpu().numpy(), v=st.v.cpu().numpy(),
                     step_state=st.step_state.cpu().numpy())
        self.collect_metrics(reset=True)
        if validation_dataset is None:
            return
        for batch in validation_dataset:
            self._step(batch, is_train=False)
        loss = self.collect_metrics(reset=True)["total_loss"]
        if loss < self.train_state.best_resconstruction_loss:
            print("Loss improved from {} to {}.".format(self.train_state.best_resconstruction_loss, loss))
            self.train_state.best_resconstruction_loss = loss
        else:
            self.train_state.num_checkpoints_not_improved += 1
            print("Loss did not improve. {} out {} unsucessful checkpoints".format(
                self.train_state.num_checkpoints_not_improved, self.config.num_checkpoints_not_improved))
        print("Checkpoint [{}] total_loss={:.3f}\n".format(n, loss))
