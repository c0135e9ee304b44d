"""Trainer with the reference's API (VarAutoEncoder/trainer.py:14-270): OptimizerConfig, TrainConfig,
TrainingState, Trainer(config, context, model, sampler).fit(dataset, model_folder, epochs, validation_dataset)
and ._step(batch, is_train).

_step is the hot path: ONE hipGraph replay runs forward, CE/BCE + kl_weight*KL, backward and the fused MXNet-rule Adam
(engine.StepPlan) on a batch that Trainer.fit's PinnedBatchPipeline packed into a page-locked ring slot and uploaded on a
side stream while the previous step ran; with several ranks the flat gradient bucket is all-reduced over RCCL in two
ranges, the first one under the rest of the backward pass. All five metrics of the reference (ppl, acc, topk, kl_loss,
total_loss) accumulate on the device and are read at log time only (the reference syncs three times per step,
trainer.py:181-186)."""
import os
from time import time

import numpy as np
import torch

from . import utils
from .. import parallel
from ..pianoroll import PinnedBatchPipeline


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
        device = getattr(context, "device", context)
        if torch.device(device).type != "cuda":
            raise RuntimeError("the VarAutoEncoder step runs on hand-written HIP kernels only: no CPU path exists "
                               "(pass --gpu / a HIP device)")
        # THIS rank's GPU becomes the current device before anything touches HIP: the process group binds to it, and
        # every launch goes to torch's current stream of the current device (ops.ptr refuses tensors of another card)
        torch.cuda.set_device(device)
        utils.limit_host_threads()
        self.dist = parallel.init_process_group(self.world, self.rank) if self.world > 1 else None
        self.reduce_fn = parallel.make_grad_allreduce(self.dist) if self.dist is not None else None
        # data parallel: asynchronous two-range all-reduce overlapped with the tail of backward (engine.StepPlan.capture)
        self.reducer = parallel.GradReducer(self.dist) if self.dist is not None else None
        # the step runs (and is captured) on its own stream: the legacy default stream cannot be captured. Everything that
        # touches the model's device state — parameter upload, checkpoint load, metric reads, the sampler — runs under it
        # too, so one stream orders all of it.
        self.stream = torch.cuda.Stream(device=device)
        with torch.cuda.stream(self.stream):
            self._initialize_model()
        self._initialize_optimizers()
        self.stream.synchronize()
        self._graphs = {}
        self._tail_gen = 0
        self.model.on_plan_evicted(self._forget_plan)
        # a failed one-launch position-0 tail (engine.ParamStore.handle_step_status) switches the store to the five-launch form:
        # the graphs recorded with the old kernel sequence are dropped and every shape runs one eager step again (lazy module
        # loads). With more than one rank a skipped update on one rank would let the replicas drift apart: stop instead.
        self.model.store.on_tail_failure(self._on_tail_failure)
        if self.world > 1:
            self.model.store.tail_policy = "raise"
        self.pipeline = PinnedBatchPipeline(self.model.store.device, lambda B, T: self._plan(B, T))
        self.train_state = TrainingState()
        self.summary_writer = utils.ScalarWriter(os.environ.get("MST_LOGDIR", "/tmp/out")) if self.rank == 0 else None
        self._grad_ranges = None

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
    def _plan(self, B, T):
        """the StepPlan of a per-rank batch shape. Every rank uses the SAME step seed; eps is drawn per global sample index
        (sample_offset) so the result does not depend on the sharding, and the dropout site ids are offset per rank so the
        ranks' masks differ."""
        plan = self.model.plan(B, T, global_batch=B * self.world, seed=1000, sample_offset=self.rank * B,
                               site_base=64 * self.rank, **self.hyper)
        if not getattr(plan, "_trainer_set", False):
            for src, dst in (("beta1", "beta1"), ("beta2", "beta2"), ("epsilon", "eps"), ("wd", "wd")):
                if src in self.opt_extra:
                    plan.opt[dst] = self.opt_extra[src]
            plan.track_token_metrics = plan.cfg.kind == "token"  # ppl / acc / topk sums ride on the CE launch
            plan._trainer_set = True
        return plan

    GUARD_POLL_STEPS = 4  # steps between two non-blocking looks at the step guard's status words

    def _poll_guard(self):
        """ParamStore.poll_status every few steps: a failed one-launch tail switches the run to the five-launch form within
        2 x GUARD_POLL_STEPS steps instead of at the next periodic log (each step in between is a skipped batch). Data
        parallel: the words are summed over the ranks first (one 8-byte all-reduce per poll), and every rank stops."""
        with torch.cuda.stream(self.stream):
            self.model.store.poll_status(reduce=(lambda t: self.dist.all_reduce(t)) if self.dist is not None else None)

    def _on_tail_failure(self):
        self._graphs.clear()
        self._tail_gen += 1

    def _forget_plan(self, plan):
        for k in [k for k in self._graphs if k[0] == id(plan)]:
            del self._graphs[k]
        self.pipeline.drop(plan)

    def _shard(self, batch):
        return parallel.shard_bounds(len(batch.data[1]), self.world, self.rank) if self.world > 1 else None

    def _run(self, plan, inbuf, is_train):
        """one step of `plan` reading the device blob `inbuf`: the first step of a shape runs eagerly (HIP modules load
        lazily and are not capturable), every (plan, input blob, mode) gets its own captured graph(s), later steps replay"""
        if plan.inbuf.data_ptr() != inbuf.data_ptr():
            plan.bind_inputs(inbuf)
        key = (id(plan), inbuf.data_ptr(), is_train)
        g = self._graphs.get(key)
        ran = False
        warm = plan.__dict__.setdefault("_warm", set())
        if (is_train, self._tail_gen) not in warm:
            plan.step_kernels(is_train, reduce_fn=self.reduce_fn)
            self.stream.synchronize()
            warm.add((is_train, self._tail_gen))
            ran = True
        if g is None:
            plan.capture(is_train, split_optimizer=self.world > 1, overlap=self.world > 1)
            g = self._graphs[key] = (plan.graph, plan.graph_late, plan.graph_opt)
        if not ran:
            plan.graph, plan.graph_late, plan.graph_opt = g
            plan.run(reduce_fn=self.reduce_fn if is_train else None, reducer=self.reducer if is_train else None)

    def _step(self, batch, is_train=True, staged=None):
        """trainer.py:155-179. `staged`: the batch as PinnedBatchPipeline uploaded it (Trainer.fit); without it the batch
        is copied into the plan's own input buffers here (validation, direct callers)."""
        with torch.cuda.stream(self.stream):
            if staged is not None:
                plan, slot = staged.plan, staged.slot
                self.stream.wait_event(slot.uploaded)
                self._run(plan, slot.dev, is_train)
                slot.consumed.record(self.stream)
            else:
                tokens, seq_lens, classes = batch.data
                labels = batch.label[0]
                shard = self._shard(batch)
                if shard is not None:  # contiguous shard of the global batch; padded length stays global (loss.py:23)
                    lo, hi = shard
                    tokens, seq_lens, classes, labels = tokens[lo:hi], seq_lens[lo:hi], classes[lo:hi], labels[lo:hi]
                plan = self._plan(tokens.shape[0], tokens.shape[1])
                plan.bind_inputs(plan.own_inbuf)
                plan.load_batch(tokens, seq_lens, classes, labels)
                self._run(plan, plan.own_inbuf, is_train)
        if self.config.verbose:
            print("Step {}: batch {} x {}".format(self.train_state.n_batches, plan.B, plan.T))
        self._last_plan = plan

    def fit(self, dataset, model_folder: str, epochs: int, validation_dataset=None):
        start_time = time()
        self.train_state = TrainingState()
        self._load_latest_checkpoint(model_folder)
        self._last_dataset = dataset
        for epoch in range(epochs):
            # the batcher runs one batch ahead: batch i+1 is packed into a page-locked ring slot and uploaded on the
            # pipeline's stream while the captured graph of step i executes (data.py:181-198 -> trainer.py:156-157)
            for staged in self.pipeline.feed(dataset, self._shard if self.world > 1 else None):
                batch = staged.batch
                self._step(batch, staged=staged)
                self.train_state.n_batches += 1
                if self.train_state.n_batches % self.GUARD_POLL_STEPS == 0:
                    self._poll_guard()
                if self.train_state.n_batches % 50 == 0:
                    self._periodic_log(epoch, start_time)
                if self.config.checkpoint_frequency > 0 and self.train_state.n_batches % self.config.checkpoint_frequency == 0:
                    self._checkpoint(model_folder, validation_dataset)
                    if self.train_state.num_checkpoints_not_improved == self.config.num_checkpoints_not_improved:
                        print("Maximum checkpoints not improved reached. Stopping training.")
                        return
                if (self.sampler is not None and self.config.sampling_frequency > 0
                        and self.train_state.n_batches % self.config.sampling_frequency == 0):
                    with torch.cuda.stream(self.stream):  # ordered behind the step that just updated the weights
                        self.sampler.update_parameters(self.model)
                        self.sampler.process_batch(batch, os.path.join(model_folder, "samples/step-{}".format(self.train_state.n_batches)),
                                                   dataset.num_classes())
                if self.config.max_steps and self.train_state.n_batches >= self.config.max_steps:
                    return

    # ------------------------------------------------------------------ metrics / logging
    def collect_metrics(self, reset=True):
        """the reference's five metrics over every step since the last reset (trainer.py:107-120,181-186): kl_loss /
        total_loss batch means and, for the token ends, masked ppl / acc / topk — all accumulated on the device by the
        steps themselves and read here with one synchronisation"""
        failure = None
        with torch.cuda.stream(self.stream):  # ordered behind every step launched so far
            try:
                m = self.model.store.read_metrics(reset)
            except RuntimeError as e:  # (tail_policy "raise": data parallel) — tell the other ranks before stopping
                if self.dist is None:
                    raise
                failure = e
                m = self.model.store.read_metrics(reset)
        keys = [k for k in ("kl_sum", "total_sum", "count", "nll_sum", "acc_hits", "topk_hits", "n_tokens") if k in m]
        if self.dist is not None:
            with torch.cuda.stream(self.stream):
                t = torch.tensor([m[k] for k in keys] + [1.0 if failure is not None else 0.0], dtype=torch.float64,
                                 device=self.model.store.device)
                self.dist.all_reduce(t)
                vals = t.tolist()
                m = dict(zip(keys, vals[:-1]))
            if vals[-1] > 0:
                raise RuntimeError(f"{int(vals[-1])} rank(s) skipped optimizer steps after a failed position-0 tail; the replicas "
                                   "are no longer identical — restart from the last checkpoint with MST_ROW_TAIL=0"
                                   + (f" (this rank: {failure})" if failure is not None else ""))
        out = {}
        if "n_tokens" in m:
            n = m["n_tokens"]
            out["ppl"] = float(np.exp(m["nll_sum"] / n)) if n else float("nan")
            out["acc"] = m["acc_hits"] / n if n else float("nan")
            out["topk"] = m["topk_hits"] / n if n else float("nan")
        n = max(m["count"], 1.0)
        out["kl_loss"] = m["kl_sum"] / n
        out["total_loss"] = m["total_sum"] / n
        return out

    def _metric_to_string_output(self, n_batches):
        """trainer.py:239-248: every metric goes to the scalar log and into the printed line, then is reset"""
        out = ""
        for name, val in self.collect_metrics(reset=True).items():
            if self.summary_writer is not None:
                self.summary_writer.add_scalar(tag=name, value=val, global_step=n_batches)
            out += "{}={:.3f} ".format(name, val)
        return out

    def _periodic_log(self, epoch, start_time):
        line = self._metric_to_string_output(self.train_state.n_batches)
        if self.rank == 0:
            print("Epoch [{}] Batch [{}] updates/sec: {:.2f} {}".format(
                epoch, self.train_state.n_batches, self.train_state.n_batches / (time() - start_time), line), flush=True)
        self._log_gradients()
        if self.summary_writer is not None:
            self.summary_writer.flush()

    def gradient_norms(self):
        """name -> L2 norm of the parameter's gradient as the last step left it (sum over the global batch, before the
        1/B rescale: what grad.grad().norm() is in trainer.py:257-265) — ONE launch over the flat bucket + one read"""
        st = self.model.store
        plan = self._last_plan
        with torch.cuda.stream(self.stream):
            if self._grad_ranges is None:
                r = [[st.offsets[n], st.offsets[n] + int(np.prod(s))] for n, s in st.shapes.items()]
                self._grad_ranges = torch.tensor(r, dtype=torch.int64, device=st.device)
                self._grad_sumsq = torch.zeros(len(r), dtype=torch.float32, device=st.device)
            from .. import ops as o
            o.segment_sumsq(st.g, self._grad_ranges, self._grad_sumsq)
            ss = self._grad_sumsq.cpu().numpy()
        return {n: float(np.sqrt(v)) / (plan.gscale_enc if n.startswith("encoder.") else plan.gscale)
                for n, v in zip(st.shapes, ss)}

    def _log_gradients(self):
        """trainer.py:257-270: one scalar per parameter + their mean as 'global_grad'"""
        if self.summary_writer is None or getattr(self, "_last_plan", None) is None:
            return
        norms = self.gradient_norms()
        for name, v in norms.items():
            self.summary_writer.add_scalar(tag=name, value=v, global_step=self.train_state.n_batches)
        self.summary_writer.add_scalar(tag="global_grad", value=sum(norms.values()) / max(1, len(norms)),
                                       global_step=self.train_state.n_batches)

    # ------------------------------------------------------------------ checkpoint / resume (trainer.py:188-233)
    def _load_latest_checkpoint(self, model_folder):
        print("Looking into folder {} for a valid training.".format(model_folder))
        try:
            latest = utils.get_latest_checkpoint_index(model_folder)
        except (FileNotFoundError, OSError):
            print("No checkpoint was found. Starting training from scratch")
            return
        print("Checkpoint {} found. Resuming training.".format(latest))
        with torch.cuda.stream(self.stream):
            utils.load_model_parameters(self.model, os.path.join(model_folder, "params.{}".format(latest)), self.context)
            self.train_state = utils.load_object(os.path.join(model_folder, "train_state.pkl"))
            opt = os.path.join(model_folder, "optimizer.{}.npz".format(latest))
            if os.path.exists(opt):  # Adam moments + step count (the reference does not save optimizer state)
                st = self.model.store
                with np.load(opt) as z:
                    st.m.copy_(torch.from_numpy(z["m"]))
                    st.v.copy_(torch.from_numpy(z["v"]))
                    st.step_state.copy_(torch.from_numpy(z["step_state"]))
        self.stream.synchronize()

    def _checkpoint(self, model_folder, validation_dataset):
        self.train_state.n_checkpoints += 1
        n = self.train_state.n_checkpoints
        print("\nCheckpoint {} reached.".format(n))
        if self.rank == 0:
            with torch.cuda.stream(self.stream):  # reads ordered behind the last step
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
        vals = self.collect_metrics(reset=True)
        loss = vals["total_loss"]
        if loss < self.train_state.best_resconstruction_loss:
            print("Loss improved from {} to {}.".format(self.train_state.best_resconstruction_loss, loss))
            self.train_state.best_resconstruction_loss = loss
        else:
            self.train_state.num_checkpoints_not_improved += 1
            print("Loss did not improve. {} out {} unsucessful checkpoints".format(
                self.train_state.num_checkpoints_not_improved, self.config.num_checkpoints_not_improved))
            print("Best loss thus far: {}".format(self.train_state.best_resconstruction_loss))
        print("Checkpoint [{}]  {}\n".format(n, " ".join("{}={:.3f}".format(k, v) for k, v in vals.items())))
        self.last_validation = vals
