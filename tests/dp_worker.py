"""One rank of the data-parallel GPU tests (tests/test_parallel_gpu.py). Started by tests/conftest.py at session start —
BEFORE the pytest process touches HIP — as `python tests/dp_worker.py --rank R --world N --port P --out DIR`; the ranks
share the box's one card (MST_FORCE_DEVICE=0) and the all-reduce travels over gloo, so what is exercised is the product's
multi-rank code path (engine.StepPlan's three-graph overlapped schedule + parallel.GradReducer, and Trainer with
WORLD_SIZE > 1), not RCCL itself. Results go to DIR/rank{R}.npz; failures to DIR/rank{R}.err."""
import argparse
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ENGINE_DIMS = ("pianoroll", 32, 32, 2, 16, 32, 2, 2, 32, 1, 2)  # two encoder layers: grad_cut() > 0, three graphs
B_GLOBAL, T_LEN, STEPS, LR, SEED = 8, 16, 3, 1e-3, 5


def engine_params(E, cfg):
    import numpy as np
    params = E.xavier_init(cfg, np.random.default_rng(77))
    Z = cfg.latent_dim
    params["encoder.latent_proj.weight"][Z:] *= 0.25  # sigma off the KL singularity (tests/test_step_gpu.py::_setup)
    params["encoder.latent_proj.bias"][Z:] += 1.5
    return params


def engine_batch(T_LEN=T_LEN):
    import numpy as np
    rng = np.random.default_rng(78)
    roll = (rng.random((B_GLOBAL, T_LEN + 1, ENGINE_DIMS[1])) < 0.1).astype(np.uint8)
    x = roll[:, :T_LEN].copy()
    x[:, 0, :] = 0
    x[:, 0, 0] = 1
    return dict(x=x, labels=roll[:, 1:].copy(), seq_lens=rng.integers(T_LEN // 2, T_LEN + 1, size=B_GLOBAL).astype(np.int64),
                classes=rng.integers(0, 2, size=B_GLOBAL).astype(np.int64))


# the headline configuration's paths under co-tenancy: encoder width 256 with heads of 32 and 6 row blocks (the K|Q|V projection inside
# the attention launch), two encoder layers (the one-launch position-0 tails, whose grid barrier needs 16 workgroups of an
# oversubscribed launch on ONE XCD — with a second process launching the same kernels on the same card)
WIDE_DIMS = ("pianoroll", 32, 32, 2, 16, 256, 2, 8, 64, 1, 4)
WIDE_T = 192


def run_engine(rank, world, reducer=None, reduce_fn=None, wide=False):
    """STEPS training steps of this rank's shard; returns what the parent compares"""
    import torch
    from musicstyletransfer_amd import engine as E
    dev = torch.device("cuda", torch.cuda.current_device())
    dims, T_LEN = (WIDE_DIMS, WIDE_T) if wide else (ENGINE_DIMS, globals()["T_LEN"])
    cfg = E.VAEConfig(*dims)
    store = E.ParamStore(cfg, dev, torch.bfloat16, params_np=engine_params(E, cfg))
    store.tail_policy = "raise"
    b = engine_batch(T_LEN)
    per = B_GLOBAL // world
    lo, hi = rank * per, (rank + 1) * per
    plan = E.StepPlan(store, per, T_LEN, lr=LR, clip_gradient=1.0, global_batch=B_GLOBAL, internal_eps=True, seed=SEED,
                      sample_offset=lo, site_base=64 * rank)
    plan.load_batch(b["x"][lo:hi], b["seq_lens"][lo:hi], b["classes"][lo:hi], b["labels"][lo:hi])
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        plan.step_kernels(True, reduce_fn=reduce_fn)  # first step of a shape: eager
        st.synchronize()
        g1, eps1, tot1 = store.g.cpu().numpy().copy(), plan.eps.cpu().numpy().copy(), plan.total.cpu().numpy().copy()
        plan.capture(True, split_optimizer=world > 1, overlap=world > 1)
        for _ in range(STEPS - 1):
            plan.run(reduce_fn=reduce_fn, reducer=reducer)
        st.synchronize()
    status = store.read_metrics(reset=False)  # (raises if a step was skipped: tail_policy)
    return dict(w=store.w.cpu().numpy(), g1=g1, eps1=eps1, total1=tot1, total_last=plan.total.cpu().numpy(),
                steps=int(store.step_state[0].item()), three_graphs=int(plan.graph_late is not None),
                tails=int(plan._tail_used["fwd"]) + int(plan._tail_used["bwd"]), tail_fused=int(store.tail_fused), skipped=int(status["skipped_steps"]))


def trainer_setup():
    from music_style_transfer.VarAutoEncoder import model, trainer
    from music_style_transfer.VarAutoEncoder.transformer import TransformerConfig
    cfg = model.ModelConfig(
        model.EncoderConfig(TransformerConfig(32, 0.0, 2, 2, 32), 16, 2, 32),
        model.DecoderConfig(TransformerConfig(32, 0.0, 1, 2, 32), 16, 2, 32), kind="pianoroll")
    tc = trainer.TrainConfig(batch_size=B_GLOBAL, sampling_frequency=0, checkpoint_frequency=0, num_checkpoints_not_improved=-1,
                             optimizer=trainer.OptimizerConfig("adam", "clip_gradient:1.0", LR), kl_loss=1.0, label_smoothing=0.0,
                             negative_label_downscaling=False, verbose=False)
    return model.Model(cfg), tc


def trainer_batches():
    from musicstyletransfer_amd.pianoroll import SyntheticPianoRollDataset
    return list(SyntheticPianoRollDataset(B_GLOBAL, T_LEN, 3 * B_GLOBAL, n_pitches=32, density=0.1, seed=9))


def run_trainer():
    """Trainer._step on three global batches (WORLD_SIZE / RANK from the environment)"""
    import torch
    from music_style_transfer.VarAutoEncoder import trainer
    from music_style_transfer.VarAutoEncoder.utils import gpu
    m, tc = trainer_setup()
    t = trainer.Trainer(tc, gpu(), m, None)
    # sigma off the KL singularity, identically on every rank
    with torch.cuda.stream(t.stream):
        p = m.store.to_numpy("w")
        p["encoder.latent_proj.weight"][16:] *= 0.25
        p["encoder.latent_proj.bias"][16:] += 1.5
        m.store.load_numpy(p)
    batches = trainer_batches()
    t._step(batches[0])                      # direct call: the plan's own input buffers
    for staged in t.pipeline.feed(batches[1:], t._shard if t.world > 1 else None):  # the pinned ring, as Trainer.fit
        t._step(staged.batch, staged=staged)
    metrics = t.collect_metrics()
    t.stream.synchronize()
    return dict(w=m.store.w.cpu().numpy(), total_loss=metrics["total_loss"], kl_loss=metrics["kl_loss"],
                steps=int(m.store.step_state[0].item()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world),
                      LOCAL_RANK=str(a.rank), MST_FORCE_DEVICE="0", MST_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        import numpy as np
        import torch
        torch.cuda.set_device(0)
        from musicstyletransfer_amd import parallel
        dist = parallel.init_process_group(a.world, a.rank)
        eng = run_engine(a.rank, a.world, reducer=parallel.GradReducer(dist), reduce_fn=parallel.make_grad_allreduce(dist))
        wide = run_engine(a.rank, a.world, reducer=parallel.GradReducer(dist), reduce_fn=parallel.make_grad_allreduce(dist), wide=True)
        tr = run_trainer()
        np.savez(os.path.join(a.out, f"rank{a.rank}.npz"), **{"eng_" + k: v for k, v in eng.items()},
                 **{"wide_" + k: v for k, v in wide.items()}, **{"tr_" + k: v for k, v in tr.items()})
        dist.barrier()
        dist.destroy_process_group()
    except BaseException:
        with open(os.path.join(a.out, f"rank{a.rank}.err"), "w") as f:
            f.write(traceback.format_exc())
        raise


if __name__ == "__main__":
    main()
