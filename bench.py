#!/usr/bin/env python3
"""bench.py — VarAutoEncoder training-step throughput on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {1,2,4}] [--data {resident,host}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full training step (forward, BCE + KL, backward, gradient all-reduce when N > 1, MXNet-rule Adam) over
one batch of synthetic piano-rolls. --config names the BASELINE.json configuration (per GPU; weak scaling):
    1  configs[1]  single-track, T 256, 128 pitches, latent 64, batch 64, bf16        (default; on 8 GPUs = configs[3])
    2  configs[2]  multi-instrument, 16 x 128 pitches, latent 256, batch 64, bf16
    4  configs[4]  long sequence T 1024, 128 pitches, fp16, batch 32 per GPU (256 over 8 GPUs)
widths from scripts/train-vae.sh (encoder 256 x 2 layers x 8 heads, decoder 128 x 1 layer x 8 heads, dropout 0.2).
--data resident (default): the batches are in HBM before the timed region (`value` is defined on this);
--data host: every step's batch travels pinned host -> HBM through PinnedBatchPipeline inside the timed region.
Rank 0 prints ONE JSON line. `value` follows the contract (K steps between barrier + synchronize, max over ranks);
`ms_per_step_median` is the median of the same K steps timed one by one with HIP events on the step's stream.

The line also carries
  roofline     : the step's heavy kernel families, each re-launched on the step's own operands and timed with HIP
                 events on its stream (`families`), and the one with the largest share of the step as the headline:
                 algorithmic bytes (or FLOPs) per launch / average launch duration against the gfx950 peak
                 (profiles/ holds the rocprofv3 summary of the same command)
  cpu_baseline : the CPU oracle (oracle/vae_oracle.py, a restatement of the reference — the MXNet reference itself
                 cannot run here) timed on this node's host cores on a bounded sample of the same workload, same
                 dropout. A reported baseline, not the optimisation target.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WIDTHS = dict(num_classes=2, e_model=256, e_layers=2, e_heads=8, d_model=128, d_layers=1, d_heads=8)  # scripts/train-vae.sh:22-28
CONFIGS = {
    1: dict(name="BASELINE configs[1]: single-track piano-roll VAE train step", P=128, Z=64, B=64, T=256, dtype="bf16"),
    2: dict(name="BASELINE configs[2]: multi-instrument piano-roll (16 tracks x 128 pitches) VAE train step", P=2048, Z=256, B=64,
            T=256, dtype="bf16"),
    4: dict(name="BASELINE configs[4]: long-sequence piano-roll VAE train step (per-GPU share of batch 256 on 8 GPUs)", P=128, Z=64,
            B=32, T=1024, dtype="fp16"),
}
DROPOUT = 0.2  # scripts/train-vae.sh:23,29
HOST_WARMUP = 48  # --data host: untimed steps before the timed region (see timed_run)
PEAK_MFMA_TFLOPS = 2500.0  # dense bf16/fp16 MFMA, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def model_dims(c):
    return dict(kind="pianoroll", in_dim=c["P"], out_dim=c["P"], latent_dim=c["Z"], **WIDTHS)


def fwd_flops(cfg, B, T):
    """SURVEY §8d algorithmic FLOPs of one forward pass; a step is 3x this."""
    De, Dd, Z, P = cfg["e_model"], cfg["d_model"], cfg["latent_dim"], cfg["in_dim"]
    Le, Ld, V = cfg["e_layers"], cfg["d_layers"], cfg["out_dim"]
    layer = lambda S, D: 24 * S * D * D + 4 * S * S * D
    per = (Le * layer(T, De) + 2 * T * P * De + 4 * De * Z + 2 * Z * Dd + Ld * layer(T + 1, Dd) + 2 * T * Dd * V)
    return B * per


def executed_fwd_flops(cfg, B, T):
    """what the forward pass actually executes: the top encoder layer is read at position 0 only (model.py:97), so after
    its dense K/Q/V projection and the key-row statistics (which run over every query) its attention output, W_proj, FFN
    and LayerNorms are computed for ONE row per sample (engine._top_encoder_layer_fwd)"""
    De = cfg["e_model"]
    full = fwd_flops(cfg, B, T)
    layer = 24 * T * De * De + 4 * T * T * De
    top = 6 * T * De * De + 2 * T * T * De + 2 * T * De + 18 * De * De  # QKV, K Q^T, P^T V for query 0, W_proj + FFN on one row
    return full - B * (layer - top)


def synthetic_batches(n, B, T, P, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        roll = (rng.random((B, T + 1, P)) < 0.04).astype(np.uint8)  # ~5 active pitches of 128 per frame
        x = roll[:, :T].copy()
        x[:, 0, :] = 0
        x[:, 0, 0] = 1  # reserved start row
        out.append(dict(x=x, labels=roll[:, 1:].copy(), seq_lens=np.full(B, T, np.int32),
                        classes=rng.integers(0, 2, size=B).astype(np.int32)))
    return out


def cpu_baseline(c, dropout, budget_s=20.0):
    """time the CPU oracle on a bounded sample: training steps of one batch of this configuration's shape, same dropout
    (explicit keep masks, drawn once), until ~budget_s of CPU work"""
    import torch
    from oracle import vae_oracle as O
    # the GPU box gives one GPU's share of the host (a cgroup quota of 16 cores; asking torch for every core the kernel
    # reports oversubscribes that share and runs ~50x slower)
    from musicstyletransfer_amd.VarAutoEncoder.utils import host_cpu_share
    cores = max(1, min(host_cpu_share(), 16))
    torch.set_num_threads(cores)
    md = model_dims(c)
    cfg = O.OracleConfig(md["kind"], md["in_dim"], md["out_dim"], md["num_classes"], md["latent_dim"], md["e_model"], md["e_layers"],
                         md["e_heads"], md["d_model"], md["d_layers"], md["d_heads"])
    B, T = c["B"], c["T"]
    if c["T"] > 256:
        B = 4  # T = 1024: the oracle materialises [B, H, S, S] attention tensors for autograd; a 4-sample slice of the batch
    rng = np.random.default_rng(1234)
    tr = O.OracleTrainer(cfg, O.init_params(cfg, rng), lr=3e-4, clip_gradient=1.0)
    batch = O.synthetic_pianoroll_batch(rng, B, T, md["in_dim"])
    eps = torch.from_numpy(rng.standard_normal((B, md["latent_dim"])).astype(np.float32))
    masks = None
    if dropout > 0:
        masks = {}
        keep = lambda *shape: torch.from_numpy((rng.random(shape) >= dropout).astype(np.float32) / (1.0 - dropout))
        for side, n_l, D, S in (("encoder", md["e_layers"], md["e_model"], T), ("decoder", md["d_layers"], md["d_model"], T + 1)):
            for i in range(n_l):
                p = f"{side}.layer{i}"
                masks[f"{p}.att"], masks[f"{p}.ffh"], masks[f"{p}.ffo"] = keep(B, S, D), keep(B, S, 4 * D), keep(B, S, D)
    tr.step(batch, eps, masks)  # warm-up (thread pool, allocator)
    t0 = time.perf_counter()
    steps = 0
    while steps < 3 or (time.perf_counter() - t0 < budget_s and steps < 50):
        tr.step(batch, eps, masks)
        steps += 1
    dt = (time.perf_counter() - t0) / steps
    return {"value": B * T / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{steps} training steps of one B={B}, T={T}, P={md['in_dim']} batch, fp32 torch-CPU restatement of the reference "
                      f"(not MXNet), dropout {dropout}, {dt * 1e3:.0f} ms/step"}


def time_launch(o, fn, iters, reps=5):
    """average duration (ms) of one launch of `fn`, HIP events on the launch stream around a captured graph of `iters` back-to-back
    launches — as the step itself runs them —, median of `reps` replays: eager launches measured the host as well (one stall of
    the Python thread between two launches once put 1.9 ms on a 0.05 ms kernel)"""
    import torch
    fn()
    torch.cuda.synchronize()
    g = o.Graph().capture(lambda: [fn() for _ in range(iters)])
    g.launch()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        e0, e1 = o.Event(), o.Event()
        e0.record()
        g.launch()
        e1.record()
        e1.sync()
        times.append(e0.elapsed_ms(e1) / iters)
    return sorted(times)[len(times) // 2]


def kernel_families(plan, o, iters=30):
    """Re-launch the step's heavy kernels on the step's own buffers, each timed with HIP events on the launch stream.
    launches_per_step follows from the layer counts (the top encoder layer runs its row-wise part on B rows and is not
    counted); algorithmic bytes = compulsory operand + result traffic of one launch, FLOPs = 2 M N K per contraction."""
    st, cfg = plan.store, plan.cfg
    De, Dd, H = cfg.e_model, cfg.d_model, cfg.e_heads
    M, B, T = plan.Me, plan.B, plan.T
    L, t = plan.enc[0], plan.be_l[0]
    pre = "encoder.layer0"
    fams = []
    full_e = max(cfg.e_layers - 1, 0)  # encoder layers that run at full width in every row
    if full_e and o.ffn_fusion_pays(De, 4 * De):
        ffn_bytes = 2.0 * (M * De + 8 * De * De + M * 4 * De + 2 * M * De) + 4.0 * (4 * De + 3 * De + 2 * M)
        fams.append(dict(
            kernel=f"ffn_ln_kernel fwd [FFN + LayerNorm, M={M}, {De} -> {4 * De} -> {De}]",
            fn=lambda: o.ffn_ln_fwd(L.x1, st.h(f"{pre}.ff1.weight"), L.a, st.h(f"{pre}.ff2.weight"), L.h2, st.p(f"{pre}.ln2.gamma"),
                                    st.p(f"{pre}.ln2.beta"), L.x2, L.mean2, L.rstd2,
                                    ff1=dict(K=De, bias=st.p(f"{pre}.ff1.bias"), act=o.ACT_RELU, **plan._drop(cfg.e_dropout, 1)),
                                    ff2=dict(K=4 * De, bias=st.p(f"{pre}.ff2.bias"), resid=L.x1, **plan._drop(cfg.e_dropout, 2))),
            flops=2.0 * 2.0 * M * 4 * De * De, bytes=ffn_bytes, launches_per_step=full_e,
            pmc_key="ffn_ln_kernel<%d,2,4,1,@%d" % (De, ((M + 63) // 64) * 512)))
        # backward form: dff in, a (gate) in, dpre out, h1 in, dh1 (+ masked copy) out
        fams.append(dict(
            kernel=f"ffn_ln_kernel bwd [FFN dgrads + LayerNorm backward, M={M}, width {De}]",
            fn=lambda: o.ffn_ln_bwd(t.dh, st.t(f"{pre}.ff2.weight"), t.dpre, L.a, st.t(f"{pre}.ff1.weight"), t.dh1, L.h1,
                                    st.p(f"{pre}.ln1.gamma"), L.mean1, L.rstd1, st.grad(f"{pre}.ln1.gamma"), st.grad(f"{pre}.ln1.beta"),
                                    resid=t.dh, partials=plan._ln_part[f"{pre}.ln1"]),
            flops=2.0 * 2.0 * M * 4 * De * De, bytes=2.0 * (M * De + 8 * De * De + 2 * M * 4 * De + 3 * M * De),
            launches_per_step=full_e, pmc_key="ffn_ln_kernel<%d,2,4,2,@%d" % (De, ((M + 63) // 64) * 512)))
    # ---- attention: all six launches of the step (VERDICT r03 #4c: the family owns ~29 % of the kernel time)
    # encoder layers below the top one: the K | Q | V projection runs inside the forward launch (mst_attn_qkv_fwd): x and the weights
    # in, qkv (kept for the backward pass) and the attention output out
    Hd, Sd, Md = cfg.d_heads, T + 1, plan.Md
    qkv_flops = lambda M_, D_: 2.0 * M_ * 3 * D_ * D_
    # PMC keys (tools/make_profile_summary.py names; alternatives '|' by the form the library picks for the shape): resident kernels at
    # T <= 512, beyond that the projection as a GEMM + the chunked forward, the streaming dV / dK kernel + the chunked dQ kernel
    dh_e = De // H
    fwd_key = f"attn_fwd_res_kernel<{dh_e},1,0>@*|gemm_nt_kernel<128,128,2,2,64>@{(M // 128) * (3 * De // 128) * 256}+attn_fwd_res_kernel<{dh_e},0,1>@*"
    bwd_key = f"attn_bwd_res_kernel<{dh_e},%(sp)d>@*|attn_bwd_kv_kernel<{dh_e},%(sp)d>@*+attn_bwd_q_chunk_kernel<{dh_e},4>@*"
    if full_e:
        fams.append(dict(
            kernel=f"attention fwd with the K,Q,V projection inside [B*H={B * H}, S={T}, dh={De // H}, D={De}]",
            fn=lambda: o.attn_qkv_fwd(plan.x0_e, st.fused(st.w16, pre, "weight"), st.fused(st.w, pre, "bias"), L.qkv, plan.keymask_e, L.lse, L.att,
                                      B, T, H, De // H, 0, De, 2 * De),
            flops=4.0 * B * T * T * De + qkv_flops(M, De), bytes=2.0 * (M * De + 3 * De * De + M * 3 * De + M * De),
            launches_per_step=full_e, pmc_key=fwd_key))
        fams.append(dict(
            kernel=f"attention bwd [B*H={B * H}, S={T}, dh={De // H}]",
            fn=lambda: o.attn_bwd(L.qkv, plan.keymask_e, L.lse, t.datt, t.dqkv, t.delta, B, T, H, De // H, 0, De, 2 * De),
            flops=2.0 * 4 * B * T * T * De, bytes=2.0 * M * (3 * De + De + 3 * De), launches_per_step=full_e,
            pmc_key=bwd_key % dict(sp=0)))
    if cfg.e_layers >= 1:
        # the top encoder layer is read at position 0 only (model.py:97): projection and key-row statistics over every query, the
        # output for ONE query per sample; backward with dO zero beyond it. Algorithmic work = the reference's dense layer.
        Lt, tt, pt = plan.enc[-1], plan.be_l[-1], f"encoder.layer{cfg.e_layers - 1}"
        x_top = plan.enc[-2].x2 if cfg.e_layers > 1 else plan.x0_e
        fams.append(dict(
            kernel=f"attention fwd, top encoder layer (projection inside, one query per sample out) [B*H={B * H}, S={T}, dh={De // H}]",
            fn=lambda: o.attn_qkv_fwd(x_top, st.fused(st.w16, pt, "weight"), st.fused(st.w, pt, "bias"), Lt.qkv, plan.keymask_e, Lt.lse, Lt.att,
                                      B, T, H, De // H, 0, De, 2 * De, q_limit=1),
            flops=4.0 * B * T * T * De + qkv_flops(M, De), bytes=2.0 * (M * De + 3 * De * De + M * 3 * De + B * De),
            launches_per_step=1, pmc_key=fwd_key))
        fams.append(dict(
            kernel=f"attention bwd, top encoder layer (dO zero beyond query 0) [B*H={B * H}, S={T}, dh={De // H}]",
            fn=lambda: o.attn_bwd(Lt.qkv, plan.keymask_e, Lt.lse, plan.sp_datt, tt.dqkv, tt.delta, B, T, H, De // H, 0, De, 2 * De, q_limit=1),
            flops=2.0 * 4 * B * T * T * De, bytes=2.0 * (M * 3 * De + B * De + M * 3 * De), launches_per_step=1,
            pmc_key=bwd_key % dict(sp=1)))
    if cfg.d_layers >= 1:
        Ld, td = plan.dec[0], plan.bd_l[0]
        fams.append(dict(
            kernel=f"attention fwd, decoder [B*H={B * Hd}, S={Sd}, dh={Dd // Hd}] (its projection: riders of the forward tail, or a GEMM launch)",
            fn=lambda: o.attn_fwd(Ld.qkv, plan.keymask_d, Ld.lse, Ld.att, B, Sd, Hd, Dd // Hd, 0, Dd, 2 * Dd),
            flops=4.0 * B * Sd * Sd * Dd, bytes=2.0 * (Md * 3 * Dd + Md * Dd), launches_per_step=cfg.d_layers,
            pmc_key="attn_fwd_res_kernel<%d,0,0>@*" % (Dd // Hd)))
        fams.append(dict(
            kernel=f"attention bwd, decoder [B*H={B * Hd}, S={Sd}, dh={Dd // Hd}]",
            fn=lambda: o.attn_bwd(Ld.qkv, plan.keymask_d, Ld.lse, td.datt, td.dqkv, td.delta, B, Sd, Hd, Dd // Hd, 0, Dd, 2 * Dd),
            flops=2.0 * 4 * B * Sd * Sd * Dd, bytes=2.0 * Md * (3 * Dd + Dd + 3 * Dd), launches_per_step=cfg.d_layers,
            pmc_key="attn_bwd_res_kernel<%d,0>@*" % (Dd // Hd)))
    wg, ps = plan.last_wgrad_launch
    if wg:
        fl = sum(2.0 * w.M * w.N * w.K for w in wg)
        by = sum(2.0 * w.M * (w.N + w.K) + 4.0 * w.N * w.K for w in wg)
        fams.append(dict(
            kernel=f"wgrad batch [the step's {len(wg)} weight-gradient problems in one launch + reduction pass]",
            fn=lambda: o.gemm_wgrad_batch(wg, scratch=plan.wgrad_scratch, sums=ps), flops=fl, bytes=by, launches_per_step=1,
            pmc_key="wgrad_kernel<256,256,4,2>@*+wgrad_reduce_kernel<256, 256>@*"))  # (both passes of the launch; any grid)
    ridge = PEAK_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
    out = []
    for f in fams:
        ms = time_launch(o, f["fn"], iters)
        tflops, gbs = f["flops"] / (ms * 1e-3) / 1e12, f["bytes"] / (ms * 1e-3) / 1e9
        hbm = f["flops"] / f["bytes"] < ridge
        out.append(dict(kernel=f["kernel"], avg_launch_ms=ms, launches_per_step=f["launches_per_step"], share_ms=ms * f["launches_per_step"],
                        bound="hbm" if hbm else "mfma", algorithmic_bytes_per_launch=f["bytes"], algorithmic_flops_per_launch=f["flops"],
                        gbs=gbs, tflops=tflops, hbm_frac=gbs / PEAK_HBM_GBS, mfma_frac=tflops / PEAK_MFMA_TFLOPS,
                        frac=(gbs / PEAK_HBM_GBS) if hbm else (tflops / PEAK_MFMA_TFLOPS), pmc_key=f["pmc_key"]))
    return out, ridge


def roofline(plan, o, config_id):
    fams, ridge = kernel_families(plan, o)
    best = max(fams, key=lambda f: f["share_ms"])
    traffic, src = None, None
    try:  # HBM bytes per launch from the committed PMC passes (tools/make_profile_summary.py), not measured here
        with open(os.path.join(ROOT, "profiles", "roofline_traffic.json")) as f:
            j = json.load(f)
        if str(config_id) in j.get("configs", {}):  # (per BASELINE config: tools/profile_round.sh <config>)
            j = j["configs"][str(config_id)]
        elif config_id != 1:
            j = {}
        if best["pmc_key"] and "bytes_per_launch" in j:
            table, src = j["bytes_per_launch"], j.get("source")

            def lookup(key):  # "name<args>@grid"; "@*": whatever the grid; a template list may be a prefix ("kernel<256,2,4,1")
                if key in table:
                    return table[key]
                name, grid = key.rsplit("@", 1)
                stem = name[:-1] if name.endswith(">") else name
                hits = [v for k, v in table.items() if k.rsplit("@", 1)[0].startswith(stem) and (grid == "*" or k.endswith("@" + grid))]
                return max(hits) if hits else None

            for alt in best["pmc_key"].split("|"):  # alternatives (the library picks the kernel form by shape): the first one found whole
                parts = [lookup(k.strip()) for k in alt.split("+")]
                if all(v is not None for v in parts):
                    traffic = sum(parts)
                    break
    except (OSError, ValueError, KeyError):
        pass
    hbm = best["bound"] == "hbm"
    return {"bound": best["bound"], "kernel": best["kernel"], "achieved": best["gbs"] if hbm else best["tflops"],
            "peak": PEAK_HBM_GBS if hbm else PEAK_MFMA_TFLOPS, "unit": "GB/s" if hbm else "TFLOP/s", "frac": best["frac"],
            "traffic": traffic, "traffic_source": src, "avg_launch_ms": best["avg_launch_ms"],
            "algorithmic_bytes_per_launch": best["algorithmic_bytes_per_launch"],
            "algorithmic_flops_per_launch": best["algorithmic_flops_per_launch"], "tflops": best["tflops"], "mfma_frac": best["mfma_frac"],
            "intensity_flop_per_byte": best["algorithmic_flops_per_launch"] / best["algorithmic_bytes_per_launch"],
            "ridge_flop_per_byte": ridge,
            "attention_share_ms": sum(f["share_ms"] for f in fams if f["kernel"].startswith("attention")),
            "attention_launches_per_step": sum(f["launches_per_step"] for f in fams if f["kernel"].startswith("attention")),
            "families": [{k: v for k, v in f.items() if k != "pmc_key"} for f in sorted(fams, key=lambda f: -f["share_ms"])]}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=1)
    ap.add_argument("--data", choices=["resident", "host"], default="resident")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dropout", type=float, default=DROPOUT)
    ap.add_argument("--dtype", choices=["bf16", "fp16"], default=None, help="default: the configuration's own")
    ap.add_argument("--ring-slots", type=int, default=None, help="--data host: slots of the pinned ring (default: MST_RING_SLOTS or the pipeline's own)")
    ap.add_argument("--decode", action="store_true",
                    help="instead of the training step: beam-search decoding with the per-layer K|Q|V caches (SURVEY §8f rank 4), "
                         "token ends at scripts/train-vae.sh's widths, batch 64 x beam 4 hypotheses; prints ONE JSON line (tokens/s)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="with --gpus N > 1 and no launcher: start the N rank processes, have each print its rendezvous environment "
                         "as one JSON line and exit (no GPU, no torch): a test of the launcher itself")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds before the self-launcher gives up on its ranks")
    return ap.parse_args(argv)


RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`python bench.py --gpus N` started WITHOUT a launcher (no WORLD_SIZE in the environment): this process becomes the
    launcher. It makes no GPU call and imports nothing that could (a process that has initialised HIP must not start
    other programs on this pool): it starts N copies of this script as child processes — one rank per GPU, the same
    environment torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free
    MASTER_PORT) —, relays rank 0's stdout (the ONE JSON line) and exits non-zero if any rank does. When one rank fails
    the others are stopped — exactly the processes started here, by PID."""
    n = args.gpus
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MST_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        # rank 0's stdout is the bench line; the other ranks have nothing to say on stdout: send it to our stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)  # drain rank 0's pipe as it fills
    reader.start()
    deadline = time.monotonic() + args.launch_timeout
    rc, pending = 0, set(range(n))
    try:
        while pending and rc == 0:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py launcher: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
            if rc == 0 and pending:
                if time.monotonic() > deadline:
                    rc = 124
                    print(f"bench.py launcher: ranks {sorted(pending)} still running after {args.launch_timeout:.0f} s", file=sys.stderr)
                else:
                    time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    out0 = b"".join(c for c in chunks if c)
    text = (out0 or b"").decode(errors="replace")
    if rc == 0:
        sys.stdout.write(text)
        sys.stdout.flush()
    else:
        sys.stderr.write(text)
    return rc


def dry_rank():
    """--dry-launch inside a rank: print the rendezvous environment, touch nothing else. (MST_BENCH_DRY_FAIL_RANK /
    MST_BENCH_DRY_SLEEP: the launcher's own tests make one rank fail while the others are still running.)"""
    if os.environ.get("MST_BENCH_DRY_FAIL_RANK") == os.environ.get("RANK"):
        return 3
    time.sleep(float(os.environ.get("MST_BENCH_DRY_SLEEP", "0")))
    print(json.dumps({k: os.environ.get(k) for k in RANK_ENV + ("HSA_ENABLE_IPC_MODE_LEGACY", "MST_BENCH_SELF_LAUNCHED")}), flush=True)
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    if args.dry_launch:
        return dry_rank()
    if args.decode:
        return run_decode(args)
    return run_rank(args)


def run_decode(args):
    """Beam-search decoding throughput (reference sampler.py:198-257 over model.py:259-272): B = 64 melodies x beam 4 = 256
    hypotheses, each position = one captured graph of the decoder's incremental step, the ranking of the beam x V continuations
    and the gather of the caches (decode.BeamSearch). A "step" here is one decoded position of all hypotheses; --steps positions are timed after
    --warmup positions (which also capture the graphs). Random-initialised weights, synthetic token batch."""
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the decode step has no CPU fallback")
    torch.cuda.set_device(0)
    from musicstyletransfer_amd import ops as o
    from musicstyletransfer_amd.MIDIUtil.defaults import NUM_EVENTS
    from musicstyletransfer_amd.VarAutoEncoder import model as M, sampler as S
    from musicstyletransfer_amd.VarAutoEncoder.data import Batch
    from musicstyletransfer_amd.VarAutoEncoder.transformer import TransformerConfig
    from musicstyletransfer_amd.VarAutoEncoder.utils import gpu, limit_host_threads
    limit_host_threads()
    B, K, T, Z = 64, 4, 64, 256  # scripts/train-vae.sh: --max-seq-len 64, --latent-dim 256
    cfg = M.ModelConfig(M.EncoderConfig(TransformerConfig(256, 0.2, 2, 8, NUM_EVENTS), Z, 2, NUM_EVENTS),
                        M.DecoderConfig(TransformerConfig(128, 0.2, 1, 8, NUM_EVENTS), Z, 2, NUM_EVENTS))
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):  # (the model prints its configuration, as the reference does: ONE JSON line on stdout)
        m = M.Model(cfg).initialize(gpu(0), seed=1234)
    rng = np.random.default_rng(1234)
    tokens = rng.integers(3, NUM_EVENTS, size=(B, T + 1))
    tokens[:, 0] = 1
    batch = Batch([tokens, np.full(B, T + 1), rng.integers(0, 2, size=B)], [np.zeros_like(tokens)])
    positions = max(8, min(args.steps, 2 * (T + 1) - 2))
    warm = max(2, min(args.warmup, 8))

    class A:
        verbose, beam_size = False, K

    smp = S.get_sampler("beam-search", None, None, None, A)
    smp.update_parameters(m)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for _ in range(warm):  # (captures every position's graph on the first pass; the second replays)
            smp.sample(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 0
        n_pos = 0
        n_tok = 0
        while reps < 3 or n_pos < positions:
            smp.sample(batch)
            n_pos += smp.positions_decoded
            n_tok += smp.tokens_decoded  # live continuations only (the device's `active` counters), not B * K per position
            reps += 1
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        # the device part alone: the captured positions (decode step + ranking + cache gather) replayed back to back
        bs = m.beam_search_plan(B, K, 2 * (T + 1), "query")
        with torch.cuda.stream(bs.plan.stream):
            ts = sorted(bs._graphs)
            e0, e1 = o.Event(), o.Event()
            e0.record()
            for t in ts:
                bs._graphs[t].launch()
            e1.record()
            e1.sync()
            dev_us = e0.elapsed_ms(e1) * 1e3 / max(len(ts), 1)
    tok_s = n_tok / elapsed
    # the decode step's roof: a position streams the cache rows it re-gathers (K and V of every hypothesis, positions 0..t, read +
    # written), attends over them (read once more) and reads the layer's weights; everything else is per-hypothesis rows
    dcfg = bs.plan.cfg
    Dd, V = dcfg.d_model, dcfg.out_dim
    t_avg = (max(ts) + 1) / 2.0 if ts else 1.0
    cache_bytes = dcfg.d_layers * B * K * t_avg * 2 * Dd * 2 * 3           # gather read + write + attention read, 16-bit K | V
    weight_bytes = 2.0 * (dcfg.d_layers * 12 * Dd * Dd + V * Dd * 2)        # the layers' 16-bit shadows + embedding and output tables
    pos_bytes = cache_bytes + weight_bytes + B * K * (V * 4 * 2 + Dd * 2 * 12)
    roof = {"bound": "hbm", "kernel": "one decoded position (decode step + beam ranking + cache gather: one captured graph)",
            "achieved": pos_bytes / (dev_us * 1e-6) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": pos_bytes / (dev_us * 1e-6) / 1e9 / PEAK_HBM_GBS,
            "traffic": None, "algorithmic_bytes_per_position": pos_bytes,
            "note": "latency-bound: ~20 dependent launches of 4-64 workgroups per position (profiles/README.md, decode)"}
    out = {"metric": "decoded tokens/s (beam search, KV-cache decode step)", "value": tok_s, "unit": "tokens/s", "n_gpus": 1,
           "steps": n_pos, "warmup": warm, "ms_per_step": elapsed / n_pos * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"beam search over the incremental decoder (sampler.py:198-257): {B} melodies x beam {K} = {B * K} hypotheses, "
                                  f"input length {T + 1}, up to {2 * (T + 1)} positions, token ends V={NUM_EVENTS}, decoder 128x1x8h, latent {Z}; "
                                  "a step = one position of all hypotheses = ONE captured graph: decode step, ranking (mst_beam_step) and cache gather "
                                  "(mst_beam_gather) on the device; the host launches graphs and polls a device counter every 8 positions",
                      "hypotheses": B * K, "beam": K, "positions_per_sequence": smp.positions_decoded,
                      "tokens_counted": "live continuations per position (device counters); PAD continuations of finished hypotheses excluded"},
           "hypothesis_positions_per_s": B * K * n_pos / elapsed,
           "device_us_per_position": dev_us, "host_share": 1.0 - dev_us * 1e-6 * n_pos / elapsed,
           "graphs_captured": len(bs._graphs), "ranking": "device" if smp.on_device else "host", "roofline": roof}
    print(json.dumps(out), flush=True)
    return 0


class CandidateRejected(RuntimeError):
    """a data-parallel candidate that is clearly slower than the baseline (decided on MAX-reduced numbers: every rank alike)"""


def run_candidates(baseline_name, baseline, candidates, budget_s, clock=time.monotonic, log=lambda msg: None, on_best=None):
    """The N > 1 bench line must not depend on anything that has never met RCCL: `baseline` (the result of the plainest
    schedule, already measured AND already written out by the caller) stays the answer unless a candidate measures
    faster. candidates: [(name, fn)], fn() -> result dict with 'ms_per_step' or raises. A candidate that raises is
    recorded and ends the experiments (after a failed collective the communicator's state is unknown); candidates are not
    started once `budget_s` of wall clock is spent. Returns (name, result, report). Pure host logic: tests/test_bench_launcher.py."""
    t0 = clock()
    report = {"baseline": baseline_name, "baseline_ms": baseline["ms_per_step"], "candidates": {}, "errors": {}, "not_run": []}
    best_name, best = baseline_name, baseline
    stop = False
    for name, fn in candidates:
        if stop or clock() - t0 > budget_s:
            report["not_run"].append(name)
            continue
        try:
            r = fn()
        except CandidateRejected as e:  # slower, decided alike on every rank: the next candidate may still run
            report["candidates"][name] = None
            report["errors"][name] = f"rejected: {e}"
            log(f"candidate '{name}' rejected: {e}")
            continue
        except Exception as e:  # noqa: BLE001 — anything: the baseline number stands
            report["candidates"][name] = None
            report["errors"][name] = f"{type(e).__name__}: {str(e).splitlines()[0][:300] if str(e) else ''}"
            log(f"candidate '{name}' failed ({report['errors'][name]}); keeping '{best_name}'")
            stop = True
            continue
        report["candidates"][name] = r["ms_per_step"]
        if r["ms_per_step"] < best["ms_per_step"]:
            best_name, best = name, r
            if on_best is not None:  # (a later candidate may build on the best schedule so far)
                on_best(name, r)
    report["chosen"] = best_name
    report["wall_s"] = clock() - t0
    return best_name, best, report


def run_rank(args):
    import torch
    c = CONFIGS[args.config]
    dtype = args.dtype or c["dtype"]
    B, T, P = c["B"], c["T"], c["P"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the training step has no CPU fallback")
    # MST_FORCE_DEVICE / MST_DIST_BACKEND exist to rehearse the multi-rank path on a one-GPU box (ranks share the
    # card, gloo carries the all-reduce); the driver's multi-GPU runs leave them unset: one rank per GPU over RCCL
    local_rank = int(os.environ.get("MST_FORCE_DEVICE", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from musicstyletransfer_amd import engine as E
    from musicstyletransfer_amd import ops as o
    from musicstyletransfer_amd import parallel
    from musicstyletransfer_amd.VarAutoEncoder.utils import limit_host_threads
    limit_host_threads()

    log = lambda msg: print(f"bench.py [rank {rank}]: {msg}", file=sys.stderr, flush=True)
    if world > 1:
        os.environ.setdefault("MST_RCCL_DEBUG", "1")  # communicator set-up lines only (INIT, ENV): nothing is logged per collective
    dist = parallel.init_process_group(world, rank) if world > 1 else None
    adt = torch.bfloat16 if dtype == "bf16" else torch.float16
    md = model_dims(c)
    cfg = E.VAEConfig(e_dropout=args.dropout, d_dropout=args.dropout, **md)
    store = E.ParamStore(cfg, dev, adt, seed=1234)  # identical initial weights on every rank
    store.tail_policy = "raise"  # a step the step guard skipped is work not done: the number would be invalid, so the run fails loudly
    plan = E.StepPlan(store, B, T, lr=3e-4, clip_gradient=1.0, kl_weight=1.0, global_batch=B * world, internal_eps=True, seed=1000,
                      sample_offset=rank * B, site_base=64 * rank)
    host = synthetic_batches(4, B, T, P, seed=1234 + rank)
    reduce_fn = parallel.make_grad_allreduce(dist, None) if world > 1 else None

    ms_of = lambda r: r["ms_per_step"]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        if args.data == "resident":
            # synthetic piano-rolls, uploaded once: inputs are resident in HBM before the timed region. One captured step
            # per resident batch buffer, each reading its batch in place (StepPlan.bind_inputs)
            blobs = [plan.pack_batch(hb["x"], hb["seq_lens"], hb["classes"], hb["labels"]).to(dev) for hb in host]
            pipe = batches = None
        else:
            # every step's batch goes pinned host -> HBM inside the timed region: the Trainer's batcher, three ring slots
            from musicstyletransfer_amd.pianoroll import PinnedBatchPipeline
            from musicstyletransfer_amd.VarAutoEncoder.data import Batch
            n_slots = args.ring_slots or PinnedBatchPipeline.DEFAULT_SLOTS
            pipe = PinnedBatchPipeline(dev, lambda b, t: plan, n_slots=n_slots)
            batches = [Batch([hb["x"], hb["seq_lens"], hb["classes"]], [hb["labels"]]) for hb in host]
            staged0 = [pipe.stage(batches[i % len(batches)]) for i in range(n_slots)]
            blobs = [s.slot.dev for s in staged0]
        plan.bind_inputs(blobs[0])
        plan.step_kernels(True, reduce_fn=reduce_fn)  # first step eager (HIP module loads), then capture
        torch.cuda.synchronize()

        def timed_run(overlap, group=None, limit_ms=None):
            """capture the step in the given data-parallel form for every batch buffer, W untimed warm-up steps, then EXACTLY
            K steps between barrier + synchronize; MAX over the ranks. limit_ms: give up (CandidateRejected, on every rank
            alike) when the warm-up steps already average above it."""
            rfn = parallel.make_grad_allreduce(dist, group) if world > 1 else None
            reducer = parallel.GradReducer(dist, group) if (world > 1 and overlap) else None
            graphs = {}
            for buf in blobs:
                plan.bind_inputs(buf)
                plan.capture(True, split_optimizer=world > 1, overlap=reducer is not None)
                graphs[buf.data_ptr()] = (plan.graph, plan.graph_late, plan.graph_opt)
            # data parallel: HIP events on the step's stream around the gap between the end of the backward graph and the
            # start of the optimizer graph = the part of the gradient all-reduce that nothing hides
            gaps = [(o.Event(), o.Event()) for _ in range(args.steps)] if world > 1 else None

            def launch(buf, stamps=None):
                plan.graph, plan.graph_late, plan.graph_opt = graphs[buf.data_ptr()]
                plan.run(reduce_fn=rfn, reducer=reducer, stamps=stamps)

            if args.data == "resident":
                def one_step(i, stamps=None):
                    launch(blobs[i % len(blobs)], stamps)
            else:
                feed = pipe.feed((batches[i % len(batches)] for i in range(max(args.warmup, HOST_WARMUP) + args.steps)))

                def one_step(i, stamps=None):
                    s = next(feed)  # batch i was staged while step i-1 ran; the generator stages batch i+1 at the next call
                    stream.wait_event(s.slot.uploaded)
                    launch(s.slot.dev, stamps)
                    s.slot.consumed.record(stream)

            torch.cuda.synchronize()
            w0 = time.perf_counter()
            # (--data host: at least HOST_WARMUP untimed steps — the host-fed loop meets a one-off stall of 4 .. 9 ms inside a copy
            # enqueue around its 20th step, whatever was copied before (64 priming copies did not move it), and never again in
            # 20 000 steps: start-up, not a training step; docs/kernel_notes.md "host-fed stall")
            n_warm = max(args.warmup, HOST_WARMUP) if args.data == "host" else args.warmup
            for i in range(n_warm):
                one_step(i)
            torch.cuda.synchronize()
            if dist is not None and limit_ms is not None and n_warm > 0:
                w = torch.tensor([(time.perf_counter() - w0) / n_warm * 1e3], dtype=torch.float64, device=dev)
                dist.all_reduce(w, op=dist.ReduceOp.MAX)
                if float(w[0].item()) > limit_ms:
                    raise CandidateRejected(f"warm-up steps average {float(w[0].item()):.3f} ms > {limit_ms:.3f} ms")
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            ev = [o.Event() for _ in range(args.steps + 1)]
            host_loop = []
            if args.data == "host":
                pipe.stamps = []
            t0 = time.perf_counter()
            ev[0].record()
            for i in range(args.steps):
                h0 = time.perf_counter()
                one_step(n_warm + i, gaps[i] if gaps else None)
                ev[i + 1].record()
                host_loop.append(time.perf_counter() - h0)
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            raw_steps = [ev[i].elapsed_ms(ev[i + 1]) for i in range(args.steps)]
            per_step = sorted(raw_steps)
            median_ms = per_step[len(per_step) // 2]
            host_pipeline = None
            if args.data == "host":
                # where a slow step came from: the host's three legs of stage() (wait for the ring slot's previous upload, pack into
                # the page-locked blob, enqueue the copy) and the whole loop iteration, per step, next to the step's own duration
                def spread(v):
                    v = sorted(v)
                    return {"median_ms": v[len(v) // 2] * 1e3, "p99_ms": v[int(0.99 * (len(v) - 1))] * 1e3, "max_ms": v[-1] * 1e3}
                legs = list(zip(*pipe.stamps)) if pipe.stamps else ([0.0], [0.0], [0.0])
                worst = max(range(args.steps), key=lambda i: raw_steps[i])
                host_pipeline = {"ring_slots": pipe.n_slots, "slot_wait": spread(legs[0]), "pack": spread(legs[1]), "enqueue": spread(legs[2]),
                                 "host_loop": spread(host_loop), "slowest_step": {"index": worst, "gpu_ms": raw_steps[worst],
                                                                                 "host_loop_ms": [t * 1e3 for t in host_loop[max(0, worst - 3): worst + 2]]},
                                 "max_over_median": per_step[-1] / median_ms, "untimed_warmup_steps": n_warm}
            exposed_us = None
            if dist is not None:
                gap = sorted(a.elapsed_ms(b) * 1e3 for a, b in gaps)
                t = torch.tensor([elapsed, median_ms, gap[len(gap) // 2]], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed, median_ms, exposed_us = float(t[0].item()), float(t[1].item()), float(t[2].item())
            cut = plan.grad_cut() if reducer is not None else 0
            return {"elapsed": elapsed, "ms_per_step": elapsed / args.steps * 1e3, "median_ms": median_ms, "per_step": per_step,
                    "host_pipeline": host_pipeline, "exposed_us": exposed_us, "cut": cut}

        step_flops = 3.0 * fwd_flops(md, B, T)
        exec_flops = 3.0 * executed_fwd_flops(md, B, T)

        def bench_line(r, extra=None):
            """the contract's JSON object from one timed run's result"""
            ms = r["ms_per_step"]
            per_step = r["per_step"]
            out = {
                "metric": "piano-roll frames/s (VAE train step)", "value": B * T * world * args.steps / r["elapsed"], "unit": "frames/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "ms_per_step_median": r["median_ms"],
                "ms_per_step_max": per_step[-1], "ms_per_step_p90": per_step[int(0.9 * (len(per_step) - 1))],
                "value_at_median": B * T * world / (r["median_ms"] * 1e-3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": dtype,
                "data": "synthetic" if args.data == "resident" else "synthetic (host batches through the pinned pipeline inside the timed region)",
                "config": {"workload": f"{c['name']}, T={T}, pitch={P}, latent={c['Z']}, batch={B}/GPU, {dtype}; encoder 256x2x8h, decoder "
                                       f"128x1x8h (scripts/train-vae.sh), dropout {args.dropout}; Xavier-initialised weights",
                           "baseline_config_index": args.config if world == 1 or args.config != 1 else 3,
                           "global_batch": B * world, "seq_len": T, "pitches": P, "latent": c["Z"], "parallelism": f"dp{world}",
                           "params": store.n_params},
                "step_tflops": step_flops / (ms * 1e-3) / 1e12,
                "step_mfma_frac": step_flops / (ms * 1e-3) / 1e12 / PEAK_MFMA_TFLOPS,
                "executed_flops_per_step": exec_flops, "algorithmic_flops_per_step": step_flops,
                "executed_tflops": exec_flops / (ms * 1e-3) / 1e12,
            }
            out.update(extra or {})
            return out

        # ---- the plainest form first: the default communicator, ONE all-reduce between backward and Adam (N = 1: the whole step
        # as one graph). It is measured and WRITTEN OUT (stderr + a file) before anything else touches RCCL.
        base_name = "one all-reduce between backward and Adam, default communicator" if world > 1 else "one captured graph"
        best_name, best = base_name, timed_run(False, None)
        dp_report = None
        final_lock, final_done = threading.Lock(), [False]
        if world > 1:
            partial_path = os.environ.get("MST_BENCH_PARTIAL") or os.path.join(tempfile.gettempdir(), f"mst_bench_partial_n{world}.json")
            if rank == 0:
                line = json.dumps(bench_line(best, {"partial": True, "schedule": base_name}))
                print("bench.py partial result (baseline schedule, before any experiment): " + line, file=sys.stderr, flush=True)
                try:
                    with open(partial_path, "w") as fh:
                        fh.write(line + "\n")
                except OSError as e:
                    log(f"could not write {partial_path}: {e}")

            # ---- a collective that never returns cannot be caught: if the experiments below do not finish in time, the baseline
            # line is the bench line and every rank leaves (os._exit: nothing of a stuck communicator is waited for)
            def give_up():
                with final_lock:
                    if final_done[0]:
                        return
                    final_done[0] = True
                    if rank == 0:
                        rep = {"baseline": base_name, "baseline_ms": ms_of(best), "chosen": base_name, "timed_out_after_s": exp_timeout}
                        print(json.dumps(bench_line(best, {"roofline": None, "rccl": {"nranks": world, "experiments": rep}})), flush=True)
                    sys.stderr.write(f"bench.py [rank {rank}]: data-parallel experiments did not finish in {exp_timeout:.0f} s; "
                                     "the baseline schedule's number stands\n")
                    sys.stderr.flush()
                    os._exit(0)
            exp_timeout = float(os.environ.get("MST_BENCH_EXPERIMENT_TIMEOUT", "240"))
            watchdog = threading.Timer(exp_timeout, give_up)
            watchdog.daemon = True
            watchdog.start()

            cands = []
            limit = 3.0 * ms_of(best)
            if os.environ.get("MST_DP_OVERLAP", "1") != "0" and plan.grad_cut() > 0:
                def overlapped():
                    hook = os.environ.get("MST_BENCH_TEST_CANDIDATE")  # rehearsals of the guard rails (docs/switches.md)
                    if hook == "raise":
                        raise RuntimeError("MST_BENCH_TEST_CANDIDATE=raise: a candidate that fails")
                    if hook == "hang":
                        time.sleep(1e6)
                    return timed_run(True, None, limit_ms=limit)
                cands.append(("two ranges, the early one on the wire during the rest of backward, default communicator", overlapped))
            tune_box = {}
            if os.environ.get("MST_RCCL_AUTOTUNE", "0") == "1" or os.environ.get("MST_RCCL_ALGO"):
                def with_tuned_group():
                    cut = plan.grad_cut()
                    g, rep = parallel.autotune_allreduce(dist, [store.n - cut, cut, store.n] if cut else [store.n], dev)
                    tune_box["report"] = rep
                    if g is None:
                        raise CandidateRejected(f"no communicator beat the default one by the margin ({rep.get('candidates')})")
                    ov = best_name.startswith("two ranges")
                    r = timed_run(ov, g, limit_ms=limit)
                    r["group"] = rep["chosen"]
                    return r
                cands.append(("best schedule so far on the communicator parallel.autotune_allreduce picked", with_tuned_group))

            def on_best(name, res):
                nonlocal best_name, best
                best_name, best = name, res
            _, _, dp_report = run_candidates(base_name, best, cands, exp_timeout * 0.75, log=log, on_best=on_best)
            if tune_box.get("report"):
                dp_report["communicator_autotune"] = tune_box["report"]
            watchdog.cancel()
        m = plan.metrics()
        if args.data == "host":
            plan.bind_inputs(blobs[0])
        roof = roofline(plan, o, args.config) if rank == 0 else None

    rccl = None
    if dist is not None:
        cut = best["cut"]
        rccl = {"nranks": world, "backend": dist.get_backend(), "bucket_bytes": 4 * store.n,
                "ranges_bytes": {"overlapped_with_backward": 4 * (store.n - cut), "exposed": 4 * cut} if cut else {"exposed": 4 * store.n},
                "schedule": best_name, "exposed_allreduce_us": best["exposed_us"], "baseline_ms": dp_report["baseline_ms"],
                "experiments": dp_report, "log": parallel.rccl_report(cleanup=True)}

        def leave():
            """the job's last collective comes AFTER rank 0 has printed its line (below), and a communicator that a failed
            experiment left in an unknown state may never finish it: bounded, then every rank just exits"""
            bye = threading.Timer(30.0, lambda: os._exit(0))
            bye.daemon = True
            bye.start()
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception as e:  # noqa: BLE001
                log(f"shutdown of the process group failed: {e}")
            bye.cancel()
    else:
        leave = lambda: None
    if rank != 0:
        leave()
        return 0
    with final_lock:
        if final_done[0]:
            return 0
        final_done[0] = True
    nonfinite = int(m.get("nonfinite_steps", 0))
    out = bench_line(best, {
        "elbo": m["total_loss"], "kl": m["kl_loss"],
        # steps (timed runs + warm-up) whose loss was not finite: the optimizer left the model alone in them (mst_step_metrics'
        # non-finite guard; seen in long free-running fp16 runs at T = 1024, never in the default run). They are not work done:
        # a run that has any reports value = null.
        "nonfinite_steps": nonfinite,
        # what precision meets what tolerance against the CPU oracle on identical weights / inputs / eps, dropout 0
        # (tests/test_configs_gpu.py, tests/test_step_gpu.py): KL = 0.5 sum(sigma^2 + mu^2 - 1 - log sigma^2) has
        # no epsilon and sigma straddles 0 at Xavier init, where a handful of |sigma| < 1e-2 elements carry the error
        "elbo_tolerance": {"bf16_raw_init": 4e-3, "fp16_raw_init": 1e-3, "bf16_conditioned": 1e-3, "fp16_conditioned": 1e-3,
                           "reconstruction_loss_any": 1e-3, "this_run": dtype + "_raw_init"},
        "roofline": roof,
    })
    if nonfinite:
        out["invalid"] = f"{nonfinite} steps were skipped by the non-finite guard: the timed region did less work than {args.steps} steps"
        out["value_if_all_steps_counted"], out["value"] = out["value"], None
    if rccl is not None:
        out["rccl"] = rccl
    if best["host_pipeline"] is not None:
        out["host_pipeline"] = best["host_pipeline"]
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(c, args.dropout)
    print(json.dumps(out), flush=True)
    leave()
    return 1 if nonfinite else 0


if __name__ == "__main__":
    sys.exit(main())
