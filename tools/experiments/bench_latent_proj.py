"""launch time of latent_fwd / latent_bwd_vec with and without the row-0 projection forms (configs[1] shapes), graph-replayed"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from musicstyletransfer_amd import engine as E, ops as o
dev = torch.device("cuda", 0)
c = bench.CONFIGS[1]
cfg = E.VAEConfig(e_dropout=0.2, d_dropout=0.2, **bench.model_dims(c))
store = E.ParamStore(cfg, dev, torch.bfloat16, seed=1234)
B, T = c["B"], c["T"]
plan = E.StepPlan(store, B, T, lr=3e-4, clip_gradient=1.0, internal_eps=True, seed=1000)
hb = bench.synthetic_batches(1, B, T, c["P"], seed=1)[0]
plan.load_batch(hb["x"], hb["seq_lens"], hb["classes"], hb["labels"])
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    plan.step_kernels(True)
    torch.cuda.synchronize()
    s, Sd, Se, Dd = store, T + 1, T, cfg.d_model
    import math
    sq_d = math.sqrt(float(Dd))
    t = plan.bd_l[0]
    dx = plan.bd_l[0].dx_a
    args = (s.p("encoder.latent_proj.weight"), plan.eps, s.p("decoder.latent2hid.weight"), plan.classes, plan.mu, plan.sigma, dx.view(B, Sd, -1), sq_d, 1.0, 1.0,
            s.grad("decoder.class2hid.weight"), plan.d_enc_out.view(B, Se, -1), plan.lat_scratch)
    proj = (t.dqkv.view(B, Sd, -1), s.t("decoder.layer0.att.W_kqv"), t.dh1.view(B, Sd, -1))
    print("latent_bwd_vec plain  us:", bench.time_launch(o, lambda: o.latent_bwd_vec(*args), 20) * 1e3)
    print("latent_bwd_vec proj   us:", bench.time_launch(o, lambda: o.latent_bwd_vec(*args, proj=proj), 20) * 1e3)
    print("latent_bwd_vec proj, no resid us:", bench.time_launch(o, lambda: o.latent_bwd_vec(*args, proj=(proj[0], proj[1], None)), 20) * 1e3)
