"""GPU diagnostic: where does the HIP step differ from the oracle? (not a test)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_step_gpu import _setup

def run(kind, dims, B, T, seed):
    O, E, ocfg, ecfg, params, batch, eps = _setup(kind, dims, B, T, seed)
    gpu = torch.device("cuda", 0)
    ot = O.OracleTrainer(ocfg, params, lr=1e-3)
    ref = ot.step(batch, torch.from_numpy(eps))
    # oracle in fp64 to size the oracle's own fp32 noise
    ot64 = O.OracleTrainer(ocfg, params, lr=1e-3, dtype=torch.float64)
    ref64 = ot64.step(batch, torch.from_numpy(eps))
    store = E.ParamStore(ecfg, gpu, torch.bfloat16, params_np=params)
    plan = E.StepPlan(store, B, T, lr=1e-3, want_probs=True)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    plan.step_kernels(True); torch.cuda.synchronize()
    tot, kl, rec = plan.total.cpu().numpy(), plan.kl.cpu().numpy(), plan.recon.cpu().numpy()
    rt, rk, rr = ref["loss"].numpy(), ref["kl"].numpy(), ref["recon"].numpy()
    print(f"--- {kind} dims={dims} B={B} T={T}")
    print(" mean total gpu %.6f ref %.6f rel %.2e | fp32-vs-fp64 oracle rel %.2e" % (tot.mean(), rt.mean(), abs(tot.mean()-rt.mean())/abs(rt.mean()), abs(rt.mean()-ref64['loss'].numpy().mean())/abs(rt.mean())))
    print(" mean kl    gpu %.6f ref %.6f rel %.2e" % (kl.mean(), rk.mean(), abs(kl.mean()-rk.mean())/abs(rk.mean())))
    print(" mean recon gpu %.6f ref %.6f rel %.2e" % (rec.mean(), rr.mean(), abs(rec.mean()-rr.mean())/abs(rr.mean())))
    mu, sg = plan.mu.cpu().numpy(), plan.sigma.cpu().numpy()
    print(" max|dmu| %.3e max|dsigma| %.3e  (|mu| max %.2f) min|sigma| %.2e" % (np.abs(mu-ref['means'].numpy()).max(), np.abs(sg-ref['stds'].numpy()).max(), np.abs(mu).max(), np.abs(ref['stds'].numpy()).min()))
    print(" per-sample |dkl| max %.3e" % np.abs(kl-rk).max())
run("token", (10, 10, 3, 16, 32, 1, 2, 32, 1, 2), 3, 5, 7)
run("pianoroll", (40, 40, 2, 16, 64, 2, 2, 32, 1, 2), 5, 19, 12)
run("token", (293, 293, 2, 32, 64, 2, 4, 32, 1, 2), 6, 23, 11)
run("pianoroll", (128, 128, 2, 64, 256, 2, 8, 128, 1, 8), 64, 256, 1234)
run("pianoroll", (128, 128, 2, 64, 256, 2, 8, 128, 1, 8), 64, 256, 99)
