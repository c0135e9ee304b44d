# the training half of tools/soak.sh at 600 steps: the reference's own CLI on the golden MIDI files (ragged, padded batches)
python - <<'PY' > gpurun_out/soak_train_short.log 2>&1
import sys, os, time, warnings
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from music_style_transfer.VarAutoEncoder import main
flags = ["--batch-size", "8", "--kl-loss", "1.0", "--validation-split", "0.0", "--max-seq-len", "64", "--slices-per-quarter-note", "4",
         "--sampling-frequency", "400", "--checkpoint-frequency", "500", "--num-checkpoints-not-improved", "32", "--epochs", "10000",
         "--optimizer", "adam", "--optimizer-params", "clip_gradient:1.0", "--learning-rate", "0.0003", "--label-smoothing", "0.0",
         "--e-n-layers", "2", "--e-dropout", "0.2", "--e-rnn-hidden-dim", "256", "--e-emb-hidden-dim", "256", "--latent-dim", "256",
         "--d-n-layers", "1", "--d-rnn-hidden-dim", "128", "--d-dropout", "0.2", "--gpu", "--data", "tests/golden/midi",
         "--model-output", "/tmp/soak_m", "--out-samples", "/tmp/soak_s", "--max-steps", "600"]
t0 = time.time()
with warnings.catch_warnings(record=True) as caught:
    warnings.simplefilter("always")
    t = main.main(flags)
    m = t.collect_metrics()
print("steps", t.train_state.n_batches, "adam t", int(t.model.store.step_state[0].item()), "metrics", m, "tail failures", t.model.store.tail_failures,
      "warnings", [str(c.message)[:120] for c in caught if "tail" in str(c.message) or "finite" in str(c.message)], "seconds", round(time.time() - t0, 1))
assert int(t.model.store.step_state[0].item()) == 600 and not t.model.store.tail_failures
PY
echo "train rc=$?"; tail -3 gpurun_out/soak_train_short.log
