"""What this box's HBM delivers to plain streaming kernels (calibration of the roofline denominators; not a test):
device-to-device copy (read + write), fill (write), sum (read) over buffers far beyond the 256 MB memory-side cache."""
import torch, time
dev = torch.device("cuda", 0)
n = 2 * 1024 ** 3 // 4  # 2 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
s = t(lambda: b.copy_(a)); print(f"copy  : {2 * a.numel() * 4 / s / 1e12:.2f} TB/s (read + write)")
s = t(lambda: b.fill_(1.0)); print(f"fill  : {a.numel() * 4 / s / 1e12:.2f} TB/s (write)")
s = t(lambda: a.sum()); print(f"sum   : {a.numel() * 4 / s / 1e12:.2f} TB/s (read)")
s = t(lambda: torch.add(a, b, out=b)); print(f"add   : {3 * a.numel() * 4 / s / 1e12:.2f} TB/s (2 reads + 1 write)")
