"""HBM ceiling calibration with plain torch kernels: fill (write), copy (read+write), sum (read)."""
import time, torch
n = 1 << 30  # 4 GiB of fp32
x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
gb = n * 4 / 1e9
print("fill  (write)      %.2f TB/s" % (gb / t(lambda: x.fill_(1.0)) / 1e3))
print("copy  (read+write) %.2f TB/s" % (2 * gb / t(lambda: y.copy_(x)) / 1e3))
print("sum   (read)       %.2f TB/s" % (gb / t(lambda: x.sum()) / 1e3))
print("add   (2r+1w)      %.2f TB/s" % (3 * gb / t(lambda: torch.add(x, y, out=y)) / 1e3))
