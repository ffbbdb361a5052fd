"""What does this MI355X stream?  torch elementwise kernels on 134 MB bf16 tensors (the size of a stage-1 activation at 4 x 512
images): read-only sum, copy (R1 W1), add (R2 W1: bn_bwd_apply's mix), and the same on fp32."""
import torch, time
d = torch.device("cuda:0")
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
for dt, name in ((torch.bfloat16, "bf16"), (torch.float32, "fp32")):
    n = 2048 * 1024 * 32
    a, b = torch.randn(n, device=d).to(dt), torch.randn(n, device=d).to(dt)
    c = torch.empty_like(a)
    by = a.element_size() * n
    t = timed(lambda: torch.add(a, b, out=c)); print("%s add  R2 W1: %6.1f us  %5.2f TB/s" % (name, t * 1e6, 3 * by / t / 1e12))
    t = timed(lambda: c.copy_(a));             print("%s copy R1 W1: %6.1f us  %5.2f TB/s" % (name, t * 1e6, 2 * by / t / 1e12))
    t = timed(lambda: a.sum());                print("%s sum  R1   : %6.1f us  %5.2f TB/s" % (name, t * 1e6, 1 * by / t / 1e12))
    t = timed(lambda: c.zero_());              print("%s zero W1   : %6.1f us  %5.2f TB/s" % (name, t * 1e6, 1 * by / t / 1e12))
