import torch, time
def t(nbytes, reps=20):
    x = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    for _ in range(3): x.zero_()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): x.zero_()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print("fill %6.1f MB: %7.1f us  %5.2f TB/s" % (nbytes / 1e6, ms * 1e3, nbytes / ms / 1e9))
for n in (185_000_000, 247_000_000, 432_000_000, 1_000_000_000, 4_000_000_000):
    t(n)
# copy (read+write)
x = torch.empty(432_000_000, dtype=torch.uint8, device='cuda'); y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): y.copy_(x)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print("copy 432 MB: %.1f us, %.2f TB/s (r+w)" % (ms * 1e3, 2 * 432e6 / ms / 1e9))
