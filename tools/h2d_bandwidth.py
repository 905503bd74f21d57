import torch, time
n = 985661440
h = torch.empty(n, dtype=torch.uint8).pin_memory()
h.fill_(3)
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print("h2d ms", dt * 1e3, "GB/s", n / dt / 1e9)
print(int(d[12345]))
# is a kernel enqueued behind the copy on the same stream ordered after it?
a = torch.randn(8192, 8192, device="cuda")
def work():
    for _ in range(6):
        a @ a
work(); torch.cuda.synchronize(); t = time.perf_counter(); work(); torch.cuda.synchronize()
print("work ms", (time.perf_counter() - t) * 1e3)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    d.copy_(h, non_blocking=True)
    work()
torch.cuda.synchronize()
print("copy+work per step ms", (time.perf_counter() - t) / 5 * 1e3)
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record(); d.copy_(h, non_blocking=True); e[1].record(); work(); e[2].record(); torch.cuda.synchronize()
print("events: copy", e[0].elapsed_time(e[1]), "work", e[1].elapsed_time(e[2]))
