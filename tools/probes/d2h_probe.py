import time, torch, numpy as np
x = torch.rand(512, 512, 3, 8, device='cuda'); m = torch.rand(512, 512, 8, device='cuda')
for name, f in (('pageable .cpu().numpy()', lambda: (x.cpu().numpy(), m.cpu().numpy())),):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    print(name, (time.perf_counter() - t0) / 10 * 1e3, 'ms')
px = torch.empty(x.shape, dtype=x.dtype, pin_memory=True); pm = torch.empty(m.shape, dtype=m.dtype, pin_memory=True)
def g():
    px.copy_(x, non_blocking=True); pm.copy_(m, non_blocking=True); torch.cuda.synchronize()
    return px.numpy().copy(), pm.numpy().copy()
for _ in range(3): g()
t0 = time.perf_counter()
for _ in range(10): g()
print('pinned + numpy copy', (time.perf_counter() - t0) / 10 * 1e3, 'ms')
def h():
    px.copy_(x, non_blocking=True); pm.copy_(m, non_blocking=True); torch.cuda.synchronize()
for _ in range(3): h()
t0 = time.perf_counter()
for _ in range(10): h()
print('pinned only', (time.perf_counter() - t0) / 10 * 1e3, 'ms')
y = np.random.rand(512, 512).astype(np.float32); P = np.random.rand(512, 512, 8).astype(np.float32)
def u():
    a = torch.from_numpy(y).cuda(); b = torch.from_numpy(P).cuda(); torch.cuda.synchronize()
for _ in range(3): u()
t0 = time.perf_counter()
for _ in range(10): u()
print('pageable upload y, Phi', (time.perf_counter() - t0) / 10 * 1e3, 'ms')
