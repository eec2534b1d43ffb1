import torch, time
dev = torch.device('cuda:0')
x = torch.empty(4 * 1024**3 // 2, dtype=torch.float16, device=dev)   # 4 GiB
y = torch.empty_like(x)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = x.numel() * 2 / 1e9
ms = t(lambda: x.fill_(1.0)); print(f'fill  {gb:.2f} GB: {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s written')
ms = t(lambda: y.copy_(x)); print(f'copy  {gb:.2f} GB: {ms:.3f} ms  {2*gb/ms*1e3:.0f} GB/s read+write')
ms = t(lambda: x.sum()); print(f'read  {gb:.2f} GB: {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s read')
