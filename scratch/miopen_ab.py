"""A/B baseline for the hand-written MFMA convolutions: the flow-update operator through torch conv2d (MIOpen, fp16
channels_last).  Diagnostic only - nothing in vipe_amd/ imports this.

    python scratch/miopen_ab.py   # on the GPU box: per-update time of both formulations at E=276, 48x64
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def forward_miopen(m, net, inp, corr, flow, ix, skip_upmask=True, n_src=None):
    """UpdateModule.forward (droid_net.py:467-499) as torch ops; NCHW [1,E,C,h,w] in and out."""
    batch, num, ch, ht, wd = net.shape
    E = batch * num
    dev = net.device
    f16 = torch.float16
    mio = {k: (c.weight.detach().to(dev, f16).contiguous(memory_format=torch.channels_last), c.bias.detach().to(dev, f16))
           for k, c in dict(corr0=m.corr_encoder[0], corr2=m.corr_encoder[2], flow0=m.flow_encoder[0],
                            flow2=m.flow_encoder[2], w=m.gru.w, convz=m.gru.convz, convr=m.gru.convr,
                            convq=m.gru.convq, zg=m.gru.convz_glo, rg=m.gru.convr_glo, qg=m.gru.convq_glo,
                            d0=m.delta[0], d2=m.delta[2], w0=m.weight[0], w2=m.weight[2],
                            a1=m.agg.conv1, a2=m.agg.conv2, eta=m.agg.eta[0], up=m.agg.upmask[0]).items()}

    def cv(x, k, pad):
        w, b = mio[k]
        return F.conv2d(x, w, b, padding=pad)

    def cl(t, c):
        return t.reshape(E, c, ht, wd).to(f16).contiguous(memory_format=torch.channels_last)

    net_ = cl(net, 128)
    inp_ = cl(inp, 128)
    c = F.relu(cv(F.relu(cv(cl(corr, 196), "corr0", 0)), "corr2", 1))
    fl = cl(flow, 4) if flow is not None else torch.zeros(E, 4, ht, wd, device=dev, dtype=f16)
    f = F.relu(cv(F.relu(cv(fl, "flow0", 3)), "flow2", 1))
    x = torch.cat([inp_, c, f], 1)
    hx = torch.cat([net_, x], 1)
    glo = (torch.sigmoid(cv(net_, "w", 0)) * net_).mean(dim=(2, 3), keepdim=True)
    z = torch.sigmoid(cv(hx, "convz", 1) + cv(glo, "zg", 0))
    r = torch.sigmoid(cv(hx, "convr", 1) + cv(glo, "rg", 0))
    q = torch.tanh(cv(torch.cat([r * net_, x], 1), "convq", 1) + cv(glo, "qg", 0))
    net_ = (1 - z) * net_ + z * q
    delta = cv(F.relu(cv(net_, "d0", 1)), "d2", 1).view(batch, num, -1, ht, wd).permute(0, 1, 3, 4, 2)[..., :2].contiguous()
    weight = torch.sigmoid(cv(F.relu(cv(net_, "w0", 1)), "w2", 1)).view(batch, num, -1, ht, wd) \
        .permute(0, 1, 3, 4, 2)[..., :2].contiguous()
    net_out = net_.view(batch, num, 128, ht, wd)
    if ix is None:
        return net_out, delta, weight
    a = F.relu(cv(net_, "a1", 1))
    if n_src is None:
        n_src = int(ix.max().item()) + 1 if ix.numel() else 0
    ixd = ix.to(dev)
    acc = torch.zeros((n_src, 128, ht, wd), dtype=torch.float32, device=dev).index_add_(0, ixd, a.float())
    cnt = torch.zeros(n_src, dtype=torch.float32, device=dev).index_add_(0, ixd, torch.ones(E, device=dev))
    a = (acc / cnt.clamp(min=1).view(-1, 1, 1, 1)).to(f16).contiguous(memory_format=torch.channels_last)
    a = F.relu(cv(a, "a2", 1))
    eta = F.softplus(cv(a, "eta", 1).float()).view(batch, n_src, ht, wd)
    upmask = None if skip_upmask else cv(a, "up", 0).view(batch, n_src, 576, ht, wd)
    return net_out, delta, weight, 0.01 * eta, upmask


if __name__ == "__main__":
    from vipe_amd.slam.networks import UpdateModule

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    um = UpdateModule().eval()
    E, ht, wd = 276, 48, 64
    net = torch.randn(1, E, 128, ht, wd, device=dev).tanh().half()
    inp = torch.randn(1, E, 128, ht, wd, device=dev).relu().half()
    corr = torch.randn(1, E, 196, ht, wd, device=dev).half()
    flow = torch.randn(1, E, 4, ht, wd, device=dev).half()
    ix = (torch.arange(E, device=dev) // 6).long()
    for name, fn in (("miopen", lambda: forward_miopen(um, net, inp, corr, flow, ix)),
                     ("hip", lambda: um.engine(dev).forward(net, inp, corr, flow, ix, skip_upmask=True))):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) * 100:.2f} ms per operator application (NCHW entry point)")
