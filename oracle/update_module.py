"""Flow-update operator (ConvGRU + heads + GraphAgg), functional fp32 restatement.

ORACLE (test infrastructure). Follows vipe/slam/networks/droid_net.py:373-499
(ConvGRU :373-400, GraphAgg :403-429, UpdateModule :432-499) and
vipe/ext/scatter.py:56-63 (scatter_mean).  `sd` is a state dict with the
reference key layout (update.* prefix stripped): corr_encoder.{0,2}, flow_encoder.{0,2},
weight.{0,2}, delta.{0,2}, gru.{convz,convr,convq,w,convz_glo,convr_glo,convq_glo},
agg.{conv1,conv2,eta.0,upmask.0}, each .weight/.bias.
"""

import torch
import torch.nn.functional as F


def _conv(sd, name, x, pad):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)


def scatter_mean(src, index, dim_size=None):
    """scatter.py:56-63 along dim 1 of [B,E,...]: sum / clamp(count,1)."""
    B, E = src.shape[:2]
    n = int(index.max()) + 1 if dim_size is None else dim_size
    out = torch.zeros((B, n) + src.shape[2:], dtype=src.dtype)
    out.index_add_(1, index, src)
    cnt = torch.zeros(n, dtype=src.dtype).index_add_(0, index, torch.ones(E, dtype=src.dtype)).clamp(min=1)
    return out / cnt.view(1, n, *([1] * (src.dim() - 2)))


def update_forward(sd, net, inp, corr, flow, ix=None):
    """droid_net.py:467-499. net, inp [1,E,128,h,w]; corr [1,E,196,h,w]; flow [1,E,4,h,w]; ix [E] int64."""
    b, num, ch, ht, wd = net.shape
    odim = (b, num, -1, ht, wd)
    net = net.reshape(b * num, -1, ht, wd)
    inp = inp.reshape(b * num, -1, ht, wd)
    corr = corr.reshape(b * num, -1, ht, wd)
    flow = flow.reshape(b * num, -1, ht, wd)

    c = F.relu(_conv(sd, "corr_encoder.2", F.relu(_conv(sd, "corr_encoder.0", corr, 0)), 1))
    f = F.relu(_conv(sd, "flow_encoder.2", F.relu(_conv(sd, "flow_encoder.0", flow, 3)), 1))

    # ConvGRU (droid_net.py:387-400)
    x = torch.cat([inp, c, f], dim=1)
    hx = torch.cat([net, x], dim=1)
    glo = torch.sigmoid(_conv(sd, "gru.w", net, 0)) * net
    glo = glo.view(b * num, ch, ht * wd).mean(-1).view(b * num, ch, 1, 1)
    z = torch.sigmoid(_conv(sd, "gru.convz", hx, 1) + _conv(sd, "gru.convz_glo", glo, 0))
    r = torch.sigmoid(_conv(sd, "gru.convr", hx, 1) + _conv(sd, "gru.convr_glo", glo, 0))
    q = torch.tanh(_conv(sd, "gru.convq", torch.cat([r * net, x], dim=1), 1) + _conv(sd, "gru.convq_glo", glo, 0))
    net = (1 - z) * net + z * q

    delta = _conv(sd, "delta.2", F.relu(_conv(sd, "delta.0", net, 1)), 1).view(*odim)
    weight = torch.sigmoid(_conv(sd, "weight.2", F.relu(_conv(sd, "weight.0", net, 1)), 1)).view(*odim)
    delta = delta.permute(0, 1, 3, 4, 2)[..., :2].contiguous()
    weight = weight.permute(0, 1, 3, 4, 2)[..., :2].contiguous()
    net = net.view(*odim)
    if ix is None:
        return net, delta, weight

    # GraphAgg (droid_net.py:414-429)
    a = F.relu(_conv(sd, "agg.conv1", net.view(b * num, 128, ht, wd), 1)).view(b, num, 128, ht, wd)
    a = scatter_mean(a, ix).view(-1, 128, ht, wd)
    a = F.relu(_conv(sd, "agg.conv2", a, 1))
    eta = F.softplus(_conv(sd, "agg.eta.0", a, 1)).view(b, -1, ht, wd)
    upmask = _conv(sd, "agg.upmask.0", a, 0).view(b, -1, 8 * 8 * 9, ht, wd)
    return net, delta, weight, 0.01 * eta, upmask
