"""How much of a Cin=128 -> 256 3x3 convolution (the `net` part of the z|r gates) and of the global-context conv hides
in the shadow of the BA kernels when it runs on a side stream forked after the operator?  Extra work is ADDED here:
the step time with it minus the step time without it is the part that does not hide."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
from vipe_amd.ext import slam_ext
from vipe_amd.slam import factor_graph as FG
from vipe_amd.slam.update_engine import _Packed

dev = torch.device("cuda:0")
g, buf, graph = bench.build_problem(dev, 48, 384, 512, 3, 0, seed=1234)
E = int(graph.ii.numel())
eng = graph.update_op.engine(dev)
torch.manual_seed(0)
pk = _Packed((0.02 * torch.randn(256, 128, 3, 3)).half(), torch.zeros(256), dev)
ybuf = torch.empty((E, 48, 64, 256), dtype=torch.float16, device=dev)
mode = {"side": None, "prio": False}
orig_finish = slam_ext.update_finish
streams = {"lo": torch.cuda.Stream(priority=0), "hi": torch.cuda.Stream(priority=-1)}


def finish_hook(*a, **k):
    orig_finish(*a, **k)
    if mode["side"] is not None:
        s = mode["side"]
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            eng._conv(pk, graph.net_n, 0, E, 48, 64, y=ybuf, act="none")
            if mode.get("glo"):
                eng._conv(eng.gw, graph.net_n, 0, E, 48, 64, mode="glo", net=graph.net_n, fout=glo)


glo = torch.zeros((E, 128), dtype=torch.float32, device=dev)
FG.slam_ext.update_finish = finish_hook


def step():
    graph.update(t0=1, t1=48, itrs=3)
    if mode["side"] is not None:
        torch.cuda.current_stream().wait_stream(mode["side"])


def run(label, n=40, use_graph=False, main=None):
    ctx = torch.cuda.stream(main) if main is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        cg = None
        if use_graph:
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg, stream=main if main is not None else None):
                step(); step()
            torch.cuda.synchronize()
            cg.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if cg is not None:
            for _ in range(n // 2):
                cg.replay()
        else:
            for _ in range(n):
                step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"{label:50s} {1e3 * dt:.3f} ms/step  {1 / dt:.1f} it/s", flush=True)
    return dt


# the conv alone
for _ in range(3):
    eng._conv(pk, graph.net_n, 0, E, 48, 64, y=ybuf, act="none")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    eng._conv(pk, graph.net_n, 0, E, 48, 64, y=ybuf, act="none")
torch.cuda.synchronize()
print(f"extra conv alone: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)

import os, ctypes, numpy as np
SM = ctypes.CDLL("/root/repo/scratch/libstreammask.so")


def masked_stream(n_reserved, pattern="low"):
    """all 256 CUs minus n_reserved (pattern: which mask bits are cleared)"""
    bits = np.ones(256, dtype=bool)
    if pattern == "low":
        bits[:n_reserved] = False
    elif pattern == "high":
        bits[256 - n_reserved:] = False
    words = np.packbits(bits.reshape(8, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).ravel()
    out = ctypes.c_void_p()
    rc = SM.sm_create(256, words.ctypes.data_as(ctypes.c_void_p), 8, ctypes.byref(out))
    assert rc == 0
    return torch.cuda.ExternalStream(out.value)


def probe(stream, label):
    n = 2048
    o = torch.zeros(2 * n, dtype=torch.int32, device=dev)
    with torch.cuda.stream(stream):
        SM.sm_where(ctypes.c_void_p(stream.cuda_stream), ctypes.c_void_p(o.data_ptr()), n)
    torch.cuda.synchronize()
    a = o.cpu().numpy().astype(np.uint32).reshape(n, 2)
    hw, xcc = a[:, 0], a[:, 1] & 0xF
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 0x7
    ids = set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_xcc = {x: len([i for i in ids if i[0] == x]) for x in sorted(set(xcc.tolist()))}
    print(f"probe {label}: {len(ids)} distinct (xcc,se,sh,cu); per xcc {per_xcc}", flush=True)


def conv_alone(stream, label):
    with torch.cuda.stream(stream):
        for _ in range(3):
            eng._conv(pk, graph.net_n, 0, E, 48, 64, y=ybuf, act="none")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng._conv(pk, graph.net_n, 0, E, 48, 64, y=ybuf, act="none")
        torch.cuda.synchronize()
        print(f"conv alone on {label}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)


if os.environ.get("SHADOW_MASK") == "2":
    conv_alone(torch.cuda.current_stream(), "default stream")
    conv_alone(streams["lo"], "torch side stream")
    for nres in (0, 8, 32):
        ms = masked_stream(nres)
        conv_alone(ms, f"masked stream -{nres}")
        mode.update(side=None)
        run(f"eager: baseline step ON masked stream -{nres}", use_graph=False, main=ms)
    sys.exit(0)
if os.environ.get("SHADOW_MASK"):
    probe(torch.cuda.current_stream(), "default stream")
    for nres, pat in ((8, "low"), (8, "high"), (16, "low"), (32, "low"), (64, "low")):
        ms = masked_stream(nres, pat)
        probe(ms, f"masked -{nres} {pat}")
        mode.update(side=ms, glo=True)
        run(f"eager: side conv + glo on masked stream (-{nres} CUs, {pat})", use_graph=False)
    mode.update(side=streams["lo"], glo=True)
    run("eager: side conv + glo, plain side stream", use_graph=False)
    mode.update(side=None)
    run("eager: baseline", use_graph=False)
    sys.exit(0)
if os.environ.get("SHADOW_ONLY"):
    m = os.environ["SHADOW_ONLY"]
    mode.update(side=None if m == "base" else streams["lo"], glo=False)
    run("eager: " + m, n=40, use_graph=False, main=streams["hi"] if m == "prio" else None)
    sys.exit(0)
for ug in (False, True):
    tag = "graph" if ug else "eager"
    mode.update(side=None)
    run(f"{tag}: baseline", use_graph=ug)
    mode.update(side=streams["lo"], glo=False)
    run(f"{tag}: + side conv (equal priority)", use_graph=ug)
    mode.update(side=streams["lo"], glo=True)
    run(f"{tag}: + side conv + glo conv", use_graph=ug)
    mode.update(side=streams["lo"], glo=False)
    run(f"{tag}: + side conv, main stream high priority", use_graph=ug, main=streams["hi"])
    mode.update(side=None)
    run(f"{tag}: baseline on high-priority stream", use_graph=ug, main=streams["hi"])
