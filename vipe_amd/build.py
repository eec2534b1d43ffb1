"""Build libvipe_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m vipe_amd.build [--force] [-j N]

Objects go to vipe_amd/lib/obj/, the library to vipe_amd/lib/libvipe_amd.so (git-ignored, but it
travels to the GPU box with the working tree).
"""

import argparse
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libvipe_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off",
         "-fno-fast-math", "-Wno-unused-result", "-munsafe-fp-atomics"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_hash(src):
    h = hashlib.sha1()
    for f in [src] + sorted(os.path.join(CSRC, x) for x in os.listdir(CSRC) if x.endswith((".cuh", ".h", ".inc"))):
        h.update(open(f, "rb").read())
    h.update(open(os.path.join(os.path.dirname(HERE), "include", "vipe_amd.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src, force):
    obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
    stamp = obj + ".sha1"
    want = _deps_hash(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    open(stamp, "w").write(want)
    return obj, True


def build(force=False, jobs=4, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res)
    if changed or not os.path.exists(LIB):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[vipe_amd.build] {len(srcs)} sources, {'rebuilt' if changed else 'up to date'} -> {LIB}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=4)
    a = ap.parse_args()
    build(a.force, a.j)
