"""Test infrastructure: build what of the reference compiles here from its own sources (build container only).

    python oracle/build_ref.py [/root/reference]

* `ref_corr` = /root/reference/csrc/corr_ext/correlation.cpp (the CPU path of `corr_ext`, torch headers only) + the
  binding stub oracle/ref_corr_bind.cpp -> oracle/_ref/ref_corr.so (git-ignored).  g++ through torch's cpp_extension.
Not buildable here (recorded in DESIGN.md section 2): lietorch_cpu.cpp (needs Eigen, absent), scatter.cpp (calls
scatter_cuda unconditionally), everything in .cu files (no nvcc).  Nothing is copied out of the reference tree.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")


def build_ref_corr(ref="/root/reference", verbose=False):
    src = os.path.join(ref, "csrc", "corr_ext", "correlation.cpp")
    if not os.path.exists(src):
        return None
    from torch.utils.cpp_extension import load
    os.makedirs(OUT, exist_ok=True)
    return load(name="ref_corr", sources=[src, os.path.join(HERE, "ref_corr_bind.cpp")], build_directory=OUT,
                extra_cflags=["-O2"], verbose=verbose)


def load_ref_corr():
    """the module if oracle/_ref/ref_corr.so exists (it travels to the GPU box with the tree), else None"""
    path = os.path.join(OUT, "ref_corr.so")
    if not os.path.exists(path):
        return None
    import importlib.util

    import torch  # noqa: F401 - libtorch must be loaded first
    spec = importlib.util.spec_from_file_location("ref_corr", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    m = build_ref_corr(sys.argv[1] if len(sys.argv) > 1 else "/root/reference", verbose=True)
    print("ref_corr:", "built" if m is not None else "reference sources not found")
