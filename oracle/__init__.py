"""CPU oracle for the ViPE dense-SLAM update iteration (TEST INFRASTRUCTURE ONLY).

Every module here is a CPU restatement (numpy / torch-CPU) of one piece of the
reference hot path, each function citing the reference file:line it follows
(paths relative to the zixunh/vipe tree).  Nothing under ``vipe_amd/`` imports
this package: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and only as the checker.

Pinning status (see DESIGN.md "Oracle"):
  * geometry / Jacobians / block Gauss-Newton BA / UpdateModule / CorrBlock /
    camera models / edge selection: pinned against outputs of the reference's
    own Python files run in the build container (``tests/golden/make_golden.py``
    loads them by path; the native ``vipe.ext.lietorch`` cannot be built there -
    no Eigen, no nvcc - so ``oracle.se3`` stands in for it, which means SE3
    closed forms themselves are pinned only by group identities and fp64
    self-consistency, not by a reference binary: "SE3 parity unpinned").
  * corr_index_forward / altcorr_forward / frame_distance / depth_filter are CUDA
    kernels with no CPU path and no reference tests: restated from the kernel
    text, cross-checked against independent formulations (F.grid_sample,
    brute-force loops); "parity unpinned" by any reference output.
"""
