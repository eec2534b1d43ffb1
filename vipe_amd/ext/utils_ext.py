"""`vipe_ext.utils_ext` (csrc/utils_ext/utils_bind.cpp:23-24): `nearest_neighbours(query, tree, knn)` - exact k nearest
neighbours, squared L2, as the reference's kd-tree search returns them (knn.cu:27-67), by an LDS-tiled brute-force HIP
kernel.  Off the update iteration: used by `SLAMMap.project_map(infill=True)`."""
import torch

from .._lib import check, check_gpu_contig, lib, ptr, require, stream_ptr


def nearest_neighbours(query, tree, knn):
    """query [M, <=3] f32, tree [N, <=3] f32 (device) -> [dist [M,knn] f32 (squared L2, ascending), indices [M,knn] int32]."""
    check_gpu_contig(query, tree)
    require(query.dtype == torch.float32 and tree.dtype == torch.float32, "query / tree must be float tensors")
    require(query.dim() == 2 and tree.dim() == 2 and 1 <= query.shape[1] <= 3 and 1 <= tree.shape[1] <= 3,
            "points of 1..3 coordinates")
    require(tree.shape[0] >= knn, "knn is too small compared to the size of point cloud!")  # knn.cu:32
    if not 1 <= knn <= 8:
        raise NotImplementedError("nearest_neighbours: 1 <= knn <= 8")
    M = query.shape[0]
    dist = torch.empty((M, knn), dtype=torch.float32, device=query.device)
    idx = torch.empty((M, knn), dtype=torch.int32, device=query.device)
    check(lib().vipe_nearest_neighbours(ptr(query), query.shape[1], ptr(tree), tree.shape[1], M, tree.shape[0], int(knn),
                                        ptr(dist), ptr(idx), stream_ptr(query)), "nearest_neighbours")
    return [dist, idx]
