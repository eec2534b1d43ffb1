"""corr_ext: spatial correlation sampler (reference: csrc/corr_ext/correlation_sampler.cpp:82-85).  Device tensors
run the HIP kernels; CPU tensors (float32) run the library's host loops, as the reference dispatches them to its CPU
implementation (correlation_sampler.cpp:44-58)."""

import torch

from .._lib import DTYPE_CODE, check, lib, ptr, require, stream_ptr


def _out_hw(H, W, kH, kW, padH, padW, dilH, dilW, dH, dW):
    oH = (H + 2 * padH - (kH - 1) * dilH - 1) // dH + 1
    oW = (W + 2 * padW - (kW - 1) * dilW - 1) // dW + 1
    return oH, oW


def _check(*tensors):
    dev = tensors[0].device
    for t in tensors:
        require(t.is_contiguous(), "tensor must be contiguous")
        require(t.device == dev, "tensors must live on one device")  # correlation_sampler.cpp:36-42


def forward(input1, input2, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dil_patchH, dil_patchW, dH, dW):
    _check(input1, input2)
    require(input1.shape == input2.shape and input1.dtype == input2.dtype, "inputs must match")
    B, C, H, W = input1.shape
    oH, oW = _out_hw(H, W, kH, kW, padH, padW, dilH, dilW, dH, dW)
    out = torch.empty((B, patchH, patchW, oH, oW), dtype=input1.dtype, device=input1.device)
    geom = (B, C, H, W, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dil_patchH, dil_patchW, dH, dW)
    if input1.is_cuda:
        require(input1.dtype in (torch.float16, torch.float32), "half/float")
        check(lib().vipe_corr_sampler_forward(ptr(input1), ptr(input2), ptr(out), *geom, DTYPE_CODE[input1.dtype],
                                              stream_ptr(input1)), "corr_ext.forward")
    else:
        require(input1.dtype == torch.float32, "CPU tensors: float32")
        check(lib().vipe_corr_sampler_forward_host(ptr(input1), ptr(input2), ptr(out), *geom), "corr_ext.forward")
    return out


def backward(input1, input2, grad_output, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dil_patchH, dil_patchW, dH, dW):
    _check(input1, input2, grad_output)
    B, C, H, W = input1.shape
    g1 = torch.zeros_like(input1)
    g2 = torch.zeros_like(input2)
    geom = (B, C, H, W, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dil_patchH, dil_patchW, dH, dW)
    if input1.is_cuda:
        check(lib().vipe_corr_sampler_backward(ptr(input1), ptr(input2), ptr(grad_output), ptr(g1), ptr(g2), *geom,
                                               DTYPE_CODE[input1.dtype], stream_ptr(input1)), "corr_ext.backward")
    else:
        require(input1.dtype == torch.float32, "CPU tensors: float32")
        check(lib().vipe_corr_sampler_backward_host(ptr(input1), ptr(input2), ptr(grad_output), ptr(g1), ptr(g2), *geom),
              "corr_ext.backward")
    return [g1, g2]
