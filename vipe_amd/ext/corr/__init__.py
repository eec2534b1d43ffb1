"""`vipe.ext.corr` (vipe/ext/corr/spatial_correlation_sampler.py:13-126): the differentiable spatial correlation sampler
on top of `corr_ext.forward / backward` (HIP kernels for device tensors, the library's host loops for CPU tensors).
Every size argument is an int or an (h, w) pair; output `[B, patchH, patchW, oH, oW]`."""
import torch

from .. import corr_ext


def _hw(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


class SpatialCorrelationSamplerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1, dilation_patch=1):
        # argument order of the native entry points: kernel, patch, padding, dilation, patch dilation, stride
        ctx.sizes = (*_hw(kernel_size), *_hw(patch_size), *_hw(padding), *_hw(dilation), *_hw(dilation_patch), *_hw(stride))
        ctx.save_for_backward(input1, input2)
        return corr_ext.forward(input1, input2, *ctx.sizes)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        g1, g2 = corr_ext.backward(*ctx.saved_tensors, grad_output.contiguous(), *ctx.sizes)
        return (g1, g2) + (None,) * 6


def spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                               dilation_patch=1):
    return SpatialCorrelationSamplerFunction.apply(input1, input2, kernel_size, patch_size, stride, padding, dilation,
                                                   dilation_patch)


class SpatialCorrelationSampler(torch.nn.Module):
    def __init__(self, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1, dilation_patch=1):
        super().__init__()
        self.kernel_size, self.patch_size, self.stride = kernel_size, patch_size, stride
        self.padding, self.dilation, self.dilation_patch = padding, dilation, dilation_patch

    def forward(self, input1, input2):
        return spatial_correlation_sample(input1, input2, self.kernel_size, self.patch_size, self.stride, self.padding,
                                          self.dilation, self.dilation_patch)
