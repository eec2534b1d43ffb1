// Test infrastructure (oracle/): python binding of the reference's OWN CPU correlation sampler, compiled from where it
// lies - /root/reference/csrc/corr_ext/correlation.cpp (correlation_cpp_forward :68-108, correlation_cpp_backward :110-148)
// - by oracle/build_ref.py into oracle/_ref/ (build container only; nothing of the reference is copied).  Used by
// tests/golden/make_golden.py to freeze input / output vectors for `vipe_amd.ext.corr_ext`.
#include <torch/extension.h>

#include <vector>

torch::Tensor correlation_cpp_forward(torch::Tensor input1, torch::Tensor input2, int kH, int kW, int patchH, int patchW,
                                      int padH, int padW, int dilationH, int dilationW, int dilation_patchH,
                                      int dilation_patchW, int dH, int dW);
std::vector<torch::Tensor> correlation_cpp_backward(torch::Tensor input1, torch::Tensor input2, torch::Tensor gradOutput,
                                                    int kH, int kW, int patchH, int patchW, int padH, int padW,
                                                    int dilationH, int dilationW, int dilation_patchH,
                                                    int dilation_patchW, int dH, int dW);

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("forward", &correlation_cpp_forward);
  m.def("backward", &correlation_cpp_backward);
}
