// Python binding for the reference's two altcorr host launchers, which live in
// /root/reference/src/altcorr_kernel.cu:290-356.  The reference binds them in src/droid.cpp
// (:193-217,246-247) together with the bundle-adjustment code, which needs Eigen and
// lietorch (absent here); this file (own code, test infrastructure) binds ONLY those two
// functions with the same contiguity checks, so that oracle/build_ref.py can build the
// reference's altcorr kernels from the source where it lies.
#include <torch/extension.h>
#include <vector>

std::vector<torch::Tensor> altcorr_cuda_forward(torch::Tensor fmap1, torch::Tensor fmap2, torch::Tensor coords,
                                                int radius);
std::vector<torch::Tensor> altcorr_cuda_backward(torch::Tensor fmap1, torch::Tensor fmap2, torch::Tensor coords,
                                                 torch::Tensor corr_grad, int radius);

static std::vector<torch::Tensor> altcorr_forward(torch::Tensor fmap1, torch::Tensor fmap2, torch::Tensor coords,
                                                  int radius) {
  TORCH_CHECK(fmap1.is_contiguous(), "fmap1 must be contiguous");
  TORCH_CHECK(fmap2.is_contiguous(), "fmap2 must be contiguous");
  TORCH_CHECK(coords.is_contiguous(), "coords must be contiguous");
  return altcorr_cuda_forward(fmap1, fmap2, coords, radius);
}

static std::vector<torch::Tensor> altcorr_backward(torch::Tensor fmap1, torch::Tensor fmap2, torch::Tensor coords,
                                                   torch::Tensor corr_grad, int radius) {
  TORCH_CHECK(fmap1.is_contiguous(), "fmap1 must be contiguous");
  TORCH_CHECK(fmap2.is_contiguous(), "fmap2 must be contiguous");
  TORCH_CHECK(coords.is_contiguous(), "coords must be contiguous");
  TORCH_CHECK(corr_grad.is_contiguous(), "corr_grad must be contiguous");
  return altcorr_cuda_backward(fmap1, fmap2, coords, corr_grad, radius);
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("altcorr_forward", &altcorr_forward, "ALTCORR forward (reference kernel)");
  m.def("altcorr_backward", &altcorr_backward, "ALTCORR backward (reference kernel)");
}
