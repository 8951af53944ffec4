// torch_binding.cpp -- the Python extension module `_pvcnn_backend`: the reference plugin's pybind11 surface
// (experiments/model/pvcnn/modules/functional/src/bindings.cpp:10-37 -- 12 functions taking and returning at::Tensor)
// as a thin host-side binding over the C ABI of libbdm_hip.so (include/bdm_hip.h, section 1).  With this module on the
// path, the reference's functional/backend.py reduces to `import _pvcnn_backend as _backend` and its functional/*.py
// wrappers, modules/*.py and networks run unchanged on MI355X.
//
// Conventions of the reference are kept: tensors must be contiguous, on the GPU, fp32 / int32 (TORCH_CHECK ->
// RuntimeError, utils.hpp:7-18); outputs are allocated here on the inputs' device; work is enqueued on torch's current
// stream.  A non-zero status of the C ABI raises RuntimeError with bdm_last_error() (the reference prints and exit(-1)s).
#include <torch/extension.h>

#include <ATen/hip/HIPContext.h>

#include <vector>

#include "../../include/bdm_hip.h"

namespace {

#define CHECK_GPU(x) TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor")
#define CHECK_CONTIGUOUS(x) TORCH_CHECK((x).is_contiguous(), #x " must be a contiguous tensor")
#define CHECK_IS_INT(x) TORCH_CHECK((x).scalar_type() == at::ScalarType::Int, #x " must be an int tensor")
#define CHECK_IS_FLOAT(x) TORCH_CHECK((x).scalar_type() == at::ScalarType::Float, #x " must be a float tensor")
#define CHECK_F(x) do { CHECK_GPU(x); CHECK_CONTIGUOUS(x); CHECK_IS_FLOAT(x); } while (0)
#define CHECK_I(x) do { CHECK_GPU(x); CHECK_CONTIGUOUS(x); CHECK_IS_INT(x); } while (0)

// Every binding makes the tensor's device current for its duration: the C ABI launches on the CURRENT device, so cuda:1 tensors
// while cuda:0 is current would otherwise be launched on the wrong GPU (the reference's shims rely on at::cuda guards likewise).
#define ON_DEVICE_OF(x) const c10::DeviceGuard bdm_device_guard((x).device())

void *stream_of(const at::Tensor &t) {
  return (void *)at::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.get_device()).stream();
}
void ok(int rc, const char *what) {
  if (rc != 0) {
    const char *msg = bdm_last_error();
    TORCH_CHECK(false, what, " failed (code ", rc, "): ", msg ? msg : "");
  }
}
at::TensorOptions fopt(const at::Tensor &like) { return at::device(like.device()).dtype(at::ScalarType::Float); }
at::TensorOptions iopt(const at::Tensor &like) { return at::device(like.device()).dtype(at::ScalarType::Int); }

// sampling.cpp:6-41
at::Tensor gather_features_forward(at::Tensor features, at::Tensor indices) {
  CHECK_F(features); CHECK_I(indices);
  ON_DEVICE_OF(features);
  const int b = features.size(0), c = features.size(1), n = features.size(2), m = indices.size(1);
  at::Tensor out = torch::empty({b, c, m}, fopt(features));
  ok(bdm_gather_features_forward(b, c, n, m, features.data_ptr<float>(), indices.data_ptr<int>(), out.data_ptr<float>(),
                                 stream_of(features)), "gather_features_forward");
  return out;
}
at::Tensor gather_features_backward(at::Tensor grad_y, at::Tensor indices, const int n) {
  CHECK_F(grad_y); CHECK_I(indices);
  ON_DEVICE_OF(grad_y);
  const int b = grad_y.size(0), c = grad_y.size(1), m = grad_y.size(2);
  at::Tensor gx = torch::empty({b, c, n}, fopt(grad_y));
  ok(bdm_gather_features_backward(b, c, n, m, grad_y.data_ptr<float>(), indices.data_ptr<int>(), gx.data_ptr<float>(),
                                  stream_of(grad_y)), "gather_features_backward");
  return gx;
}
// sampling.cpp:43-58
at::Tensor furthest_point_sampling_forward(at::Tensor coords, const int num_samples) {
  CHECK_F(coords);
  ON_DEVICE_OF(coords);
  const int b = coords.size(0), n = coords.size(2);
  at::Tensor idx = torch::zeros({b, num_samples}, iopt(coords));
  ok(bdm_furthest_point_sampling(b, n, num_samples, coords.data_ptr<float>(), idx.data_ptr<int>(), nullptr, stream_of(coords)),
     "furthest_point_sampling");
  return idx;
}
// ball_query.cpp:6-30
at::Tensor ball_query_forward(at::Tensor centers_coords, at::Tensor points_coords, const float radius, const int num_neighbors) {
  CHECK_F(centers_coords); CHECK_F(points_coords);
  ON_DEVICE_OF(centers_coords);
  const int b = centers_coords.size(0), m = centers_coords.size(2), n = points_coords.size(2);
  at::Tensor out = torch::empty({b, m, num_neighbors}, iopt(points_coords));
  ok(bdm_ball_query(b, n, m, radius, num_neighbors, centers_coords.data_ptr<float>(), points_coords.data_ptr<float>(),
                    out.data_ptr<int>(), stream_of(points_coords)), "ball_query");
  return out;
}
// grouping.cpp:6-44
at::Tensor grouping_forward(at::Tensor features, at::Tensor indices) {
  CHECK_F(features); CHECK_I(indices);
  ON_DEVICE_OF(features);
  const int b = features.size(0), c = features.size(1), n = features.size(2), m = indices.size(1), u = indices.size(2);
  at::Tensor out = torch::empty({b, c, m, u}, fopt(features));
  ok(bdm_grouping_forward(b, c, n, m, u, features.data_ptr<float>(), indices.data_ptr<int>(), out.data_ptr<float>(),
                          stream_of(features)), "grouping_forward");
  return out;
}
at::Tensor grouping_backward(at::Tensor grad_y, at::Tensor indices, const int n) {
  CHECK_F(grad_y); CHECK_I(indices);
  ON_DEVICE_OF(grad_y);
  const int b = grad_y.size(0), c = grad_y.size(1), m = indices.size(1), u = indices.size(2);
  at::Tensor gx = torch::empty({b, c, n}, fopt(grad_y));
  ok(bdm_grouping_backward(b, c, n, m, u, grad_y.data_ptr<float>(), indices.data_ptr<int>(), gx.data_ptr<float>(),
                           stream_of(grad_y)), "grouping_backward");
  return gx;
}
// neighbor_interpolate.cpp:6-66
std::vector<at::Tensor> three_nearest_neighbors_interpolate_forward(at::Tensor points_coords, at::Tensor centers_coords,
                                                                    at::Tensor centers_features) {
  CHECK_F(points_coords); CHECK_F(centers_coords); CHECK_F(centers_features);
  ON_DEVICE_OF(points_coords);
  const int b = centers_features.size(0), c = centers_features.size(1), m = centers_features.size(2), n = points_coords.size(2);
  at::Tensor idx = torch::empty({b, 3, n}, iopt(points_coords)), w = torch::empty({b, 3, n}, fopt(points_coords));
  at::Tensor out = torch::empty({b, c, n}, fopt(points_coords));
  ok(bdm_three_nn_interpolate_forward(b, c, m, n, points_coords.data_ptr<float>(), centers_coords.data_ptr<float>(),
                                      centers_features.data_ptr<float>(), out.data_ptr<float>(), idx.data_ptr<int>(),
                                      w.data_ptr<float>(), stream_of(points_coords)), "three_nearest_neighbors_interpolate_forward");
  return {out, idx, w};
}
at::Tensor three_nearest_neighbors_interpolate_backward(at::Tensor grad_y, at::Tensor indices, at::Tensor weights, const int m) {
  CHECK_F(grad_y); CHECK_I(indices); CHECK_F(weights);
  ON_DEVICE_OF(grad_y);
  const int b = grad_y.size(0), c = grad_y.size(1), n = grad_y.size(2);
  at::Tensor gx = torch::empty({b, c, m}, fopt(grad_y));
  ok(bdm_three_nn_interpolate_backward(b, c, n, m, grad_y.data_ptr<float>(), indices.data_ptr<int>(), weights.data_ptr<float>(),
                                       gx.data_ptr<float>(), stream_of(grad_y)), "three_nearest_neighbors_interpolate_backward");
  return gx;
}
// trilinear_devox.cpp:18-83
std::vector<at::Tensor> trilinear_devoxelize_forward(const int r, const bool is_training, const at::Tensor coords,
                                                     const at::Tensor features) {
  CHECK_F(features); CHECK_F(coords);
  ON_DEVICE_OF(features);
  const int b = features.size(0), c = features.size(1), n = coords.size(2);
  at::Tensor outs = torch::empty({b, c, n}, fopt(features));
  if (is_training) {
    at::Tensor inds = torch::empty({b, 8, n}, iopt(features)), wgts = torch::empty({b, 8, n}, fopt(features));
    ok(bdm_trilinear_devoxelize_forward_training(b, c, n, r, coords.data_ptr<float>(), features.data_ptr<float>(),
                                                 outs.data_ptr<float>(), inds.data_ptr<int>(), wgts.data_ptr<float>(),
                                                 stream_of(features)), "trilinear_devoxelize_forward");
    return {outs, inds, wgts};
  }
  ok(bdm_trilinear_devoxelize_forward(b, c, n, r, coords.data_ptr<float>(), features.data_ptr<float>(), outs.data_ptr<float>(),
                                      stream_of(features)), "trilinear_devoxelize_forward");
  return {outs, torch::zeros({1}, iopt(features)), torch::zeros({1}, fopt(features))};
}
at::Tensor trilinear_devoxelize_backward(const at::Tensor grad_y, const at::Tensor indices, const at::Tensor weights, const int r) {
  CHECK_F(grad_y); CHECK_I(indices); CHECK_F(weights);
  ON_DEVICE_OF(grad_y);
  const int b = grad_y.size(0), c = grad_y.size(1), n = grad_y.size(2);
  at::Tensor gx = torch::empty({b, c, r * r * r}, fopt(grad_y));
  ok(bdm_trilinear_devoxelize_backward(b, c, n, r, indices.data_ptr<int>(), weights.data_ptr<float>(), grad_y.data_ptr<float>(),
                                       gx.data_ptr<float>(), stream_of(grad_y)), "trilinear_devoxelize_backward");
  return gx;
}
// vox.cpp:17-69
std::vector<at::Tensor> avg_voxelize_forward(const at::Tensor features, const at::Tensor coords, const int resolution) {
  CHECK_F(features); CHECK_I(coords);
  ON_DEVICE_OF(features);
  const int b = features.size(0), c = features.size(1), n = features.size(2), r = resolution, r3 = r * r * r;
  at::Tensor ind = torch::empty({b, n}, iopt(features)), cnt = torch::empty({b, r3}, iopt(features));
  at::Tensor out = torch::empty({b, c, r3}, fopt(features));
  at::Tensor ws = torch::empty({(int64_t)bdm_voxelize_workspace_bytes(b, n, r)}, at::device(features.device()).dtype(at::ScalarType::Byte));
  ok(bdm_avg_voxelize_forward(b, c, n, r, features.data_ptr<float>(), coords.data_ptr<int>(), out.data_ptr<float>(),
                              ind.data_ptr<int>(), cnt.data_ptr<int>(), ws.data_ptr(), stream_of(features)), "avg_voxelize_forward");
  return {out, ind, cnt};
}
at::Tensor avg_voxelize_backward(const at::Tensor grad_y, const at::Tensor indices, const at::Tensor cnt) {
  CHECK_F(grad_y); CHECK_I(indices); CHECK_I(cnt);
  ON_DEVICE_OF(grad_y);
  const int b = grad_y.size(0), c = grad_y.size(1), s = grad_y.size(2), n = indices.size(1);
  int r = 1;
  while (r * r * r < s) ++r;
  TORCH_CHECK(r * r * r == s, "grad_y must be (B, C, r^3)");
  at::Tensor gx = torch::empty({b, c, n}, fopt(grad_y));
  ok(bdm_avg_voxelize_backward(b, c, n, r, indices.data_ptr<int>(), cnt.data_ptr<int>(), grad_y.data_ptr<float>(),
                               gx.data_ptr<float>(), stream_of(grad_y)), "avg_voxelize_backward");
  return gx;
}

}  // namespace

// the same 12 names as bindings.cpp:10-37
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("gather_features_forward", &gather_features_forward, "Gather Centers' Features forward (HIP)");
  m.def("gather_features_backward", &gather_features_backward, "Gather Centers' Features backward (HIP)");
  m.def("furthest_point_sampling", &furthest_point_sampling_forward, "Furthest Point Sampling (HIP)");
  m.def("ball_query", &ball_query_forward, "Ball Query (HIP)");
  m.def("grouping_forward", &grouping_forward, "Grouping Features forward (HIP)");
  m.def("grouping_backward", &grouping_backward, "Grouping Features backward (HIP)");
  m.def("three_nearest_neighbors_interpolate_forward", &three_nearest_neighbors_interpolate_forward,
        "3 Nearest Neighbors Interpolate forward (HIP)");
  m.def("three_nearest_neighbors_interpolate_backward", &three_nearest_neighbors_interpolate_backward,
        "3 Nearest Neighbors Interpolate backward (HIP)");
  m.def("trilinear_devoxelize_forward", &trilinear_devoxelize_forward, "Trilinear Devoxelization forward (HIP)");
  m.def("trilinear_devoxelize_backward", &trilinear_devoxelize_backward, "Trilinear Devoxelization backward (HIP)");
  m.def("avg_voxelize_forward", &avg_voxelize_forward, "Voxelization forward with average pooling (HIP)");
  m.def("avg_voxelize_backward", &avg_voxelize_backward, "Voxelization backward (HIP)");
}
