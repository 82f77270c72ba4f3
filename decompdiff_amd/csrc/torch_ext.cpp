// Compiled torch extension: the op-level boundary of the C ABI (BASELINE.json north_star "exposed as torch
// extensions") registered with the dispatcher from C++ -- TORCH_LIBRARY(decompdiff_hip, ...) with CUDA(=HIP)-key kernels
// that launch libdecompdiff_hip.so's entry points on torch's current stream.  Host-only C++ (no device code here): built
// with g++ against the installed torch headers by decompdiff_amd/build.py (build_torch_ext), loaded with
// torch.ops.load_library.  decompdiff_amd/functional.py routes through these ops when the extension is present (the
// ctypes binding of the same entry points otherwise); the reference call sites they replace are cited in
// include/decompdiff_hip.h.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/decompdiff_hip.h"

namespace {

void* cur_stream(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check(int rc, const char* what) { TORCH_CHECK(rc == DD_OK, "decompdiff_hip ", what, " failed: ", dd_status_string(rc)); }

const float* fptr(const at::Tensor& t) { return t.data_ptr<float>(); }

at::Tensor f32c(const at::Tensor& t) { return t.to(at::kFloat).contiguous(); }

// torch_cluster.knn per sample (uni_transformer_edge.py:353): x [B,N,3] -> neighbour lists [B,N,K] int32, ascending (d2, index)
at::Tensor knn(const at::Tensor& x, int64_t k) {
  TORCH_CHECK(x.is_cuda() && x.dim() == 3 && x.size(2) == 3, "knn: x must be a [B,N,3] tensor on a HIP device");
  const at::Tensor xc = f32c(x);
  at::Tensor nbr = at::empty({x.size(0), x.size(1), k}, x.options().dtype(at::kInt));
  check(dd_knn(fptr(xc), (int)x.size(0), (int)x.size(1), (int)k, nbr.data_ptr<int32_t>(), cur_stream(x)), "dd_knn");
  return nbr;
}

// scatter_sum / mean / min / max over dim 0 of [E,F] rows grouped by destination (CSR seg_ptr [n+1] int32); op 0..3
std::tuple<at::Tensor, at::Tensor> segment_reduce(const at::Tensor& src, const at::Tensor& seg_ptr, int64_t op) {
  TORCH_CHECK(src.is_cuda() && src.dim() == 2 && seg_ptr.is_cuda() && seg_ptr.scalar_type() == at::kInt, "segment_reduce: [E,F] fp32 + int32 seg_ptr on HIP");
  const at::Tensor s = f32c(src);
  const int64_t n = seg_ptr.numel() - 1, F = s.size(1), E = s.size(0);
  at::Tensor out = at::empty({n, F}, s.options());
  at::Tensor arg = (op >= 2) ? at::empty({n, F}, s.options().dtype(at::kLong)) : at::empty({0}, s.options().dtype(at::kLong));
  if (n > 0 && F > 0)
    check(dd_segment_reduce(E ? fptr(s) : nullptr, seg_ptr.data_ptr<int32_t>(), (int)n, (int)F, (int)op, (long)E, out.data_ptr<float>(),
                            op >= 2 ? arg.data_ptr<int64_t>() : nullptr, cur_stream(src)), "dd_segment_reduce");
  return {out, arg};
}

at::Tensor segment_softmax(const at::Tensor& src, const at::Tensor& seg_ptr) {
  TORCH_CHECK(src.is_cuda() && src.dim() == 2 && seg_ptr.scalar_type() == at::kInt, "segment_softmax: [E,F] fp32 + int32 seg_ptr on HIP");
  const at::Tensor s = f32c(src);
  at::Tensor out = at::empty_like(s);
  if (s.numel() > 0)
    check(dd_segment_softmax(fptr(s), seg_ptr.data_ptr<int32_t>(), (int)(seg_ptr.numel() - 1), (int)s.size(1), out.data_ptr<float>(),
                             cur_stream(src)), "dd_segment_softmax");
  return out;
}

// scatter_softmax + scatter_sum pair of NodeUpdateLayer / BondUpdateLayer (uni_transformer_edge.py:63-68,158-164)
at::Tensor attn_aggregate_node(const at::Tensor& q, bool q_per_edge, const at::Tensor& k, const at::Tensor& v,
                               const c10::optional<at::Tensor>& e_w, const at::Tensor& seg_ptr) {
  const at::Tensor qc = f32c(q), kc = f32c(k), vc = f32c(v);
  const int64_t n = seg_ptr.numel() - 1;
  at::Tensor out = at::empty({n, 128}, kc.options());
  at::Tensor ew;
  if (e_w.has_value()) ew = f32c(*e_w).reshape({-1});
  check(dd_attn_aggregate_node(fptr(qc), q_per_edge ? 1 : 0, fptr(kc), fptr(vc), e_w.has_value() ? fptr(ew) : nullptr,
                               seg_ptr.data_ptr<int32_t>(), (int)n, out.data_ptr<float>(), cur_stream(k)), "dd_attn_aggregate_node");
  return out;
}

// PosUpdateLayer's pair (uni_transformer_edge.py:205-211): v16 [E,16], rel_x [E,3] -> [n,3]
at::Tensor attn_aggregate_pos(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v16, const c10::optional<at::Tensor>& e_w,
                              const at::Tensor& rel_x, const at::Tensor& seg_ptr) {
  const at::Tensor qc = f32c(q), kc = f32c(k), vc = f32c(v16), rc = f32c(rel_x);
  const int64_t n = seg_ptr.numel() - 1;
  at::Tensor out = at::empty({n, 3}, kc.options());
  at::Tensor ew;
  if (e_w.has_value()) ew = f32c(*e_w).reshape({-1});
  check(dd_attn_aggregate_pos(fptr(qc), fptr(kc), fptr(vc), e_w.has_value() ? fptr(ew) : nullptr, fptr(rc), seg_ptr.data_ptr<int32_t>(),
                              (int)n, out.data_ptr<float>(), cur_stream(k)), "dd_attn_aggregate_pos");
  return out;
}

}  // namespace

TORCH_LIBRARY(decompdiff_hip, m) {
  m.def("knn(Tensor x, int k) -> Tensor");
  m.def("segment_reduce(Tensor src, Tensor seg_ptr, int op) -> (Tensor, Tensor)");
  m.def("segment_softmax(Tensor src, Tensor seg_ptr) -> Tensor");
  m.def("attn_aggregate_node(Tensor q, bool q_per_edge, Tensor k, Tensor v, Tensor? e_w, Tensor seg_ptr) -> Tensor");
  m.def("attn_aggregate_pos(Tensor q, Tensor k, Tensor v16, Tensor? e_w, Tensor rel_x, Tensor seg_ptr) -> Tensor");
  m.def("abi_version() -> int", []() -> int64_t { return dd_abi_version(); });
}

TORCH_LIBRARY_IMPL(decompdiff_hip, CUDA, m) {     // (the CUDA dispatch key is the HIP device on a ROCm build)
  m.impl("knn", &knn);
  m.impl("segment_reduce", &segment_reduce);
  m.impl("segment_softmax", &segment_softmax);
  m.impl("attn_aggregate_node", &attn_aggregate_node);
  m.impl("attn_aggregate_pos", &attn_aggregate_pos);
}
