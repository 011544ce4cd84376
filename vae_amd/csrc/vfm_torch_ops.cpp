// vfm_torch_ops.cpp -- torch.ops.vfm_hip.*: a thin TORCH_LIBRARY shim over the C ABI of
// libvfm_hip.so (include/vfm_hip.h).  It only validates tensors (device, dtype, contiguity, shapes),
// takes the current HIP stream of the tensors' device and forwards raw pointers; all arithmetic is in
// the HIP kernels.  Schemas mirror the C entry points one to one.
#include <ATen/ATen.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <string.h>

#include "vfm_hip.h"

namespace {

using at::Tensor;
using c10::optional;

void check(int rc, const char* what) {
  TORCH_CHECK(rc == 0, what, " failed (code ", rc, "): ", vfm_last_error());
}

const Tensor& dev_tensor(const Tensor& t, at::ScalarType dt, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must be on the GPU (vae_amd has no CPU fallback)");
  TORCH_CHECK(t.scalar_type() == dt, name, " has the wrong dtype");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
  return t;
}
const float* fptr(const optional<Tensor>& t, const char* name) {
  if (!t.has_value() || !t->defined()) return nullptr;
  return dev_tensor(*t, at::kFloat, name).data_ptr<float>();
}
float* fptr_mut(const optional<Tensor>& t, const char* name) { return const_cast<float*>(fptr(t, name)); }

vfm_problem_t problem(const Tensor& x_like, int64_t B, int64_t B_global, int64_t T, int64_t F, int64_t d,
                      int64_t nb_train, int64_t likelihood, int64_t id_bits, int64_t flags,
                      at::IntArrayRef group_hi, at::ArrayRef<double> group_n, int64_t seed, int64_t step,
                      int64_t n_samples = 1) {
  TORCH_CHECK(F >= 1 && F <= VFM_MAX_FIELDS, "F out of range");
  TORCH_CHECK(n_samples >= 1 && n_samples <= 64, "n_samples out of range [1,64]");
  TORCH_CHECK((int64_t)group_hi.size() == F && (int64_t)group_n.size() == F, "group_hi / group_n need F entries");
  vfm_problem_t p;
  VFM_STRUCT_INIT(p);
  p.B = B; p.B_global = B_global; p.T = T; p.nb_train = nb_train;
  p.F = (int32_t)F; p.d = (int32_t)d; p.likelihood = (int32_t)likelihood; p.id_bits = (int32_t)id_bits;
  p.n_samples = (int32_t)n_samples; p.flags = (int32_t)flags;
  for (int64_t g = 0; g < F; ++g) { p.group_hi[g] = group_hi[g]; p.group_n[g] = group_n[g]; }
  p.seed = (uint64_t)seed; p.step = (uint64_t)step;
  return p;
}

// optional device-side extensions of vfm_problem_t: packed first-order records [T,4], device step state (64 bytes)
void extensions(vfm_problem_t& p, const optional<Tensor>& wrec, const optional<Tensor>& dev_step, int64_t T) {
  if (wrec.has_value() && wrec->defined()) {
    TORCH_CHECK(dev_tensor(*wrec, at::kFloat, "wrec").numel() >= 4 * T, "wrec needs [T,4] floats");
    p.wrec = wrec->data_ptr<float>();
  }
  if (dev_step.has_value() && dev_step->defined()) {
    TORCH_CHECK(dev_tensor(*dev_step, at::kLong, "dev_step").numel() >= 8, "dev_step needs 8 int64 (vfm_dev_step_t)");
    p.dev_step = reinterpret_cast<vfm_dev_step_t*>(dev_step->data_ptr<int64_t>());
  }
}

struct Ids {
  const void* ptr; int64_t B, F; int id_bits;
};
Ids ids_of(const Tensor& x) {
  TORCH_CHECK(x.is_cuda() && x.is_contiguous() && x.dim() == 2, "x must be a contiguous [B,F] GPU tensor");
  TORCH_CHECK(x.scalar_type() == at::kLong || x.scalar_type() == at::kInt, "x must be int64 or int32");
  return {x.data_ptr(), x.size(0), x.size(1), x.scalar_type() == at::kLong ? 64 : 32};
}

void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

void elbo_fwd(const Tensor& x, const optional<Tensor>& y, const Tensor& entity, const Tensor& bias,
              const optional<Tensor>& inv_occ, const Tensor& scalars, const optional<Tensor>& W,
              const optional<Tensor>& eps_entity, const optional<Tensor>& eps_bias,
              const optional<Tensor>& eps_global, Tensor pred, Tensor partials,
              const optional<Tensor>& sumz, const optional<Tensor>& grow, at::IntArrayRef group_hi,
              at::ArrayRef<double> group_n, int64_t nb_train, int64_t B_global, int64_t likelihood,
              int64_t flags, int64_t seed, int64_t step, int64_t n_samples,
              const optional<Tensor>& wrec, const optional<Tensor>& dev_step) {
  const Ids id = ids_of(x);
  dev_tensor(entity, at::kFloat, "entity_params"); dev_tensor(bias, at::kFloat, "bias_params");
  dev_tensor(scalars, at::kFloat, "scalars"); dev_tensor(pred, at::kFloat, "pred");
  dev_tensor(partials, at::kDouble, "partials");
  TORCH_CHECK(entity.dim() == 2 && bias.dim() == 2 && bias.size(1) == 2 && bias.size(0) == entity.size(0) &&
              entity.size(1) % 2 == 0, "table shapes");
  TORCH_CHECK(n_samples >= 1 && pred.numel() >= n_samples * id.B && partials.numel() >= VFM_PARTIALS_LEN &&
              scalars.numel() >= 3, "output sizes");
  const int64_t T = entity.size(0), d = entity.size(1) / 2;
  if (sumz.has_value() && sumz->defined()) TORCH_CHECK(sumz->numel() >= n_samples * id.B * d, "sumz too small");
  if (eps_entity.has_value() && eps_entity->defined())
    TORCH_CHECK(eps_entity->numel() >= n_samples * T * d && eps_bias.has_value() && eps_bias->defined() &&
                eps_bias->numel() >= n_samples * T && eps_global.has_value() && eps_global->defined() &&
                eps_global->numel() >= n_samples, "eps tables too small for n_samples");
  c10::hip::HIPGuard guard(x.get_device());
  vfm_problem_t p = problem(x, id.B, B_global, T, id.F, d, nb_train, likelihood, id.id_bits, flags, group_hi,
                            group_n, seed, step, n_samples);
  extensions(p, wrec, dev_step, T);
  const double* Wp = (W.has_value() && W->defined()) ? dev_tensor(*W, at::kDouble, "W").data_ptr<double>() : nullptr;
  check(vfm_elbo_fwd_f32(&p, id.ptr, fptr(y, "y"), entity.data_ptr<float>(), bias.data_ptr<float>(),
                         fptr(inv_occ, "inv_occ"), scalars.data_ptr<float>(), Wp, fptr(eps_entity, "eps_entity"),
                         fptr(eps_bias, "eps_bias"), fptr(eps_global, "eps_global"), pred.data_ptr<float>(),
                         partials.data_ptr<double>(), fptr_mut(sumz, "sumz"), fptr_mut(grow, "grow"),
                         stream_of(x)),
        "vfm_elbo_fwd_f32");
}

void elbo_finalize(Tensor partials, const Tensor& scalars, Tensor loss, int64_t nb_train, int64_t B_global,
                   int64_t flags, int64_t n_samples) {
  dev_tensor(partials, at::kDouble, "partials"); dev_tensor(scalars, at::kFloat, "scalars");
  dev_tensor(loss, at::kFloat, "loss");
  TORCH_CHECK(loss.numel() >= 3, "loss needs 3 entries");
  c10::hip::HIPGuard guard(partials.get_device());
  vfm_problem_t p;
  VFM_STRUCT_INIT(p);
  p.B = 0; p.B_global = B_global; p.T = 1; p.nb_train = nb_train; p.F = 1; p.d = 4; p.id_bits = 64;
  TORCH_CHECK(n_samples >= 1 && n_samples <= 64, "n_samples out of range [1,64]");
  p.n_samples = (int32_t)n_samples; p.flags = (int32_t)flags; p.group_hi[0] = 1; p.group_n[0] = 1;
  check(vfm_elbo_finalize_f32(&p, partials.data_ptr<double>(), scalars.data_ptr<float>(), loss.data_ptr<float>(),
                              stream_of(partials)),
        "vfm_elbo_finalize_f32");
}

int64_t rec_len(int64_t d) { return 4 + ((d + 3) / 4) * 4; }

// index = [occ_ptr, occ_rows, status] (+ [heavy_ids, heavy_items, heavy_acc]) (+ [touched_ids]) (+ a HOST int32 tensor
// [max work items of one heavy entity, work-item length, heavy threshold]): status = one DEVICE int32 word (vfm_index_t.status: the kernels count the index
// entries they had to clamp there), touched_ids = the batch's entities as a sorted list (vfm_index_t.touched_ids)
vfm_index_t index_of(at::TensorList index_all, int64_t T, int64_t B, int64_t F, int64_t d, int64_t n_samples = 1) {
  int32_t max_items = 0, heavy_list = 0, heavy_threshold = 0;
  if (index_all.size() > 0 && index_all[index_all.size() - 1].is_cpu()) {
    const at::Tensor& meta = index_all[index_all.size() - 1];
    TORCH_CHECK(meta.scalar_type() == at::kInt && meta.numel() >= 3,
                "index: the trailing host tensor holds int32 [max_items, heavy_list, heavy_threshold]");
    max_items = meta.data_ptr<int32_t>()[0];
    heavy_list = meta.data_ptr<int32_t>()[1];
    heavy_threshold = meta.data_ptr<int32_t>()[2];
    index_all = index_all.slice(0, index_all.size() - 1);
  }
  TORCH_CHECK(index_all.size() == 3 || index_all.size() == 4 || index_all.size() == 6 || index_all.size() == 7,
              "index = [occ_ptr, occ_rows, status] (+ [heavy_ids, heavy_items, heavy_acc]) (+ [touched_ids])");
  const bool has_touched = index_all.size() == 4 || index_all.size() == 7;
  const bool has_heavy = index_all.size() >= 6;
  dev_tensor(index_all[0], at::kInt, "occ_ptr"); dev_tensor(index_all[1], at::kInt, "occ_rows");
  dev_tensor(index_all[2], at::kInt, "index status");
  TORCH_CHECK(index_all[0].numel() == T + 1 && index_all[1].numel() == B * F && index_all[2].numel() == 1, "inverted index sizes");
  vfm_index_t ix;
  VFM_STRUCT_INIT(ix);
  ix.occ_ptr = index_all[0].data_ptr<int32_t>(); ix.occ_rows = index_all[1].data_ptr<int32_t>();
  ix.status = index_all[2].data_ptr<int32_t>();
  ix.heavy_list = heavy_list; ix.heavy_threshold = heavy_threshold;
  if (has_heavy && index_all[3].numel() > 0) {
    const at::Tensor &hid = index_all[3], &items = index_all[4], &hacc = index_all[5];
    dev_tensor(hid, at::kInt, "heavy_ids"); dev_tensor(items, at::kInt, "heavy_items"); dev_tensor(hacc, at::kFloat, "heavy_acc");
    TORCH_CHECK(items.numel() % 4 == 0 && hacc.numel() >= n_samples * (hid.numel() + items.numel() / 4) * rec_len(d),
                "heavy index sizes");
    ix.heavy_ids = hid.data_ptr<int32_t>(); ix.heavy_items = items.data_ptr<int32_t>(); ix.heavy_acc = hacc.data_ptr<float>();
    ix.n_heavy = (int32_t)hid.numel(); ix.n_items = (int32_t)(items.numel() / 4);
    ix.max_items = max_items;
  }
  if (has_touched) {
    const at::Tensor& t = index_all[index_all.size() - 1];
    dev_tensor(t, at::kInt, "touched_ids");
    ix.touched_ids = t.numel() > 0 ? t.data_ptr<int32_t>() : nullptr; ix.n_touched = t.numel();
  }
  return ix;
}

struct BwdCommon {
  vfm_problem_t p; vfm_index_t ix;
};

BwdCommon bwd_common(at::TensorList index, const Tensor& entity, const Tensor& bias,
                     int64_t B, int64_t F, int64_t B_global, int64_t nb_train, int64_t likelihood, int64_t flags,
                     at::IntArrayRef group_hi, at::ArrayRef<double> group_n, int64_t seed, int64_t step,
                     int64_t n_samples, const Tensor& sumz, const optional<Tensor>& eps_entity,
                     const optional<Tensor>& eps_bias, const optional<Tensor>& eps_global) {
  dev_tensor(entity, at::kFloat, "entity_params"); dev_tensor(bias, at::kFloat, "bias_params");
  const int64_t T = entity.size(0), d = entity.size(1) / 2;
  TORCH_CHECK(n_samples >= 1 && sumz.numel() >= n_samples * B * d, "sumz too small for n_samples");
  if (eps_entity.has_value() && eps_entity->defined())
    TORCH_CHECK(eps_entity->numel() >= n_samples * T * d && eps_bias.has_value() && eps_bias->defined() &&
                eps_bias->numel() >= n_samples * T && eps_global.has_value() && eps_global->defined() &&
                eps_global->numel() >= n_samples, "eps tables too small for n_samples");
  return {problem(entity, B, B_global, T, F, d, nb_train, likelihood, 64, flags, group_hi, group_n, seed, step,
                  n_samples),
          index_of(index, T, B, F, d, n_samples)};
}

void elbo_bwd(at::TensorList index, const Tensor& entity, const Tensor& bias,
              const Tensor& inv_occ, const Tensor& scalars, const Tensor& W, const optional<Tensor>& eps_entity,
              const optional<Tensor>& eps_bias, const optional<Tensor>& eps_global, const Tensor& sumz,
              const Tensor& grow, const Tensor& partials, const Tensor& grad_out, Tensor g_entity, Tensor g_bias,
              Tensor g_scalars, int64_t F, at::IntArrayRef group_hi, at::ArrayRef<double> group_n,
              int64_t nb_train, int64_t B_global, int64_t likelihood, int64_t flags, int64_t seed, int64_t step,
              int64_t n_samples) {
  const int64_t B = grow.numel();
  BwdCommon c = bwd_common(index, entity, bias, B, F, B_global, nb_train, likelihood, flags, group_hi,
                           group_n, seed, step, n_samples, sumz, eps_entity, eps_bias, eps_global);
  TORCH_CHECK(g_entity.sizes() == entity.sizes() && g_bias.sizes() == bias.sizes() && g_scalars.numel() >= 3,
              "gradient shapes");
  c10::hip::HIPGuard guard(entity.get_device());
  check(vfm_elbo_bwd_f32(&c.p, &c.ix, entity.data_ptr<float>(), bias.data_ptr<float>(),
                         dev_tensor(inv_occ, at::kFloat, "inv_occ").data_ptr<float>(),
                         dev_tensor(scalars, at::kFloat, "scalars").data_ptr<float>(),
                         dev_tensor(W, at::kDouble, "W").data_ptr<double>(), fptr(eps_entity, "eps_entity"),
                         fptr(eps_bias, "eps_bias"), fptr(eps_global, "eps_global"),
                         dev_tensor(sumz, at::kFloat, "sumz").data_ptr<float>(),
                         dev_tensor(grow, at::kFloat, "grow").data_ptr<float>(),
                         dev_tensor(partials, at::kDouble, "partials").data_ptr<double>(),
                         dev_tensor(grad_out, at::kFloat, "grad_out").data_ptr<float>(),
                         dev_tensor(g_entity, at::kFloat, "g_entity").data_ptr<float>(),
                         dev_tensor(g_bias, at::kFloat, "g_bias").data_ptr<float>(),
                         dev_tensor(g_scalars, at::kFloat, "g_scalars").data_ptr<float>(), stream_of(entity)),
        "vfm_elbo_bwd_f32");
}

void elbo_bwd_adam(at::TensorList index, Tensor entity, Tensor bias, Tensor scalars,
                   const Tensor& inv_occ, const Tensor& W, const optional<Tensor>& eps_entity,
                   const optional<Tensor>& eps_bias, const optional<Tensor>& eps_global, const Tensor& sumz,
                   const Tensor& grow, const Tensor& partials, Tensor m_entity, Tensor v_entity, Tensor m_bias,
                   Tensor v_bias, Tensor m_scalars, Tensor v_scalars, int64_t F, at::IntArrayRef group_hi,
                   at::ArrayRef<double> group_n, int64_t nb_train, int64_t B_global, int64_t likelihood,
                   int64_t flags, int64_t seed, int64_t step, double lr, double beta1, double beta2,
                   double eps_adam, int64_t adam_step, const optional<Tensor>& loss, int64_t n_samples,
                   const optional<Tensor>& wrec, const optional<Tensor>& dev_step) {
  const int64_t B = grow.numel();
  BwdCommon c = bwd_common(index, entity, bias, B, F, B_global, nb_train, likelihood, flags, group_hi,
                           group_n, seed, step, n_samples, sumz, eps_entity, eps_bias, eps_global);
  extensions(c.p, wrec, dev_step, entity.size(0));
  TORCH_CHECK(m_entity.numel() == entity.numel() && v_entity.numel() == entity.numel() &&
              m_bias.numel() == bias.numel() && v_bias.numel() == bias.numel() && m_scalars.numel() >= 3 &&
              v_scalars.numel() >= 3, "Adam moment shapes");
  c10::hip::HIPGuard guard(entity.get_device());
  check(vfm_elbo_bwd_adam_f32(
            &c.p, &c.ix, entity.data_ptr<float>(), bias.data_ptr<float>(),
            dev_tensor(scalars, at::kFloat, "scalars").data_ptr<float>(),
            dev_tensor(inv_occ, at::kFloat, "inv_occ").data_ptr<float>(),
            dev_tensor(W, at::kDouble, "W").data_ptr<double>(), fptr(eps_entity, "eps_entity"),
            fptr(eps_bias, "eps_bias"), fptr(eps_global, "eps_global"),
            dev_tensor(sumz, at::kFloat, "sumz").data_ptr<float>(), dev_tensor(grow, at::kFloat, "grow").data_ptr<float>(),
            dev_tensor(partials, at::kDouble, "partials").data_ptr<double>(),
            dev_tensor(m_entity, at::kFloat, "m_entity").data_ptr<float>(),
            dev_tensor(v_entity, at::kFloat, "v_entity").data_ptr<float>(),
            dev_tensor(m_bias, at::kFloat, "m_bias").data_ptr<float>(),
            dev_tensor(v_bias, at::kFloat, "v_bias").data_ptr<float>(),
            dev_tensor(m_scalars, at::kFloat, "m_scalars").data_ptr<float>(),
            dev_tensor(v_scalars, at::kFloat, "v_scalars").data_ptr<float>(), (float)lr, (float)beta1, (float)beta2,
            (float)eps_adam, adam_step, fptr_mut(loss, "loss"), stream_of(entity)),
        "vfm_elbo_bwd_adam_f32");
}

void elbo_bwd_acc(at::TensorList index, const Tensor& sumz, const Tensor& grow,
                  const Tensor& partials, Tensor acc, Tensor sums, int64_t T, int64_t F, int64_t d, int64_t e_lo,
                  int64_t e_hi) {
  dev_tensor(acc, at::kFloat, "acc"); dev_tensor(sums, at::kFloat, "sums");
  const int64_t B = grow.numel();
  TORCH_CHECK(acc.numel() >= T * rec_len(d) && sums.numel() >= 2, "bwd_acc sizes");
  vfm_index_t ix = index_of(index, T, B, F, d);
  c10::hip::HIPGuard guard(acc.get_device());
  vfm_problem_t p;
  VFM_STRUCT_INIT(p);
  p.B = B; p.B_global = B; p.T = T; p.nb_train = 1; p.F = (int32_t)F; p.d = (int32_t)d; p.id_bits = 64;
  p.n_samples = 1; p.e_lo = e_lo; p.e_hi = e_hi;
  for (int64_t g = 0; g < F; ++g) { p.group_hi[g] = T; p.group_n[g] = 1; }
  check(vfm_elbo_bwd_acc_f32(&p, &ix,
                             dev_tensor(sumz, at::kFloat, "sumz").data_ptr<float>(),
                             dev_tensor(grow, at::kFloat, "grow").data_ptr<float>(),
                             dev_tensor(partials, at::kDouble, "partials").data_ptr<double>(),
                             acc.data_ptr<float>(), sums.data_ptr<float>(), stream_of(acc)),
        "vfm_elbo_bwd_acc_f32");
}

void elbo_apply_adam(const Tensor& acc, const Tensor& sums, Tensor entity, Tensor bias,
                     Tensor scalars, const Tensor& inv_occ, const Tensor& W, const optional<Tensor>& eps_entity,
                     const optional<Tensor>& eps_bias, const optional<Tensor>& eps_global, Tensor m_entity,
                     Tensor v_entity, Tensor m_bias, Tensor v_bias, Tensor m_scalars, Tensor v_scalars, int64_t F,
                     at::IntArrayRef group_hi, at::ArrayRef<double> group_n, int64_t nb_train, int64_t B_global,
                     int64_t likelihood, int64_t flags, int64_t seed, int64_t step, double lr, double beta1,
                     double beta2, double eps_adam, int64_t adam_step, int64_t e_lo, int64_t e_hi) {
  dev_tensor(entity, at::kFloat, "entity_params"); dev_tensor(bias, at::kFloat, "bias_params");
  const int64_t T = entity.size(0), d = entity.size(1) / 2;
  TORCH_CHECK(acc.numel() >= T * rec_len(d), "apply_adam sizes");
  TORCH_CHECK(sums.numel() >= 2, "sums size");
  TORCH_CHECK(m_entity.numel() == entity.numel() && v_entity.numel() == entity.numel() &&
              m_bias.numel() == bias.numel() && v_bias.numel() == bias.numel(), "Adam moment shapes");
  c10::hip::HIPGuard guard(entity.get_device());
  vfm_problem_t p = problem(entity, 0, B_global, T, F, d, nb_train, likelihood, 64, flags, group_hi, group_n, seed, step);
  p.e_lo = e_lo; p.e_hi = e_hi;
  check(vfm_elbo_apply_adam_f32(
            &p, dev_tensor(acc, at::kFloat, "acc").data_ptr<float>(), dev_tensor(sums, at::kFloat, "sums").data_ptr<float>(),
            entity.data_ptr<float>(), bias.data_ptr<float>(), dev_tensor(scalars, at::kFloat, "scalars").data_ptr<float>(),
            dev_tensor(inv_occ, at::kFloat, "inv_occ").data_ptr<float>(), dev_tensor(W, at::kDouble, "W").data_ptr<double>(),
            fptr(eps_entity, "eps_entity"), fptr(eps_bias, "eps_bias"), fptr(eps_global, "eps_global"),
            dev_tensor(m_entity, at::kFloat, "m_entity").data_ptr<float>(),
            dev_tensor(v_entity, at::kFloat, "v_entity").data_ptr<float>(),
            dev_tensor(m_bias, at::kFloat, "m_bias").data_ptr<float>(), dev_tensor(v_bias, at::kFloat, "v_bias").data_ptr<float>(),
            dev_tensor(m_scalars, at::kFloat, "m_scalars").data_ptr<float>(),
            dev_tensor(v_scalars, at::kFloat, "v_scalars").data_ptr<float>(), (float)lr, (float)beta1, (float)beta2,
            (float)eps_adam, adam_step, stream_of(entity)),
        "vfm_elbo_apply_adam_f32");
}

void adam(Tensor p, const Tensor& g, Tensor m, Tensor v, double lr, double beta1, double beta2, double eps,
          int64_t step) {
  dev_tensor(p, at::kFloat, "p"); dev_tensor(g, at::kFloat, "g"); dev_tensor(m, at::kFloat, "m");
  dev_tensor(v, at::kFloat, "v");
  TORCH_CHECK(g.numel() >= p.numel() && m.numel() == p.numel() && v.numel() == p.numel(), "adam sizes");
  c10::hip::HIPGuard guard(p.get_device());
  check(vfm_adam_f32(p.data_ptr<float>(), g.data_ptr<float>(), m.data_ptr<float>(), v.data_ptr<float>(), p.numel(),
                     (float)lr, (float)beta1, (float)beta2, (float)eps, step, stream_of(p)),
        "vfm_adam_f32");
}

void moments_rescale(Tensor m, Tensor v, double beta1, double beta2, int64_t step, bool to_scaled) {
  dev_tensor(m, at::kFloat, "m"); dev_tensor(v, at::kFloat, "v");
  TORCH_CHECK(m.numel() == v.numel(), "moment sizes");
  c10::hip::HIPGuard guard(m.get_device());
  check(vfm_moments_rescale_f32(m.data_ptr<float>(), v.data_ptr<float>(), m.numel(), (float)beta1, (float)beta2, step,
                                to_scaled ? 1 : 0, stream_of(m)),
        "vfm_moments_rescale_f32");
}

int64_t abi_version() { return vfm_abi_version(); }

}  // namespace

TORCH_LIBRARY(vfm_hip, m) {
  m.def("abi_version() -> int", &abi_version);
  m.def("elbo_fwd(Tensor x, Tensor? y, Tensor entity_params, Tensor bias_params, Tensor? inv_occ, Tensor scalars, "
        "Tensor? W, Tensor? eps_entity, Tensor? eps_bias, Tensor? eps_global, Tensor(a!) pred, Tensor(b!) partials, "
        "Tensor(c!)? sumz, Tensor(d!)? grow, int[] group_hi, float[] group_n, int nb_train, int B_global, "
        "int likelihood, int flags, int seed, int step, int n_samples=1, Tensor(e!)? wrec=None, "
        "Tensor(f!)? dev_step=None) -> ()", &elbo_fwd);
  m.def("elbo_finalize(Tensor(a!) partials, Tensor scalars, Tensor(b!) loss, int nb_train, int B_global, int flags, "
        "int n_samples=1) -> ()",
        &elbo_finalize);
  m.def("elbo_bwd(Tensor[] index, Tensor entity_params, Tensor bias_params, Tensor inv_occ, "
        "Tensor scalars, Tensor W, Tensor? eps_entity, Tensor? eps_bias, Tensor? eps_global, Tensor sumz, Tensor grow, "
        "Tensor partials, Tensor grad_out, Tensor(a!) g_entity, Tensor(b!) g_bias, Tensor(c!) g_scalars, int F, "
        "int[] group_hi, float[] group_n, int nb_train, int B_global, int likelihood, int flags, int seed, int step, "
        "int n_samples=1) -> ()",
        &elbo_bwd);
  m.def("elbo_bwd_adam(Tensor[] index, Tensor(a!) entity_params, Tensor(b!) bias_params, "
        "Tensor(c!) scalars, Tensor inv_occ, Tensor W, Tensor? eps_entity, Tensor? eps_bias, Tensor? eps_global, "
        "Tensor sumz, Tensor grow, Tensor partials, Tensor(d!) m_entity, Tensor(e!) v_entity, Tensor(f!) m_bias, "
        "Tensor(g!) v_bias, Tensor(h!) m_scalars, Tensor(i!) v_scalars, int F, int[] group_hi, float[] group_n, "
        "int nb_train, int B_global, int likelihood, int flags, int seed, int step, float lr, float beta1, "
        "float beta2, float eps_adam, int adam_step, Tensor(j!)? loss, int n_samples=1, "
        "Tensor(k!)? wrec=None, Tensor(l!)? dev_step=None) -> ()",
        &elbo_bwd_adam);
  m.def("elbo_bwd_acc(Tensor[] index, Tensor sumz, Tensor grow, Tensor partials, Tensor(a!) acc, "
        "Tensor(c!) sums, int T, int F, int d, int e_lo, int e_hi) -> ()", &elbo_bwd_acc);
  m.def("elbo_apply_adam(Tensor acc, Tensor sums, Tensor(a!) entity_params, Tensor(b!) bias_params, "
        "Tensor(c!) scalars, Tensor inv_occ, Tensor W, Tensor? eps_entity, Tensor? eps_bias, Tensor? eps_global, "
        "Tensor(d!) m_entity, Tensor(e!) v_entity, Tensor(f!) m_bias, Tensor(g!) v_bias, Tensor(h!) m_scalars, "
        "Tensor(i!) v_scalars, int F, int[] group_hi, float[] group_n, int nb_train, int B_global, int likelihood, "
        "int flags, int seed, int step, float lr, float beta1, float beta2, float eps_adam, int adam_step, "
        "int e_lo, int e_hi) -> ()",
        &elbo_apply_adam);
  m.def("moments_rescale(Tensor(a!) m, Tensor(b!) v, float beta1, float beta2, int step, bool to_scaled) -> ()",
        &moments_rescale);
  m.def("adam(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, float eps, "
        "int step) -> ()", &adam);
}
