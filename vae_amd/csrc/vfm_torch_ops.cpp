// vfm_torch_ops.cpp -- torch.ops.vfm_hip.*: a thin TORCH_LIBRARY shim over the C ABI of
// libvfm_hip.so (include/vfm_hip.h).  It only validates tensors (device, dtype, contiguity, shapes),
// takes the current HIP stream of the tensors' device and forwards raw pointers; all arithmetic is in
// the HIP kernels.  Schemas mirror the C entry points one to one.
#include <ATen/ATen.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

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
                      int64_t n_samples = 1, int64_t coord_off = 0) {
  TORCH_CHECK(F >= 1 && F <= VFM_MAX_FIELDS, "F out of range");
  TORCH_CHECK(n_samples >= 1 && n_samples <= 64, "n_samples out of range [1,64]");
  TORCH_CHECK((int64_t)group_hi.size() == F && (int64_t)group_n.size() == F, "group_hi / group_n need F entries");
  vfm_problem_t p{};
  p.B = B; p.B_global = B_global; p.T = T; p.nb_train = nb_train;
  p.F = (int32_t)F; p.d = (int32_t)d; p.likelihood = (int32_t)likelihood; p.id_bits = (int32_t)id_bits;
  p.n_samples = (int32_t)n_samples; p.flags = (int32_t)flags;
  for (int64_t g = 0; g < F; ++g) { p.group_hi[g] = group_hi[g]; p.group_n[g] = group_n[g]; }
  p.seed = (uint64_t)seed; p.step = (uint64_t)step;
  p.coord_off = (int32_t)coord_off;
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
              int64_t flags, int64_t seed, int64_t step, int64_t n_samples, int64_t coord_off,
              const optional<Tensor>& wrec, const optional<Tensor>& dev_step) {
  const Ids id = ids_of(x);
  dev_tensor(entity, at::kFloat, "entity_params"); dev_tensor(bias, at::kFloat, "bias_params");
  dev_tensor(scalars, at::kFloat, "scalars"); dev_tensor(pred, at::kFloat, "pred");
  dev_tensor(partials, at::kDouble, "partials");
  TORCH_CHECK(entity.dim() == 2 && bias.dim() == 2 && bias.size(1) == 2 && bias.size(0) == entity.size(0) &&
              entity.size(1) % 2 == 0, "table shapes");
  TORCH_CHECK(n_samples >= 1 && pred.numel() >= n_samples * id.B && partials.numel() >= VFM_PARTIALS_LEN &&
              scalars.numel() >= 3, "output sizes");
  if (flags & VFM_FLAG_PARTIAL_PRED)
    TORCH_CHECK(pred.numel() >= id.B + VFM_MAX_FWD_BLOCKS, "VFM_FLAG_PARTIAL_PRED: pred needs B + VFM_MAX_FWD_BLOCKS floats");
  const int64_t T = entity.size(0), d = entity.size(1) / 2;
  if (sumz.has_value() && sumz->defined()) TORCH_CHECK(sumz->numel() >= n_samples * id.B * d, "sumz too small");
  if (eps_entity.has_value() && eps_entity->defined())
    TORCH_CHECK(eps_entity->numel() >= n_samples * T * d && eps_bias.has_value() && eps_bias->defined() &&
                eps_bias->numel() >= n_samples * T && eps_global.has_value() && eps_global->defined() &&
                eps_global->numel() >= n_samples, "eps tables too small for n_samples");
  c10::hip::HIPGuard guard(x.get_device());
  vfm_problem_t p = problem(x, id.B, B_global, T, id.F, d, nb_train, likelihood, id.id_bits, flags, group_hi,
                            group_n, seed, step, n_samples, coord_off);
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
  vfm_problem_t p{};
  p.B = 0; p.B_global = B_global; p.T = 1; p.nb_train = nb_train; p.F = 1; p.d = 4; p.id_bits = 64;
  TORCH_CHECK(n_samples >= 1 && n_samples <= 64, "n_samples out of range [1,64]");
  p.n_samples = (int32_t)n_samples; p.flags = (int32_t)flags; p.group_hi[0] = 1; p.group_n[0] = 1;
  check(vfm_elbo_finalize_f32(&p, partials.data_ptr<double>(), scalars.data_ptr<float>(), loss.data_ptr<float>(),
                              stream_of(partials)),
        "vfm_elbo_finalize_f32");
}

int64_t rec_len(int64_t d) { return 4 + ((d + 3) / 4) * 4; }

// index = [occ_ptr, occ_rows] or [occ_ptr, occ_rows, heavy_ids, heavy_items, heavy_acc], each optionally followed by
// [touched_ids] (3 or 6 tensors: the batch's entities as a sorted list, vfm_index_t.touched_ids)
vfm_index_t index_of(at::TensorList index_all, int64_t T, int64_t B, int64_t F, int64_t d, int64_t n_samples = 1) {
  // (a trailing HOST int32 tensor, if any: [max work items of one heavy entity] -- vfm_index_t.max_items)
  int32_t max_items = 0;
  if (index_all.size() > 0 && index_all[index_all.size() - 1].is_cpu()) {
    const at::Tensor& meta = index_all[index_all.size() - 1];
    TORCH_CHECK(meta.scalar_type() == at::kInt && meta.numel() >= 1, "index: the trailing host tensor holds int32 [max_items]");
    max_items = meta.data_ptr<int32_t>()[0];
    index_all = index_all.slice(0, index_all.size() - 1);
  }
  TORCH_CHECK(index_all.size() == 2 || index_all.size() == 3 || index_all.size() == 5 || index_all.size() == 6,
              "index = [occ_ptr, occ_rows] (+ [heavy_ids, heavy_items, heavy_acc]) (+ [touched_ids])");
  const bool has_touched = index_all.size() == 3 || index_all.size() == 6;
  at::TensorList index = index_all.slice(0, index_all.size() - (has_touched ? 1 : 0));
  dev_tensor(index[0], at::kInt, "occ_ptr"); dev_tensor(index[1], at::kInt, "occ_rows");
  TORCH_CHECK(index[0].numel() == T + 1 && index[1].numel() == B * F, "inverted index sizes");
  vfm_index_t ix{};
  ix.occ_ptr = index[0].data_ptr<int32_t>(); ix.occ_rows = index[1].data_ptr<int32_t>();
  if (index.size() == 5 && index[2].numel() > 0) {
    dev_tensor(index[2], at::kInt, "heavy_ids"); dev_tensor(index[3], at::kInt, "heavy_items");
    dev_tensor(index[4], at::kFloat, "heavy_acc");
    TORCH_CHECK(index[3].numel() % 4 == 0 && index[4].numel() >= n_samples * (index[2].numel() + index[3].numel() / 4) * rec_len(d),
                "heavy index sizes");
    ix.heavy_ids = index[2].data_ptr<int32_t>(); ix.heavy_items = index[3].data_ptr<int32_t>();
    ix.heavy_acc = index[4].data_ptr<float>();
    ix.n_heavy = (int32_t)index[2].numel(); ix.n_items = (int32_t)(index[3].numel() / 4);
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
                     const optional<Tensor>& eps_bias, const optional<Tensor>& eps_global, int64_t coord_off = 0) {
  dev_tensor(entity, at::kFloat, "entity_params"); dev_tensor(bias, at::kFloat, "bias_params");
  const int64_t T = entity.size(0), d = entity.size(1) / 2;
  TORCH_CHECK(n_samples >= 1 && sumz.numel() >= n_samples * B * d, "sumz too small for n_samples");
  if (eps_entity.has_value() && eps_entity->defined())
    TORCH_CHECK(eps_entity->numel() >= n_samples * T * d && eps_bias.has_value() && eps_bias->defined() &&
                eps_bias->numel() >= n_samples * T && eps_global.has_value() && eps_global->defined() &&
                eps_global->numel() >= n_samples, "eps tables too small for n_samples");
  return {problem(entity, B, B_global, T, F, d, nb_train, likelihood, 64, flags, group_hi, group_n, seed, step,
                  n_samples, coord_off),
          index_of(index, T, B, F, d, n_samples)};
}

void elbo_bwd(at::TensorList index, const Tensor& entity, const Tensor& bias,
              const Tensor& inv_occ, const Tensor& scalars, const Tensor& W, const optional<Tensor>& eps_entity,
              const optional<Tensor>& eps_bias, const optional<Tensor>& eps_global, const Tensor& sumz,
              const Tensor& grow, const Tensor& partials, const Tensor& grad_out, Tensor g_entity, Tensor g_bias,
              Tensor g_scalars, int64_t F, at::IntArrayRef group_hi, at::ArrayRef<double> group_n,
              int64_t nb_train, int64_t B_global, int64_t likelihood, int64_t flags, int64_t seed, int64_t step,
              int64_t n_samples, int64_t coord_off) {
  const int64_t B = grow.numel();
  BwdCommon c = bwd_common(index, entity, bias, B, F, B_global, nb_train, likelihood, flags, group_hi,
                           group_n, seed, step, n_samples, sumz, eps_entity, eps_bias, eps_global, coord_off);
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
                   int64_t coord_off, const optional<Tensor>& wrec, const optional<Tensor>& dev_step) {
  const int64_t B = grow.numel();
  BwdCommon c = bwd_common(index, entity, bias, B, F, B_global, nb_train, likelihood, flags, group_hi,
                           group_n, seed, step, n_samples, sumz, eps_entity, eps_bias, eps_global, coord_off);
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
  vfm_problem_t p{};
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
                     double beta2, double eps_adam, int64_t adam_step, int64_t e_lo, int64_t e_hi,
                     int64_t own_mod, int64_t own_rank, const optional<Tensor>& kl_ws,
                     const optional<Tensor>& rec_ptr, const optional<Tensor>& rec_pos) {
  dev_tensor(entity, at::kFloat, "entity_params"); dev_tensor(bias, at::kFloat, "bias_params");
  const int64_t T = entity.size(0), d = entity.size(1) / 2;
  const int64_t n_rec = own_mod > 1 ? (T - own_rank + own_mod - 1) / own_mod : T;
  const bool gather = rec_ptr.has_value() && rec_ptr->defined();
  const int32_t* rp = nullptr; const int32_t* rq = nullptr;
  if (gather) {
    TORCH_CHECK(rec_pos.has_value() && rec_pos->defined() && rec_ptr->numel() == n_rec + 1 &&
                acc.numel() >= rec_pos->numel() * rec_len(d), "gather index sizes");
    rp = dev_tensor(*rec_ptr, at::kInt, "rec_ptr").data_ptr<int32_t>();
    rq = dev_tensor(*rec_pos, at::kInt, "rec_pos").data_ptr<int32_t>();
  } else {
    TORCH_CHECK(acc.numel() >= n_rec * rec_len(d), "apply_adam sizes");
  }
  TORCH_CHECK(sums.numel() >= 2, "sums size");
  double* klp = nullptr;
  if (kl_ws.has_value() && kl_ws->defined()) {
    TORCH_CHECK(kl_ws->numel() >= 4097, "kl_ws needs 4097 doubles");
    klp = dev_tensor(*kl_ws, at::kDouble, "kl_ws").data_ptr<double>();
  }
  TORCH_CHECK(m_entity.numel() == entity.numel() && v_entity.numel() == entity.numel() &&
              m_bias.numel() == bias.numel() && v_bias.numel() == bias.numel(), "Adam moment shapes");
  c10::hip::HIPGuard guard(entity.get_device());
  vfm_problem_t p = problem(entity, 0, B_global, T, F, d, nb_train, likelihood, 64, flags, group_hi, group_n, seed, step);
  p.e_lo = e_lo; p.e_hi = e_hi; p.own_mod = (int32_t)own_mod; p.own_rank = (int32_t)own_rank;
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
            (float)eps_adam, adam_step, klp, rp, rq, stream_of(entity)),
        "vfm_elbo_apply_adam_f32");
}

// ---- entity-sharded mode ----
void elbo_fwd_zpre(const Tensor& x, const Tensor& y, const Tensor& zbuf, const Tensor& scalars,
                   const optional<Tensor>& eps_global, Tensor pred, Tensor partials, Tensor sumz, Tensor grow,
                   int64_t d, int64_t nb_train, int64_t B_global, int64_t likelihood, int64_t flags, int64_t seed,
                   int64_t step) {
  const Ids id = ids_of(x);
  dev_tensor(zbuf, at::kFloat, "zbuf"); dev_tensor(scalars, at::kFloat, "scalars");
  dev_tensor(pred, at::kFloat, "pred"); dev_tensor(partials, at::kDouble, "partials");
  TORCH_CHECK(zbuf.numel() % rec_len(d) == 0, "zbuf must hold whole records");
  const int64_t U = zbuf.numel() / rec_len(d);
  TORCH_CHECK(pred.numel() >= id.B && partials.numel() >= VFM_PARTIALS_LEN && sumz.numel() >= id.B * d &&
              grow.numel() >= id.B, "output sizes");
  c10::hip::HIPGuard guard(x.get_device());
  std::vector<int64_t> hi((size_t)id.F, U > 0 ? U : 1);
  std::vector<double> gn((size_t)id.F, 1.0);
  vfm_problem_t p = problem(x, id.B, B_global, U > 0 ? U : 1, id.F, d, nb_train, likelihood, id.id_bits,
                            flags | VFM_FLAG_ZPRE, hi, gn, seed, step);
  check(vfm_elbo_fwd_f32(&p, id.ptr, dev_tensor(y, at::kFloat, "y").data_ptr<float>(), zbuf.data_ptr<float>(),
                         nullptr, nullptr, scalars.data_ptr<float>(), nullptr, nullptr, nullptr,
                         fptr(eps_global, "eps_global"), pred.data_ptr<float>(), partials.data_ptr<double>(),
                         dev_tensor(sumz, at::kFloat, "sumz").data_ptr<float>(),
                         dev_tensor(grow, at::kFloat, "grow").data_ptr<float>(), stream_of(x)),
        "vfm_elbo_fwd_f32 (ZPRE)");
}

void shard_sample(const Tensor& ids, const Tensor& entity, const Tensor& bias, const optional<Tensor>& eps_entity,
                  const optional<Tensor>& eps_bias, Tensor out, int64_t seed, int64_t step, int64_t flags) {
  dev_tensor(ids, at::kInt, "ids"); dev_tensor(entity, at::kFloat, "entity_params");
  dev_tensor(bias, at::kFloat, "bias_params"); dev_tensor(out, at::kFloat, "out");
  const int64_t T = entity.size(0), d = entity.size(1) / 2, n = ids.numel();
  TORCH_CHECK(out.numel() >= n * rec_len(d), "out too small");
  c10::hip::HIPGuard guard(entity.get_device());
  std::vector<int64_t> hi(1, T);
  std::vector<double> gn(1, 1.0);
  vfm_problem_t p = problem(entity, 0, 0, T, 1, d, 1, 0, 64, flags, hi, gn, seed, step);
  check(vfm_shard_sample_f32(&p, ids.data_ptr<int32_t>(), n, entity.data_ptr<float>(), bias.data_ptr<float>(),
                             fptr(eps_entity, "eps_entity"), fptr(eps_bias, "eps_bias"), out.data_ptr<float>(),
                             stream_of(entity)),
        "vfm_shard_sample_f32");
}

void shard_pack(Tensor small, const Tensor& loss_local, const Tensor& kl_ws) {
  c10::hip::HIPGuard guard(small.get_device());
  check(vfm_shard_pack_f32(dev_tensor(small, at::kFloat, "small").data_ptr<float>(),
                           dev_tensor(loss_local, at::kFloat, "loss_local").data_ptr<float>(),
                           dev_tensor(kl_ws, at::kDouble, "kl_ws").data_ptr<double>(), stream_of(small)),
        "vfm_shard_pack_f32");
}

void shard_loss(const Tensor& small, Tensor loss3) {
  c10::hip::HIPGuard guard(small.get_device());
  check(vfm_shard_loss_f32(dev_tensor(small, at::kFloat, "small").data_ptr<float>(),
                           dev_tensor(loss3, at::kFloat, "loss3").data_ptr<float>(), stream_of(small)),
        "vfm_shard_loss_f32");
}

void records_add(Tensor dst, const Tensor& idx, const Tensor& src, int64_t d, bool atomic) {
  dev_tensor(dst, at::kFloat, "dst"); dev_tensor(idx, at::kInt, "idx"); dev_tensor(src, at::kFloat, "src");
  TORCH_CHECK(src.numel() >= idx.numel() * rec_len(d), "src too small");
  c10::hip::HIPGuard guard(dst.get_device());
  check(vfm_records_add_f32(dst.data_ptr<float>(), idx.data_ptr<int32_t>(), src.data_ptr<float>(), idx.numel(),
                            (int32_t)d, atomic ? 1 : 0, stream_of(dst)),
        "vfm_records_add_f32");
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

void elbo_lik(const Tensor& y, const Tensor& scalars, const optional<Tensor>& eps_global, Tensor pred, Tensor grow,
              Tensor partials, int64_t nb_train, int64_t B_global, int64_t likelihood, int64_t flags, int64_t seed,
              int64_t step) {
  dev_tensor(y, at::kFloat, "y"); dev_tensor(scalars, at::kFloat, "scalars"); dev_tensor(pred, at::kFloat, "pred");
  dev_tensor(grow, at::kFloat, "grow"); dev_tensor(partials, at::kDouble, "partials");
  const int64_t B = y.numel();
  TORCH_CHECK(pred.numel() >= B + VFM_MAX_FWD_BLOCKS && grow.numel() >= B && partials.numel() >= VFM_PARTIALS_LEN &&
              scalars.numel() >= 3, "elbo_lik sizes (pred holds B row values + VFM_MAX_FWD_BLOCKS KL shares)");
  c10::hip::HIPGuard guard(y.get_device());
  vfm_problem_t p{};
  p.B = B; p.B_global = B_global; p.T = 1; p.nb_train = nb_train; p.F = 1; p.d = 4; p.id_bits = 64;
  p.likelihood = (int32_t)likelihood; p.n_samples = 1; p.flags = (int32_t)flags; p.group_hi[0] = 1; p.group_n[0] = 1;
  p.seed = (uint64_t)seed; p.step = (uint64_t)step;
  check(vfm_elbo_lik_f32(&p, y.data_ptr<float>(), scalars.data_ptr<float>(), fptr(eps_global, "eps_global"),
                         pred.data_ptr<float>(), grow.data_ptr<float>(), partials.data_ptr<double>(), stream_of(y)),
        "vfm_elbo_lik_f32");
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
        "int likelihood, int flags, int seed, int step, int n_samples=1, int coord_off=0, Tensor(e!)? wrec=None, "
        "Tensor(f!)? dev_step=None) -> ()", &elbo_fwd);
  m.def("elbo_finalize(Tensor(a!) partials, Tensor scalars, Tensor(b!) loss, int nb_train, int B_global, int flags, "
        "int n_samples=1) -> ()",
        &elbo_finalize);
  m.def("elbo_bwd(Tensor[] index, Tensor entity_params, Tensor bias_params, Tensor inv_occ, "
        "Tensor scalars, Tensor W, Tensor? eps_entity, Tensor? eps_bias, Tensor? eps_global, Tensor sumz, Tensor grow, "
        "Tensor partials, Tensor grad_out, Tensor(a!) g_entity, Tensor(b!) g_bias, Tensor(c!) g_scalars, int F, "
        "int[] group_hi, float[] group_n, int nb_train, int B_global, int likelihood, int flags, int seed, int step, "
        "int n_samples=1, int coord_off=0) -> ()",
        &elbo_bwd);
  m.def("elbo_bwd_adam(Tensor[] index, Tensor(a!) entity_params, Tensor(b!) bias_params, "
        "Tensor(c!) scalars, Tensor inv_occ, Tensor W, Tensor? eps_entity, Tensor? eps_bias, Tensor? eps_global, "
        "Tensor sumz, Tensor grow, Tensor partials, Tensor(d!) m_entity, Tensor(e!) v_entity, Tensor(f!) m_bias, "
        "Tensor(g!) v_bias, Tensor(h!) m_scalars, Tensor(i!) v_scalars, int F, int[] group_hi, float[] group_n, "
        "int nb_train, int B_global, int likelihood, int flags, int seed, int step, float lr, float beta1, "
        "float beta2, float eps_adam, int adam_step, Tensor(j!)? loss, int n_samples=1, int coord_off=0, "
        "Tensor(k!)? wrec=None, Tensor(l!)? dev_step=None) -> ()",
        &elbo_bwd_adam);
  m.def("elbo_bwd_acc(Tensor[] index, Tensor sumz, Tensor grow, Tensor partials, Tensor(a!) acc, "
        "Tensor(c!) sums, int T, int F, int d, int e_lo, int e_hi) -> ()", &elbo_bwd_acc);
  m.def("elbo_apply_adam(Tensor acc, Tensor sums, Tensor(a!) entity_params, Tensor(b!) bias_params, "
        "Tensor(c!) scalars, Tensor inv_occ, Tensor W, Tensor? eps_entity, Tensor? eps_bias, Tensor? eps_global, "
        "Tensor(d!) m_entity, Tensor(e!) v_entity, Tensor(f!) m_bias, Tensor(g!) v_bias, Tensor(h!) m_scalars, "
        "Tensor(i!) v_scalars, int F, int[] group_hi, float[] group_n, int nb_train, int B_global, int likelihood, "
        "int flags, int seed, int step, float lr, float beta1, float beta2, float eps_adam, int adam_step, "
        "int e_lo, int e_hi, int own_mod, int own_rank, Tensor(j!)? kl_ws, Tensor? rec_ptr, Tensor? rec_pos) -> ()",
        &elbo_apply_adam);
  m.def("elbo_fwd_zpre(Tensor x, Tensor y, Tensor zbuf, Tensor scalars, Tensor? eps_global, Tensor(a!) pred, "
        "Tensor(b!) partials, Tensor(c!) sumz, Tensor(d!) grow, int d, int nb_train, int B_global, int likelihood, "
        "int flags, int seed, int step) -> ()", &elbo_fwd_zpre);
  m.def("shard_sample(Tensor ids, Tensor entity_params, Tensor bias_params, Tensor? eps_entity, Tensor? eps_bias, "
        "Tensor(a!) out, int seed, int step, int flags=0) -> ()", &shard_sample);
  m.def("records_add(Tensor(a!) dst, Tensor idx, Tensor src, int d, bool atomic) -> ()", &records_add);
  m.def("shard_pack(Tensor(a!) small, Tensor loss_local, Tensor kl_ws) -> ()", &shard_pack);
  m.def("shard_loss(Tensor small, Tensor(a!) loss3) -> ()", &shard_loss);
  m.def("elbo_lik(Tensor y, Tensor scalars, Tensor? eps_global, Tensor(a!) pred, Tensor(b!) grow, Tensor(c!) partials, "
        "int nb_train, int B_global, int likelihood, int flags, int seed, int step) -> ()", &elbo_lik);
  m.def("moments_rescale(Tensor(a!) m, Tensor(b!) v, float beta1, float beta2, int step, bool to_scaled) -> ()",
        &moments_rescale);
  m.def("adam(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) v, float lr, float beta1, float beta2, float eps, "
        "int step) -> ()", &adam);
}
