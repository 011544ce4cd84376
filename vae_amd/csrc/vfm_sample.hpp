// vfm_sample.hpp -- k_sample_rec: the sample records of the software-pipelined step, made from the tables.
// Included inside `namespace vfm { namespace {` of vfm_bwd.hip (one object per link function).
#pragma once

// Software-pipelined step (VFM_FLAG_ZREC): the records of a batch's entities, indexed BY ENTITY ID --
// zrec[e] = (w, weighted KL of e, 0, 0 | z[0..d-1]), weighted KL = KL(q_e || N(0,1)) / occ(e) * n_g / W_g (what every
// occurrence of e adds to the loss).  This kernel makes them from the tables (first step of a run, or whenever no
// backward has prepared them); in steady state the fused backward of the previous step writes them while the
// updated rows are still in its registers (k_bwd<PIPE>).
template <int LPE, int CPL, int VEC, int LINK>
__global__ __launch_bounds__(BLOCK) void k_sample_rec(const KArgs a, const int32_t* __restrict__ ids, int n,
                                                      float* __restrict__ zrec) {
  constexpr int GPB = BLOCK / LPE;
  __shared__ float sh_cs[VFM_MAX_FIELDS];
  __shared__ int64_t sh_hi[VFM_MAX_FIELDS];
  if ((int)threadIdx.x < a.G) {
    sh_cs[threadIdx.x] = (float)(a.group_n[threadIdx.x] / a.W[threadIdx.x]);
    sh_hi[threadIdx.x] = a.group_hi[threadIdx.x];
  }
  __syncthreads();
  const int lig = threadIdx.x % LPE;
  const int d = a.d;
  const int C = (d + VEC - 1) / VEC;
  const int64_t xs = 4 + (((int64_t)d + 3) & ~(int64_t)3);
  for (int i = blockIdx.x * GPB + threadIdx.x / LPE; i < n; i += gridDim.x * GPB) {
    const uint32_t e = (uint32_t)ids[i];
    const float* row = a.entity + (size_t)e * (2 * (size_t)d);
    float* rec = zrec + (size_t)e * xs;
    float epw = 0.f, klv = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int j = lig + c * LPE;
      if (j < C) {
        const Chunk<VEC> mu = ld_chunk<VEC>(row + (size_t)j * VEC);
        const Chunk<VEC> s = ld_chunk<VEC>(row + d + (size_t)j * VEC);
        Chunk<VEC> ep, z;
        float nb;
        eps_of_chunk<VEC>(a.key, e, j, ep.v, nb);
        if (c == 0) epw = nb;
#pragma unroll
        for (int t = 0; t < VEC; ++t) {
          const float sg = link_f<LINK>(s.v[t]);
          z.v[t] = fmaf(sg, ep.v[t], mu.v[t]);
          klv += kl_std_normal(mu.v[t], sg);
        }
        st_chunk<VEC>(rec + 4 + (size_t)j * VEC, z);
      }
    }
    const float2 th = *reinterpret_cast<const float2*>(a.bias + 2 * (size_t)e);
    const float sgw = link_f<LINK>(th.y);
    if (lig == 0) klv += kl_std_normal(th.x, sgw);
    klv = group_sum<LPE>(klv);
    if (lig == 0) {
      const float cs = sh_cs[group_index(sh_hi, a.G, (int64_t)e)];
      *reinterpret_cast<float4*>(rec) = make_float4(fmaf(sgw, epw, th.x), klv * (cs * a.inv_occ[e]), 0.f, 0.f);
    }
  }
}
