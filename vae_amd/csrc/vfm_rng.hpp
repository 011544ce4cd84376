// vfm_rng.hpp -- counter-based eps stream (Philox4x32-10 + Box-Muller).
// Included inside `namespace vfm { namespace {` of a translation unit (see vfm_args.hpp).
#pragma once

// ---------------------------------------------------------------------------------------
// Counter-based RNG: Philox4x32-10 (Salmon et al. 2011) + Box-Muller on the hardware
// transcendental units (v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32).  The draw of entity e at
// step `step` depends on (seed, step, e, coordinate) only: every row -- and every rank -- that
// touches e regenerates the same eps, no eps tensor is ever stored.
// ---------------------------------------------------------------------------------------
// (struct RngKey: vfm_args.hpp.)  Variational sample s > 0 of a step uses counter word 3 =
// step_hi | (s << 16) (steps stay below 2^48 when S > 1; checked on the host): one scalar OR per launch.
__device__ __forceinline__ RngKey key_of_sample(RngKey k, int s) {
  k.step_hi |= (uint32_t)s << 16;
  return k;
}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Box-Muller pair from RB radius bits and AB angle bits (v_sin / v_cos take revolutions).  Both
// uniforms sit at bin centres: u1 in (0,1) so log is finite, and no angle lands exactly on an axis.
template <int RB, int AB>
__device__ __forceinline__ void box_muller_bits(uint32_t f, float& n0, float& n1) {
  const float u1 = fmaf((float)((f >> AB) & ((1u << RB) - 1u)), 1.0f / (float)(1u << RB), 0.5f / (float)(1u << RB));
  const float u2 = fmaf((float)(f & ((1u << AB) - 1u)), 1.0f / (float)(1u << AB), 0.5f / (float)(1u << AB));
  const float r = __builtin_amdgcn_sqrtf(-2.0f * LN2 * __builtin_amdgcn_logf(u1));
  n0 = r * __builtin_amdgcn_cosf(u2);
  n1 = r * __builtin_amdgcn_sinf(u2);
}

// One Philox4x32-10 call with counter (p, e, step_lo, step_hi) and key (seed_lo, seed_hi) yields
// 128 bits = four 26-bit fields (16-bit radius, 10-bit angle) + one 24-bit field (16 + 8):
//   n[0..7] : eps of embedding coordinates 8p .. 8p+7 of entity e at this step (four Box-Muller pairs)
//   nb      : eps of the entity's first-order weight (first normal of the fifth pair), used for p == 0
// The global-bias eps is n[0] of the reserved id e = 0xFFFFFFFF, p = 0 (one draw per entity per
// step, as the reference's per-unique-entity rsample, vfm-torch.py:207-208,238-245).
__device__ __forceinline__ void normal8b(const RngKey& k, uint32_t e, uint32_t p, float n[8], float& nb) {
  uint32_t o[4];
  philox4x32_10(p, e, k.step_lo, k.step_hi, k.seed_lo, k.seed_hi, o);
  box_muller_bits<16, 10>(o[0], n[0], n[1]);                                        // o0[25:0]
  box_muller_bits<16, 10>(__builtin_amdgcn_alignbit(o[1], o[0], 26), n[2], n[3]);   // o1[19:0] : o0[31:26]
  box_muller_bits<16, 10>(__builtin_amdgcn_alignbit(o[2], o[1], 20), n[4], n[5]);   // o2[13:0] : o1[31:20]
  box_muller_bits<16, 10>(__builtin_amdgcn_alignbit(o[3], o[2], 14), n[6], n[7]);   // o3[7:0]  : o2[31:14]
  float unused;
  box_muller_bits<16, 8>(o[3] >> 8, nb, unused);                                    // o3[31:8]
}

// eps of chunk j (VEC coordinates from j*VEC) for the lane that owns it, plus the bias eps
template <int VEC>
__device__ __forceinline__ void eps_of_chunk(const RngKey& k, uint32_t e, int j, float (&ep)[VEC], float& nb) {
  float n[8];
  if constexpr (VEC == 4) {
    const uint32_t jg = (uint32_t)j + k.chunk_off;      // global chunk (chunk_off is even: same parity)
    normal8b(k, e, jg >> 1, n, nb);
    const bool odd = j & 1;
#pragma unroll
    for (int t = 0; t < 4; ++t) ep[t] = odd ? n[4 + t] : n[t];
  } else {
    normal8b(k, e, (uint32_t)j >> 3, n, nb);
    float v = n[0];
#pragma unroll
    for (int t = 1; t < 8; ++t) v = ((j & 7) == t) ? n[t] : v;
    ep[0] = v;
  }
}
