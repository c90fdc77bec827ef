// Fused cross-attention on the 16-bit matrix cores: the kernels of
// prh_attn.hpp - same wave-per-(segment, head) structure, same online softmax, same dropout
// hash, same in-place K/V column blocks and dK/dV gradient arena - with every product moved
// from v_mfma_f32_32x32x2_f32 (64 cycles per 2-deep step) to v_mfma_f32_32x32x16_{f16,bf16}:
//   PREC 0: every operand split into two fp16 planes, three products (h h' + h l' + l h'), fp32
//           accumulate - fp32-level error (the split of prh_gemm_s3.hpp).  fp16's exponent is
//           narrow, so every 32 x 32 operand tile (Q, K, V, dO as loaded; dS as computed) carries
//           a power-of-two scale taken from ITS OWN largest magnitude (wave-wide max: gradients of
//           1e-6 keep their 22 bits) and each product is unscaled exactly; probabilities (<= 1)
//           are split as they are;
//   PREC 1: one bf16 plane, one product (bf16 mode, BASELINE config 3).
// A key tile costs 12 (forward) / 42 (backward) MFMAs of 32 cycles instead of 32 / 112 of 64:
// the kernels become bound by what they read and write (K, V once; dK, dV once).
//
// Operand maps, v_mfma_f32_32x32x16 (lane l, r = l & 31, h = l >> 5, element j = 0..7):
//   A[row r][k = 8h + j]   B[k = 8h + j][col r]   D[row = crow(reg, h)][col r]
// "Row on the lane" operands (K, Q, V, dO as they lie in memory) are converted from the lane's
// own registers: the lane keeps channels {8h..8h+7} (k-step 0) and {16+8h..16+8h+7} (k-step 1).
// An accumulator tile X is the B operand of a following product that sums over X's ROWS: registers
// 8s..8s+7 form the fragment of k-step s, holding rows crow(8s + j, h) = 16s + 8(j >> 2) + 4h + (j & 3).
// The matching A operand (the transposed tile: V^T, dO^T, Q^T, K^T) is read from a row-major
// 16-bit LDS image with ds_read_b64_tr_b16, whose 4-row blocks are addressed at exactly those rows.
#pragma once
#include "prh_attn.hpp"
#include "prh_gemm_h2.hpp"

namespace prh {

constexpr int A16_ROWB = 80;                    // bytes per image row: 32 x 2 B + 16 B pad (conflict-free tr reads)
constexpr int A16_IMG = 32 * A16_ROWB;          // one plane of a 32 x 32 tile

template <int PREC> struct A16 {
  static constexpr int NPL = PREC == 0 ? 2 : 1;
  using frag = typename std::conditional<PREC == 0, f16x8, bf16x8>::type;
};

// power-of-two scale of a tile from the lane's 16 values (wave-wide maximum); 1 for bf16
template <int PREC>
__device__ __forceinline__ float a16_scale(const float (&x)[16]) {
  if constexpr (PREC != 0) return 1.f;
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) m = fmaxf(m, fabsf(x[t]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return pow2_scale(m);
}
__device__ __forceinline__ float a16_scale_acc(const f32x16& x) {
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) m = fmaxf(m, fabsf(x[t]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return pow2_scale(m);
}
// 8 floats (times S) -> fragment plane(s)
// (NP = 1 with PREC 0: the values are known to fit one fp16 plane exactly - bf16-stored K / V)
template <int PREC, int NP>
__device__ __forceinline__ void a16_split(const float* x_, typename A16<PREC>::frag (&o)[NP], float S = 1.f) {
  if constexpr (PREC == 0 && NP == 1) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    uint4 h;
    unsigned* hp = reinterpret_cast<unsigned*>(&h);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v2f v = {x_[2 * j] * S, x_[2 * j + 1] * S};
      f16x2 hb = __builtin_convertvector(v, f16x2);
      hp[j] = __builtin_bit_cast(unsigned, hb);
    }
    o[0] = __builtin_bit_cast(f16x8, h);
  } else if constexpr (PREC == 0) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = x_[j] * S;
    uint4 h, l;
    split2h(x[0], x[1], h.x, l.x); split2h(x[2], x[3], h.y, l.y);
    split2h(x[4], x[5], h.z, l.z); split2h(x[6], x[7], h.w, l.w);
    o[0] = __builtin_bit_cast(f16x8, h);
    o[1] = __builtin_bit_cast(f16x8, l);
  } else {
    const uint4 h = make_uint4(pack_bf16x2(x_[0], x_[1]), pack_bf16x2(x_[2], x_[3]), pack_bf16x2(x_[4], x_[5]),
                               pack_bf16x2(x_[6], x_[7]));
    o[0] = __builtin_bit_cast(bf16x8, h);
  }
}
template <int PREC, int NA, int NB>
__device__ __forceinline__ f32x16 a16_mma(const typename A16<PREC>::frag (&a)[NA],
                                          const typename A16<PREC>::frag (&b)[NB], f32x16 c) {
  if constexpr (PREC == 0) {
    if constexpr (NA > 1) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[NA - 1], b[0], c, 0, 0, 0);
    if constexpr (NB > 1) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[NB - 1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  }
  return c;
}
// row-on-the-lane load of a 32 x 32 fp32 block: the lane's 16 channels {8h..8h+7, 16+8h..16+8h+7}
__device__ __forceinline__ void a16_load_rows(const float* base, long ld, long row0, int rows_valid, int col0,
                                              int lane, float (&x)[16]) {
  const int r = lane & 31, h2 = lane >> 5;
  const bool ok = r < rows_valid;
  const float* p = base + (size_t)(row0 + (ok ? r : 0)) * ld + col0 + h2 * 8;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4 t = ok ? ldg4(p + s * 16 + 4 * j) : zero4();
      x[s * 8 + 4 * j] = t.x; x[s * 8 + 4 * j + 1] = t.y; x[s * 8 + 4 * j + 2] = t.z; x[s * 8 + 4 * j + 3] = t.w;
    }
}
// K / V rows from fp32 or (KV16) bf16 storage: bf16 values are exact in one fp16 plane once placed by
// the tile's power-of-two scale (8 significant bits of 11), so only the bytes change
template <bool KV16>
__device__ __forceinline__ void a16_load_kv(const float* base, long ld, long row0, int rows_valid, int col0,
                                            int lane, float (&x)[16]) {
  if constexpr (!KV16) {
    a16_load_rows(base, ld, row0, rows_valid, col0, lane, x);
  } else {
    const int r = lane & 31, h2 = lane >> 5;
    const bool ok = r < rows_valid;
    const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + (size_t)(row0 + (ok ? r : 0)) * ld + col0 + h2 * 8;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint4 u = ok ? *reinterpret_cast<const uint4*>(p + s * 16) : make_uint4(0u, 0u, 0u, 0u);
      x[s * 8 + 0] = bf16_lo(u.x); x[s * 8 + 1] = bf16_hi(u.x); x[s * 8 + 2] = bf16_lo(u.y); x[s * 8 + 3] = bf16_hi(u.y);
      x[s * 8 + 4] = bf16_lo(u.z); x[s * 8 + 5] = bf16_hi(u.z); x[s * 8 + 6] = bf16_lo(u.w); x[s * 8 + 7] = bf16_hi(u.w);
    }
  }
}
// the lane's row (16 channels as above) -> 16-bit row-major image plane(s): row r, channel c at r*ROWB + 2c
template <int PREC, int NP>
__device__ __forceinline__ void a16_store_rows(char* img, int lane, const typename A16<PREC>::frag (&f0)[NP],
                                               const typename A16<PREC>::frag (&f1)[NP]) {
  const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) {
    char* q = img + pl * A16_IMG + r * A16_ROWB + h2 * 16;
    *reinterpret_cast<uint4*>(q) = __builtin_bit_cast(uint4, f0[pl]);
    *reinterpret_cast<uint4*>(q + 32) = __builtin_bit_cast(uint4, f1[pl]);
  }
  // the image is wave-private and DS operations of one wave execute in order; this only keeps the
  // compiler from moving the transposed reads (an intrinsic) above the stores
  asm volatile("" ::: "memory");
}
// A operand = the TRANSPOSE of a row-major image: lane (column c = l & 31, half h) receives rows
// 16s + 4h + {0..3} and 16s + 8 + 4h + {0..3} of column c - the rows an accumulator fragment holds
template <int PREC, int NP>
__device__ __forceinline__ void a16_tr(const char* img, int s, int lane, typename A16<PREC>::frag (&o)[NP]) {
  const int g = lane >> 4, j = lane & 15;
  const char* a = img + (16 * s + 4 * (g >> 1) + (j >> 2)) * A16_ROWB + (16 * (g & 1) + 4 * (j & 3)) * 2;
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) {
    const fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(a + pl * A16_IMG));
    const fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
        (__attribute__((address_space(3))) fp16x4*)(a + pl * A16_IMG + 8 * A16_ROWB));
    struct Pair { fp16x4 a, b; } pr = {lo, hi};
    o[pl] = __builtin_bit_cast(typename A16<PREC>::frag, pr);
  }
}
// registers 8s..8s+7 of an accumulator tile -> B-operand fragment plane(s)
template <int PREC, int NP>
__device__ __forceinline__ void a16_acc_frag(const f32x16& x, int s, typename A16<PREC>::frag (&o)[NP],
                                             float S = 1.f) {
  float t[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = x[8 * s + j];
  a16_split<PREC>(t, o, S);
}
__device__ __forceinline__ f32x16 a16_zero() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// ---------------------------------------------------------------------------------------
// forward.  LDS per wave: the V tile image (NPL planes).
// ---------------------------------------------------------------------------------------
// Key-split mode (p.ksplit > 0; small batches, where B x H waves would leave the chip idle): the
// workgroup's waves share one (segment, head), each takes p.ksplit keys, and the partial softmax
// states (running maximum, sum, unnormalised O^T) are merged through LDS once per query tile.
template <int PREC> constexpr int a16_fwd_wave_lds() {
  return A16<PREC>::NPL * A16_IMG > 18 * 64 * 4 ? A16<PREC>::NPL * A16_IMG : 18 * 64 * 4;
}
template <int PREC, bool KV16 = false>
__global__ __launch_bounds__(512) void attn16_fwd_kernel(const AttnParams p) {
  using frag = typename A16<PREC>::frag;
  constexpr int NPL = A16<PREC>::NPL;
  constexpr int NKV = KV16 ? 1 : NPL;      // bf16-stored K / V are exact in one fp16 plane
  constexpr int WL = a16_fwd_wave_lds<PREC>();
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h2 = lane >> 5, l31 = lane & 31;
  const int wpb = blockDim.x >> 6;
  const int hpb = p.H / wpb;
  const bool split = p.ksplit > 0;
  const int b = split ? blockIdx.x / p.H : blockIdx.x / hpb;
  const int h = split ? blockIdx.x % p.H : (blockIdx.x % hpb) * wpb + wave;
  const int kbeg = split ? wave * p.ksplit : 0;
  int kend = split ? kbeg + p.ksplit : p.N;
  kend = kend > p.N ? p.N : kend;
  const unsigned seed_eff = effective_seed(p.seed, p.seed_src);
  char* vimg = smem + wave * WL;
  const int col0 = h * 32;
  const unsigned bh = (unsigned)(b * p.H + h);

  for (int qt = 0; qt < p.M; qt += 32) {
    float x[16];
    a16_load_rows(p.q, p.ldq, (long)b * p.M + qt, p.M - qt, col0, lane, x);
#pragma unroll
    for (int t = 0; t < 16; ++t) x[t] *= p.scale;
    const float sq = a16_scale<PREC>(x);
    frag qf[2][NPL];
    a16_split<PREC>(x, qf[0], sq);
    a16_split<PREC>(x + 8, qf[1], sq);

    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    float kn[16], vn[16];
    if (kbeg < kend) {
      a16_load_kv<KV16>(p.k, p.ldk, (long)b * p.N + kbeg, p.N - kbeg, col0, lane, kn);
      a16_load_kv<KV16>(p.v, p.ldv, (long)b * p.N + kbeg, p.N - kbeg, col0, lane, vn);
    }
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
      frag kf[2][NKV], vf[2][NKV];
      const float sk = a16_scale<PREC>(kn), sv = a16_scale<PREC>(vn);
      a16_split<PREC>(kn, kf[0], sk); a16_split<PREC>(kn + 8, kf[1], sk);
      a16_split<PREC>(vn, vf[0], sv); a16_split<PREC>(vn + 8, vf[1], sv);
      if (k0 + 32 < kend) {
        a16_load_kv<KV16>(p.k, p.ldk, (long)b * p.N + k0 + 32, p.N - k0 - 32, col0, lane, kn);
        a16_load_kv<KV16>(p.v, p.ldv, (long)b * p.N + k0 + 32, p.N - k0 - 32, col0, lane, vn);
      }
      a16_store_rows<PREC>(vimg, lane, vf[0], vf[1]);
      // S^T[key][q]: rows = keys in registers, column = query on the lane
      f32x16 s = a16_zero();
      s = a16_mma<PREC>(kf[0], qf[0], s);
      s = a16_mma<PREC>(kf[1], qf[1], s);
      const float us = 1.f / (sk * sq);
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[r] = (k0 + crow(r, h2) >= p.N) ? -INFINITY : s[r] * us;
        mx = fmaxf(mx, s[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __expf(m_run - m_new);
      float psum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __expf(s[r] - m_new);
        psum += e;
        float pd = e;
        if (p.drop_thresh != 0u)
          pd = attn_keep(seed_eff, bh, (unsigned)(qt + l31), (unsigned)(k0 + crow(r, h2)), p.drop_thresh)
                   ? e * p.keep_scale : 0.f;
        s[r] = pd;
        oacc[r] *= alpha;
      }
      psum += __shfl_xor(psum, 32);
      l_run = l_run * alpha + psum;
      m_run = m_new;
      // O^T[d][q] += V^T P^T : A = V^T from the image (transposed reads), B = P^T registers
      f32x16 ot = a16_zero();
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        frag pf[NPL], va[NKV];
        a16_acc_frag<PREC>(s, st, pf);
        a16_tr<PREC>(vimg, st, lane, va);
        ot = a16_mma<PREC>(va, pf, ot);
      }
      const float uv = 1.f / sv;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] = fmaf(ot[r], uv, oacc[r]);
    }
    const int q = qt + l31;
    if (split) {
      // merge the waves' partial states: each leaves (O^T registers, m, l) in its own image space;
      // waves 0..3 each finish one group of four channels
      asm volatile("" ::: "memory");
      float* part = reinterpret_cast<float*>(vimg);
#pragma unroll
      for (int r = 0; r < 16; ++r) part[r * 64 + lane] = oacc[r];
      part[16 * 64 + lane] = m_run;
      part[17 * 64 + lane] = l_run;
      __syncthreads();
      for (int g = wave; g < 4; g += wpb) {
        float mt = -INFINITY;
        for (int sw = 0; sw < wpb; ++sw) mt = fmaxf(mt, reinterpret_cast<const float*>(smem + sw * WL)[16 * 64 + lane]);
        float lt = 0.f;
        float4 ov = zero4();
        for (int sw = 0; sw < wpb; ++sw) {
          const float* ps = reinterpret_cast<const float*>(smem + sw * WL);
          const float ms = ps[16 * 64 + lane];
          const float w = ms == -INFINITY ? 0.f : __expf(ms - mt);
          lt = fmaf(ps[17 * 64 + lane], w, lt);
          ov.x = fmaf(ps[(4 * g) * 64 + lane], w, ov.x); ov.y = fmaf(ps[(4 * g + 1) * 64 + lane], w, ov.y);
          ov.z = fmaf(ps[(4 * g + 2) * 64 + lane], w, ov.z); ov.w = fmaf(ps[(4 * g + 3) * 64 + lane], w, ov.w);
        }
        const float inv = 1.f / lt;
        if (q < p.M) {
          float* op = p.o + (size_t)((long)b * p.M + q) * p.ldo + col0;
          *reinterpret_cast<float4*>(op + 8 * g + 4 * h2) = make_float4(ov.x * inv, ov.y * inv, ov.z * inv, ov.w * inv);
          if (g == 0 && h2 == 0) p.lse[((size_t)b * p.H + h) * p.M + q] = mt + __logf(lt);
        }
      }
      __syncthreads();
      continue;
    }
    const float inv = 1.f / l_run;
    if (q < p.M) {
      float* op = p.o + (size_t)((long)b * p.M + q) * p.ldo + col0;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(op + 8 * g + 4 * h2) =
            make_float4(oacc[4 * g] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv);
      if (h2 == 0) p.lse[((size_t)b * p.H + h) * p.M + q] = m_run + __logf(l_run);
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward.  LDS per wave: Q, dO, K, dS images (NPL planes each) + one fp32 transpose scratch.
// ---------------------------------------------------------------------------------------
// Two workgroups per CU: 80 KB of LDS each (the fp32 transpose scratch shares the dS image, the
// per-query scalars sit in the pad bytes of the Q image) and at most 256 registers per wave -
// the kernel is bound by vector-instruction issue, and one wave alone on a SIMD gets half of it.
// Key-split mode as in the forward: dK / dV tiles belong to one wave each, the dQ^T partial sums
// are added through LDS (up to 8 waves = the whole 160 KB).
template <int PREC, bool KV16 = false, bool KSPLIT = false>
__global__ __launch_bounds__(KSPLIT ? 512 : 256, KSPLIT ? 1 : 2) void attn16_bwd_kernel(const AttnParams p) {
  using frag = typename A16<PREC>::frag;
  constexpr int NPL = A16<PREC>::NPL;
  constexpr int NKV = KV16 ? 1 : NPL;
  constexpr bool SCR_IN_DS = AT_TILE * 4 <= NPL * A16_IMG;      // two planes: yes; one bf16 plane: own space
  constexpr int WAVE_LDS = 4 * NPL * A16_IMG + (SCR_IN_DS ? 0 : AT_TILE * 4);
  extern __shared__ __attribute__((aligned(16))) char dsm16[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h2 = lane >> 5, l31 = lane & 31;
  const int wpb = blockDim.x >> 6;
  const int hpb = p.H / wpb;
  constexpr bool split = KSPLIT;
  const int b = split ? blockIdx.x / p.H : blockIdx.x / hpb;
  const int h = split ? blockIdx.x % p.H : (blockIdx.x % hpb) * wpb + wave;
  const int kbeg = split ? wave * p.ksplit : 0;
  int kend = split ? kbeg + p.ksplit : p.N;
  kend = kend > p.N ? p.N : kend;
  const unsigned seed_eff = effective_seed(p.seed, p.seed_src);
  char* qimg = dsm16 + wave * WAVE_LDS;
  char* doimg = qimg + NPL * A16_IMG;
  char* kimg = doimg + NPL * A16_IMG;
  char* dsimg = kimg + NPL * A16_IMG;
  float* scr = reinterpret_cast<float*>(SCR_IN_DS ? dsimg : dsimg + NPL * A16_IMG);   // the dS image is dead by then
  const int col0 = h * 32;
  const unsigned bh = (unsigned)(b * p.H + h);
  float mx_dv = 0.f, mx_dk = 0.f;

  for (int qt = 0; qt < p.M; qt += 32) {
    float xq[16], xdo[16], xo[16];
    a16_load_rows(p.q, p.ldq, (long)b * p.M + qt, p.M - qt, col0, lane, xq);
    a16_load_rows(p.dout, p.lddo, (long)b * p.M + qt, p.M - qt, col0, lane, xdo);
    a16_load_rows(p.o, p.ldo, (long)b * p.M + qt, p.M - qt, col0, lane, xo);
#pragma unroll
    for (int t = 0; t < 16; ++t) xq[t] *= p.scale;
    const float sq = a16_scale<PREC>(xq), sdo = a16_scale<PREC>(xdo);
    frag qf[2][NPL], dof[2][NPL];
    a16_split<PREC>(xq, qf[0], sq); a16_split<PREC>(xq + 8, qf[1], sq);
    a16_split<PREC>(xdo, dof[0], sdo); a16_split<PREC>(xdo + 8, dof[1], sdo);
    a16_store_rows<PREC>(qimg, lane, qf[0], qf[1]);
    a16_store_rows<PREC>(doimg, lane, dof[0], dof[1]);
    float dl = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) dl = fmaf(xdo[t], xo[t], dl);
    dl += __shfl_xor(dl, 32);
    const int qcol = qt + l31;
    const float lse_col = qcol < p.M ? p.lse[((size_t)b * p.H + h) * p.M + qcol] : 0.f;
    const float dl_col = dl;
    // per-query scalars in row form (query = crow(r, h2)) are read back from the 16 pad bytes of the Q
    // image's rows - 32 registers less than keeping them, and a broadcast LDS read each
    if (h2 == 0) *reinterpret_cast<float2*>(qimg + l31 * A16_ROWB + 64) = make_float2(lse_col, dl_col);
    asm volatile("" ::: "memory");
    f32x16 dqacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[r] = 0.f;

    float kn[16], vn[16];
    if (kbeg < kend) {
      a16_load_kv<KV16>(p.k, p.ldk, (long)b * p.N + kbeg, p.N - kbeg, col0, lane, kn);
      a16_load_kv<KV16>(p.v, p.ldv, (long)b * p.N + kbeg, p.N - kbeg, col0, lane, vn);
    }
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
      frag kf[2][NKV], vf[2][NKV];
      const float sk = a16_scale<PREC>(kn), sv = a16_scale<PREC>(vn);
      a16_split<PREC>(kn, kf[0], sk); a16_split<PREC>(kn + 8, kf[1], sk);
      a16_split<PREC>(vn, vf[0], sv); a16_split<PREC>(vn + 8, vf[1], sv);
      if (k0 + 32 < kend) {
        a16_load_kv<KV16>(p.k, p.ldk, (long)b * p.N + k0 + 32, p.N - k0 - 32, col0, lane, kn);
        a16_load_kv<KV16>(p.v, p.ldv, (long)b * p.N + k0 + 32, p.N - k0 - 32, col0, lane, vn);
      }
      a16_store_rows<PREC>(kimg, lane, kf[0], kf[1]);
      const bool key_ok = (k0 + l31) < p.N;
      const float us = 1.f / (sq * sk), udp = 1.f / (sdo * sv);

      // ---- keys on lanes: S[q][key], dP[q][key]
      f32x16 s = a16_zero(), dp = a16_zero();
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        s = a16_mma<PREC>(qf[st], kf[st], s);
        dp = a16_mma<PREC>(dof[st], vf[st], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qt + crow(r, h2);
        const float2 qs = *reinterpret_cast<const float2*>(qimg + crow(r, h2) * A16_ROWB + 64);      // (lse, delta) of query q
        float pr = (key_ok && q < p.M) ? __expf(s[r] * us - qs.x) : 0.f;
        float keepf = 1.f;
        if (p.drop_thresh != 0u)
          keepf = attn_keep(seed_eff, bh, (unsigned)q, (unsigned)(k0 + l31), p.drop_thresh) ? p.keep_scale : 0.f;
        const float ds = pr * (dp[r] * udp * keepf - qs.y);
        s[r] = pr * keepf;                            // Pd
        dp[r] = ds;                                   // dS
      }
      // dV^T[d][key] = dO^T Pd ;  dK^T[d][key] = Q^T dS
      f32x16 dvt = a16_zero(), dkt = a16_zero();
      const float sds = PREC == 0 ? a16_scale_acc(dp) : 1.f;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        frag pf[NPL], dsf[NPL], at[NPL];
        a16_acc_frag<PREC>(s, st, pf);
        a16_tr<PREC>(doimg, st, lane, at);
        dvt = a16_mma<PREC>(at, pf, dvt);
        a16_acc_frag<PREC>(dp, st, dsf, sds);
        a16_tr<PREC>(qimg, st, lane, at);
        dkt = a16_mma<PREC>(at, dsf, dkt);
        // row `key` of the dS image: elements j = 0..3 are queries 16 st + 4 h2 + j, j = 4..7 are 8 further
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          const uint4 u = __builtin_bit_cast(uint4, dsf[pl]);
          char* q = dsimg + pl * A16_IMG + l31 * A16_ROWB + (16 * st + 4 * h2) * 2;
          *reinterpret_cast<uint2*>(q) = make_uint2(u.x, u.y);
          *reinterpret_cast<uint2*>(q + 16) = make_uint2(u.z, u.w);
        }
      }
      asm volatile("" ::: "memory");
      {
        const float uv = 1.f / sdo, uk = 1.f / (sq * sds);
#pragma unroll
        for (int r = 0; r < 16; ++r) { dvt[r] *= uv; dkt[r] *= uk; }
      }
      // ---- dQ^T[d][q] += K^T dS^T.  dS is in hand with the KEY on the lane; the product wants the
      // query there.  Its fragments (already split for dK^T) went to a row-major [key][q] image
      // above; transposed reads of that image and of the K image deliver both operands with the
      // same key order - no second evaluation of S^T / dP^T (12 MFMAs, 16 exp and 16 dropout
      // hashes per tile, which is what bound this kernel once the products were cheap)
      {
        f32x16 dqt = a16_zero();
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          frag bt[NPL], at[NKV];
          a16_tr<PREC>(dsimg, st, lane, bt);
          a16_tr<PREC>(kimg, st, lane, at);
          dqt = a16_mma<PREC>(at, bt, dqt);
        }
        const float uq = 1.f / (sk * sds);
#pragma unroll
        for (int r = 0; r < 16; ++r) dqacc[r] = fmaf(dqt[r], uq, dqacc[r]);
      }
      asm volatile("" ::: "memory");      // the scratch below overwrites the dS image
      // transpose through LDS and store rows: lane -> (key = l>>3 + 8*it, 4 floats at (l&7)*4)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const f32x16& acc = pass == 0 ? dvt : dkt;
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[l31 * AT_LD + crow(r, h2)] = acc[r];   // scr[key][d]
        float* dst = pass == 0 ? p.dv : p.dk;
        const long ldd = pass == 0 ? p.lddv : p.lddk;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int key = (lane >> 3) + 8 * it;
          float4 vv = *reinterpret_cast<const float4*>(scr + key * AT_LD + (lane & 7) * 4);
          if (k0 + key < p.N) {
            const size_t eo = (size_t)((long)b * p.N + k0 + key) * ldd + col0 + (lane & 7) * 4;
            if constexpr (KV16) {        // gradient arena in bf16 (bf16 mode)
              uint2* gp = reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dst) + eo);
              if (qt > 0) {
                const uint2 old = *gp;
                vv.x += bf16_lo(old.x); vv.y += bf16_hi(old.x); vv.z += bf16_lo(old.y); vv.w += bf16_hi(old.y);
              }
              *gp = make_uint2(pack_bf16x2(vv.x, vv.y), pack_bf16x2(vv.z, vv.w));
            } else {
              float* gp = dst + eo;
              if (qt > 0) {
                const float4 old = ldg4(gp);
                vv.x += old.x; vv.y += old.y; vv.z += old.z; vv.w += old.w;
              }
              *reinterpret_cast<float4*>(gp) = vv;
            }
            const float m4 = fmaxf(fmaxf(fabsf(vv.x), fabsf(vv.y)), fmaxf(fabsf(vv.z), fabsf(vv.w)));
            if (pass == 0) mx_dv = fmaxf(mx_dv, m4); else mx_dk = fmaxf(mx_dk, m4);
          }
        }
      }
    }
    if (split) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int r = 0; r < 16; ++r) scr[r * 64 + lane] = dqacc[r];
      __syncthreads();
      for (int g = wave; g < 4 && qcol < p.M; g += wpb) {
        float4 dv4 = zero4();
        for (int sw = 0; sw < wpb; ++sw) {
          const float* ps = reinterpret_cast<const float*>(dsm16 + sw * WAVE_LDS + (scr - reinterpret_cast<float*>(qimg)) * 4);
          dv4.x += ps[(4 * g) * 64 + lane]; dv4.y += ps[(4 * g + 1) * 64 + lane];
          dv4.z += ps[(4 * g + 2) * 64 + lane]; dv4.w += ps[(4 * g + 3) * 64 + lane];
        }
        float* dqp = p.dq + (size_t)((long)b * p.M + qcol) * p.lddq + col0;
        *reinterpret_cast<float4*>(dqp + 8 * g + 4 * h2) =
            make_float4(dv4.x * p.scale, dv4.y * p.scale, dv4.z * p.scale, dv4.w * p.scale);
      }
      __syncthreads();
    } else if (qcol < p.M) {
      float* dqp = p.dq + (size_t)((long)b * p.M + qcol) * p.lddq + col0;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dqp + 8 * g + 4 * h2) =
            make_float4(dqacc[4 * g] * p.scale, dqacc[4 * g + 1] * p.scale, dqacc[4 * g + 2] * p.scale,
                        dqacc[4 * g + 3] * p.scale);
    }
  }
  if (p.kv_amax_part != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mx_dv = fmaxf(mx_dv, __shfl_xor(mx_dv, o));
      mx_dk = fmaxf(mx_dk, __shfl_xor(mx_dk, o));
    }
    if (split) {     // one pair per workgroup (= per head), as many entries as the other mode writes
      float* red = reinterpret_cast<float*>(dsm16);
      if (lane == 0) { red[wave * 2] = mx_dv; red[wave * 2 + 1] = mx_dk; }
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int sw = 1; sw < wpb; ++sw) { mx_dv = fmaxf(mx_dv, red[sw * 2]); mx_dk = fmaxf(mx_dk, red[sw * 2 + 1]); }
        p.kv_amax_part[(size_t)blockIdx.x * 2] = mx_dv;
        p.kv_amax_part[(size_t)blockIdx.x * 2 + 1] = mx_dk;
      }
    } else if (lane == 0) {
      p.kv_amax_part[(size_t)(blockIdx.x * wpb + wave) * 2] = mx_dv;
      p.kv_amax_part[(size_t)(blockIdx.x * wpb + wave) * 2 + 1] = mx_dk;
    }
  }
}

}  // namespace prh
