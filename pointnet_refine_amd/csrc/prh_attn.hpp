// Fused cross-attention for the DETR decoder layers of LineRefineNet (src/model.py:119-128):
// M = 32 line-point queries against N context keys, heads of 32 channels, exact fp32 on
// v_mfma_f32_32x32x2_f32, online softmax, optional dropout on the attention weights.
//
// One wave owns one (segment b, head h): the 32 x 32 query block is resident, key/value tiles
// of 32 rows stream through.  K and V are read straight from the batched projection buffers
// ([B*N, 6*256], row stride ldk) and - in the backward - dK and dV are written straight into
// the matching gradient buffers, so neither the (B,8,32,N) score tensor nor per-layer copies
// of K/V and their gradients ever exist.
//
// MFMA 32x32x2 operand maps used below (lane l, h2 = l>>5):
//   A[i = l&31][k = h2]   B[k = h2][j = l&31]   D[row = crow(r,h2)][col = l&31], r = 0..15
// An accumulator therefore IS a valid B operand of a following MFMA whose summation index
// is the accumulator's ROW (step t takes register t: k-pair = rows crow(t,0), crow(t,1)); the
// other operand is fetched for exactly those rows, so no shuffles or transposes are needed:
//   S^T[key][q]  = K Q^T            A = K rows (lane = key),   B = Q rows (lane = q)
//   O^T[d][q]    = V^T P^T          A = V[crow(t,h2)][d],      B = P^T registers
//   S[q][key], dP[q][key]           operands swapped            (keys on lanes)
//   dV^T[d][key] = dO^T Pd          A = dO[crow(t,h2)][d],     B = Pd registers
//   dK^T[d][key] = Q^T dS           A = Q[crow(t,h2)][d],      B = dS registers
//   dQ^T[d][q]   = K^T dS^T         A = K[crow(t,h2)][d],      B = dS^T registers
// Row-indexed operands (second column) come from wave-private LDS tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prh_gemm.hpp"

namespace prh {

constexpr int AT_LD = 36;                       // LDS row stride (floats) of a 32 x 32 tile
constexpr int AT_TILE = 32 * AT_LD;             // floats per tile

struct AttnParams {
  const float* q;  long ldq;     // [B*M, H*32] projected queries
  const float* k;  long ldk;     // [B*N, ..] key projections (column offset applied by caller)
  const float* v;  long ldv;
  float* o;        long ldo;     // [B*M, H*32]
  float* lse;                    // [B, H, M]  log-sum-exp of the scaled scores
  const float* dout; long lddo;  // backward
  float* dq; long lddq;
  float* dk; long lddk;
  float* dv; long lddv;
  int B, M, N, H;
  float scale;                   // 1/sqrt(32)
  float keep_scale;              // 1/(1-p)
  unsigned drop_thresh;          // drop iff hash < thresh (0: no dropout)
  unsigned seed;
  float* kv_amax_part;        // backward, optional: [blocks*4][2] largest |dV|, |dK| a wave stored (an
                              // upper bound of the final values when M > 32 accumulates) - the
                              // operand maxima of the K/V projection's backward GEMMs, for free
  const unsigned* seed_src;   // optional device word mixed into the seed (graph replays: a counter
                              // the caller advances on the device, so every replay draws new masks)
  int ksplit;                 // 16-bit cores, small batches: > 0 = keys per wave (multiple of 32); the
                              // waves of a workgroup then share ONE (segment, head) and split its keys
};

// effective seed of a launch: the host value, mixed with the device word when one is registered
// (read through to L2: the word is rewritten between launches)
__device__ __forceinline__ unsigned effective_seed(unsigned seed, const unsigned* src) {
  if (src == nullptr) return seed;
  const unsigned w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return seed ^ (__builtin_amdgcn_readfirstlane(w) * 0x9E3779B9u);
}

// counter-based dropout decision, a function of (seed, segment*head, query, key) only, so the
// forward and the two orientations of the backward agree (tests re-create it in torch)
__device__ __forceinline__ bool attn_keep(unsigned seed, unsigned bh, unsigned q, unsigned key,
                                          unsigned thresh) {
  unsigned x = seed ^ (bh * 0xC2B2AE3Du) ^ (q * 0x9E3779B1u) ^ (key * 0x85EBCA77u);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x >= thresh;
}

// stage a 32 x 32 block (rows row0.., 32 floats at column col0) into an LDS tile; rows beyond
// `rows` are zero.  lane = (row = l&31, h2): 16 contiguous floats each.  Returns the lane's 16
// values (the "row on the lane" operand form).
__device__ __forceinline__ void load_rows16(const float* base, long ld, long row0, int rows_valid,
                                            int col0, int lane, float (&x)[16]) {
  const int r = lane & 31, h2 = lane >> 5;
  const bool ok = r < rows_valid;
  const float* p = base + (size_t)(row0 + (ok ? r : 0)) * ld + col0 + h2 * 16;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 t = ok ? ldg4(p + 4 * j) : zero4();
    x[4 * j] = t.x; x[4 * j + 1] = t.y; x[4 * j + 2] = t.z; x[4 * j + 3] = t.w;
  }
}
__device__ __forceinline__ void tile_store(float* tile, int lane, const float (&x)[16]) {
  float* p = tile + (lane & 31) * AT_LD + (lane >> 5) * 16;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    *reinterpret_cast<float4*>(p + 4 * j) = make_float4(x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]);
}

// ---------------------------------------------------------------------------------------
// forward: block = WPB waves = WPB heads of one segment (WPB = blockDim.x / 64: 4 normally, 1 for
// small batches, where B * H / 4 workgroups would leave most of the 256 CUs idle); grid = B * H / WPB
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) float smem[4 * AT_TILE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h2 = lane >> 5, l31 = lane & 31;
  const int wpb = blockDim.x >> 6;
  const int hpb = p.H / wpb;                        // blocks per segment
  const int b = blockIdx.x / hpb, h = (blockIdx.x % hpb) * wpb + wave;
  const unsigned seed_eff = effective_seed(p.seed, p.seed_src);
  float* vt = smem + wave * AT_TILE;                // this wave's V tile
  const int col0 = h * 32;
  const unsigned bh = (unsigned)(b * p.H + h);

  for (int qt = 0; qt < p.M; qt += 32) {            // query tiles of 32 (M = 32 in the model)
    float qa[16];
    load_rows16(p.q, p.ldq, (long)b * p.M + qt, p.M - qt, col0, lane, qa);
#pragma unroll
    for (int t = 0; t < 16; ++t) qa[t] *= p.scale;

    f32x16 oacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;           // per query = per lane column

    float kn[16], vn[16];                           // next tile, loaded one tile ahead
    load_rows16(p.k, p.ldk, (long)b * p.N, p.N, col0, lane, kn);
    load_rows16(p.v, p.ldv, (long)b * p.N, p.N, col0, lane, vn);
    for (int k0 = 0; k0 < p.N; k0 += 32) {
      float ka[16], va[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) { ka[t] = kn[t]; va[t] = vn[t]; }
      if (k0 + 32 < p.N) {
        load_rows16(p.k, p.ldk, (long)b * p.N + k0 + 32, p.N - k0 - 32, col0, lane, kn);
        load_rows16(p.v, p.ldv, (long)b * p.N + k0 + 32, p.N - k0 - 32, col0, lane, vn);
      }
      tile_store(vt, lane, va);
      // S^T[key][q]: rows = keys in registers, column = query on the lane
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int t = 0; t < 16; ++t) s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t], qa[t], s, 0, 0, 0);
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (k0 + crow(r, h2) >= p.N) s[r] = -INFINITY;
        mx = fmaxf(mx, s[r]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __expf(m_run - m_new);    // first tile: exp(-inf) = 0
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
      // O^T[d][q] += V^T P^T : A = V[key = crow(t,h2)][d = l31] from the LDS tile
#pragma unroll
      for (int t = 0; t < 16; ++t)
        oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vt[crow(t, h2) * AT_LD + l31], s[t], oacc, 0, 0, 0);
    }
    const float inv = 1.f / l_run;
    const int q = qt + l31;
    if (q < p.M) {
      float* op = p.o + (size_t)((long)b * p.M + q) * p.ldo + col0;
#pragma unroll
      for (int r = 0; r < 16; ++r) op[crow(r, h2)] = oacc[r] * inv;
      if (h2 == 0) p.lse[((size_t)b * p.H + h) * p.M + q] = m_run + __logf(l_run);
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward: block = 4 waves = 4 heads of one segment; each wave streams the key tiles once,
// writes dK/dV tiles (row-major, 16 B per lane through an LDS transpose) and keeps dQ^T in
// registers.  LDS per wave: Q, dO, K tiles + one transpose scratch (4 x 4.5 KB).
// ---------------------------------------------------------------------------------------
// (242 VGPRs + 96 AGPRs: one wave per SIMD.  Forcing two - __launch_bounds__(256, 2) - spills
// 64 VGPRs and measured 4 % slower; the kernel runs the fp32 matrix pipe at ~80 % of its peak
// by the 14-product count below, so occupancy is not what holds it back.)
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h2 = lane >> 5, l31 = lane & 31;
  const int wpb = blockDim.x >> 6;
  const int hpb = p.H / wpb;
  const int b = blockIdx.x / hpb, h = (blockIdx.x % hpb) * wpb + wave;
  const unsigned seed_eff = effective_seed(p.seed, p.seed_src);
  float* qtile = dsm + wave * (4 * AT_TILE);
  float* dotile = qtile + AT_TILE;
  float* ktile = dotile + AT_TILE;
  float* scr = ktile + AT_TILE;
  const int col0 = h * 32;
  const unsigned bh = (unsigned)(b * p.H + h);
  float mx_dv = 0.f, mx_dk = 0.f;

  for (int qt = 0; qt < p.M; qt += 32) {
    float qa[16], doa[16];
    load_rows16(p.q, p.ldq, (long)b * p.M + qt, p.M - qt, col0, lane, qa);
    load_rows16(p.dout, p.lddo, (long)b * p.M + qt, p.M - qt, col0, lane, doa);
    float oa[16];
    load_rows16(p.o, p.ldo, (long)b * p.M + qt, p.M - qt, col0, lane, oa);
#pragma unroll
    for (int t = 0; t < 16; ++t) qa[t] *= p.scale;
    tile_store(qtile, lane, qa);
    tile_store(dotile, lane, doa);
    // delta[q] = sum_d dO[q][d] * O[q][d]   (lane = q row form: 16 of 32 d per lane half)
    float dl = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) dl = fmaf(doa[t], oa[t], dl);
    dl += __shfl_xor(dl, 32);
    // per-query scalars in both orientations: column form (this lane's q = l31) and row form
    const int qcol = qt + l31;
    const float lse_col = qcol < p.M ? p.lse[((size_t)b * p.H + h) * p.M + qcol] : 0.f;
    const float dl_col = dl;                         // lane l31 and l31+32 both hold q = l31
    float lse_row[16], dl_row[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int src = crow(r, h2);                   // row q = crow(r,h2): fetch from that lane
      lse_row[r] = __shfl(lse_col, src);
      dl_row[r] = __shfl(dl_col, src);
    }

    f32x16 dqacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) dqacc[r] = 0.f;

    float kn[16], vn[16];                           // next tile, loaded one tile ahead
    load_rows16(p.k, p.ldk, (long)b * p.N, p.N, col0, lane, kn);
    load_rows16(p.v, p.ldv, (long)b * p.N, p.N, col0, lane, vn);
    for (int k0 = 0; k0 < p.N; k0 += 32) {
      float ka[16], va[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) { ka[t] = kn[t]; va[t] = vn[t]; }
      if (k0 + 32 < p.N) {
        load_rows16(p.k, p.ldk, (long)b * p.N + k0 + 32, p.N - k0 - 32, col0, lane, kn);
        load_rows16(p.v, p.ldv, (long)b * p.N + k0 + 32, p.N - k0 - 32, col0, lane, vn);
      }
      tile_store(ktile, lane, ka);
      const bool key_ok = (k0 + l31) < p.N;          // this lane's key (keys-on-lanes form)

      // ---- keys on lanes: S[q][key], dP[q][key]
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[t], ka[t], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(doa[t], va[t], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = qt + crow(r, h2);
        float pr = (key_ok && q < p.M) ? __expf(s[r] - lse_row[r]) : 0.f;
        float keepf = 1.f;
        if (p.drop_thresh != 0u)
          keepf = attn_keep(seed_eff, bh, (unsigned)q, (unsigned)(k0 + l31), p.drop_thresh) ? p.keep_scale : 0.f;
        const float ds = pr * (dp[r] * keepf - dl_row[r]);
        s[r] = pr * keepf;                            // Pd
        dp[r] = ds;                                   // dS
      }
      // dV^T[d][key] = dO^T Pd ;  dK^T[d][key] = Q^T dS      (A rows from the LDS tiles)
      f32x16 dvt, dkt;
#pragma unroll
      for (int r = 0; r < 16; ++r) { dvt[r] = 0.f; dkt[r] = 0.f; }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int row = crow(t, h2) * AT_LD + l31;
        dvt = __builtin_amdgcn_mfma_f32_32x32x2f32(dotile[row], s[t], dvt, 0, 0, 0);
        dkt = __builtin_amdgcn_mfma_f32_32x32x2f32(qtile[row], dp[t], dkt, 0, 0, 0);
      }
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
            float* gp = dst + (size_t)((long)b * p.N + k0 + key) * ldd + col0 + (lane & 7) * 4;
            if (qt > 0) {       // M > 32: later query tiles add to what the first one stored
              const float4 old = ldg4(gp);
              vv.x += old.x; vv.y += old.y; vv.z += old.z; vv.w += old.w;
            }
            *reinterpret_cast<float4*>(gp) = vv;
            const float m4 = fmaxf(fmaxf(fabsf(vv.x), fabsf(vv.y)), fmaxf(fabsf(vv.z), fabsf(vv.w)));
            if (pass == 0) mx_dv = fmaxf(mx_dv, m4); else mx_dk = fmaxf(mx_dk, m4);
          }
        }
      }

      // ---- queries on lanes: S^T[key][q], dP^T[key][q] -> dS^T -> dQ^T += K^T dS^T
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[t], qa[t], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(va[t], doa[t], dp, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + crow(r, h2);
        const float pr = (key < p.N && qcol < p.M) ? __expf(s[r] - lse_col) : 0.f;
        float keepf = 1.f;
        if (p.drop_thresh != 0u)
          keepf = attn_keep(seed_eff, bh, (unsigned)qcol, (unsigned)key, p.drop_thresh) ? p.keep_scale : 0.f;
        dp[r] = pr * (dp[r] * keepf - dl_col);       // dS^T
      }
#pragma unroll
      for (int t = 0; t < 16; ++t)
        dqacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ktile[crow(t, h2) * AT_LD + l31], dp[t], dqacc, 0, 0, 0);
    }
    if (qcol < p.M) {
      float* dqp = p.dq + (size_t)((long)b * p.M + qcol) * p.lddq + col0;
#pragma unroll
      for (int r = 0; r < 16; ++r) dqp[crow(r, h2)] = dqacc[r] * p.scale;
    }
  }
  if (p.kv_amax_part != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mx_dv = fmaxf(mx_dv, __shfl_xor(mx_dv, o));
      mx_dk = fmaxf(mx_dk, __shfl_xor(mx_dk, o));
    }
    if (lane == 0) {
      p.kv_amax_part[(size_t)(blockIdx.x * wpb + wave) * 2] = mx_dv;
      p.kv_amax_part[(size_t)(blockIdx.x * wpb + wave) * 2 + 1] = mx_dk;
    }
  }
}

}  // namespace prh
