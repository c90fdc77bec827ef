// Split-precision GEMM cores: fp32 operands are split ON LOAD into three bf16 planes
//     a = h + m + l   (h = bf16(a), m = bf16(a-h), l = bf16(a-h-m): 3 x 8 = 24 significant
//     bits, i.e. the fp32 value exactly)
// and the product is assembled from the six partial products whose weight is >= 2^-16
//     a*w = h*h' + h*m' + m*h' + m*m' + h*l' + l*h'      (dropped: m*l', l*m', l*l' <= 2^-24)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Every bf16*bf16 product is exact in
// fp32, so the result carries fp32-level error (measured: same 1e-6 class as the fp32 MFMA
// path against the fp64 oracle) with NO range assumption (bf16 has the fp32 exponent), while
// issuing 6 bf16 MFMAs (6 x 32 cycles per 32x32x16 block) instead of 8 fp32 MFMAs (8 x 64
// cycles): a 2.67x higher matrix ceiling for fp32-accurate results, since gfx950 has no
// TF32/xf32 path.
//
// Tiling: 256 x 256 x 16 per 512-thread workgroup (8 waves as 2(M) x 4(N), wave tile
// 128 x 64 = 4 x 2 MFMA blocks, 128 accumulator registers), LDS double-buffered
// (2 x 48 KB of bf16 planes), ONE barrier per k-tile: global loads for tile t+1 are issued
// before the MFMAs of tile t and converted/written to the other LDS buffer after them.
// LDS images are [plane][row][16 k] bf16 (32 B rows) with the two 16-B chunks of a row
// swapped on rows with bit 3 set, which makes every ds_read_b128 fragment read conflict-free.
// Weights are split once per launch into that tiled, pre-swizzled image
// (prep_weights_s3_kernel), so their staging is a straight 16-B-per-lane copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prh_gemm.hpp"

namespace prh {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int S3_BM = 256, S3_BN = 256, S3_BK = 16;
constexpr int S3_PLANE = 256 * S3_BK * 2;          // bytes of one [256][16] bf16 plane = 8 KB
constexpr int S3_OPER = 3 * S3_PLANE;              // one operand, three planes = 24 KB
constexpr int S3_STAGE = 2 * S3_OPER;              // A + W = 48 KB
constexpr int S3_LDS = 2 * S3_STAGE;               // double buffered = 96 KB

// byte offset of element (row, k) inside one plane
__device__ __forceinline__ int s3_off(int row, int k) {
  return row * 32 + ((((k >> 3) ^ (row >> 3)) & 1) << 4) + ((k & 7) << 1);
}

// split two floats into three packed bf16 pairs (h, m, l)
__device__ __forceinline__ void split2(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
  f32x2 v = {a0, a1};
  bf16x2 hb = __builtin_convertvector(v, bf16x2);
  h = *reinterpret_cast<unsigned*>(&hb);
  f32x2 r = {a0 - __uint_as_float(h << 16), a1 - __uint_as_float(h & 0xffff0000u)};
  bf16x2 mb = __builtin_convertvector(r, bf16x2);
  m = *reinterpret_cast<unsigned*>(&mb);
  f32x2 r2 = {r[0] - __uint_as_float(m << 16), r[1] - __uint_as_float(m & 0xffff0000u)};
  bf16x2 lb = __builtin_convertvector(r2, bf16x2);
  l = *reinterpret_cast<unsigned*>(&lb);
}

// ---------------------------------------------------------------------------------------
// Weight preparation: W'[n][k] (= W[n*ldw + k], or W[k*ldw + n] when transposed) ->
// tiled planes out[((n_tile*KT + k_tile)*3 + plane) * 8 KB + s3_off(row, k)], zero padded.
// One thread per (row, 8-k chunk).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_weights_s3_kernel(const float* __restrict__ W, int N,
                                                              int K, long ldw, int transposed,
                                                              char* __restrict__ out) {
  const int KT = (K + S3_BK - 1) / S3_BK;
  const int NT_ = (N + S3_BN - 1) / S3_BN;
  const long total = (long)NT_ * 256 * KT * 2;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int chunk = (int)(i & 1);
  long t = i >> 1;
  const int k_tile = (int)(t % KT); t /= KT;
  const int row = (int)(t & 255);
  const int n_tile = (int)(t >> 8);
  const int n = n_tile * 256 + row;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k_tile * S3_BK + chunk * 8 + j;
    v[j] = (n < N && k < K) ? (transposed ? W[(size_t)k * ldw + n] : W[(size_t)n * ldw + k]) : 0.f;
  }
  uint4 h, m, l;
  split2(v[0], v[1], h.x, m.x, l.x);
  split2(v[2], v[3], h.y, m.y, l.y);
  split2(v[4], v[5], h.z, m.z, l.z);
  split2(v[6], v[7], h.w, m.w, l.w);
  char* base = out + ((size_t)n_tile * KT + k_tile) * S3_OPER + s3_off(row, chunk * 8);
  *reinterpret_cast<uint4*>(base) = h;
  *reinterpret_cast<uint4*>(base + S3_PLANE) = m;
  *reinterpret_cast<uint4*>(base + 2 * S3_PLANE) = l;
}

inline size_t s3_weight_bytes(int N, int K) {
  return (size_t)((N + S3_BN - 1) / S3_BN) * ((K + S3_BK - 1) / S3_BK) * S3_OPER;
}

// six-product MFMA block: acc += (ah+am+al)*(wh+wm+wl) without the three smallest terms
__device__ __forceinline__ void mfma6(f32x16& acc, const bf16x8& ah, const bf16x8& am,
                                      const bf16x8& al, const bf16x8& wh, const bf16x8& wm,
                                      const bf16x8& wl) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, wh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc, 0, 0, 0);
}

// one k-tile of MFMAs for a wave tile of 128 x 64 out of the stage at `st`
__device__ __forceinline__ void s3_compute(f32x16 (&acc)[4][2], const char* st, int wm, int wn,
                                           int l31, int half) {
  bf16x8 w[2][3];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wn + j * 32 + l31;
    const char* q = st + S3_OPER + s3_off(row, half * 8);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) w[j][pl] = *reinterpret_cast<const bf16x8*>(q + pl * S3_PLANE);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wm + i * 32 + l31;
    const char* q = st + s3_off(row, half * 8);
    bf16x8 a[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8*>(q + pl * S3_PLANE);
#pragma unroll
    for (int j = 0; j < 2; ++j) mfma6(acc[i][j], a[0], a[1], a[2], w[j][0], w[j][1], w[j][2]);
  }
}

// ---------------------------------------------------------------------------------------
// NT: C[M,N] = pro(A)[M,K] * W'[N,K]^T, W' pre-split (Wp).  Same prologues/epilogues as
// gemm_nt_kernel (EPI_GATE excluded).
// ---------------------------------------------------------------------------------------
template <int PRO, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_s3_kernel(const NTParams p,
                                                            const char* __restrict__ Wp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
  const int m0 = tile_m * S3_BM, n0 = tile_n * S3_BN;
  const int KT = (p.K + S3_BK - 1) / S3_BK;

  const int sc = (tid & 3) * 4;      // staging k offset (float4)
  const int sr = tid >> 2;           // staging row 0..127 (+128)

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[2], ra2[2];
  const char* wsrc = Wp + (size_t)tile_n * KT * S3_OPER + tid * 16;
  // wave-uniform LDS offset of this wave's 1-KB slice of a W plane (LDS-DMA adds lane*16)
  const int wdst = S3_OPER + __builtin_amdgcn_readfirstlane(wave) * 1024;

  // W planes: pre-split, pre-swizzled image -> straight global->LDS DMA (no registers, no
  // ds_write); the __syncthreads() that ends the iteration waits for it (vmcnt).
  auto dma_w = [&](int kt, char* st) {
    const char* q = wsrc + (size_t)kt * S3_OPER;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(q + pl * S3_PLANE),
          (__attribute__((address_space(3))) void*)(st + wdst + pl * S3_PLANE), 16, 0, 0);
  };
  auto load_tile = [&](int kt) {
    const int k = kt * S3_BK + sc;
    const bool kok = k < p.K;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = m0 + sr + 128 * j;
      const bool ok = kok && row < p.M;
      ra[j] = ok ? ldg4(p.A + (size_t)row * p.lda + k) : zero4();
      if (PRO == PRO_BNBWD) ra2[j] = ok ? ldg4(p.A2 + (size_t)row * p.lda2 + k) : zero4();
    }
  };
  auto store_tile = [&](int kt, char* st) {
    const int k = kt * S3_BK + sc;
    const bool kok = k < p.K;
    float4 ka = zero4(), kb = zero4(), kc = zero4();
    if (PRO != PRO_NONE && kok) {
      ka = ldg4(p.pa + k);
      kb = ldg4(p.pb + k);
      if (PRO == PRO_BNBWD) kc = ldg4(p.pc + k);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = sr + 128 * j;
      const bool ok = kok && (m0 + r) < p.M;
      const float4 v = ok ? pro_apply<PRO>(ra[j], ra2[j], ka, kb, kc) : zero4();
      uint2 h, m, l;
      split2(v.x, v.y, h.x, m.x, l.x);
      split2(v.z, v.w, h.y, m.y, l.y);
      char* q = st + s3_off(r, sc);
      *reinterpret_cast<uint2*>(q) = h;
      *reinterpret_cast<uint2*>(q + S3_PLANE) = m;
      *reinterpret_cast<uint2*>(q + 2 * S3_PLANE) = l;
    }
  };

  dma_w(0, smem);
  load_tile(0);
  store_tile(0, smem);
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    char* cur = smem + (kt & 1) * S3_STAGE;
    char* nxt = smem + ((kt + 1) & 1) * S3_STAGE;
    if (kt + 1 < KT) {
      dma_w(kt + 1, nxt);
      load_tile(kt + 1);
    }
    s3_compute(acc, cur, wm, wn, l31, half);
    if (kt + 1 < KT) store_tile(kt + 1, nxt);
    __syncthreads();
  }
  nt_epilogue<EPI, 4, 2>(acc, p, m0 + wm, n0 + wn, tile_m * 2 + (wave >> 2), lane);
}

// ---------------------------------------------------------------------------------------
// TN (wgrad): C[Mo,Ni] = sum_p proA(A)[p,Mo] * proB(B)[p,Ni] over this split's rows.
// Staging: thread = (column c, 8-row octet): 8 scalar loads down a column (a wave reads
// 256 contiguous bytes per row), per-column prologue coefficients live in registers, the
// 8 values become one 16-B bf16 fragment per plane in the same [row = column][16 k] image
// the NT kernel uses, so the MFMA loop is shared.
// ---------------------------------------------------------------------------------------
template <int PROA, int PROB>
__global__ __launch_bounds__(512, 2) void gemm_tn_s3_kernel(const TNParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  int b = blockIdx.x;
  const int tile_n = b % p.tiles_n; b /= p.tiles_n;
  const int tile_m = b % p.tiles_m; b /= p.tiles_m;
  const int split = b;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int p_begin = split * p.rows_per_split;
  int p_end = p_begin + p.rows_per_split;
  if (p_end > p.P) p_end = p.P;
  const int KT = (p_end - p_begin + S3_BK - 1) / S3_BK;

  const int c = tid & 255, oct = tid >> 8;
  const bool aok = (m0 + c) < p.Mo, bok = (n0 + c) < p.Ni;
  float ka = 0.f, kb = 0.f, kc = 0.f, qa = 0.f, qb = 0.f;
  if (PROA != PRO_NONE && aok) {
    ka = p.pa[m0 + c];
    kb = p.pb[m0 + c];
    if (PROA == PRO_BNBWD) kc = p.pc[m0 + c];
  }
  if (PROB != PRO_NONE && bok) {
    qa = p.qa[n0 + c];
    qb = p.qb[n0 + c];
  }
  const float* Ap = p.A + m0 + c;
  const float* A2p = (PROA == PRO_BNBWD) ? p.A2 + m0 + c : nullptr;
  const float* Bp = p.B + n0 + c;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float va[8], va2[8], vb[8];
  float csum = 0.f;

  auto load_tile = [&](int kt) {
    const int r0 = p_begin + kt * S3_BK + oct * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = r0 + j;
      const bool rok = row < p_end;
      va[j] = (rok && aok) ? Ap[(size_t)row * p.lda] : 0.f;
      if (PROA == PRO_BNBWD) va2[j] = (rok && aok) ? A2p[(size_t)row * p.lda2] : 0.f;
      vb[j] = (rok && bok) ? Bp[(size_t)row * p.ldb] : 0.f;
    }
  };
  auto store_tile = [&](int kt, char* st) {
    const int r0 = p_begin + kt * S3_BK + oct * 8;
    float ta[8], tb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool rok = (r0 + j) < p_end;
      float x = va[j];
      if (PROA == PRO_BNBWD) x = fmaf(ka, va[j], fmaf(kb, va2[j], kc));
      else if (PROA == PRO_BNRELU) x = fmaxf(fmaf(va[j], ka, kb), 0.f);
      ta[j] = (rok && aok) ? x : 0.f;
      float y = vb[j];
      if (PROB == PRO_BNRELU) y = fmaxf(fmaf(vb[j], qa, qb), 0.f);
      tb[j] = (rok && bok) ? y : 0.f;
      csum += ta[j];
    }
    uint4 h, m, l;
    split2(ta[0], ta[1], h.x, m.x, l.x);
    split2(ta[2], ta[3], h.y, m.y, l.y);
    split2(ta[4], ta[5], h.z, m.z, l.z);
    split2(ta[6], ta[7], h.w, m.w, l.w);
    char* q = st + s3_off(c, oct * 8);
    *reinterpret_cast<uint4*>(q) = h;
    *reinterpret_cast<uint4*>(q + S3_PLANE) = m;
    *reinterpret_cast<uint4*>(q + 2 * S3_PLANE) = l;
    split2(tb[0], tb[1], h.x, m.x, l.x);
    split2(tb[2], tb[3], h.y, m.y, l.y);
    split2(tb[4], tb[5], h.z, m.z, l.z);
    split2(tb[6], tb[7], h.w, m.w, l.w);
    q += S3_OPER;
    *reinterpret_cast<uint4*>(q) = h;
    *reinterpret_cast<uint4*>(q + S3_PLANE) = m;
    *reinterpret_cast<uint4*>(q + 2 * S3_PLANE) = l;
  };

  if (KT > 0) {
    load_tile(0);
    store_tile(0, smem);
  }
  __syncthreads();
  for (int kt = 0; kt < KT; ++kt) {
    char* cur = smem + (kt & 1) * S3_STAGE;
    char* nxt = smem + ((kt + 1) & 1) * S3_STAGE;
    if (kt + 1 < KT) load_tile(kt + 1);
    s3_compute(acc, cur, wm, wn, l31, half);
    if (kt + 1 < KT) store_tile(kt + 1, nxt);
    __syncthreads();
  }

  float* out = p.slab + (size_t)split * p.Mo * p.Ni;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = n0 + wn + nt * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + mt * 32 + crow(r, half);
        if (row < p.Mo && col < p.Ni) out[(size_t)row * p.Ni + col] = acc[mt][nt][r];
      }
    }

  if (p.colsum != nullptr && tile_n == 0) {
    float* red = reinterpret_cast<float*>(smem);   // all LDS reads are behind the last barrier
    red[tid] = csum;
    __syncthreads();
    if (tid < 256 && (m0 + tid) < p.Mo)
      p.colsum[(size_t)split * p.Mo + m0 + tid] = red[tid] + red[tid + 256];
  }
}

}  // namespace prh
