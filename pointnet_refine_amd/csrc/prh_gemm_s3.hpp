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

#include <type_traits>

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
// NPL = 2: two fp16 planes and three products.  fp16 carries 11 significant bits, so
//     a*S = h + l   (h = fp16(a*S), l = fp16(a*S - h): 22 bits)
//     a*w = (h*h' + h*l' + l*h') / (S*S')          (dropped: l*l' <= 2^-22)
// has a split error of ~7e-8 relative (numpy emulation), below the error of the
// fp32 accumulation itself, for HALF the MFMA issue of the six-product bf16 form.  fp16 has a
// narrow exponent, so each operand carries a power-of-two scale S chosen from its largest
// magnitude (absmax_kernel below; the result is unscaled exactly in the epilogue):
// amax*S lies in [2^13, 2^14), a factor 4 under the fp16 maximum; elements down to 2^-16 of
// the largest keep all 22 bits, smaller ones degrade gracefully (absolute floor 2^-25 * S^-1).
// ---------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float pow2_scale(float amax) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);   // biased exponent, amax >= 0
  if (e == 0 || e == 255) return 1.f;                            // 0 / denormal / inf / nan
  int se = 127 + 13 - (e - 127);
  se = se < 67 ? 67 : (se > 187 ? 187 : se);                     // S within 2^-60 .. 2^60
  return __uint_as_float((unsigned)se << 23);
}

// The operand maxima are written by the kernel launched just before (atomic max) at an address
// that is reused launch after launch.  A plain load of a uniform, read-only address becomes a
// scalar load, and the scalar cache may still hold the PREVIOUS launch's value (observed:
// run-to-run differences at rounding level, i.e. a neighbouring power-of-two scale): read it
// with an agent-scope atomic load, which is served by L2.
__device__ __forceinline__ float load_amax(const float* p) {
  const unsigned u = __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
  return __uint_as_float(__builtin_amdgcn_readfirstlane(u));
}

// split two (already scaled) floats into packed fp16 pairs (h, l)
__device__ __forceinline__ void split2h(float a0, float a1, unsigned& h, unsigned& l) {
  f32x2 v = {a0, a1};
  f16x2 hb = __builtin_convertvector(v, f16x2);
  h = *reinterpret_cast<unsigned*>(&hb);
  f32x2 r = {a0 - (float)hb[0], a1 - (float)hb[1]};
  f16x2 lb = __builtin_convertvector(r, f16x2);
  l = *reinterpret_cast<unsigned*>(&lb);
}

// Largest magnitude of pro(A) over [rows, cols]: per-block maxima -> part[blockIdx.x], reduced
// by absmax_final_kernel (plain kernels on the launch stream: no atomics, nothing to
// pre-zero, so no memset whose ordering against queued kernels would have to be trusted).
// One pass at HBM rate: block = 64 column vectors x 4 rows, a thread keeps its column's
// prologue coefficients in registers and walks down the rows (no per-element index
// arithmetic).  vec = 1: cols and the leading dimensions are multiples of 4 (16-B loads).
constexpr int ABSMAX_MAX_BLOCKS = 2048;
template <int PRO>
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ A, long lda,
                                                     const float* __restrict__ A2, long lda2,
                                                     const float* __restrict__ pa,
                                                     const float* __restrict__ pb,
                                                     const float* __restrict__ pc, long rows,
                                                     int cols, int vec, float* __restrict__ part) {
  __shared__ float red[4];
  float m = 0.f;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * 4 + ty, rstep = (long)gridDim.x * 4;
  if (vec) {
    const int c4 = cols >> 2;
    for (int cv = tx; cv < c4; cv += 64) {
      const int c = cv * 4;
      float4 ka = zero4(), kb = zero4(), kc = zero4();
      if (PRO != PRO_NONE) { ka = ldg4(pa + c); kb = ldg4(pb + c); }
      if (PRO == PRO_BNBWD) kc = ldg4(pc + c);
#pragma unroll 4
      for (long r = r0; r < rows; r += rstep) {
        const float4 a = PRO == PRO_GATE1 ? make_float4(A[r * lda], 0.f, 0.f, 0.f) : ldg4(A + r * lda + c);
        const float4 a2 = PRO == PRO_BNBWD ? ldg4(A2 + r * lda2 + c) : zero4();
        const float4 v = pro_apply<PRO>(a, a2, ka, kb, kc);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    }
  } else {
    for (int c = tx; c < cols; c += 64) {
      float ka = 0.f, kb = 0.f, kc = 0.f;
      if (PRO != PRO_NONE) { ka = pa[c]; kb = pb[c]; }
      if (PRO == PRO_BNBWD) kc = pc[c];
      for (long r = r0; r < rows; r += rstep) {
        float x = PRO == PRO_GATE1 ? A[r * lda] : A[r * lda + c];
        if (PRO == PRO_BNRELU || PRO == PRO_GATE1) x = fmaxf(fmaf(x, ka, kb), 0.f);
        if (PRO == PRO_BNBWD) x = fmaf(ka, x, fmaf(kb, A2[r * lda2 + c], kc));
        m = fmaxf(m, fabsf(x));
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (tx == 0) red[ty] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// BatchNorm backward applied in place:  dy <- ka*dy + kb*z + kc  (= dz), with the per-block
// maxima of |dz| for the operand scale.  The split-fp16 wgrad and dgrad of a layer then both
// read dz as a plain operand (4 B/element each instead of dy and z, 8 B, twice), and the
// largest magnitude is exact and free.  Same walk as absmax_kernel (vec = 1 only: cols, ld % 4).
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ dy, long lddy,
                                                           const float* __restrict__ z, long ldz,
                                                           const float* __restrict__ pa,
                                                           const float* __restrict__ pb,
                                                           const float* __restrict__ pc, long rows,
                                                           int cols, float* __restrict__ part) {
  __shared__ float red[4];
  float m = 0.f;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * 4 + ty, rstep = (long)gridDim.x * 4;
  const int c4 = cols >> 2;
  for (int cv = tx; cv < c4; cv += 64) {
    const int c = cv * 4;
    const float4 ka = ldg4(pa + c), kb = ldg4(pb + c), kc = ldg4(pc + c);
#pragma unroll 4
    for (long r = r0; r < rows; r += rstep) {
      float* q = dy + r * lddy + c;
      const float4 v = pro_apply<PRO_BNBWD>(ldg4(q), ldg4(z + r * ldz + c), ka, kb, kc);
      *reinterpret_cast<float4*>(q) = v;
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (tx == 0) red[ty] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void absmax_final_kernel(const float* __restrict__ part, int n,
                                                           float* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  // part[] is rewritten by every measurement and this one-block kernel tends to land on the CU
  // that reduced the previous one: read through to L2 (agent-scope loads) rather than trust
  // that the vector L1 holds no line of the previous contents (observed: stale maxima, i.e. the
  // scale of the previously measured operand, with overflow to inf in the fp16 planes)
  for (int i = threadIdx.x; i < n; i += 256)
    m = fmaxf(m, __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(part) + i,
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) *out = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---------------------------------------------------------------------------------------
// Weight preparation: W'[n][k] (= W[n*ldw + k], or W[k*ldw + n] when transposed) ->
// tiled planes out[((n_tile*KT + k_tile)*3 + plane) * 8 KB + s3_off(row, k)], zero padded.
// One thread per (row, 8-k chunk).
// ---------------------------------------------------------------------------------------
// amax != null: two-plane fp16 image of W * pow2_scale(*amax) (planes 0 and 1)
__global__ __launch_bounds__(256) void prep_weights_s3_kernel(const float* __restrict__ W, int N,
                                                              int K, long ldw, int transposed,
                                                              char* __restrict__ out,
                                                              const float* __restrict__ amax) {
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
  char* base = out + ((size_t)n_tile * KT + k_tile) * S3_OPER + s3_off(row, chunk * 8);
  if (amax != nullptr) {
    const float S = pow2_scale(load_amax(amax));
    split2h(v[0] * S, v[1] * S, h.x, l.x);
    split2h(v[2] * S, v[3] * S, h.y, l.y);
    split2h(v[4] * S, v[5] * S, h.z, l.z);
    split2h(v[6] * S, v[7] * S, h.w, l.w);
    *reinterpret_cast<uint4*>(base) = h;
    *reinterpret_cast<uint4*>(base + S3_PLANE) = l;
    return;
  }
  split2(v[0], v[1], h.x, m.x, l.x);
  split2(v[2], v[3], h.y, m.y, l.y);
  split2(v[4], v[5], h.z, m.z, l.z);
  split2(v[6], v[7], h.w, m.w, l.w);
  *reinterpret_cast<uint4*>(base) = h;
  *reinterpret_cast<uint4*>(base + S3_PLANE) = m;
  *reinterpret_cast<uint4*>(base + 2 * S3_PLANE) = l;
}

// head of a weight image (floats): [0] largest |W|, [1] largest |pro(A)|, [64..) per-block maxima
constexpr int S3_HDR_FLOATS = 64 + ABSMAX_MAX_BLOCKS;
constexpr int S3_WHDR = ((S3_HDR_FLOATS * 4 + 255) / 256) * 256;
inline size_t s3_weight_bytes(int N, int K) {
  return S3_WHDR + (size_t)((N + S3_BN - 1) / S3_BN) * ((K + S3_BK - 1) / S3_BK) * S3_OPER;
}

// LDS-DMA of 16 B per lane, hidden from the compiler's wait-count bookkeeping: with the builtin
// form hipcc drains vmcnt(0) before the next ds_read (it must assume the DMA aliases it), i.e.
// right behind the issue, exposing the full memory latency every k-tile.  Issued from inline
// asm the DMA is ours to wait for: a counted s_waitcnt vmcnt(N) before the barrier that
// publishes the stage (N = younger loads that may stay in flight).  M0 (the LDS base of the
// wave's 1-KB slice; the hardware adds lane*16) is saved and restored in the same statement.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_addr)
               : "memory");
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

// one k-tile of MFMAs for a wave tile of 128 x 64 out of the stage at `st`.
// NPL = 3: fp32-accurate six-product form; NPL = 1: plain bf16 operands (h plane only),
// one product - the reduced-precision mode of BASELINE config 3.
template <int NPL, int I0 = 0, int I1 = 4>
__device__ __forceinline__ void s3_compute(f32x16 (&acc)[4][2], const char* st, int wm, int wn,
                                           int l31, int half) {
  if constexpr (NPL == 2) {      // fp16 planes (h, l): l*h' + h*l' + h*h'
    f16x8 w[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const char* q = st + S3_OPER + s3_off(wn + j * 32 + l31, half * 8);
      w[j][0] = *reinterpret_cast<const f16x8*>(q);
      w[j][1] = *reinterpret_cast<const f16x8*>(q + S3_PLANE);
    }
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      const char* q = st + s3_off(wm + i * 32 + l31, half * 8);
      const f16x8 ah = *reinterpret_cast<const f16x8*>(q);
      const f16x8 al = *reinterpret_cast<const f16x8*>(q + S3_PLANE);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, w[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, w[j][0], acc[i][j], 0, 0, 0);
      }
    }
    return;
  }
  bf16x8 w[2][NPL];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wn + j * 32 + l31;
    const char* q = st + S3_OPER + s3_off(row, half * 8);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) w[j][pl] = *reinterpret_cast<const bf16x8*>(q + pl * S3_PLANE);
  }
#pragma unroll
  for (int i = I0; i < I1; ++i) {
    const int row = wm + i * 32 + l31;
    const char* q = st + s3_off(row, half * 8);
    bf16x8 a[NPL];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) a[pl] = *reinterpret_cast<const bf16x8*>(q + pl * S3_PLANE);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (NPL == 3) mfma6(acc[i][j], a[0], a[1], a[2], w[j][0], w[j][1], w[j][2]);
      else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], w[j][0], acc[i][j], 0, 0, 0);
    }
  }
}

// NPL-plane variant of split2 for packing: only the planes that are used are produced
template <int NPL>
__device__ __forceinline__ void splitn(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
  if constexpr (NPL == 3) {
    split2(a0, a1, h, m, l);
  } else if constexpr (NPL == 2) {
    split2h(a0, a1, h, m);
    l = 0;
  } else {
    f32x2 v = {a0, a1};
    bf16x2 hb = __builtin_convertvector(v, bf16x2);
    h = *reinterpret_cast<unsigned*>(&hb);
    m = 0; l = 0;
  }
}

// ---------------------------------------------------------------------------------------
// NT: C[M,N] = pro(A)[M,K] * W'[N,K]^T, W' pre-split (Wp).  Same prologues/epilogues as
// gemm_nt_kernel (EPI_GATE excluded).
// ---------------------------------------------------------------------------------------
template <int PRO, int EPI, int NPL>
__global__ __launch_bounds__(512, 2) void gemm_nt_s3_kernel(const NTParams p,
                                                            const char* __restrict__ Wp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = vb % p.tiles_n, tile_m = vb / p.tiles_n;
  const int m0 = tile_m * S3_BM, n0 = tile_n * S3_BN;
  const int KT = (p.K + S3_BK - 1) / S3_BK;

  const int sc = (tid & 3) * 4;      // staging k offset (float4)
  const int sr = tid >> 2;           // staging row 0..127 (+128)
  const int KP = (KT + 2) * S3_BK;   // padded length of the coefficient vectors in LDS
  float* coef = reinterpret_cast<float*>(smem + S3_LDS);
  // fp16 planes: operand scales (powers of two), folded into the prologue coefficients
  float sA = 1.f, unscale = 1.f;
  if (NPL == 2) {
    sA = pow2_scale(load_amax(p.amaxA));
    unscale = 1.f / (sA * pow2_scale(load_amax(p.amaxW)));
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Three staging register sets: tile t lives in set t%3.  Iteration kt converts tile kt+1
  // (loaded TWO iterations earlier: a k-tile of the three-product core is ~0.8 us of MFMA
  // issue, less than an HBM round trip under load, so one k-tile of distance left the loads
  // exposed - measured +25% on cache-resident A) and issues the loads of tile kt+3.
  float4 ra[3][2], ra2[3][2];
  const char* wsrc = Wp + (size_t)tile_n * KT * S3_OPER + tid * 16;
  // wave-uniform LDS byte address of this wave's 1-KB slice of a W plane in stage 0
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
  const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + S3_OPER + wave * 1024);

  // W planes: pre-split, pre-swizzled image -> straight global->LDS DMA (no registers, no
  // ds_write), issued from inline asm and waited for by the counted vmcnt that ends the k-tile.
  auto dma_w = [&](int kt, int stage) {
    const char* q = wsrc + (size_t)(kt < KT ? kt : KT - 1) * S3_OPER;   // tail: harmless re-copy
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) glds16(q + pl * S3_PLANE, wdst + stage * S3_STAGE + pl * S3_PLANE);
  };
  // A loads issued after the DMA in one k-tile: they may stay in flight across the barrier
  constexpr int NA = PRO == PRO_GATE1 ? 0 : (PRO == PRO_BNBWD ? 4 : 2);
  // A via buffer loads: descriptor = this block's 256 rows (hardware returns 0 beyond them,
  // so no exec-mask branches and no address clamps), per-lane 32-bit byte offsets computed
  // once; the k advance is added to the per-lane offset (NOT the scalar offset, which the
  // hardware leaves out of its range check: the prefetch of tile kt+2 runs past K).
  int arows = p.M - m0; arows = arows > S3_BM ? S3_BM : arows;
  // extent = up to column K of the last row (A may be a column slice of a wider tensor, so
  // "arows * lda" would reach past the end of the allocation on the last row)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.A + (size_t)m0 * p.lda), 0,
      PRO == PRO_GATE1 ? 0 : (int)(((size_t)(arows - 1) * p.lda + p.K) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((PRO == PRO_BNBWD ? p.A2 : p.A) + (size_t)m0 * (PRO == PRO_BNBWD ? p.lda2 : p.lda)), 0,
      (int)(((size_t)(arows - 1) * (PRO == PRO_BNBWD ? p.lda2 : p.lda) + p.K) * 4), 0x00020000);
  int voA[2], voA2[2];
  float gi[2] = {0.f, 0.f};          // PRO_GATE1: the row's intensity (A has one value per row)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    voA[j] = ((sr + 128 * j) * (int)p.lda + sc) * 4;
    voA2[j] = ((sr + 128 * j) * (int)p.lda2 + sc) * 4;
    if (PRO == PRO_GATE1) {
      const int row = m0 + sr + 128 * j;
      gi[j] = row < p.M ? p.A[(size_t)row * p.lda] : 0.f;
    }
  }
  auto load_tile = [&](int kt, float4 (&r)[2], float4 (&r2)[2]) {
    const int so = kt * (S3_BK * 4);
    if (PRO == PRO_GATE1) return;     // operand is generated from gi[], nothing to load
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      r[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[j] + so, 0, 0));
      if (PRO == PRO_BNBWD)
        r2[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA2, voA2[j] + so, 0, 0));
    }
  };
  auto store_tile = [&](int kt, char* st, const float4 (&r_)[2], const float4 (&r2_)[2]) {
    const int k = kt * S3_BK + sc;
    const bool kok = k < p.K || PRO != PRO_NONE;   // coefficient image is zero beyond K
    float4 ka = zero4(), kb = zero4(), kc = zero4();
    if (PRO != PRO_NONE) {        // prologue coefficients from their LDS image (no vmcnt wait)
      ka = *reinterpret_cast<const float4*>(coef + k);
      kb = *reinterpret_cast<const float4*>(coef + KP + k);
      if (PRO == PRO_BNBWD) kc = *reinterpret_cast<const float4*>(coef + 2 * KP + k);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = sr + 128 * j;
      // rows beyond M need no mask: they only feed output rows the epilogue never stores or
      // counts; k beyond K must be exact zeros (the W image is zero there, but 0*inf = nan)
      float4 v;
      if (PRO == PRO_GATE1) v = pro_apply<PRO>(make_float4(gi[j], 0.f, 0.f, 0.f), zero4(), ka, kb, kc);
      else v = pro_apply<PRO>(r_[j], r2_[j], ka, kb, kc);
      v.x = kok ? v.x : 0.f; v.y = kok ? v.y : 0.f; v.z = kok ? v.z : 0.f; v.w = kok ? v.w : 0.f;
      if (NPL == 2 && PRO == PRO_NONE) { v.x *= sA; v.y *= sA; v.z *= sA; v.w *= sA; }
      uint2 h, m, l;
      splitn<NPL>(v.x, v.y, h.x, m.x, l.x);
      splitn<NPL>(v.z, v.w, h.y, m.y, l.y);
      char* q = st + s3_off(r, sc);
      *reinterpret_cast<uint2*>(q) = h;
      if (NPL >= 2) *reinterpret_cast<uint2*>(q + S3_PLANE) = m;
      if (NPL == 3) *reinterpret_cast<uint2*>(q + 2 * S3_PLANE) = l;
    }
  };
  // one k-tile: compute tile kt from its stage, convert tile kt+1 out of register set CS into
  // the other stage, refill set CS+2 (mod 3, the set tile kt was converted from) with tile
  // kt+3.  Branch free (tail tiles load zeros).
  // TAIL (the peeled iterations behind the unrolled loop): no prefetch is issued - its data
  // would never be used, and the compiler deletes such loads anyway, which silently breaks a
  // counted wait that assumes them (seen: the W image of the last k-tile read before it
  // landed, on layers with KT % 3 == 2, when the DMA was slower than the k-tile) - and the
  // k-tile ends on vmcnt(0).
  auto iter = [&](int kt, auto cs, auto tail) {
    constexpr int CS = decltype(cs)::value;
    constexpr int FS = (CS + 2) % 3;
    constexpr bool TAIL = decltype(tail)::value;
    char* cur = smem + (kt & 1) * S3_STAGE;
    char* nxt = smem + ((kt + 1) & 1) * S3_STAGE;
    // First "use" of the registers converted in this k-tile, BEFORE any new memory operation is
    // issued: hipcc waits vmcnt(0) at the first use of a load result while an LDS-DMA is in
    // flight; here that wait only covers loads issued a whole k-tile ago.  Without it the
    // wait lands right behind the loads issued below and exposes their full latency.
    if (PRO != PRO_GATE1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        asm volatile("" : "+v"(ra[CS][j].x), "+v"(ra[CS][j].y), "+v"(ra[CS][j].z), "+v"(ra[CS][j].w));
        if (PRO == PRO_BNBWD)
          asm volatile("" : "+v"(ra2[CS][j].x), "+v"(ra2[CS][j].y), "+v"(ra2[CS][j].z), "+v"(ra2[CS][j].w));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    dma_w(kt + 1, (kt + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);   // DMA strictly before the A loads (vmcnt is in order)
    if (!TAIL) load_tile(kt + 3, ra[FS], ra2[FS]);
    // keep the memory operations at the head of the k-tile
    __builtin_amdgcn_sched_barrier(0);
    s3_compute<NPL>(acc, cur, wm, wn, l31, half);
    store_tile(kt + 1, nxt, ra[CS], ra2[CS]);
    // Interleave: the conversion VALU work of tile kt+1 is independent of the MFMAs of tile
    // kt; in-order issue only overlaps them if they alternate in program order, so ask the
    // scheduler for 1 MFMA : 3 VALU groups (MFMA issue occupies 8 of its 32 cycles).
    if constexpr (NPL == 3) {
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
    } else if constexpr (NPL == 2) {
#pragma unroll
      for (int g = 0; g < 24; ++g) {      // 24 MFMAs, ~70 conversion VALU per k-tile
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      }
    }
    // the DMA of this k-tile (older than its NA register loads) must have landed
    if (TAIL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NA) : "memory");
    __syncthreads();
  };

  dma_w(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load_tile(0, ra[0], ra2[0]);
  load_tile(1, ra[1], ra2[1]);
  load_tile(2, ra[2], ra2[2]);
  if (PRO != PRO_NONE) {
    // per-k prologue coefficient vectors -> LDS once per block, zero beyond K (which also
    // zeroes the prologue output there: relu(a*0+0) = 0, 0*dy + 0*z + 0 = 0); two tiles of
    // slack for the prefetch distance
    for (int i = tid; i < KP; i += 512) {
      const bool in = i < p.K;
      coef[i] = in ? p.pa[i] * sA : 0.f;      // every prologue is positively homogeneous in its coefficients
      coef[KP + i] = in ? p.pb[i] * sA : 0.f;
      if (PRO == PRO_BNBWD) coef[2 * KP + i] = in ? p.pc[i] * sA : 0.f;
    }
    __syncthreads();
  }
  store_tile(0, smem, ra[0], ra2[0]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first W image landed (prologue only)
  __syncthreads();
  int kt = 0;
  for (; kt + 2 < KT; kt += 3) {
    iter(kt, std::integral_constant<int, 1>{}, std::false_type{});
    iter(kt + 1, std::integral_constant<int, 2>{}, std::false_type{});
    iter(kt + 2, std::integral_constant<int, 0>{}, std::false_type{});
  }
  if (kt < KT) iter(kt, std::integral_constant<int, 1>{}, std::true_type{});
  if (kt + 1 < KT) iter(kt + 1, std::integral_constant<int, 2>{}, std::true_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (NPL == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= unscale;
  }
  // the stage buffers are free now (every DMA drained, every wave past its last fragment read):
  // use them as wave-private scratch for the row-major, 16-B-per-lane epilogue
  const bool vec_ok = ((p.N | (int)p.ldc) & 3) == 0 && (p.E1 == nullptr || ((int)p.lde1 & 3) == 0) &&
                      (p.flags & F_E1_ROWVEC) == 0 && (p.C2 == nullptr || ((int)p.ldc2 & 3) == 0);
  if (vec_ok)
    nt_epilogue_vec<EPI, 4>(acc, p, m0 + wm, n0 + wn, tile_m * 2 + (wave >> 2), lane,
                            reinterpret_cast<float*>(smem) + wave * (32 * EPI_LDW));
  else
    nt_epilogue<EPI, 4, 2>(acc, p, m0 + wm, n0 + wn, tile_m * 2 + (wave >> 2), lane);
}

// ---------------------------------------------------------------------------------------
// TN (wgrad): C[Mo,Ni] = sum_p proA(A)[p,Mo] * proB(B)[p,Ni] over this split's rows.
// Staging: thread = (column c, 8-row octet): 8 scalar loads down a column (a wave reads
// 256 contiguous bytes per row), per-column prologue coefficients live in registers, the
// 8 values become one 16-B bf16 fragment per plane in the same [row = column][16 k] image
// the NT kernel uses, so the MFMA loop is shared.
// ---------------------------------------------------------------------------------------
template <int PROA, int PROB, int NPL>
__global__ __launch_bounds__(512, 2) void gemm_tn_s3_kernel(const TNParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  // XCD-aware id: the tiles of one row split run on one XCD, so the split's rows of A and B
  // are fetched into that L2 once and shared by its tiles_m x tiles_n blocks
  int b = xcd_remap(blockIdx.x, gridDim.x);
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
  // fp16 planes: operand scales (powers of two) folded into the per-column coefficients
  float sA = 1.f, sB = 1.f;
  if (NPL == 2) {
    sA = pow2_scale(load_amax(p.amaxA));
    sB = pow2_scale(load_amax(p.amaxB));
    ka *= sA; kb *= sA; kc *= sA; qa *= sB; qb *= sB;
  }
  // Plain A operand (PRO_NONE): rows beyond the split and columns beyond Mo need no select -
  // the buffer descriptor returns 0 for the rows, and the column mask rides on the scale
  // (finite neighbours x 0).  With A exactly zero there, B needs no row mask either (finite
  // x 0), and its column mask is the zeroed coefficient pair / scale.
  constexpr bool LEAN = PROA == PRO_NONE && NPL == 2;
  const float sAm = aok ? sA : 0.f, sBm = bok ? sB : 0.f;
  const bool want_cs = p.colsum != nullptr && tile_n == 0;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // plain A operand: two staging register sets, loads issued two k-tiles ahead (a k-tile of
  // MFMAs is far shorter than an HBM round trip; the BN-backward form has a third array to
  // stage and stays at distance 1)
  constexpr bool D2 = PROA == PRO_NONE;
  float va[D2 ? 2 : 1][8], va2[1][8], vb[D2 ? 2 : 1][8];
  float csum = 0.f;
  // Buffer descriptors over this split's rows, based at column m0 / n0: rows at or beyond
  // p_end are out of range and read as 0 (no clamps, no exec-mask branches); per-lane byte
  // offsets are 32-bit.
  const int nrows = p_end - p_begin > 0 ? p_end - p_begin : 0;
  int acols = p.Mo - m0; acols = acols > 256 ? 256 : acols;
  int bcols = p.Ni - n0; bcols = bcols > 256 ? 256 : bcols;
  const size_t abytes = nrows > 0 ? ((size_t)(nrows - 1) * p.lda + acols) * 4 : 0;
  const size_t a2bytes = nrows > 0 ? ((size_t)(nrows - 1) * p.lda2 + acols) * 4 : 0;
  const size_t bbytes = nrows > 0 ? ((size_t)(nrows - 1) * p.ldb + bcols) * 4 : 0;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.A + (size_t)p_begin * p.lda + m0), 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((PROA == PRO_BNBWD ? p.A2 + (size_t)p_begin * p.lda2 + m0 : p.A)), 0,
      (int)(PROA == PRO_BNBWD ? a2bytes : 0), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.B + (size_t)p_begin * p.ldb + n0), 0, (int)bbytes, 0x00020000);
  const int voA = (oct * 8 * (int)p.lda + c) * 4, voA2 = (oct * 8 * (int)p.lda2 + c) * 4;
  const int voB = (oct * 8 * (int)p.ldb + c) * 4;
  const int stepA = (int)p.lda * 4, stepA2 = (int)p.lda2 * 4, stepB = (int)p.ldb * 4;

  auto load_tile = [&](int kt, float (&xa)[8], float (&xa2)[8], float (&xb)[8]) {
    const int oa = voA + kt * S3_BK * stepA, oa2 = voA2 + kt * S3_BK * stepA2;
    const int ob = voB + kt * S3_BK * stepB;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xa[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA, oa + j * stepA, 0, 0));
      if (PROA == PRO_BNBWD)
        xa2[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsA2, oa2 + j * stepA2, 0, 0));
      xb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsB, ob + j * stepB, 0, 0));
    }
  };
  auto store_tile = [&](int kt, char* st, const float (&xa)[8], const float (&xa2)[8],
                        const float (&xb)[8]) {
    const int r0 = p_begin + kt * S3_BK + oct * 8;
    float ta[8], tb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool rok = (r0 + j) < p_end;
      if (LEAN) {
        ta[j] = xa[j] * sAm;
        tb[j] = PROB == PRO_BNRELU ? fmaxf(fmaf(xb[j], qa, qb), 0.f) : xb[j] * sBm;
        if (want_cs) csum += ta[j];
        continue;
      }
      float x = xa[j];
      if (PROA == PRO_BNBWD) x = fmaf(ka, xa[j], fmaf(kb, xa2[j], kc));
      else if (PROA == PRO_BNRELU) x = fmaxf(fmaf(xa[j], ka, kb), 0.f);
      else if (NPL == 2) x *= sA;
      ta[j] = (rok && aok) ? x : 0.f;
      float y = xb[j];
      if (PROB == PRO_BNRELU) y = fmaxf(fmaf(xb[j], qa, qb), 0.f);
      else if (NPL == 2) y *= sB;
      tb[j] = (rok && bok) ? y : 0.f;
      csum += ta[j];
    }
    uint4 h, m, l;
    splitn<NPL>(ta[0], ta[1], h.x, m.x, l.x);
    splitn<NPL>(ta[2], ta[3], h.y, m.y, l.y);
    splitn<NPL>(ta[4], ta[5], h.z, m.z, l.z);
    splitn<NPL>(ta[6], ta[7], h.w, m.w, l.w);
    char* q = st + s3_off(c, oct * 8);
    *reinterpret_cast<uint4*>(q) = h;
    if (NPL >= 2) *reinterpret_cast<uint4*>(q + S3_PLANE) = m;
    if (NPL == 3) *reinterpret_cast<uint4*>(q + 2 * S3_PLANE) = l;
    splitn<NPL>(tb[0], tb[1], h.x, m.x, l.x);
    splitn<NPL>(tb[2], tb[3], h.y, m.y, l.y);
    splitn<NPL>(tb[4], tb[5], h.z, m.z, l.z);
    splitn<NPL>(tb[6], tb[7], h.w, m.w, l.w);
    q += S3_OPER;
    *reinterpret_cast<uint4*>(q) = h;
    if (NPL >= 2) *reinterpret_cast<uint4*>(q + S3_PLANE) = m;
    if (NPL == 3) *reinterpret_cast<uint4*>(q + 2 * S3_PLANE) = l;
  };
  // distance-1 software pipeline (a second staging register set does not fit next to the
  // 128 accumulators): loads of tile kt+1 are issued ahead of the MFMAs of tile kt and
  // converted after them.
  if (KT > 0) {
    load_tile(0, va[0], va2[0], vb[0]);
    if (D2) load_tile(1, va[D2 ? 1 : 0], va2[0], vb[D2 ? 1 : 0]);
    store_tile(0, smem, va[0], va2[0], vb[0]);
  }
  __syncthreads();
  if constexpr (D2) {
    // tile t lives in set t & 1: iteration kt converts tile kt+1 (set CS) and refills the set
    // tile kt came from with tile kt+2
    auto iter2 = [&](int kt, auto cs) {
      constexpr int CS = decltype(cs)::value;
      char* cur = smem + (kt & 1) * S3_STAGE;
      char* nxt = smem + ((kt + 1) & 1) * S3_STAGE;
      load_tile(kt + 2, va[CS ^ 1], va2[0], vb[CS ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
      s3_compute<NPL, 0, 2>(acc, cur, wm, wn, l31, half);
      __builtin_amdgcn_sched_barrier(0);
      s3_compute<NPL, 2, 4>(acc, cur, wm, wn, l31, half);
      store_tile(kt + 1, nxt, va[CS], va2[0], vb[CS]);
      if constexpr (NPL == 3) {
#pragma unroll
        for (int g = 0; g < 24; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
        }
      } else if constexpr (NPL == 2) {
#pragma unroll
        for (int g = 0; g < 12; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
        }
      }
      __syncthreads();
    };
    int kt = 0;
    for (; kt + 1 < KT; kt += 2) {
      iter2(kt, std::integral_constant<int, 1>{});
      iter2(kt + 1, std::integral_constant<int, 0>{});
    }
    if (kt < KT) iter2(kt, std::integral_constant<int, 1>{});
  }
  for (int kt = 0; kt < (D2 ? 0 : KT); ++kt) {
    char* cur = smem + (kt & 1) * S3_STAGE;
    char* nxt = smem + ((kt + 1) & 1) * S3_STAGE;
    load_tile(kt + 1, va[0], va2[0], vb[0]);       // past the last tile: out of range -> zeros
    __builtin_amdgcn_sched_barrier(0);
    // first half of the MFMAs covers the latency of the loads issued above ...
    s3_compute<NPL, 0, 2>(acc, cur, wm, wn, l31, half);
    __builtin_amdgcn_sched_barrier(0);
    // ... the second half is interleaved with the conversion of the tile they brought in
    // (1 MFMA : 8 VALU - the wgrad core splits both operands, ~200 VALU per k-tile)
    s3_compute<NPL, 2, 4>(acc, cur, wm, wn, l31, half);
    store_tile(kt + 1, nxt, va[0], va2[0], vb[0]);
    if constexpr (NPL == 3) {
#pragma unroll
      for (int g = 0; g < 24; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
    } else if constexpr (NPL == 2) {
#pragma unroll
      for (int g = 0; g < 12; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
      }
    }
    __syncthreads();
  }
  const float unscale = NPL == 2 ? 1.f / (sA * sB) : 1.f;

  float* out = p.slab + (size_t)split * p.Mo * p.Ni;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = n0 + wn + nt * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + mt * 32 + crow(r, half);
        if (row < p.Mo && col < p.Ni) out[(size_t)row * p.Ni + col] = acc[mt][nt][r] * unscale;
      }
    }

  if (p.colsum != nullptr && tile_n == 0) {
    float* red = reinterpret_cast<float*>(smem);   // all LDS reads are behind the last barrier
    red[tid] = csum;
    __syncthreads();
    if (tid < 256 && (m0 + tid) < p.Mo)
      p.colsum[(size_t)split * p.Mo + m0 + tid] = (red[tid] + red[tid + 256]) * (NPL == 2 ? 1.f / sA : 1.f);
  }
}

}  // namespace prh
