// Small-problem GEMM cores (exact fp32 MFMA).  The query side of the decoder works on B x 32
// tokens: at the reference's own batch (B = 32, train_dist.py:118-126) every Linear there is a
// [1024, 256..1024] x [256..1024] product, which the 128 x 128 tiling of gemm_nt_kernel turns
// into 16-64 workgroups of 8-32 k-steps each - 15-87 us per launch on a chip that is 90 % idle,
// 108 times per step.  These kernels trade tile size for workgroup count and latency:
//   * 32 x 32 MFMA blocks, one per wave; the four waves of a workgroup are arranged
//     TM x TN x KS (row blocks x column blocks x splits of every 64-deep k-tile), so a launch can
//     be cut into 64 x 64, 32 x 64 or 32 x 32 output tiles until it fills the chip;
//   * operands go global -> LDS by DMA (global_load_lds_dwordx4), two stages, no staging
//     registers and no conversion: one wait per 64-deep k-tile;
//   * NN form (B operand given as [K][N]) for the dgrad, so no transposed copy of W is made;
//   * the wgrad (TN) form reads both operands as they lie in memory ([rows][columns]) and splits
//     the rows over workgroups in chunks of 64..; partial sums go to the slab the existing
//     reduction kernels read.
// Same arithmetic as the fp32 core (v_mfma_f32_32x32x2_f32, fp32 accumulation); only the order
// of the k-sum differs when KS > 1.
#pragma once
#include "prh_gemm_s3.hpp"

namespace prh {

constexpr int SM_BK = 64;

// lane -> (row in a 1-KB piece, 16-B slot) for tiles whose rows are 256 B (64 floats)
// NT operand tile: [rows][64 k] floats, 16-B slots of a row XOR-swizzled by (row & 15) so that
// the ds_read_b128 fragment reads (16 lanes = 16 consecutive rows, same logical slot) are
// conflict free.  `src_row0` = pointer to (first row of the tile, k of this k-tile); rows
// beyond `rows_ok` are clamped (their products are never stored).
__device__ __forceinline__ void sm_dma_rows256(const float* base, long ld, int row0, int row_max, int k0,
                                               int piece, unsigned lds_dst, int lane, bool swz) {
  const int r = piece * 4 + (lane >> 4);
  const int ps = lane & 15;
  const int gs = swz ? (ps ^ (r & 15)) : ps;
  int gr = row0 + r;
  gr = gr > row_max ? row_max : gr;
  glds16(base + (size_t)gr * ld + k0 + gs * 4, lds_dst + piece * 1024);
}
// [rows][32 columns] floats (128-B rows): 8 rows per piece
__device__ __forceinline__ void sm_dma_rows128(const float* base, long ld, int row0, int row_max, int c0,
                                               int piece, unsigned lds_dst, int lane) {
  const int r = piece * 8 + (lane >> 3);
  const int ps = lane & 7;
  int gr = row0 + r;
  gr = gr > row_max ? row_max : gr;
  glds16(base + (size_t)gr * ld + c0 + ps * 4, lds_dst + piece * 1024);
}

// C[M,N] = A[M,K] * op(W) + bias (+ E1) (relu), EPI_BIAS flags as gemm_nt_kernel.
//   NN = false: W is [N][K] (ld ldw)           - the forward of a Linear
//   NN = true : W is [K][N] (ld ldw)           - its dgrad with the weight as stored
// Requires K % 64 == 0, lda % 4 == 0, ldw % 4 == 0; NN also N % (32 TN) == 0.
template <int TM, int TN, int KS, bool NN, int EPI = EPI_BIAS>
__global__ __launch_bounds__(256) void gemm_small_kernel(const NTParams p) {
  static_assert(TM * TN * KS == 4, "four waves");
  constexpr int BMs = TM * 32, BNs = TN * 32;
  constexpr int A_BYTES = BMs * 256, B_BYTES = BNs * 256;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int PA = BMs / 4, PB = BNs / 4;           // 1-KB pieces per stage
  constexpr int NI = (PA + PB) / 4;                   // DMA instructions per wave and stage
  static_assert((PA + PB) % 4 == 0, "pieces deal evenly");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int ks = wave / (TM * TN), wt = wave % (TM * TN);
  const int wm = (wt / TN) * 32, wn = (wt % TN) * 32;
  const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
  const int m0 = tile_m * BMs, n0 = tile_n * BNs;
  const int nk = p.K / SM_BK;
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);

  auto dma = [&](int kt, int stage) {
    const unsigned sb = lds0 + stage * STAGE;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int q = __builtin_amdgcn_readfirstlane(wave + 4 * i);
      if (q < PA) {
        sm_dma_rows256(p.A, p.lda, m0, p.M - 1, kt * SM_BK, q, sb, lane, true);
      } else if (!NN) {
        sm_dma_rows256(p.W, p.ldw, n0, p.N - 1, kt * SM_BK, q - PA, sb + A_BYTES, lane, true);
      } else if (TN == 2) {   // [64 k][64 n]
        sm_dma_rows256(p.W, p.ldw, kt * SM_BK, p.K - 1, n0, q - PA, sb + A_BYTES, lane, false);
      } else {                // [64 k][32 n]
        sm_dma_rows128(p.W, p.ldw, kt * SM_BK, p.K - 1, n0, q - PA, sb + A_BYTES, lane);
      }
    }
  };

  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;

  dma(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      dma(kt + 1, (kt + 1) & 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NI) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const char* As = smem + (kt & 1) * STAGE;
    const char* Bs = As + A_BYTES;
    const int ar = wm + l31, br = wn + l31;
#pragma unroll
    for (int gg = 0; gg < 8 / KS; ++gg) {
      const int g = ks * (8 / KS) + gg;
      const int ls = g * 2 + half;                        // logical 16-B slot = k 4*ls .. 4*ls+3
      const float4 a4 = *reinterpret_cast<const float4*>(As + ar * 256 + ((ls ^ (ar & 15)) << 4));
      float4 b4;
      if (!NN) {
        b4 = *reinterpret_cast<const float4*>(Bs + br * 256 + ((ls ^ (br & 15)) << 4));
      } else {
        const float* bq = reinterpret_cast<const float*>(Bs) + (ls * 4) * BNs + br;
        b4.x = bq[0]; b4.y = bq[BNs]; b4.z = bq[2 * BNs]; b4.w = bq[3 * BNs];
      }
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc[0][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc[0][0], 0, 0, 0);
    }
    __syncthreads();
  }
  if (KS > 1) {     // sum the k-splits: waves ks > 0 hand their block to wave (0, wt) through LDS
    float* red = reinterpret_cast<float*>(smem);
    if (ks > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((ks - 1) * TM * TN + wt) * 1024 + r * 64 + lane] = acc[0][0][r];
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll
    for (int s = 0; s < KS - 1; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][r] += red[(s * TM * TN + wt) * 1024 + r * 64 + lane];
  }
  nt_epilogue<EPI, 1, 1>(acc, p, m0 + wm, n0 + wn, (m0 + wm) >> 5, lane);
}
template <int TM, int TN, int KS>
constexpr int small_lds() {
  constexpr int st = 2 * (TM + TN) * 32 * 256, red = (KS - 1) * TM * TN * 4096;
  return st > red ? st : red;
}

// wgrad: slab[split][Mo][Ni] = sum over this split's rows of A[p][Mo]^T B[p][Ni], 64 x 64 tiles,
// rows_per_split a multiple of 64 (the launcher requires P, Mo and Ni to be multiples of 64).
// colsum[split][Mo] = column sums of A (tile_n == 0).
__global__ __launch_bounds__(256) void gemm_tn_small_kernel(const TNParams p) {
  constexpr int T_BYTES = 64 * 256, STAGE = 2 * T_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  int b = blockIdx.x;
  const int tile_n = b % p.tiles_n; b /= p.tiles_n;
  const int tile_m = b % p.tiles_m; b /= p.tiles_m;
  const int split = b;
  const int m0 = tile_m * 64, n0 = tile_n * 64;
  const int p_begin = split * p.rows_per_split;
  int p_end = p_begin + p.rows_per_split;
  if (p_end > p.P) p_end = p.P;
  const int nk = (p_end - p_begin) / SM_BK;
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
  // columns beyond Mo / Ni: the launcher requires Mo % 64 == 0 and Ni % 64 == 0
  auto dma = [&](int kt, int stage) {
    const unsigned sb = lds0 + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = __builtin_amdgcn_readfirstlane(wave + 4 * i);
      if (q < 16) sm_dma_rows256(p.A, p.lda, p_begin + kt * SM_BK, p.P - 1, m0, q, sb, lane, false);
      else sm_dma_rows256(p.B, p.ldb, p_begin + kt * SM_BK, p.P - 1, n0, q - 16, sb + T_BYTES, lane, false);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool do_colsum = p.colsum != nullptr && tile_n == 0;
  float csum = 0.f;
  if (nk > 0) dma(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      dma(kt + 1, (kt + 1) & 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const float* As = reinterpret_cast<const float*>(smem + (kt & 1) * STAGE);
    const float* Bs = As + 64 * 64;
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      const float a = As[(kk * 2 + half) * 64 + wm + l31];
      const float bb = Bs[(kk * 2 + half) * 64 + wn + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
    }
    if (do_colsum) {     // thread = (column tid & 63, rows 16 * wave ..)
#pragma unroll
      for (int r = 0; r < 16; ++r) csum += As[(wave * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  float* out = p.slab + (size_t)split * p.Mo * p.Ni;
  const int col = n0 + wn + l31;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm + crow(r, half);
    out[(size_t)row * p.Ni + col] = acc[r];
  }
  if (do_colsum) {
    float* red = reinterpret_cast<float*>(smem);
    red[wave * 64 + lane] = csum;
    __syncthreads();
    if (tid < 64) p.colsum[(size_t)split * p.Mo + m0 + tid] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
  }
}


// wgrad without a slab: C[Mo,Ni] (ld ldc) = A[P,Mo]^T B[P,Ni], colsum_out[Mo] = column sums of A
// (optional), for short row ranges (P <= 4096).  One 32 x 32 output tile per workgroup; the four
// waves split every 64-row k-tile (16 rows each) and add their blocks through LDS, so the result
// is written once, in a fixed order, and no reduction launch follows.  P % 64 == 0, Mo % 32 == 0,
// Ni % 32 == 0.
__global__ __launch_bounds__(256) void gemm_tn_direct_kernel(const TNParams p, float* __restrict__ C, long ldc,
                                                             float* __restrict__ colsum_out) {
  constexpr int T_BYTES = 64 * 128, STAGE = 2 * T_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int tile_n = blockIdx.x % p.tiles_n, tile_m = blockIdx.x / p.tiles_n;
  const int m0 = tile_m * 32, n0 = tile_n * 32;
  const int nk = p.P / SM_BK;
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
  auto dma = [&](int kt, int stage) {
    const unsigned sb = lds0 + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = __builtin_amdgcn_readfirstlane(wave + 4 * i);
      if (q < 8) sm_dma_rows128(p.A, p.lda, kt * SM_BK, p.P - 1, m0, q, sb, lane);
      else sm_dma_rows128(p.B, p.ldb, kt * SM_BK, p.P - 1, n0, q - 8, sb + T_BYTES, lane);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool do_colsum = colsum_out != nullptr && tile_n == 0;
  float csum = 0.f;
  if (nk > 0) dma(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      dma(kt + 1, (kt + 1) & 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const float* As = reinterpret_cast<const float*>(smem + (kt & 1) * STAGE) + wave * 16 * 32;
    const float* Bs = As + 64 * 32;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float a = As[(kk * 2 + half) * 32 + l31];
      const float bb = Bs[(kk * 2 + half) * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
    }
    if (do_colsum) {     // lane = (column l31, rows 8 half .. of this wave's 16)
#pragma unroll
      for (int r = 0; r < 8; ++r) csum += As[(half * 8 + r) * 32 + l31];
    }
    __syncthreads();
  }
  float* red = reinterpret_cast<float*>(smem);          // [3][16][64] blocks + [4][64] column sums
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  if (do_colsum) red[3 * 1024 + wave * 64 + lane] = csum;
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += red[(s * 16 + r) * 64 + lane];
    const int col = n0 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) C[(size_t)(m0 + crow(r, half)) * ldc + col] = acc[r];
  } else if (wave == 1 && do_colsum && lane < 32) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += red[3 * 1024 + w * 64 + lane] + red[3 * 1024 + w * 64 + 32 + lane];
    colsum_out[m0 + lane] = s;
  }
}

}  // namespace prh
