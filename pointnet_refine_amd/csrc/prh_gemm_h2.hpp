// Split-fp16 NT core, second generation: C[M,N] = pro(A)[M,K] * W[N,K]^T with two scaled fp16
// planes per operand and three products per MAC (prh_gemm_s3.hpp, NPL = 2, for the numerics),
// re-tiled around what the measurements of the first generation showed:
//   * v_mfma_f32_16x16x32_f16 instead of 32x32x16: the same FLOPs per cycle, but the chip holds
//     a higher clock under it once the matrix pipe is the power limit (MI355X_MICROARCH.md,
//     'DVFS give-back' item 7: 1.12-1.15x FLOP/s with operands re-read from LDS) - the forward
//     fusion GEMM ran at 1.64 GHz with the 32x32x16 form;
//   * BK = 32: one barrier and one fragment-read restart per 32-deep k-tile instead of per 16,
//     and the activation tile is fetched in full 128-B lines (a 16-deep fp32 k-tile is half a
//     line per row; re-fetching the other half from L2 one k-tile later cost 20-25 %, measured
//     with the operand pinned in L1).
// Tile 256 x 256 x 32, 512 threads (8 waves as 2(M) x 4(N)), wave tile 128 x 64 = 8 x 4 MFMA
// blocks (128 accumulator registers), LDS 2 stages x (A 2 planes x 16 KB + W 2 planes x 16 KB)
// = 128 KB + the prologue coefficient vectors.  LDS rows are 64 B (32 fp16); the four 16-B
// chunks of a row are permuted by chunk ^ (-(row >> 2) & 3), which makes every ds_read_b128
// fragment read (lane = row 0..15, k-chunk = lane >> 4) conflict-free for the hardware's
// 16-lane service groups.  Weights: pre-split, pre-scaled, pre-swizzled image, LDS-DMA.
// Activations: buffer loads two register sets deep, converted (prologue + split) behind the
// MFMAs of the previous k-tile.  Prologues: NONE / BNRELU / GATE1 (BN-backward operands are
// materialised by bn_bwd_apply_kernel in this mode).  Epilogue: the shared vector epilogue.
#pragma once
#include "prh_gemm_s3.hpp"

namespace prh {

#ifndef PRH_H2_DIRECT_EPILOGUE
#define PRH_H2_DIRECT_EPILOGUE 0
#endif
// [r03] 1: transposed MFMA blocks + stores straight from registers (nt_epilogue_t, prh_gemm.hpp).  Built,
// parity-green on every GEMM / encoder / model test, measured on one box against the LDS-transposing vector
// epilogue (scripts/ab_epilogue.sh, B=4096): fusion dgrad 49.2 vs 46.5 ms, fusion fwd 40.9 vs 40.4, conv5 fwd
// 14.6 vs 13.9, gate 11.9 vs 8.7, K/V projections 5.86 vs 5.65, step 453.5 vs 443.1 ms - SLOWER everywhere: a
// store or operand read of 16 rows x 64 B per instruction costs more than the scratch round trip that turns
// it into 4 rows x 256 B.  Kept as a compile-time option for that comparison; the default stays 0.
constexpr bool H2_DIRECT_EPILOGUE = PRH_H2_DIRECT_EPILOGUE != 0;
constexpr int H2_BK = 32;
constexpr int H2_PLANE = 256 * H2_BK * 2;         // one [256][32] fp16 plane = 16 KB
constexpr int H2_OPER = 2 * H2_PLANE;             // two planes = 32 KB
constexpr int H2_STAGE = 2 * H2_OPER;             // A + W = 64 KB
constexpr int H2_LDS = 2 * H2_STAGE;              // 128 KB

// byte offset of element (row, k) inside one plane
__device__ __forceinline__ int h2_off(int row, int k) {
  return row * 64 + ((((k >> 3) ^ (-(row >> 2))) & 3) << 4) + ((k & 7) << 1);
}

inline size_t h2_weight_bytes(int N, int K) {
  return S3_WHDR + (size_t)((N + 255) / 256) * ((K + H2_BK - 1) / H2_BK) * H2_OPER;
}

// W[n][k] -> tiled planes out[(n_tile*KT + k_tile) * 32 KB + plane * 16 KB + h2_off(row, k)],
// scaled by pow2_scale(*amax), zero padded.  One thread per (row, 8-k chunk).
__global__ __launch_bounds__(256) void prep_weights_h2_kernel(const float* __restrict__ W, int N,
                                                              int K, long ldw, char* __restrict__ out,
                                                              const float* __restrict__ amax) {
  const int KT = (K + H2_BK - 1) / H2_BK;
  const int NT_ = (N + 255) / 256;
  const long total = (long)NT_ * 256 * KT * 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int chunk = (int)(i & 3);
  long t = i >> 2;
  const int k_tile = (int)(t % KT); t /= KT;
  const int row = (int)(t & 255);
  const int n_tile = (int)(t >> 8);
  const int n = n_tile * 256 + row;
  const float S = pow2_scale(load_amax(amax));
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k_tile * H2_BK + chunk * 8 + j;
    v[j] = (n < N && k < K) ? W[(size_t)n * ldw + k] * S : 0.f;
  }
  uint4 h, l;
  split2h(v[0], v[1], h.x, l.x);
  split2h(v[2], v[3], h.y, l.y);
  split2h(v[4], v[5], h.z, l.z);
  split2h(v[6], v[7], h.w, l.w);
  char* base = out + ((size_t)n_tile * KT + k_tile) * H2_OPER + h2_off(row, chunk * 8);
  *reinterpret_cast<uint4*>(base) = h;
  *reinterpret_cast<uint4*>(base + H2_PLANE) = l;
}

// MFMAs of one 32-deep k-tile for a wave tile of 128 x 64 (row blocks I0..I1-1 of 8): the W
// fragments of the wave's 4 column blocks stay in registers, the A fragments stream through.
// (A wgrad core on this loop - 32 rows per k-tile, 16 scalar loads per operand and thread - was
// built and measured: 13.3 vs 12.9 ms on the fusion wgrad.  The TN core is bound by its
// column-wise staging, not by the matrix pipe's clock, so it stays on the 32x32x16 loop.)
// [r03] TR: the two operands of every MFMA are swapped (weights as the A operand), so a block comes out
// transposed - acc[i][j][r] = C[row 16 i + (lane & 15)][col 16 j + 4 (lane >> 4) + r] - the layout
// nt_epilogue_t stores straight from registers (prh_gemm.hpp).  Same fragments, same LDS reads, same sums.
template <int I0 = 0, int I1 = 8, bool TR = false>
__device__ __forceinline__ void h2_compute(f32x4 (&acc)[8][4], const char* st, int wm, int wn, int l15,
                                           int kc) {
  f16x8 wh[4], wl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const char* q = st + H2_OPER + h2_off(wn + j * 16 + l15, kc * 8);
    wh[j] = *reinterpret_cast<const f16x8*>(q);
    wl[j] = *reinterpret_cast<const f16x8*>(q + H2_PLANE);
  }
#pragma unroll
  for (int i = I0; i < I1; ++i) {
    const char* q = st + h2_off(wm + i * 16 + l15, kc * 8);
    const f16x8 ah = *reinterpret_cast<const f16x8*>(q);
    const f16x8 al = *reinterpret_cast<const f16x8*>(q + H2_PLANE);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (TR) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], al, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[j], ah, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[j], ah, acc[i][j], 0, 0, 0);
      } else {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[j], acc[i][j], 0, 0, 0);
      }
    }
  }
}

// PP ("ping-pong", [r03]): the same tile, staging and numerics on a phase-split k-loop.  The two wave rows
// of the workgroup (waves 0-3 and 4-7: one wave of each on every SIMD) run ONE barrier apart: while a row
// issues the 24 MFMAs of two of its eight row blocks, the other row is in its memory phase (the fragment reads
// of its next two row blocks, one row pass of the next k-tile's conversion, its share of the loads), then
// they swap - 8 raw s_barriers per 32-deep k-tile instead of one __syncthreads, no MFMA and no staging
// instruction of the same wave in the same phase (`cdna_hip_programming.md`, "The 256^2 8-phase template").
// Hazards: a memory phase ends on lgkmcnt(0) before its barrier (its ds_reads are done before the other row
// may overwrite the stage, its ds_writes are visible); the W image of tile kt+1 is DMA-ed in phase 0 and
// awaited (counted vmcnt) before the barrier of phase 3; tile kt+1 is first read two barriers later.
// Measured (PRH_H2_PP=1, scripts/ab_pp.sh + scripts/pmc_pp.sh, one box each): parity-green on every GEMM / encoder /
// model / fp64-oracle test; plain-operand launches 3-4 % faster (fusion dgrad 45.8 vs 47.7 ms, K=1536 8.25 vs 8.55),
// BN+ReLU-prologue launches 3-4 % slower (fusion fwd 42.5 vs 40.9), step 437.5-438.4 vs 437.6-438.5 ms.  The SQ
// counters say why it cannot matter: the matrix pipe's busy share moves (fusion dgrad 0.57 -> 0.62, fusion fwd 0.59 ->
// 0.57) and the clock the chip holds moves the other way (1.76 -> 1.70, 1.75 -> 1.83 GHz): busy x clock stays at
// 1.01-1.05 GHz-equivalent of 2.4 on both loops.  This core runs at the board's power limit for its instruction
// mix (fp16 MFMA on random mantissas + fp32 operand stream from HBM + split arithmetic); a denser issue stream is
// paid back in clock (MI355X_MICROARCH.md, "DVFS give-back" item 3).  Off by default.
template <int PRO, int EPI, bool PP = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_h2_kernel(const NTParams p,
                                                            const char* __restrict__ Wp) {
  static_assert(PRO == PRO_NONE || PRO == PRO_BNRELU || PRO == PRO_GATE1, "prologue not supported");
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PRH_STAMP
  unsigned t_acc[4] = {0u, 0u, 0u, 0u};
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kc = lane >> 4;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = vb % p.tiles_n, tile_m = vb / p.tiles_n;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int KT = (p.K + H2_BK - 1) / H2_BK;

  const int sc = (tid & 7) * 4;      // staging k offset (float4): 8 lanes cover a 128-B row
  const int sr = tid >> 3;           // staging row 0..63 (+64 j)
  const int KP = (KT + 2) * H2_BK;   // padded length of the coefficient vectors in LDS
  float* coef = reinterpret_cast<float*>(smem + H2_LDS);
  const float sA = pow2_scale(load_amax(p.amaxA));
  const float unscale = 1.f / (sA * pow2_scale(load_amax(p.amaxW)));

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // two staging register sets of 4 float4: tile t lives in set t & 1, loaded one (32-deep)
  // k-tile ahead of its conversion
  float4 ra[2][4];
  const char* wsrc = Wp + (size_t)tile_n * KT * H2_OPER + tid * 16;
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
  const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + H2_OPER + wave * 1024);
  // W: 2 planes x 16 KB per k-tile = 4 LDS-DMA instructions of 8 KB (512 lanes x 16 B)
  auto dma_w = [&](int kt, int stage) {
    const char* q = wsrc + (size_t)(kt < KT ? kt : KT - 1) * H2_OPER;   // tail: harmless re-copy
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) glds16(q + pc * 8192, wdst + stage * H2_STAGE + pc * 8192);
  };
  constexpr int NA = PRO == PRO_GATE1 ? 0 : 4;   // A loads issued behind the DMA in one k-tile
  int arows = p.M - m0; arows = arows > 256 ? 256 : arows;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.A + (size_t)m0 * p.lda), 0,
      PRO == PRO_GATE1 ? 0 : (int)(((size_t)(arows - 1) * p.lda + p.K) * 4), 0x00020000);
  int voA[4];
  float gi[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    voA[j] = ((sr + 64 * j) * (int)p.lda + sc) * 4;
    if (PRO == PRO_GATE1) {
      const int row = m0 + sr + 64 * j;
      gi[j] = row < p.M ? p.A[(size_t)row * p.lda] : 0.f;
    }
  }
  auto load_tile = [&](int kt, float4 (&r)[4]) {
    if (PRO == PRO_GATE1) return;
    const int so = kt * (H2_BK * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      r[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[j] + so, 0, 0));
  };
  auto store_tile = [&](int kt, char* st, const float4 (&r_)[4]) {
    const int k = kt * H2_BK + sc;
    const bool kok = k < p.K || PRO != PRO_NONE;   // coefficient image is zero beyond K
    float4 ka = zero4(), kb = zero4();
    if (PRO != PRO_NONE) {
      ka = *reinterpret_cast<const float4*>(coef + k);
      kb = *reinterpret_cast<const float4*>(coef + KP + k);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = sr + 64 * j;
      float4 v;
      if (PRO == PRO_GATE1) v = pro_apply<PRO>(make_float4(gi[j], 0.f, 0.f, 0.f), zero4(), ka, kb, zero4());
      else v = pro_apply<PRO>(r_[j], zero4(), ka, kb, zero4());
      v.x = kok ? v.x : 0.f; v.y = kok ? v.y : 0.f; v.z = kok ? v.z : 0.f; v.w = kok ? v.w : 0.f;
      if (PRO == PRO_NONE) { v.x *= sA; v.y *= sA; v.z *= sA; v.w *= sA; }
      uint2 h, l;
      split2h(v.x, v.y, h.x, l.x);
      split2h(v.z, v.w, h.y, l.y);
      char* q = st + h2_off(r, sc);
      *reinterpret_cast<uint2*>(q) = h;
      *reinterpret_cast<uint2*>(q + H2_PLANE) = l;
    }
  };
  // one k-tile: compute tile kt, convert tile kt+1 (set CS) into the other stage, refill set
  // CS^1 (the set tile kt came from) with tile kt+2.  TAIL: no prefetch, ends on vmcnt(0)
  // (the compiler deletes prefetches nobody reads, which would break the counted wait).
  auto iter = [&](int kt, auto cs, auto tail) {
    constexpr int CS = decltype(cs)::value;
    constexpr bool TAIL = decltype(tail)::value;
    char* cur = smem + (kt & 1) * H2_STAGE;
    char* nxt = smem + ((kt + 1) & 1) * H2_STAGE;
    if (PRO != PRO_GATE1) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        asm volatile("" : "+v"(ra[CS][j].x), "+v"(ra[CS][j].y), "+v"(ra[CS][j].z), "+v"(ra[CS][j].w));
      __builtin_amdgcn_sched_barrier(0);
    }
    dma_w(kt + 1, (kt + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);   // DMA strictly before the A loads (vmcnt is in order)
    if (!TAIL) load_tile(kt + 2, ra[CS ^ 1]);
    __builtin_amdgcn_sched_barrier(0);
    h2_compute<0, 8, H2_DIRECT_EPILOGUE>(acc, cur, wm, wn, l15, kc);
    store_tile(kt + 1, nxt, ra[CS]);
    // interleave the conversion (~130 VALU) with the 96 MFMAs
#pragma unroll
    for (int g = 0; g < 48; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
    if (TAIL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NA) : "memory");
    __syncthreads();
  };

  dma_w(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load_tile(0, ra[0]);
  load_tile(1, ra[1]);
  if (PRO != PRO_NONE) {
    for (int i = tid; i < KP; i += 512) {
      const bool in = i < p.K;
      coef[i] = in ? p.pa[i] * sA : 0.f;
      coef[KP + i] = in ? p.pb[i] * sA : 0.f;
    }
    __syncthreads();
  }
  store_tile(0, smem, ra[0]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // tile 0 was converted out of set 0 above: iteration kt converts tile kt+1 from set (kt+1)&1
  PRH_TICK(0)                        // prologue
  int kt = 0;
  if constexpr (!PP) {
    for (; kt + 2 < KT; kt += 2) {
      iter(kt, std::integral_constant<int, 1>{}, std::false_type{});
      iter(kt + 1, std::integral_constant<int, 0>{}, std::false_type{});
    }
    if (kt < KT) iter(kt, std::integral_constant<int, 1>{}, std::true_type{});
    if (kt + 1 < KT) iter(kt + 1, std::integral_constant<int, 0>{}, std::true_type{});
  } else {
    const int wr = __builtin_amdgcn_readfirstlane(wave >> 2);
    const int aoff = h2_off(wm + l15, kc * 8);                  // + i * 1024 (+ H2_PLANE): row block i
    const int woff = H2_OPER + h2_off(wn + l15, kc * 8);        // + j * 1024 (+ H2_PLANE): column block j
    f16x8 wh[4], wl[4], ah[2], al[2];
    // one row pass (64 rows x this thread's 4 k-values) of tile kt+1: prologue, scale, split, two 8-B stores
    auto store_pass = [&](int kt1, char* st, const float4& r_, const int j) {
      const int k = kt1 * H2_BK + sc;
      const bool kok = k < p.K || PRO != PRO_NONE;
      float4 ka = zero4(), kb = zero4();
      if (PRO != PRO_NONE) {
        ka = *reinterpret_cast<const float4*>(coef + k);
        kb = *reinterpret_cast<const float4*>(coef + KP + k);
      }
      float4 v;
      if (PRO == PRO_GATE1) v = pro_apply<PRO>(make_float4(gi[j], 0.f, 0.f, 0.f), zero4(), ka, kb, zero4());
      else v = pro_apply<PRO>(r_, zero4(), ka, kb, zero4());
      v.x = kok ? v.x : 0.f; v.y = kok ? v.y : 0.f; v.z = kok ? v.z : 0.f; v.w = kok ? v.w : 0.f;
      if (PRO == PRO_NONE) { v.x *= sA; v.y *= sA; v.z *= sA; v.w *= sA; }
      uint2 h, l;
      split2h(v.x, v.y, h.x, l.x);
      split2h(v.z, v.w, h.y, l.y);
      char* q = st + h2_off(sr + 64 * j, sc);
      *reinterpret_cast<uint2*>(q) = h;
      *reinterpret_cast<uint2*>(q + H2_PLANE) = l;
    };
    auto phase = [&](int kt_, auto mp, auto cs, auto tail) {
      constexpr int MP = decltype(mp)::value;
      constexpr int CS = decltype(cs)::value;
      constexpr bool TAIL = decltype(tail)::value;
      const char* cur = smem + (kt_ & 1) * H2_STAGE;
      char* nxt = smem + ((kt_ + 1) & 1) * H2_STAGE;
      // ---- memory phase MP: fragments of row blocks 2 MP, 2 MP + 1 (and, in phase 0, of the 4 column blocks)
      __builtin_amdgcn_sched_barrier(0);
      if (MP == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          wh[j] = *reinterpret_cast<const f16x8*>(cur + woff + j * 1024);
          wl[j] = *reinterpret_cast<const f16x8*>(cur + woff + j * 1024 + H2_PLANE);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const f16x8*>(cur + aoff + (2 * MP + i) * 1024);
        al[i] = *reinterpret_cast<const f16x8*>(cur + aoff + (2 * MP + i) * 1024 + H2_PLANE);
      }
      __builtin_amdgcn_sched_barrier(0);
      // row pass MP of tile kt+1 out of register set CS (loaded one k-tile ago)
      store_pass(kt_ + 1, nxt, ra[CS][MP], MP);
      __builtin_amdgcn_sched_barrier(0);
      if (MP == 0) dma_w(kt_ + 1, (kt_ + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!TAIL && PRO != PRO_GATE1)
        ra[CS ^ 1][MP] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[MP] + (kt_ + 2) * (H2_BK * 4), 0, 0));
      __builtin_amdgcn_sched_barrier(0);
      if (MP == 3) {      // the W image of tile kt+1 (4 DMA pieces of phase 0) has landed; the row passes behind it may fly on
        if (TAIL || PRO == PRO_GATE1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA phase MP: 2 row blocks x 4 column blocks x 3 products, product-outer (8 independent accumulators
      // between two MFMAs on the same one)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[2 * MP + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], wh[j], acc[2 * MP + i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[2 * MP + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], wl[j], acc[2 * MP + i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[2 * MP + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], wh[j], acc[2 * MP + i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto tile = [&](int kt_, auto cs, auto tail) {
      phase(kt_, std::integral_constant<int, 0>{}, cs, tail);
      phase(kt_, std::integral_constant<int, 1>{}, cs, tail);
      phase(kt_, std::integral_constant<int, 2>{}, cs, tail);
      phase(kt_, std::integral_constant<int, 3>{}, cs, tail);
    };
    if (wr == 1) __builtin_amdgcn_s_barrier();       // the second wave row runs one barrier behind the first
    for (; kt + 2 < KT; kt += 2) {
      tile(kt, std::integral_constant<int, 1>{}, std::false_type{});
      tile(kt + 1, std::integral_constant<int, 0>{}, std::false_type{});
    }
    if (kt < KT) tile(kt, std::integral_constant<int, 1>{}, std::true_type{});
    if (kt + 1 < KT) tile(kt + 1, std::integral_constant<int, 0>{}, std::true_type{});
    if (wr == 0) __builtin_amdgcn_s_barrier();       // same number of barriers on both rows
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  PRH_TICK(1)                        // k-loop
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] *= unscale;
  if (H2_DIRECT_EPILOGUE)
    nt_epilogue_t<EPI>(acc, p, m0 + wm, n0 + wn, tile_m * 2 + (wave >> 2), lane);
  else
    nt_epilogue_vec<EPI, 4>(acc, p, m0 + wm, n0 + wn, tile_m * 2 + (wave >> 2), lane,
                            reinterpret_cast<float*>(smem) + wave * (32 * EPI_LDW));
#ifdef PRH_STAMP
  PRH_TICK(2)                        // epilogue: instructions issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PRH_TICK(3)                        // its stores acknowledged
  // one workgroup of: fusion dgrad (slot block 0), fusion forward (1), K/V projection forward (2), conv5 forward (3)
  const int which = (p.K == 1024 && p.N == 1984) ? 0 : (p.K == 1984 && p.N == 1024) ? 1 : (p.K == 256 && p.N == 1536) ? 2 : (p.K == 512 && p.N == 1024) ? 3 : -1;
  if (g_prh_stamp != nullptr && which >= 0 && blockIdx.x == gridDim.x / 2 + 3 && lane == 0 && (wave & 3) == 0) {
    unsigned* o = g_prh_stamp + 64 + which * 16 + (wave >> 2) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = t_acc[i];
    o[4] = (unsigned)KT;
  }
#endif
}

// ---------------------------------------------------------------------------------------
// TN (wgrad) with ROW-MAJOR staging and transposed fragment reads:
//   C[Mo,Ni] = sum_p A[p,Mo] * proB(B)[p,Ni] over this split's rows, two scaled fp16 planes.
// The first TN core stages column-wise (each lane walks down a column with 4-byte loads) so
// that eight k-values of one column form a 16-B fragment in LDS; its limit is that global
// staging.  Here the k-tile (16 rows x 256 columns) is loaded as it lies in memory - 16-B
// loads, a wave per 1-KB row - split, and written ROW-major ([k][column] fp16, 8-B stores,
// rows padded to 576 B), and the MFMA operands are gathered by ds_read_b64_tr_b16: each
// 16-lane group reads a 4 (k) x 16 (columns) block and receives it column-major, two reads per
// 8-deep fragment.  The 576-B row stride puts the four rows of a 32-lane half on disjoint
// 16-bank ranges.  Plain A operand (dz is materialised), B plain or BN+ReLU; needs Mo, Ni and
// the leading dimensions to be multiples of 4.  32x32x16 MFMA, 256 x 256 tile, 8 waves as
// 2 x 4, two staging register sets (loads two k-tiles ahead).
// (A 16x16x32 / 32-row-k-tile form of this core - 544-B rows with the k-rows permuted so the
// eight 4-row blocks of a half's gather sit on eight 8-bank ranges - was built and measured:
// 11.9 vs 12.1 ms on the fusion wgrad, slower on the smaller ones; not kept.)
// ---------------------------------------------------------------------------------------
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
constexpr int TR_ROW = 576;                      // bytes per image row: 256 fp16 + 64 B pad
constexpr int TR_PLANE = 16 * TR_ROW;            // 9216
constexpr int TR_OPER = 2 * TR_PLANE;            // h, l
constexpr int TR_STAGE = 2 * TR_OPER;            // A, B
constexpr int TR_LDS = 2 * TR_STAGE;             // 73728

__device__ __forceinline__ f16x8 tr_fragment(const char* plane, int col0, int lane) {
  const int g = lane >> 4, j = lane & 15;
  const char* a = plane + ((g >> 1) * 8 + (j >> 2)) * TR_ROW + (col0 + (g & 1) * 16 + 4 * (j & 3)) * 2;
  const fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
      (__attribute__((address_space(3))) fp16x4*)(a));
  const fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
      (__attribute__((address_space(3))) fp16x4*)(a + 4 * TR_ROW));
  struct Pair { fp16x4 a, b; } pr = {lo, hi};      // k 0..3 | k 4..7 of this lane's half
  return __builtin_bit_cast(f16x8, pr);
}

// Pacing.  The tiles of one split run together on one XCD (xcd_remap) and share the split's
// rows of A and B through its 4 MB L2 - but only while they stay within ~340 rows of each
// other.  Nothing holds them there, and the launch is bistable: in step, the fusion wgrad at
// B=4096 fetches 52 GB for 50.5 GB of operands and takes 45 ms; once the tiles have drifted
// apart the misses keep them apart, and the same launch fetches 160-180 GB and takes 49 ms
// (both seen, box to box, with rocprofv3 --pmc FETCH_SIZE; profiles/r01u_pmc_wgrad_pacing.txt
// has both settings on a box that drifts, plus an induced skew).  So every TR_PACE k-tiles thread
// 0 publishes the block's progress and, when the block is more than one unit ahead of its
// split's mean, sleeps until the others catch up (the other waves wait at the next barrier).
// The wait is bounded and abandoned for good after one time-out: blocks that are not
// co-resident cost a fraction of a millisecond once, never a hang.
constexpr int TR_PACE = 8;            // k-tiles (128 rows) between progress reports
constexpr int TR_PACE_SPINS = 256;    // x (s_sleep 16 + one L2 load): ~0.3 ms at most, once
template <int PROB>
__global__ __launch_bounds__(512, 2) void gemm_tn_tr_kernel(const TNParams p) {
  static_assert(PROB == PRO_NONE || PROB == PRO_BNRELU, "prologue not supported");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = b % p.tiles_n; b /= p.tiles_n;
  const int tile_m = b % p.tiles_m; b /= p.tiles_m;
  const int split = b;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int p_begin = split * p.rows_per_split;
  int p_end = p_begin + p.rows_per_split;
  if (p_end > p.P) p_end = p.P;
  const int KT = (p_end - p_begin + S3_BK - 1) / S3_BK;

  const int c4 = (tid & 63) * 4, r8 = tid >> 6;        // staging: 4 columns, rows r8 and r8 + 8
  const bool aok = (m0 + c4) < p.Mo, bok = (n0 + c4) < p.Ni;
  const float sA = pow2_scale(load_amax(p.amaxA)), sB = pow2_scale(load_amax(p.amaxB));
  const float sAm = aok ? sA : 0.f, sBm = bok ? sB : 0.f;
  float4 qa = zero4(), qb = zero4();
  if (PROB == PRO_BNRELU && bok) {
    qa = ldg4(p.qa + n0 + c4); qb = ldg4(p.qb + n0 + c4);
    qa.x *= sB; qa.y *= sB; qa.z *= sB; qa.w *= sB; qb.x *= sB; qb.y *= sB; qb.z *= sB; qb.w *= sB;
  }
  const bool want_cs = p.colsum != nullptr && tile_n == 0;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nrows = p_end - p_begin > 0 ? p_end - p_begin : 0;
  int acols = p.Mo - m0; acols = acols > 256 ? 256 : acols;
  int bcols = p.Ni - n0; bcols = bcols > 256 ? 256 : bcols;
  const size_t abytes = nrows > 0 ? ((size_t)(nrows - 1) * p.lda + acols) * 4 : 0;
  const size_t bbytes = nrows > 0 ? ((size_t)(nrows - 1) * p.ldb + bcols) * 4 : 0;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.A + (size_t)p_begin * p.lda + m0), 0, (int)abytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.B + (size_t)p_begin * p.ldb + n0), 0, (int)bbytes, 0x00020000);
  const int stepA = (int)p.lda * 4, stepB = (int)p.ldb * 4;
  const int voA = r8 * stepA + c4 * 4, voB = r8 * stepB + c4 * 4;

  float4 va[2][2], vb[2][2];
  float4 cs = zero4();
  auto load_tile = [&](int kt, float4 (&xa)[2], float4 (&xb)[2]) {
    const int oa = voA + kt * S3_BK * stepA, ob = voB + kt * S3_BK * stepB;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      xa[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, oa + j * 8 * stepA, 0, 0));
      xb[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsB, ob + j * 8 * stepB, 0, 0));
    }
  };
  auto store_tile = [&](char* st, const float4 (&xa)[2], const float4 (&xb)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float4 a = xa[j];
      a.x *= sAm; a.y *= sAm; a.z *= sAm; a.w *= sAm;
      if (want_cs) { cs.x += a.x; cs.y += a.y; cs.z += a.z; cs.w += a.w; }
      float4 y = xb[j];
      if (PROB == PRO_BNRELU) {
        y.x = fmaxf(fmaf(y.x, qa.x, qb.x), 0.f); y.y = fmaxf(fmaf(y.y, qa.y, qb.y), 0.f);
        y.z = fmaxf(fmaf(y.z, qa.z, qb.z), 0.f); y.w = fmaxf(fmaf(y.w, qa.w, qb.w), 0.f);
      } else {
        y.x *= sBm; y.y *= sBm; y.z *= sBm; y.w *= sBm;
      }
      uint2 h, l;
      char* q = st + (r8 + 8 * j) * TR_ROW + c4 * 2;
      split2h(a.x, a.y, h.x, l.x);
      split2h(a.z, a.w, h.y, l.y);
      *reinterpret_cast<uint2*>(q) = h;
      *reinterpret_cast<uint2*>(q + TR_PLANE) = l;
      split2h(y.x, y.y, h.x, l.x);
      split2h(y.z, y.w, h.y, l.y);
      *reinterpret_cast<uint2*>(q + TR_OPER) = h;
      *reinterpret_cast<uint2*>(q + TR_OPER + TR_PLANE) = l;
    }
  };
  auto compute = [&](const char* st, auto i0c, auto i1c) {
    constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
    f16x8 wh[2], wl[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      wh[j] = tr_fragment(st + TR_OPER, wn + j * 32, lane);
      wl[j] = tr_fragment(st + TR_OPER + TR_PLANE, wn + j * 32, lane);
    }
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      const f16x8 ah = tr_fragment(st, wm + i * 32, lane);
      const f16x8 al = tr_fragment(st + TR_PLANE, wm + i * 32, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wh[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wl[j], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wh[j], acc[i][j], 0, 0, 0);
      }
    }
  };
  if (p.skew > 0 && ((tile_m * p.tiles_n + tile_n) & 1))      // diagnostic: knock the tiles out of step
    for (int i = 0; i < p.skew; ++i) __builtin_amdgcn_s_sleep(64);
  if (KT > 0) {
    load_tile(0, va[0], vb[0]);
    load_tile(1, va[1], vb[1]);
    store_tile(smem, va[0], vb[0]);
  }
  __syncthreads();
  auto iter = [&](int kt, auto cs_) {
    constexpr int CS = decltype(cs_)::value;
    char* cur = smem + (kt & 1) * TR_STAGE;
    char* nxt = smem + ((kt + 1) & 1) * TR_STAGE;
    load_tile(kt + 2, va[CS ^ 1], vb[CS ^ 1]);
    __builtin_amdgcn_sched_barrier(0);
    compute(cur, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    compute(cur, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
    store_tile(nxt, va[CS], vb[CS]);
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    }
    __syncthreads();
  };
  const int pace_tiles = p.tiles_m * p.tiles_n;
  bool pacing = p.pace != nullptr && tid == 0;
  int kt = 0;
  for (; kt + 1 < KT; kt += 2) {
    if ((kt & (TR_PACE - 1)) == 0 && pacing) {
      const int mine = (kt / TR_PACE + 1) * pace_tiles;
      int tot = __hip_atomic_fetch_add(p.pace + split, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
      int spins = 0;
      while (mine - tot > pace_tiles && spins < TR_PACE_SPINS) {
        __builtin_amdgcn_s_sleep(16);
        tot = __hip_atomic_load(p.pace + split, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++spins;
      }
      if (spins == TR_PACE_SPINS) pacing = false;
    }
    iter(kt, std::integral_constant<int, 1>{});
    iter(kt + 1, std::integral_constant<int, 0>{});
  }
  if (kt < KT) iter(kt, std::integral_constant<int, 1>{});

  const float unscale = 1.f / (sA * sB);
  const int half = lane >> 5, l31 = lane & 31;
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
  if (want_cs) {       // column sums of A: 8 row groups per column quad
    float4* red = reinterpret_cast<float4*>(smem);      // all LDS reads are behind the last barrier
    red[tid] = cs;
    __syncthreads();
    if (tid < 64) {
      float4 s = red[tid];
#pragma unroll
      for (int g = 1; g < 8; ++g) {
        const float4 t = red[tid + 64 * g];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      const float inv = 1.f / sA;
      float* o = p.colsum + (size_t)split * p.Mo + m0 + c4;
      if (m0 + c4 + 3 < p.Mo) { o[0] = s.x * inv; o[1] = s.y * inv; o[2] = s.z * inv; o[3] = s.w * inv; }
    }
  }
}

}  // namespace prh
