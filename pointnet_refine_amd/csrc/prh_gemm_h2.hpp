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
template <int I0 = 0, int I1 = 8>
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
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[j], acc[i][j], 0, 0, 0);
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[j], acc[i][j], 0, 0, 0);
    }
  }
}

template <int PRO, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_h2_kernel(const NTParams p,
                                                            const char* __restrict__ Wp) {
  static_assert(PRO == PRO_NONE || PRO == PRO_BNRELU || PRO == PRO_GATE1, "prologue not supported");
  extern __shared__ __attribute__((aligned(16))) char smem[];
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
    h2_compute(acc, cur, wm, wn, l15, kc);
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
  int kt = 0;
  for (; kt + 2 < KT; kt += 2) {
    iter(kt, std::integral_constant<int, 1>{}, std::false_type{});
    iter(kt + 1, std::integral_constant<int, 0>{}, std::false_type{});
  }
  if (kt < KT) iter(kt, std::integral_constant<int, 1>{}, std::true_type{});
  if (kt + 1 < KT) iter(kt + 1, std::integral_constant<int, 0>{}, std::true_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] *= unscale;
  nt_epilogue_vec<EPI, 4>(acc, p, m0 + wm, n0 + wn, tile_m * 2 + (wave >> 2), lane,
                          reinterpret_cast<float*>(smem) + wave * (32 * EPI_LDW));
}

}  // namespace prh
