// bf16 mode (PRH_GEMM=bf16, mode 4; BASELINE config 3 "bf16 training"): ONE bf16 plane per
// operand, one MFMA product per MAC, and the encoder's activations (z_cat, z_fus, gate, fused)
// and their gradients (dy_cat, dy_f) STORED in bf16.  bf16 has fp32's exponent, so there are no
// operand scales and none of the operand-maximum passes of the split-fp16 mode.  Accumulation,
// BatchNorm statistics, parameters, their gradients and the optimiser stay fp32.
//
//   gemm_nt_b16_kernel  C[M,N] = pro(A)[M,K] W[N,K]^T.  256 x 256 x 64 tile, 512 threads (8
//       waves as 2 x 4, wave tile 128 x 64 = 8 x 4 blocks of v_mfma_f32_16x16x32_bf16), LDS
//       2 stages x (A 32 KB + W 32 KB).  The LDS image of one operand is the split-fp16 core's
//       (prh_gemm_h2.hpp) with the two "planes" now holding k 0..31 and k 32..63 of the 64-deep
//       k-tile: 64-B rows, 16-B chunks permuted by chunk ^ (-(row >> 2) & 3), conflict-free
//       ds_read_b128 fragment reads; a k-tile is a full 128-B line of every activation row.
//       Weights: bf16 image prepared per launch (prep_weights_b16_kernel), LDS-DMA.  A: bf16
//       (16 B = 8 k-values per lane) or fp32 (A16 = false: decoder Linears, converted in
//       flight), prologues NONE / BN+ReLU / GATE1 in fp32 registers.  Epilogue: the shared
//       vector epilogue with 16-bit stores (C16) - statistics are taken from the ROUNDED
//       values, i.e. of the tensor the next kernel reads.
//   gemm_tn_b16_kernel  C[Mo,Ni] = sum_p A[p,Mo] proB(B)[p,Ni] (wgrad), both operands bf16
//       row-major; 32-row k-tiles staged as they lie in memory and gathered by
//       ds_read_b64_tr_b16 (the transposed-read core of prh_gemm_h2.hpp with one plane);
//       proB: NONE / BN+ReLU / GATE1 (B generated from one fp32 scalar per row).
#pragma once
#include "prh_gemm_h2.hpp"

namespace prh {

constexpr int B16_BK = 64;
inline size_t b16_weight_bytes(int N, int K) {
  return S3_WHDR + (size_t)((N + 255) / 256) * ((K + B16_BK - 1) / B16_BK) * H2_OPER;
}

// W[n][k] fp32 -> bf16 tiled image out[(n_tile*KT + k_tile) * 32 KB + half * 16 KB + h2_off(row, k & 31)],
// half = (k >> 5) & 1, zero padded.  One thread per (row, 8-k chunk).
__global__ __launch_bounds__(256) void prep_weights_b16_kernel(const float* __restrict__ W, int N, int K,
                                                               long ldw, char* __restrict__ out) {
  const int KT = (K + B16_BK - 1) / B16_BK;
  const int NT_ = (N + 255) / 256;
  const long total = (long)NT_ * 256 * KT * 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int chunk = (int)(i & 7);
  long t = i >> 3;
  const int k_tile = (int)(t % KT); t /= KT;
  const int row = (int)(t & 255);
  const int n_tile = (int)(t >> 8);
  const int n = n_tile * 256 + row;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k_tile * B16_BK + chunk * 8 + j;
    v[j] = (n < N && k < K) ? W[(size_t)n * ldw + k] : 0.f;
  }
  const uint4 h = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                             pack_bf16x2(v[6], v[7]));
  char* base = out + ((size_t)n_tile * KT + k_tile) * H2_OPER + (chunk >> 2) * H2_PLANE + h2_off(row, (chunk & 3) * 8);
  *reinterpret_cast<uint4*>(base) = h;
}

// fp32 [rows, cs] (ld lds_) -> bf16 [rows, cd] (ld ldd), columns >= cs zero: operand casts
// (context rows padded to 8 channels, gradients arriving in fp32)
__global__ __launch_bounds__(256) void cast_b16_kernel(const float* __restrict__ src, long lds_, int cs,
                                                       unsigned short* __restrict__ dst, long ldd, int cd, size_t rows) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per 2 output columns
  const int half = cd >> 1;
  if (i >= rows * (size_t)half) return;
  const size_t r = i / half;
  const int c = (int)(i - r * half) * 2;
  const float a = c < cs ? src[r * lds_ + c] : 0.f, b = c + 1 < cs ? src[r * lds_ + c + 1] : 0.f;
  *reinterpret_cast<unsigned*>(dst + r * ldd + c) = pack_bf16x2(a, b);
}

// ---------------------------------------------------------------------------------------
// First layer of the bf16 mode on fp32 OPERANDS (src/model.py:10,15,43: conv1 + bn1): the
// context rows (x, y, z in metres, raw intensity) are never cast to bf16.  K = in_channel <= 8 is
// VALU work, not a GEMM: one workgroup = one 128-row statistics tile, a thread = one row x CO/2
// output channels, z = b + W x in fp32 (fmaf chain in channel order), ROUNDED to bf16 as stored,
// and the tile's per-column sum / centred M2 taken of the rounded values (the tensor the next
// kernels read back), in the ws_a / ws_b [tile][CO] format of the GEMM statistics epilogue.
// ---------------------------------------------------------------------------------------
template <int CO>
__global__ __launch_bounds__(256) void conv_in_b16_kernel(const float* __restrict__ x, int C, const float* __restrict__ W,
                                                          const float* __restrict__ b, unsigned short* __restrict__ z,
                                                          long ldz, int P, int want_stats, float* __restrict__ ws_a,
                                                          float* __restrict__ ws_b) {
  static_assert(CO % 16 == 0 && CO <= 128, "conv_in_b16: output width");
  constexpr int HC = CO / 2;
  __shared__ float sw[CO * 8 + CO];
  __shared__ float sv[128][CO + 1];
  __shared__ float red[4][CO];
  const int tid = threadIdx.x;
  for (int i = tid; i < CO * 8; i += 256) {
    const int co = i >> 3, c = i & 7;
    sw[i] = c < C ? W[(size_t)co * C + c] : 0.f;
  }
  for (int i = tid; i < CO; i += 256) sw[CO * 8 + i] = b != nullptr ? b[i] : 0.f;
  __syncthreads();
  const int lr = tid >> 1, half = tid & 1;
  const long row = (long)blockIdx.x * 128 + lr;
  const bool rok = row < P;
  float xv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) xv[c] = (rok && c < C) ? x[(size_t)row * C + c] : 0.f;
  float out[HC];
#pragma unroll
  for (int j = 0; j < HC; ++j) {
    const int co = half * HC + j;
    float a = sw[CO * 8 + co];
#pragma unroll
    for (int c = 0; c < 8; ++c) a = fmaf(sw[co * 8 + c], xv[c], a);
    out[j] = bf16_round(a);
  }
  if (rok) {
    unsigned short* q = z + (size_t)row * ldz + half * HC;
#pragma unroll
    for (int j = 0; j < HC; j += 8)
      *reinterpret_cast<uint4*>(q + j) = make_uint4(pack_bf16x2(out[j], out[j + 1]), pack_bf16x2(out[j + 2], out[j + 3]),
                                                    pack_bf16x2(out[j + 4], out[j + 5]), pack_bf16x2(out[j + 6], out[j + 7]));
  }
  if (!want_stats) return;
#pragma unroll
  for (int j = 0; j < HC; ++j) sv[lr][half * HC + j] = out[j];
  __syncthreads();
  int nrows = P - (int)((long)blockIdx.x * 128);
  nrows = nrows > 128 ? 128 : nrows;
  // 256 threads = CO columns x (256 / CO) row groups
  constexpr int G = 256 / CO, RG = 128 / G;
  const int col = tid % CO, g = tid / CO;
  float s = 0.f;
  for (int r = g * RG; r < (g + 1) * RG; ++r)
    if (r < nrows) s += sv[r][col];
  red[g][col] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < G; ++k) tot += red[k][col];
  const float mean = tot / (float)nrows;
  float m2 = 0.f;
  for (int r = g * RG; r < (g + 1) * RG; ++r)
    if (r < nrows) { const float d = sv[r][col] - mean; m2 = fmaf(d, d, m2); }
  __syncthreads();
  red[g][col] = m2;
  __syncthreads();
  if (g == 0) {
    float m2t = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) m2t += red[k][col];
    ws_a[(size_t)blockIdx.x * CO + col] = tot;
    ws_b[(size_t)blockIdx.x * CO + col] = m2t;
  }
}

// Its weight gradient, dW[co][c] = sum_p dz[p][co] x[p][c], with the fp32 context rows as the second
// operand: a thread = one output channel x one of four row lanes, C <= 8 running sums in fp32,
// per-workgroup partials [block][CO][C] summed in fp64 by slab_reduce_kernel.
template <int CO>
__global__ __launch_bounds__(256) void conv_in_wgrad_b16_kernel(const unsigned short* __restrict__ dz, long lddz,
                                                                const float* __restrict__ x, int C, int P,
                                                                int rows_per_block, float* __restrict__ part) {
  static_assert(CO == 64, "conv_in_wgrad_b16: one wave per row lane");
  __shared__ float red[4][CO][8];
  const int tid = threadIdx.x, co = tid & 63;
  const int rg = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  r1 = r1 > P ? P : r1;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // eight rows per trip, their loads issued together (a 2-byte load per lane and row: one row per trip is
  // latency-bound - 0.96 ms for 0.5 GB at B=4096, profiles/r03m_*bf16_prh_by_grid.csv)
  long r = r0 + rg;
  for (; r + 28 < r1; r += 32) {
    float d[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) d[u] = __uint_as_float((unsigned)dz[(size_t)(r + 4 * u) * lddz + co] << 16);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* xr = x + (size_t)(r + 4 * u) * C;       // wave-uniform row: broadcast loads
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < C) acc[c] = fmaf(d[u], xr[c], acc[c]);
    }
  }
  for (; r < r1; r += 4) {
    const float d = __uint_as_float((unsigned)dz[(size_t)r * lddz + co] << 16);
    const float* xr = x + (size_t)r * C;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < C) acc[c] = fmaf(d, xr[c], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) red[rg][co][c] = acc[c];
  __syncthreads();
  for (int i = tid; i < CO * C; i += 256) {
    const int o = i / C, c = i - o * C;
    part[(size_t)blockIdx.x * CO * C + i] = red[0][o][c] + red[1][o][c] + red[2][o][c] + red[3][o][c];
  }
}

__device__ __forceinline__ void unpack8(const uint4 u, float (&f)[8]) {
  f[0] = bf16_lo(u.x); f[1] = bf16_hi(u.x); f[2] = bf16_lo(u.y); f[3] = bf16_hi(u.y);
  f[4] = bf16_lo(u.z); f[5] = bf16_hi(u.z); f[6] = bf16_lo(u.w); f[7] = bf16_hi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

// MFMAs of one 64-deep k-tile for a 128 x 64 wave tile: per 32-deep half the W fragments of the
// wave's 4 column blocks stay in registers and the 8 A fragments stream through
__device__ __forceinline__ void b16_compute(f32x4 (&acc)[8][4], const char* st, int wm, int wn, int l15, int kc) {
  const char* wb = st + H2_OPER + h2_off(wn + l15, kc * 8);     // + j * 1024 + hf * H2_PLANE
  const char* ab = st + h2_off(wm + l15, kc * 8);               // + i * 1024 + hf * H2_PLANE
  // eight steps (k-half x group of 2 row blocks), software-pipelined one step deep: the fragment
  // reads of step s+1 are issued in front of the 8 MFMAs of step s and pinned there - hoisted
  // further, the fragments of a whole k-tile (96 registers) push the accumulator tile into scratch
  bf16x8 w[2][4], a[2][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) w[0][j] = *reinterpret_cast<const bf16x8*>(wb + j * 1024);
#pragma unroll
  for (int i = 0; i < 2; ++i) a[0][i] = *reinterpret_cast<const bf16x8*>(ab + i * 1024);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int hf = s >> 2, g = s & 3;
    if (s < 7) {
      const int hn = (s + 1) >> 2, gn = (s + 1) & 3;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(ab + hn * H2_PLANE + (gn * 2 + i) * 1024);
      if (s == 2) {
#pragma unroll
        for (int j = 0; j < 4; ++j) w[1][j] = *reinterpret_cast<const bf16x8*>(wb + H2_PLANE + j * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[g * 2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][i], w[hf][j], acc[g * 2 + i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int PRO, int EPI, bool A16, bool C16>
__global__ __launch_bounds__(512, 2) void gemm_nt_b16_kernel(const NTParams p, const char* __restrict__ Wp) {
  static_assert(PRO == PRO_NONE || PRO == PRO_BNRELU || PRO == PRO_GATE1, "prologue not supported");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kc = lane >> 4;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = vb % p.tiles_n, tile_m = vb / p.tiles_n;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int KT = (p.K + B16_BK - 1) / B16_BK;

  const int sc = tid & 7;            // staging k-chunk: 8 values, 8 lanes cover the 64-deep row
  const int sr = tid >> 3;           // staging row 0..63 (+64 j)
  const int KP = (KT + 2) * B16_BK;  // padded length of the coefficient vectors in LDS
  float* coef = reinterpret_cast<float*>(smem + H2_LDS);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // staging registers: A16: one uint4 (8 bf16) per row pass, A32: two float4
  constexpr int RW = A16 ? 1 : 2;
  uint4 ra[A16 ? 2 : 1][4][RW];
  const char* wsrc = Wp + (size_t)tile_n * KT * H2_OPER + tid * 16;
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
  const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + H2_OPER + wave * 1024);
  auto dma_w = [&](int kt, int stage) {
    const char* q = wsrc + (size_t)(kt < KT ? kt : KT - 1) * H2_OPER;   // tail: harmless re-copy
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) glds16(q + pc * 8192, wdst + stage * H2_STAGE + pc * 8192);
  };
  constexpr int ES = A16 ? 2 : 4;
  constexpr int NA = PRO == PRO_GATE1 ? 0 : 4 * RW;   // A loads issued behind the DMA in one k-tile
  int arows = p.M - m0; arows = arows > 256 ? 256 : arows;
  const char* Abase = reinterpret_cast<const char*>(p.A) + (size_t)m0 * p.lda * ES;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)Abase, 0, PRO == PRO_GATE1 ? 0 : (int)(((size_t)(arows - 1) * p.lda + p.K) * ES), 0x00020000);
  int voA[4];
  float gi[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    voA[j] = ((sr + 64 * j) * (int)p.lda + sc * 8) * ES;
    if (PRO == PRO_GATE1) {
      const int row = m0 + sr + 64 * j;
      gi[j] = row < p.M ? p.A[(size_t)row * p.lda] : 0.f;      // GATE1: A is fp32, one scalar per row
    }
  }
  auto load_tile = [&](int kt, uint4 (&r)[4][RW]) {
    if (PRO == PRO_GATE1) return;
    const int so = kt * (B16_BK * ES);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int w = 0; w < RW; ++w)
        r[j][w] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voA[j] + so + w * 16, 0, 0));
  };
  auto store_tile = [&](int kt, char* st, const uint4 (&r_)[4][RW]) {
    const int k = kt * B16_BK + sc * 8;
    const bool kok = k < p.K;          // K is a multiple of 8: a chunk is inside or outside as a whole
    float ka[8], kb[8];
    if (PRO != PRO_NONE) {
      *reinterpret_cast<float4*>(ka) = *reinterpret_cast<const float4*>(coef + k);
      *reinterpret_cast<float4*>(ka + 4) = *reinterpret_cast<const float4*>(coef + k + 4);
      *reinterpret_cast<float4*>(kb) = *reinterpret_cast<const float4*>(coef + KP + k);
      *reinterpret_cast<float4*>(kb + 4) = *reinterpret_cast<const float4*>(coef + KP + k + 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = sr + 64 * j;
      uint4 o;
      if (PRO == PRO_NONE && A16) {
        o = r_[j][0];                  // already bf16: no arithmetic at all
      } else {
        float f[8];
        if (PRO == PRO_GATE1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = gi[j];
        } else if (A16) {
          unpack8(r_[j][0], f);
        } else {
          *reinterpret_cast<uint4*>(f) = r_[j][0];
          *reinterpret_cast<uint4*>(f + 4) = r_[j][RW - 1];
        }
        if (PRO != PRO_NONE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], ka[e], kb[e]), 0.f);
        }
        o = pack8(f);
      }
      if (!kok) o = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(st + (sc >> 2) * H2_PLANE + h2_off(r, (sc & 3) * 8)) = o;
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
#pragma unroll
        for (int w = 0; w < RW; ++w)
          asm volatile("" : "+v"(ra[CS][j][w].x), "+v"(ra[CS][j][w].y), "+v"(ra[CS][j][w].z), "+v"(ra[CS][j][w].w));
      __builtin_amdgcn_sched_barrier(0);
    }
    dma_w(kt + 1, (kt + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);   // DMA strictly before the A loads (vmcnt is in order)
    if constexpr (!A16) {
      // fp32 A (8 registers per row pass): ONE register set.  Tile kt+1 is converted into the free
      // stage first, then its registers take tile kt+2, which has the whole k-tile of MFMAs to land
      // (two sets of 32 registers next to the accumulator tile spilled 99 VGPRs into the k-loop)
      store_tile(kt + 1, nxt, ra[0]);
      __builtin_amdgcn_sched_barrier(0);
      if (!TAIL) load_tile(kt + 2, ra[0]);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      if (!TAIL) load_tile(kt + 2, ra[CS ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    b16_compute(acc, cur, wm, wn, l15, kc);
    if constexpr (A16) {
      store_tile(kt + 1, nxt, ra[CS]);
      if (PRO != PRO_NONE) {
        // interleave the conversion with the 64 MFMAs
#pragma unroll
        for (int g = 0; g < 32; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
      }
    }
    if (TAIL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NA) : "memory");
    __syncthreads();
  };

  dma_w(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  load_tile(0, ra[0]);
  if constexpr (A16) load_tile(1, ra[1]);
  if (PRO != PRO_NONE) {
    for (int i = tid; i < KP; i += 512) {
      const bool in = i < p.K;
      coef[i] = in ? p.pa[i] : 0.f;
      coef[KP + i] = in ? p.pb[i] : 0.f;
    }
    __syncthreads();
  }
  store_tile(0, smem, ra[0]);
  if constexpr (!A16) load_tile(1, ra[0]);
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
  bool bias_done = false;
  if (C16 && (EPI == EPI_BIAS || EPI == EPI_BIAS_STATS)) {
    // bias first, then round: the statistics (and the F_RELU_OUT / residual arithmetic) see the
    // values that will be read back from the bf16 buffer
    const bool plain = EPI == EPI_BIAS_STATS || (p.flags & (F_RESID | F_RELU_OUT)) == 0;
    if (plain) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn + j * 16 + l15;
        const float b = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = bf16_round(acc[i][j][r] + b);
      }
      bias_done = true;
    }
  }
  nt_epilogue_vec<EPI, 4, f32x4[8][4], C16>(acc, p, m0 + wm, n0 + wn, tile_m * 2 + (wave >> 2), lane,
                                            reinterpret_cast<float*>(smem) + wave * (32 * EPI_LDW), bias_done);
}

// ---------------------------------------------------------------------------------------
// [r03] gemm_nt_b16d_kernel: the same GEMM for a PLAIN bf16 A operand (dgrads, plain Linears of the bf16 mode),
// BOTH operands by LDS-DMA and a phase-split k-loop (`cdna_hip_programming.md`, "The 256^2 8-phase template"):
//   * no staging registers, no VALU and no ds_write in the loop: a lane's 16 B of A go from the activation
//     tensor straight into the swizzled LDS image (the permutation of h2_off is applied to the SOURCE address:
//     the LDS side of a DMA is lane-linear); every vector-memory instruction of the loop is an LDS-DMA, so ONE
//     counted s_waitcnt per k-tile is exact;
//   * the W fragments of a k-tile (4 column blocks x 2 k-halves) are read in phase 0 and stay in registers, so
//     the W image of tile kt+2 is DMA-ed over tile kt's as soon as both wave rows have passed their phase 0;
//     A lives in a ring of three 32-KB images: both operands are in flight two k-tiles ahead
//     (LDS: 2 x 32 KB W + 3 x 32 KB A = 160 KB, the epilogue scratch aliases it);
//   * the two wave rows (waves 0-3, 4-7: one of each per SIMD) run one barrier apart: while one issues the 16
//     MFMAs of two row blocks the other reads its next fragments and issues its share of the DMA.
// Needs K % 64 == 0 (the image of a partial k-tile could not be zero-filled by a DMA), lda % 8 == 0.
// (Tried on top of this, no change - fusion dgrad 19.8-20.1 vs 19.9-20.0 ms on one box: the W fragments of k-half 1
// read one k-tile ahead in phase 3, so that no memory phase holds more than 8 fragment reads.)
// ---------------------------------------------------------------------------------------
constexpr int B16D_W = 0;                       // two W stages
constexpr int B16D_A = 2 * H2_OPER;             // ring of three A images
constexpr int B16D_LDS = 2 * H2_OPER + 3 * H2_OPER;     // 163840

// (PRH_TICK: diagnostic build only, prh_gemm.hpp)

template <int EPI, bool C16>
__global__ __launch_bounds__(512, 2) void gemm_nt_b16d_kernel(const NTParams p, const char* __restrict__ Wp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef PRH_STAMP
  unsigned t_acc[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) t_acc[i] = 0u;
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kc = lane >> 4;
  const int wm = (wave >> 2) * 128, wn = (wave & 3) * 64;
  const int wr = __builtin_amdgcn_readfirstlane(wave >> 2);
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = vb % p.tiles_n, tile_m = vb / p.tiles_n;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int KT = p.K / B16_BK;
  if ((p.flags & F_STAGGER) != 0 && blockIdx.x < 256 && (blockIdx.x & 1) != 0) {
    const int n = __builtin_amdgcn_readfirstlane((KT * 5) >> 3);      // x 4096 cycles: about half a tile
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(64);
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);
  const unsigned wslice = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);     // + image base + piece * 8192
  // W: the prepared image is the LDS image, piece pc = bytes [pc * 8 KB, +8 KB) of the 32-KB tile
  const char* wsrc = Wp + (size_t)tile_n * KT * H2_OPER + tid * 16;
  // A: piece pc = half (pc >> 1), rows (pc & 1) * 128 + wave * 16 + (lane >> 2); the lane's stored chunk
  // position lane & 3 holds logical chunk (lane & 3) ^ (-(row >> 2) & 3)
  const char* asrc[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = h * 128 + wave * 16 + (lane >> 2);
    int gr = m0 + row; gr = gr < p.M ? gr : p.M - 1;
    const int ch = (lane & 3) ^ (-(row >> 2) & 3);
    asrc[h] = reinterpret_cast<const char*>(p.A) + ((size_t)gr * p.lda + ch * 8) * 2;
  }
  auto dma_a = [&](int kt_, int slot, int pc) {      // pc 0..3
    const int kq = kt_ < KT ? kt_ : KT - 1;          // beyond K: harmless re-copy
    glds16(asrc[pc & 1] + (size_t)kq * (B16_BK * 2) + (pc >> 1) * 64, wslice + B16D_A + slot * H2_OPER + pc * 8192);
  };
  auto dma_wp = [&](int kt_, int stage, int pc) {
    const int kq = kt_ < KT ? kt_ : KT - 1;
    glds16(wsrc + (size_t)kq * H2_OPER + pc * 8192, wslice + B16D_W + stage * H2_OPER + pc * 8192);
  };
  const int aoff = h2_off(wm + l15, kc * 8);          // + i * 1024 + hf * H2_PLANE
  const int woff = h2_off(wn + l15, kc * 8);          // + j * 1024 + hf * H2_PLANE
  bf16x8 w[2][4], a[2][2];

  auto phase = [&](int kt_, int aslot, int anext, auto mp, auto tail) {
    constexpr int MP = decltype(mp)::value;
    constexpr bool TAIL = decltype(tail)::value;
    const char* ws = smem + B16D_W + (kt_ & 1) * H2_OPER;
    const char* as = smem + B16D_A + aslot * H2_OPER;
    __builtin_amdgcn_sched_barrier(0);
    if (MP == 0) {
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[hf][j] = *reinterpret_cast<const bf16x8*>(ws + woff + j * 1024 + hf * H2_PLANE);
    }
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int i = 0; i < 2; ++i) a[hf][i] = *reinterpret_cast<const bf16x8*>(as + aoff + (2 * MP + i) * 1024 + hf * H2_PLANE);
    __builtin_amdgcn_sched_barrier(0);
    if (!TAIL) {
      dma_a(kt_ + 2, anext, MP);                     // A piece MP of tile kt+2 -> the ring slot tile kt-1 left
      if (MP == 1) { dma_wp(kt_ + 2, kt_ & 1, 0); dma_wp(kt_ + 2, kt_ & 1, 1); }    // W of tile kt+2 over tile kt's:
      if (MP == 2) dma_wp(kt_ + 2, kt_ & 1, 2);                                      // both rows are past phase 0
      if (MP == 3) dma_wp(kt_ + 2, kt_ & 1, 3);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MP == 3) {      // tile kt+1 (8 pieces issued one k-tile ago, this tile's 8 behind them) has landed
      if (TAIL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    PRH_TICK(MP * 4 + 0)            // memory phase (reads, DMA issue, waits)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PRH_TICK(MP * 4 + 1)            // barrier after the memory phase
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[2 * MP + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[hf][i], w[hf][j], acc[2 * MP + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    PRH_TICK(MP * 4 + 2)            // 16 MFMAs issued
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PRH_TICK(MP * 4 + 3)            // barrier after the MFMA phase
  };
  auto tile = [&](int kt_, int aslot, int anext, auto tail) {
    phase(kt_, aslot, anext, std::integral_constant<int, 0>{}, tail);
    phase(kt_, aslot, anext, std::integral_constant<int, 1>{}, tail);
    phase(kt_, aslot, anext, std::integral_constant<int, 2>{}, tail);
    phase(kt_, aslot, anext, std::integral_constant<int, 3>{}, tail);
  };

  // prologue: tiles 0 and 1 (16 pieces); tile 0 has landed when at most 8 are outstanding
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) { dma_a(0, 0, pc); dma_wp(0, 0, pc); }
#pragma unroll
  for (int pc = 0; pc < 4; ++pc) { dma_a(1, 1, pc); dma_wp(1, 1, pc); }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __syncthreads();
  PRH_TICK(16)                                       // prologue
  if (wr == 1) __builtin_amdgcn_s_barrier();         // the second wave row runs one barrier behind the first
  int kt = 0, aslot = 0;
  for (; kt + 2 < KT; ++kt) {
    const int anext = aslot == 0 ? 2 : aslot - 1;    // (kt + 2) % 3 == (kt - 1) % 3
    tile(kt, aslot, anext, std::false_type{});
    aslot = aslot == 2 ? 0 : aslot + 1;
  }
  for (; kt < KT; ++kt) {
    tile(kt, aslot, 0, std::true_type{});
    aslot = aslot == 2 ? 0 : aslot + 1;
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  PRH_TICK(17)                                       // loop exit
  bool bias_done = false;
  if (C16 && (EPI == EPI_BIAS || EPI == EPI_BIAS_STATS)) {
    const bool plain = EPI == EPI_BIAS_STATS || (p.flags & (F_RESID | F_RELU_OUT)) == 0;
    if (plain) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn + j * 16 + l15;
        const float b = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = bf16_round(acc[i][j][r] + b);
      }
      bias_done = true;
    }
  }
  nt_epilogue_vec<EPI, 4, f32x4[8][4], C16>(acc, p, m0 + wm, n0 + wn, tile_m * 2 + (wave >> 2), lane,
                                            reinterpret_cast<float*>(smem) + wave * (32 * EPI_LDW), bias_done);
#ifdef PRH_STAMP
  PRH_TICK(18)                                       // epilogue: instructions issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PRH_TICK(19)                                       // its stores acknowledged
  if (g_prh_stamp != nullptr && p.K == 1024 && p.N == 1984 && blockIdx.x == gridDim.x / 2 + 3 && lane == 0 && (wave & 3) == 0) {
    unsigned* o = g_prh_stamp + (wave >> 2) * 32;
#pragma unroll
    for (int i = 0; i < 24; ++i) o[i] = t_acc[i];
    o[24] = (unsigned)KT;
  }
#endif
}

// ---------------------------------------------------------------------------------------
// wgrad, bf16 operands: C[Mo,Ni] = sum_p A[p,Mo] * proB(B)[p,Ni] over this split's rows.
// k-tile = 32 rows; a row of the 256-column tile is 512 B = 32 lanes x 16 B, so a 512-thread
// workgroup stages one operand's k-tile with two 16-B loads per thread (rows r and r + 16).
// LDS image [k][column] bf16, rows padded to 576 B (see gemm_tn_tr_kernel), fragments by
// ds_read_b64_tr_b16; v_mfma_f32_32x32x16_bf16, two k-steps per k-tile.
// ---------------------------------------------------------------------------------------
constexpr int TB_BK = 32;
constexpr int TB_OPER = TB_BK * TR_ROW;          // 18432
constexpr int TB_STAGE = 2 * TB_OPER;            // A, B
constexpr int TB_LDS = 2 * TB_STAGE;             // 73728

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 tb_fragment(const char* plane, int col0, int lane) {
  const int g = lane >> 4, j = lane & 15;
  const char* a = plane + ((g >> 1) * 8 + (j >> 2)) * TR_ROW + (col0 + (g & 1) * 16 + 4 * (j & 3)) * 2;
  const fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(a));
  const fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(a + 4 * TR_ROW));
  struct Pair { fp16x4 a, b; } pr = {lo, hi};      // k 0..3 | k 4..7 of this lane's half (16-bit payload, type-blind)
  return __builtin_bit_cast(bf16x8, pr);
}

template <int PROB>
__global__ __launch_bounds__(512, 2) void gemm_tn_b16_kernel(const TNParams p) {
  static_assert(PROB == PRO_NONE || PROB == PRO_BNRELU || PROB == PRO_GATE1, "prologue not supported");
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
  const int nrows = p_end - p_begin > 0 ? p_end - p_begin : 0;
  const int KT = (nrows + TB_BK - 1) / TB_BK;

  const int c8 = (tid & 31) * 8, r16 = tid >> 5;        // staging: 8 columns, rows r16 and r16 + 16
  const bool aok = (m0 + c8) < p.Mo, bok = (n0 + c8) < p.Ni;
  float qa[8], qb[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { qa[e] = 0.f; qb[e] = 0.f; }
  if (PROB != PRO_NONE && bok) {
    *reinterpret_cast<float4*>(qa) = ldg4(p.qa + n0 + c8); *reinterpret_cast<float4*>(qa + 4) = ldg4(p.qa + n0 + c8 + 4);
    *reinterpret_cast<float4*>(qb) = ldg4(p.qb + n0 + c8); *reinterpret_cast<float4*>(qb + 4) = ldg4(p.qb + n0 + c8 + 4);
  }
  const bool want_cs = p.colsum != nullptr && tile_n == 0;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int acols = p.Mo - m0; acols = acols > 256 ? 256 : acols;
  int bcols = p.Ni - n0; bcols = bcols > 256 ? 256 : bcols;
  const size_t abytes = nrows > 0 ? ((size_t)(nrows - 1) * p.lda + acols) * 2 : 0;
  const char* Ab = reinterpret_cast<const char*>(p.A) + ((size_t)p_begin * p.lda + m0) * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, (int)abytes, 0x00020000);
  // B: bf16 matrix, or (GATE1) one fp32 scalar per row at p.B[row * ldb]
  const size_t bbytes = nrows > 0 ? (PROB == PRO_GATE1 ? ((size_t)(nrows - 1) * p.ldb + 1) * 4
                                                       : ((size_t)(nrows - 1) * p.ldb + bcols) * 2) : 0;
  const char* Bb = reinterpret_cast<const char*>(p.B) +
                   (PROB == PRO_GATE1 ? (size_t)p_begin * p.ldb * 4 : ((size_t)p_begin * p.ldb + n0) * 2);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)Bb, 0, (int)bbytes, 0x00020000);
  const int stepA = (int)p.lda * 2, stepB = (int)p.ldb * (PROB == PRO_GATE1 ? 4 : 2);
  const int voA = r16 * stepA + c8 * 2, voB = r16 * stepB + (PROB == PRO_GATE1 ? 0 : c8 * 2);

  uint4 va[2][2], vb[2][2];
  float cs[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) cs[e] = 0.f;
  auto load_tile = [&](int kt, uint4 (&xa)[2], uint4 (&xb)[2]) {
    const int oa = voA + kt * TB_BK * stepA, ob = voB + kt * TB_BK * stepB;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      xa[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, oa + j * 16 * stepA, 0, 0));
      if (PROB == PRO_GATE1) {
        xb[j].x = __builtin_bit_cast(unsigned, __builtin_amdgcn_raw_buffer_load_b32(rsB, ob + j * 16 * stepB, 0, 0));
      } else {
        xb[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsB, ob + j * 16 * stepB, 0, 0));
      }
    }
  };
  auto store_tile = [&](char* st, const uint4 (&xa)[2], const uint4 (&xb)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint4 a = xa[j];
      if (!aok) a = make_uint4(0u, 0u, 0u, 0u);
      if (want_cs) {
        float f[8];
        unpack8(a, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] += f[e];
      }
      uint4 y;
      if (PROB == PRO_NONE) {
        y = xb[j];
      } else {
        float f[8];
        if (PROB == PRO_GATE1) {
          const float s = __uint_as_float(xb[j].x);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = s;
        } else {
          unpack8(xb[j], f);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], qa[e], qb[e]), 0.f);
        y = pack8(f);
      }
      if (!bok) y = make_uint4(0u, 0u, 0u, 0u);
      char* q = st + (r16 + 16 * j) * TR_ROW + c8 * 2;
      *reinterpret_cast<uint4*>(q) = a;
      *reinterpret_cast<uint4*>(q + TB_OPER) = y;
    }
  };
  auto compute = [&](const char* st) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const char* sa = st + ks * 16 * TR_ROW;
      bf16x8 w[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) w[j] = tb_fragment(sa + TB_OPER, wn + j * 32, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8 a = tb_fragment(sa, wm + i * 32, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w[j], acc[i][j], 0, 0, 0);
      }
    }
  };
  if (KT > 0) {
    load_tile(0, va[0], vb[0]);
    load_tile(1, va[1], vb[1]);
    store_tile(smem, va[0], vb[0]);
  }
  __syncthreads();
  auto iter = [&](int kt, auto cs_) {
    constexpr int CS = decltype(cs_)::value;
    char* cur = smem + (kt & 1) * TB_STAGE;
    char* nxt = smem + ((kt + 1) & 1) * TB_STAGE;
    load_tile(kt + 2, va[CS ^ 1], vb[CS ^ 1]);
    __builtin_amdgcn_sched_barrier(0);
    compute(cur);
    store_tile(nxt, va[CS], vb[CS]);
    __syncthreads();
  };
  const int pace_tiles = p.tiles_m * p.tiles_n;
  bool pacing = p.pace != nullptr && tid == 0;
  constexpr int PACE = 4;            // k-tiles (128 rows) between progress reports, as in the fp16 core
  int kt = 0;
  for (; kt + 1 < KT; kt += 2) {
    if ((kt & (PACE - 1)) == 0 && pacing) {
      const int mine = (kt / PACE + 1) * pace_tiles;
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
        if (row < p.Mo && col < p.Ni) out[(size_t)row * p.Ni + col] = acc[mt][nt][r];
      }
    }
  if (want_cs) {       // column sums of A: 16 row groups per 8-column chunk
    float* red = reinterpret_cast<float*>(smem);          // all LDS reads are behind the last barrier
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = cs[e];
    __syncthreads();
    if (tid < 256) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) s += red[(g * 32 + (tid >> 3)) * 8 + (tid & 7)];
      if (m0 + tid < p.Mo) p.colsum[(size_t)split * p.Mo + m0 + tid] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------
// elementwise passes over bf16 buffers
// ---------------------------------------------------------------------------------------
// BatchNorm backward applied in place: dy <- bf16(ka*dy + kb*z + kc)  (6 B per element)
__global__ __launch_bounds__(256) void bn_bwd_apply_b16_kernel(unsigned short* __restrict__ dy, long lddy,
                                                               const unsigned short* __restrict__ z, long ldz,
                                                               const float* __restrict__ pa,
                                                               const float* __restrict__ pb,
                                                               const float* __restrict__ pc, long rows, int cols) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * 4 + ty, rstep = (long)gridDim.x * 4;
  const int c8n = cols >> 3;
  for (int cv = tx; cv < c8n; cv += 64) {
    const int c = cv * 8;
    float ka[8], kb[8], kc[8];
    *reinterpret_cast<float4*>(ka) = ldg4(pa + c); *reinterpret_cast<float4*>(ka + 4) = ldg4(pa + c + 4);
    *reinterpret_cast<float4*>(kb) = ldg4(pb + c); *reinterpret_cast<float4*>(kb + 4) = ldg4(pb + c + 4);
    *reinterpret_cast<float4*>(kc) = ldg4(pc + c); *reinterpret_cast<float4*>(kc + 4) = ldg4(pc + c + 4);
#pragma unroll 4
    for (long r = r0; r < rows; r += rstep) {
      uint4* q = reinterpret_cast<uint4*>(dy + r * lddy + c);
      float d[8], zz[8];
      unpack8(*q, d);
      unpack8(*reinterpret_cast<const uint4*>(z + r * ldz + c), zz);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = fmaf(ka[e], d[e], fmaf(kb[e], zz[e], kc[e]));
      *q = pack8(d);
    }
  }
}

// dual pooling over bf16 `fused` (pool_kernel of prh_kernels.hpp; first-index ties)
__global__ __launch_bounds__(256) void pool_b16_kernel(const unsigned short* __restrict__ F, int N, int C,
                                                       float* gfeat, int32_t* argmax) {
  __shared__ float smax[4][64];
  __shared__ int sidx[4][64];
  __shared__ float ssum[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c, b = blockIdx.y;
  float mx = -INFINITY, sum = 0.f;
  int ix = 0;
  if (col < C) {
    const unsigned short* base = F + (size_t)b * N * C + col;
    for (int n = g; n < N; n += 4) {
      const float v = __uint_as_float((unsigned)base[(size_t)n * C] << 16);
      sum += v;
      if (v > mx) { mx = v; ix = n; }
    }
  }
  smax[g][c] = mx; sidx[g][c] = ix; ssum[g][c] = sum;
  __syncthreads();
  if (g == 0 && col < C) {
    for (int j = 1; j < 4; ++j) {
      const float v = smax[j][c];
      const int i2 = sidx[j][c];
      if (v > mx || (v == mx && i2 < ix)) { mx = v; ix = i2; }
      sum += ssum[j][c];
    }
    gfeat[(size_t)b * 2 * C + col] = mx;
    gfeat[(size_t)b * 2 * C + C + col] = sum / (float)N;
    if (argmax != nullptr) argmax[(size_t)b * C + col] = ix;
  }
}

// combine_bwd_kernel of prh_kernels.hpp over bf16 buffers: dF (bf16 or null), zf, gate (-> dG in
// place), dy_out bf16; partial sums taken from the rounded dy
__global__ __launch_bounds__(256) void combine_bwd_b16_kernel(
    const unsigned short* dF, const float* __restrict__ d_gfeat, const int32_t* __restrict__ argmax,
    const unsigned short* __restrict__ zf, unsigned short* gate, const float* __restrict__ s,
    const float* __restrict__ t, int P, int N, int C, unsigned short* dy_out, float* ws_a, float* ws_b) {
  __shared__ float r1[4][64], r2[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + c;
  const int row0 = blockIdx.x * 64 + g * 16;
  float s1 = 0.f, s2 = 0.f;
  auto f = [](unsigned short h) { return __uint_as_float((unsigned)h << 16); };
  auto h16 = [](float x) { return (unsigned short)(pack_bf16x2(x, 0.f) & 0xffffu); };
  if (col < C) {
    const float sc = s[col], sh = t[col];
    for (int i = 0; i < 16; ++i) {
      const int row = row0 + i;
      if (row >= P) break;
      const size_t off = (size_t)row * C + col;
      float d = dF != nullptr ? f(dF[off]) : 0.f;
      if (d_gfeat != nullptr) {
        const int b = row / N, n = row - b * N;
        d += d_gfeat[(size_t)b * 2 * C + C + col] / (float)N;
        if (argmax[(size_t)b * C + col] == n) d += d_gfeat[(size_t)b * 2 * C + col];
      }
      const float z = f(zf[off]), m = f(gate[off]);
      const float pre = fmaf(z, sc, sh);
      const float r = fmaxf(pre, 0.f);
      const float sig = 2.f * m - 1.f;
      const unsigned short dyh = h16(pre > 0.f ? d * m : 0.f);
      gate[off] = h16(d * r * 0.5f * sig * (1.f - sig));
      dy_out[off] = dyh;
      const float dy = f(dyh);
      s1 += dy;
      s2 = fmaf(dy, z, s2);
    }
  }
  r1[g][c] = s1; r2[g][c] = s2;
  __syncthreads();
  if (g == 0 && col < C) {
    for (int j = 1; j < 4; ++j) { s1 += r1[j][c]; s2 += r2[j][c]; }
    ws_a[(size_t)blockIdx.x * C + col] = s1;
    ws_b[(size_t)blockIdx.x * C + col] = s2;
  }
}

// combine_bwd_b16_kernel with 16-byte accesses: a thread owns 8 consecutive channels of a row
// (C / 8 threads per row, 256 / (C / 8) rows in flight per block, 64 rows per block = one
// statistics tile).  C / 8 must be 32, 64, 128 or 256.
__global__ __launch_bounds__(256) void combine_bwd_b16v_kernel(
    const unsigned short* dF, const float* __restrict__ d_gfeat, const int32_t* __restrict__ argmax,
    const unsigned short* __restrict__ zf, unsigned short* gate, const float* __restrict__ s,
    const float* __restrict__ t, int P, int N, int C, unsigned short* dy_out, float* ws_a, float* ws_b) {
  __shared__ float red[2][256][8];
  const int ng = C >> 3;                       // channel groups per row
  const int cg = threadIdx.x % ng, rl = threadIdx.x / ng, nrl = 256 / ng;
  const int c0 = cg * 8;
  const int row0 = blockIdx.x * 64;
  float sc[8], sh[8], s1[8], s2[8];
  *reinterpret_cast<float4*>(sc) = ldg4(s + c0); *reinterpret_cast<float4*>(sc + 4) = ldg4(s + c0 + 4);
  *reinterpret_cast<float4*>(sh) = ldg4(t + c0); *reinterpret_cast<float4*>(sh + 4) = ldg4(t + c0 + 4);
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  for (int i = rl; i < 64; i += nrl) {
    const int row = row0 + i;
    if (row >= P) break;
    const size_t off = (size_t)row * C + c0;
    float d[8], z[8], m[8], dy[8], dg[8];
    if (dF != nullptr) unpack8(*reinterpret_cast<const uint4*>(dF + off), d);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) d[e] = 0.f;
    }
    if (d_gfeat != nullptr) {
      const int b = row / N, n = row - b * N;
      const float invn = 1.f / (float)N;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        d[e] += d_gfeat[(size_t)b * 2 * C + C + c0 + e] * invn;
        if (argmax[(size_t)b * C + c0 + e] == n) d[e] += d_gfeat[(size_t)b * 2 * C + c0 + e];
      }
    }
    unpack8(*reinterpret_cast<const uint4*>(zf + off), z);
    unpack8(*reinterpret_cast<const uint4*>(gate + off), m);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float pre = fmaf(z[e], sc[e], sh[e]);
      const float r = fmaxf(pre, 0.f);
      const float sig = 2.f * m[e] - 1.f;
      dy[e] = bf16_round(pre > 0.f ? d[e] * m[e] : 0.f);
      dg[e] = d[e] * r * 0.5f * sig * (1.f - sig);
      s1[e] += dy[e];
      s2[e] = fmaf(dy[e], z[e], s2[e]);
    }
    *reinterpret_cast<uint4*>(gate + off) = pack8(dg);
    *reinterpret_cast<uint4*>(dy_out + off) = pack8(dy);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
  __syncthreads();
  if (rl == 0) {
    for (int j = 1; j < nrl; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] += red[0][j * ng + cg][e]; s2[e] += red[1][j * ng + cg][e]; }
    float* a = ws_a + (size_t)blockIdx.x * C + c0;
    float* b = ws_b + (size_t)blockIdx.x * C + c0;
    *reinterpret_cast<float4*>(a) = make_float4(s1[0], s1[1], s1[2], s1[3]);
    *reinterpret_cast<float4*>(a + 4) = make_float4(s1[4], s1[5], s1[6], s1[7]);
    *reinterpret_cast<float4*>(b) = make_float4(s2[0], s2[1], s2[2], s2[3]);
    *reinterpret_cast<float4*>(b + 4) = make_float4(s2[4], s2[5], s2[6], s2[7]);
  }
}

}  // namespace prh
