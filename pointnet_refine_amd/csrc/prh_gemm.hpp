// fp32 MFMA GEMM cores for the LineRefineNet shared-MLP path on gfx950 (MI355X).
//
// Two cores, both on v_mfma_f32_32x32x2_f32 (exact fp32 = an fmaf chain, so the
// 1e-4 parity gate against the CPU reference holds with ~1e-6 to spare):
//
//   gemm_nt : C[M,N] = pro(A)[M,K] * W[N,K]^T   rows = points, K-contiguous operands.
//             Every forward 1x1 conv / Linear and every dgrad (with W^T materialised).
//   gemm_tn : C[Mo,Ni] = sum_p proA(A)[p,Mo] * proB(B)[p,Ni]   (wgrad, reduce over points,
//             split over row ranges into slabs that a second kernel sums).
//
// What makes them the "fused shared MLP" rather than library GEMMs is what rides on the
// operand staging (prologue) and on the accumulator tile (epilogue):
//   prologue  BNRELU : a = relu(z*s[k] + t[k])      BatchNorm+ReLU applied while staging,
//                                                   so post-activation tensors never exist
//             BNBWD  : a = pa[k]*dy + pb[k]*z + pc[k]  BatchNorm backward applied on load
//             GATE1  : a = relu(i*w1[k] + b1[k])    intensity-gate hidden layer from 1 scalar
//   epilogue  bias, per-column batch statistics (sum, centred M2 per 64-row wave tile),
//             gate combine F = relu(zf*s+t) * (0.5+0.5*sigmoid(acc+b)), ReLU-mask +
//             accumulate + BN-backward statistics for dgrad.
//
// Layout: everything is POINT-MAJOR [rows = B*N points][channels]; a wave's global loads
// are 16 B per lane along channels, 128 B contiguous per 8 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

namespace prh {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { PRO_NONE = 0, PRO_BNRELU = 1, PRO_BNBWD = 2, PRO_GATE1 = 3 };
enum { EPI_BIAS = 0, EPI_BIAS_STATS = 1, EPI_GATE = 2, EPI_DGRAD = 3 };

// flags
enum {
  F_RELU_OUT = 1,     // EPI_BIAS: C = relu(acc+bias)
  F_ACCUM = 2,        // EPI_DGRAD: v = acc + C_old
  F_MASK = 4,         // EPI_DGRAD: v *= (E1*es+et > 0)
  F_STATS = 8,        // EPI_DGRAD: write per-wave column sums of v and v*E1
  F_STORE_GATE = 16,  // EPI_GATE: also store m = 0.5+0.5*sigmoid() to C2
  F_E1_ROWVEC = 32,   // EPI_DGRAD: E1 is one value per row (E1[row*lde1]), not a matrix
  F_RESID = 64,       // EPI_BIAS: C = acc + bias + E1 (residual input, before the optional ReLU)
  F_DROPOUT = 256,    // EPI_BIAS: after the optional ReLU, C = keep(row, col) ? C * drop_scale : 0 - the dropout of
                      //   the FFN hidden layer (src/model.py:131: linear2(dropout(activation(linear1(tgt))))) on the
                      //   epilogue that produces it; the decision is a counter hash of (seed, row, column), so no
                      //   mask is stored: the backward reads it off the output (C > 0)
  F_STAGGER = 512,    // NT cores: the odd workgroups of the first generation (blockIdx < 256) start about half a
                      //   tile late, so that the epilogues of half the chip (HBM traffic, no MFMA) fall into the main
                      //   loops of the other half (MFMA, little HBM traffic) instead of all 256 CUs alternating in step
  F_POOL = 128,       // EPI_GATE (vector epilogue): per 128-row wave tile and column, the largest output,
                      //   the tile-local row of its FIRST occurrence and the column sum -> ws_a / ws_c (int
                      //   bits) / ws_b [2*row_tiles][N]: the dual pooling (src/model.py:58-60) rides on the
                      //   epilogue that produces `fused`; pool_tiles_kernel combines the tiles of a segment
};

struct NTParams {
  const float* A;  long lda;     // [M,K]
  const float* A2; long lda2;    // PRO_BNBWD: z
  const float* W;  long ldw;     // [N,K]
  float* C;        long ldc;     // [M,N]
  int M, N, K;
  const float* bias;             // [N] or null
  const float* pa; const float* pb; const float* pc;   // prologue vectors over K
  const float* E1; long lde1;    // epilogue operand [M,N] (zf for GATE, z for DGRAD mask)
  const float* es; const float* et;  // epilogue per-column scale/shift
  float* C2;       long ldc2;    // EPI_GATE second output
  float* ws_a; float* ws_b;      // stats partials [2*row_tiles][N]
  float* ws_c; float* ws_d;      // EPI_BIAS_STATS: per-column max / min of the outputs per row block
                                 //   (null: not produced) - the exact maximum of relu(bn(z)) follows
  int flags;
  int tiles_n;
  char* wprep;                   // scratch for the split weight image (null: fp32 MFMA core)
  const float* amaxA;            // fp16-plane cores: largest |pro(A)| (device; null: the launch
  const float* amaxW;            //   measures it) and largest |W| (set by the launch)
  int mask_col0;                      // EPI_DGRAD (vector epilogue): F_MASK / F_STATS apply to columns >= mask_col0 only
                                      //   (a multiple of 64).  The fusion conv's dgrad writes the gradient of all five
                                      //   conv blocks; blocks 1..4 are only PARTIAL there - the conv dgrad that later
                                      //   completes a block accumulates into it and masks the sum (mask (a + b) =
                                      //   mask a + mask b) and takes its statistics - so their z need not be read here
  unsigned drop_seed, drop_thresh;    // F_DROPOUT: host seed, p * 2^32
  float drop_scale;                   //   1 / (1 - p)
  const unsigned* seed_src;           //   optional device word mixed into the seed (graph replays)
};

// dropout decision of element (row, col): counter hash, the same for every kernel that needs it again
__device__ __forceinline__ bool epi_keep(unsigned seed, unsigned row, unsigned col, unsigned thresh) {
  unsigned x = seed ^ (row * 0x9E3779B1u) ^ (col * 0x85EBCA77u);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x >= thresh;
}
__device__ __forceinline__ unsigned epi_seed(const NTParams& p) {
  if (p.seed_src == nullptr) return p.drop_seed;
  const unsigned w = __hip_atomic_load(p.seed_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return p.drop_seed ^ (__builtin_amdgcn_readfirstlane(w) * 0x9E3779B9u);
}

constexpr int BM = 128, BN = 128, BK = 32;

__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// XCD-aware block id: workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the
// blocks that share an L2), so give each XCD a CONTIGUOUS range of tile ids.  With the column
// tile fastest in the id, the column tiles of one row tile then run side by side on one L2 and
// the activation tile (the large, HBM-resident operand) is fetched once instead of once per
// column tile.  Bijective for any grid size; speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int q = nb >> 3, r = nb & 7, x = b & 7, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// swizzled LDS float offset of 16-B slot `slot` (0..7) in row `row` of a [rows][32] tile.
// ds_read_b128 lane groups hold 16 lanes with distinct row mod 16 -> conflict free.
__device__ __forceinline__ int lds_off(int row, int slot) {
  return row * BK + ((slot ^ ((row >> 1) & 7)) << 2);
}

template <int PRO>
__device__ __forceinline__ float4 pro_apply(float4 a, float4 a2, float4 ka, float4 kb, float4 kc) {
  float4 o;
  if (PRO == PRO_NONE) {
    o = a;
  } else if (PRO == PRO_BNRELU) {
    o.x = fmaxf(fmaf(a.x, ka.x, kb.x), 0.f); o.y = fmaxf(fmaf(a.y, ka.y, kb.y), 0.f);
    o.z = fmaxf(fmaf(a.z, ka.z, kb.z), 0.f); o.w = fmaxf(fmaf(a.w, ka.w, kb.w), 0.f);
  } else if (PRO == PRO_BNBWD) {
    o.x = fmaf(ka.x, a.x, fmaf(kb.x, a2.x, kc.x)); o.y = fmaf(ka.y, a.y, fmaf(kb.y, a2.y, kc.y));
    o.z = fmaf(ka.z, a.z, fmaf(kb.z, a2.z, kc.z)); o.w = fmaf(ka.w, a.w, fmaf(kb.w, a2.w, kc.w));
  } else {  // PRO_GATE1: a.x holds the row's intensity
    o.x = fmaxf(fmaf(a.x, ka.x, kb.x), 0.f); o.y = fmaxf(fmaf(a.x, ka.y, kb.y), 0.f);
    o.z = fmaxf(fmaf(a.x, ka.z, kb.z), 0.f); o.w = fmaxf(fmaf(a.x, ka.w, kb.w), 0.f);
  }
  return o;
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---------------------------------------------------------------------------------------
// Shared epilogue over a wave's accumulator tile acc[MT][NT] of 32x32 MFMA blocks
// (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); dtype independent).
//   rbase/cbase : global row / column of the wave tile's origin
//   rb          : index of this wave's row tile in the statistics partial arrays
//                 (partials cover MT*32 rows each)
// ---------------------------------------------------------------------------------------
template <int EPI, int MT, int NT>
__device__ __forceinline__ void nt_epilogue(f32x16 (&acc)[MT][NT], const NTParams& p, int rbase_,
                                            int cbase, int rb, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
  // Wave-uniform row base in SGPRs + 32-bit in-tile offsets: one VGPR per address instead of
  // a 64-bit pair (the 128-register accumulator tile leaves no room for 64 address pairs).
  const int rbase = __builtin_amdgcn_readfirstlane(rbase_);
  float* __restrict__ Cb = p.C + (size_t)rbase * p.ldc;
  const float* __restrict__ Eb = p.E1 != nullptr ? p.E1 + (size_t)rbase * p.lde1 : nullptr;
  float* __restrict__ C2b = p.C2 != nullptr ? p.C2 + (size_t)rbase * p.ldc2 : nullptr;
  const int ldc = (int)p.ldc, lde1 = (int)p.lde1, ldc2 = (int)p.ldc2;
  const int mrows = p.M - rbase;   // rows of this wave tile that exist
  const unsigned dseed = (EPI == EPI_BIAS && (p.flags & F_DROPOUT) != 0) ? epi_seed(p) : 0u;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = cbase + nt * 32 + l31;
    const bool cok = col < p.N;
    const float bias = (p.bias != nullptr && cok) ? p.bias[col] : 0.f;

    if (EPI == EPI_BIAS || EPI == EPI_BIAS_STATS) {
      float s = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = mt * 32 + crow(r, half);
          float v = acc[mt][nt][r] + bias;
          if (EPI == EPI_BIAS && (p.flags & F_RESID) != 0 && lr < mrows && cok) v += Eb[lr * lde1 + col];
          if ((p.flags & F_RELU_OUT) != 0) v = fmaxf(v, 0.f);
          if (EPI == EPI_BIAS && (p.flags & F_DROPOUT) != 0)
            v = epi_keep(dseed, (unsigned)(rbase + lr), (unsigned)col, p.drop_thresh) ? v * p.drop_scale : 0.f;
          acc[mt][nt][r] = v;
          if (lr < mrows && cok) {
            Cb[lr * ldc + col] = v;
            s += v;
          }
        }
      if (EPI == EPI_BIAS_STATS) {
        s += __shfl_xor(s, 32);
        int nrows = mrows < 0 ? 0 : (mrows > MT * 32 ? MT * 32 : mrows);
        const float mean = nrows > 0 ? s / (float)nrows : 0.f;
        float m2 = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lr = mt * 32 + crow(r, half);
            const float d = acc[mt][nt][r] - mean;
            if (lr < mrows) m2 = fmaf(d, d, m2);
          }
        m2 += __shfl_xor(m2, 32);
        if (half == 0 && cok) {
          p.ws_a[(size_t)rb * p.N + col] = s;
          p.ws_b[(size_t)rb * p.N + col] = m2;
        }
        if (p.ws_c != nullptr) {
          float mx = -3.0e38f, mn = 3.0e38f;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (mt * 32 + crow(r, half) < mrows) { mx = fmaxf(mx, acc[mt][nt][r]); mn = fminf(mn, acc[mt][nt][r]); }
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          mn = fminf(mn, __shfl_xor(mn, 32));
          if (half == 0 && cok) {
            p.ws_c[(size_t)rb * p.N + col] = mx;
            p.ws_d[(size_t)rb * p.N + col] = mn;
          }
        }
      }
    } else if (EPI == EPI_GATE) {
      const float es = cok ? p.es[col] : 0.f, et = cok ? p.et[col] : 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = mt * 32 + crow(r, half);
          if (lr < mrows && cok) {
            const float g = acc[mt][nt][r] + bias;
            const float m = 0.5f + 0.5f / (1.f + __expf(-g));
            const float zf = Eb[lr * lde1 + col];
            const float rl = fmaxf(fmaf(zf, es, et), 0.f);
            Cb[lr * ldc + col] = rl * m;
            if ((p.flags & F_STORE_GATE) != 0) C2b[lr * ldc2 + col] = m;
          }
        }
    } else {  // EPI_DGRAD
      const bool mask = (p.flags & F_MASK) != 0, accum = (p.flags & F_ACCUM) != 0;
      const bool need_z = mask || (p.flags & F_STATS) != 0;
      const int ecol = (p.flags & F_E1_ROWVEC) ? 0 : col;
      const float es = (mask && cok) ? p.es[col] : 0.f, et = (mask && cok) ? p.et[col] : 0.f;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int lr = mt * 32 + crow(r, half);
          if (lr < mrows && cok) {
            float v = acc[mt][nt][r] + bias;
            if (accum) v += Cb[lr * ldc + col];
            float z = 0.f;
            if (need_z) z = Eb[lr * lde1 + ecol];
            if (mask && !(fmaf(z, es, et) > 0.f)) v = 0.f;
            Cb[lr * ldc + col] = v;
            s1 += v;
            s2 = fmaf(v, z, s2);
          }
        }
      if ((p.flags & F_STATS) != 0) {
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (half == 0 && cok) {
          p.ws_a[(size_t)rb * p.N + col] = s1;
          p.ws_b[(size_t)rb * p.N + col] = s2;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Vectorised epilogue for wave tiles of NT = 2 MFMA blocks (64 columns): the accumulator
// layout puts one COLUMN on a lane, so direct stores are 4 B per lane to two 128-B row
// segments per instruction.  Here each 32 x 64 block of the wave tile goes through a
// wave-private LDS scratch (row stride 68 floats) and comes back row-major, 16 B per lane,
// four full 256-B row segments per instruction: 4x fewer memory instructions for the
// store and for every epilogue operand read (z for the ReLU mask, the old gradient).
// `scratch` = this wave's 32*68 floats.  Requires N, ldc, lde1 multiples of 4.
// ---------------------------------------------------------------------------------------
// Diagnostic build only (-DPRH_STAMP, scripts/diag_b16d_stamps.py; the shipped library has none of it): wave-uniform
// cycle accumulators per part of a kernel, s_memtime deltas taken where lgkmcnt(0) holds anyway, dumped by lane 0 of
// waves 0 and 4 of ONE workgroup of a chosen launch after its epilogue.
#ifdef PRH_STAMP
__device__ unsigned* g_prh_stamp = nullptr;
#define PRH_TICK(slot)                                                     \
  {                                                                        \
    const unsigned long long t_now_ = __builtin_amdgcn_s_memtime();        \
    t_acc[slot] += (unsigned)(t_now_ - t_prev);                            \
    t_prev = t_now_;                                                       \
  }
#else
#define PRH_TICK(slot)
#endif
typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
#ifndef PRH_EPI_RB16
#define PRH_EPI_RB16 8      // 4: the round-2 batching (A/B builds: scripts/ab_epi_rb.sh)
#endif
constexpr int EPI_LDW = 68;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The two accumulator layouts of a 128 x 64 wave tile the vector epilogue accepts:
//   f32x16 acc[MT][2]   : 32x32 MFMA blocks (col = lane&31, row = crow(r, lane>>5))
//   f32x4  acc[2*MT][4] : 16x16 MFMA blocks (col = lane&15, row = 4*(lane>>4) + r)
// (a) 32-row block `mt` -> scratch[row][col] (conflict-free 4-B writes), (b) per-column sum and
// centred M2 of acc+bias over the tile's valid rows -> statistics partials.
template <int MT>
__device__ __forceinline__ void epi_block_to_scratch(const f32x16 (&acc)[MT][2], int mt, float* scratch, int lane) {
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) scratch[crow(r, half) * EPI_LDW + nt * 32 + l31] = acc[mt][nt][r];
}
template <int MB>
__device__ __forceinline__ void epi_block_to_scratch(const f32x4 (&acc)[MB][4], int mt, float* scratch, int lane) {
  const int q = lane >> 4, l15 = lane & 15;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        scratch[(h * 16 + q * 4 + r) * EPI_LDW + nb * 16 + l15] = acc[2 * mt + h][nb][r];
}
template <int MT>
__device__ __forceinline__ void epi_col_stats(const f32x16 (&acc)[MT][2], const NTParams& p, int cbase,
                                              int mrows, int rb, int lane, const float* biasp) {
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int col = cbase + nt * 32 + l31;
    const bool cok = col < p.N;
    const float bias = (biasp != nullptr && cok) ? biasp[col] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (mt * 32 + crow(r, half) < mrows) s += acc[mt][nt][r] + bias;
    s += __shfl_xor(s, 32);
    const int nrows = mrows < 0 ? 0 : (mrows > MT * 32 ? MT * 32 : mrows);
    const float mean = nrows > 0 ? s / (float)nrows : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dlt = acc[mt][nt][r] + bias - mean;
        if (mt * 32 + crow(r, half) < mrows) m2 = fmaf(dlt, dlt, m2);
      }
    m2 += __shfl_xor(m2, 32);
    if (half == 0 && cok) {
      p.ws_a[(size_t)rb * p.N + col] = s;
      p.ws_b[(size_t)rb * p.N + col] = m2;
    }
    if (p.ws_c != nullptr) {
      float mx = -3.0e38f, mn = 3.0e38f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (mt * 32 + crow(r, half) < mrows) {
            mx = fmaxf(mx, acc[mt][nt][r] + bias); mn = fminf(mn, acc[mt][nt][r] + bias);
          }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      mn = fminf(mn, __shfl_xor(mn, 32));
      if (half == 0 && cok) {
        p.ws_c[(size_t)rb * p.N + col] = mx;
        p.ws_d[(size_t)rb * p.N + col] = mn;
      }
    }
  }
}
template <int MB>
__device__ __forceinline__ void epi_col_stats(const f32x4 (&acc)[MB][4], const NTParams& p, int cbase,
                                              int mrows, int rb, int lane, const float* biasp) {
  const int q = lane >> 4, l15 = lane & 15;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int col = cbase + nb * 16 + l15;
    const bool cok = col < p.N;
    const float bias = (biasp != nullptr && cok) ? biasp[col] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (mb * 16 + q * 4 + r < mrows) s += acc[mb][nb][r] + bias;
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const int nrows = mrows < 0 ? 0 : (mrows > MB * 16 ? MB * 16 : mrows);
    const float mean = nrows > 0 ? s / (float)nrows : 0.f;
    float m2 = 0.f;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dlt = acc[mb][nb][r] + bias - mean;
        if (mb * 16 + q * 4 + r < mrows) m2 = fmaf(dlt, dlt, m2);
      }
    m2 += __shfl_xor(m2, 16);
    m2 += __shfl_xor(m2, 32);
    if (q == 0 && cok) {
      p.ws_a[(size_t)rb * p.N + col] = s;
      p.ws_b[(size_t)rb * p.N + col] = m2;
    }
    if (p.ws_c != nullptr) {
      float mx = -3.0e38f, mn = 3.0e38f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (mb * 16 + q * 4 + r < mrows) {
            mx = fmaxf(mx, acc[mb][nb][r] + bias); mn = fminf(mn, acc[mb][nb][r] + bias);
          }
      mx = fmaxf(fmaxf(mx, __shfl_xor(mx, 16)), 0.f - 3.0e38f);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      mn = fminf(mn, __shfl_xor(mn, 16));
      mn = fminf(mn, __shfl_xor(mn, 32));
      if (q == 0 && cok) {
        p.ws_c[(size_t)rb * p.N + col] = mx;
        p.ws_d[(size_t)rb * p.N + col] = mn;
      }
    }
  }
}

// 16-bit storage helpers of the bf16-storage mode (prh_b16.hpp): element offsets, H = the
// buffer holds bf16 (2 bytes per element) instead of fp32
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef __bf16 v2b __attribute__((ext_vector_type(2)));
  v2f v = {a, b};
  v2b h = __builtin_convertvector(v, v2b);        // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ float bf16_round(float x) { return bf16_lo(pack_bf16x2(x, 0.f)); }
template <bool H>
__device__ __forceinline__ float4 ld4e(const char* base, int off) {
  if (H) {
    const uint2 u = *reinterpret_cast<const uint2*>(base + (size_t)off * 2);
    return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
  }
  return *reinterpret_cast<const float4*>(base + (size_t)off * 4);
}
template <bool H>
__device__ __forceinline__ void st4e(char* base, int off, float4 v) {
  if (H) *reinterpret_cast<uint2*>(base + (size_t)off * 2) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  else *reinterpret_cast<float4*>(base + (size_t)off * 4) = v;
}

// C16: C, C2, the accumulated-into C and the matrix-shaped E1 are bf16 buffers (leading
// dimensions in elements); a row-vector E1 (F_E1_ROWVEC: one fp32 value per row) stays fp32.
// Round-2 form of the vector epilogue (global loads / stores, clamped rows, flag branches), kept for fp32 storage:
// with the buffer-addressed form below the split-fp16 launches take the same time (they are power-bound in their
// k-loops) but the fusion dgrad FETCHES 103-106 GB instead of 87 GB per launch (PMC passes on the same sources,
// `profiles/r03p_epilogue_experiments.txt`): the shorter epilogue changes how far the workgroups of an XCD drift
// apart, and with them the re-reads of the 8 MB weight image out of the Infinity Cache.  The bf16-storage launches
// (k-loops a third as long, not power-bound) take the buffer-addressed form: 42.8 instead of 48.4 GB and 1-4 % less time.
template <int EPI, int MT, class ACC, bool C16>
__device__ __forceinline__ void nt_epilogue_vec_r2(ACC& acc, const NTParams& p, int rbase_,
                                                int cbase, int rb, int lane, float* scratch,
                                                bool bias_done = false) {     // bias already in acc
  const int rbase = __builtin_amdgcn_readfirstlane(rbase_);
  const float* biasp = bias_done ? nullptr : p.bias;
  constexpr int ES = C16 ? 2 : 4;
  const bool rowvec = EPI == EPI_DGRAD && (p.flags & F_E1_ROWVEC) != 0;
  const int ees = (C16 && !rowvec) ? 2 : 4;
  char* __restrict__ Cb = reinterpret_cast<char*>(p.C) + (size_t)rbase * p.ldc * ES;
  const char* __restrict__ Eb = p.E1 != nullptr ? reinterpret_cast<const char*>(p.E1) + (size_t)rbase * p.lde1 * ees : nullptr;
  const int ldc = (int)p.ldc, lde1 = (int)p.lde1;
  const int mrows = p.M - rbase;
  const int rr = lane >> 4, c4 = (lane & 15) * 4;     // row-major phase: 4 rows x 16 float4
  const int col4 = cbase + c4;
  const bool c4ok = col4 < p.N;
  float4 bias4 = zero4(), es4 = zero4(), et4 = zero4();
  if (c4ok) {
    if (biasp != nullptr) bias4 = ldg4(biasp + col4);
    if ((EPI == EPI_DGRAD && (p.flags & F_MASK) != 0) || EPI == EPI_GATE) {
      es4 = ldg4(p.es + col4);
      et4 = ldg4(p.et + col4);
    }
  }
  char* __restrict__ C2b = (EPI == EPI_GATE && p.C2 != nullptr) ? reinterpret_cast<char*>(p.C2) + (size_t)rbase * p.ldc2 * ES : nullptr;
  const int ldc2 = (int)p.ldc2;

  // statistics straight from the accumulators (column on the lane): sum, then centred M2
  if (EPI == EPI_BIAS_STATS) epi_col_stats(acc, p, cbase, mrows, rb, lane, biasp);

  const bool tile_masked = EPI != EPI_DGRAD || cbase >= p.mask_col0;      // wave-uniform (64-column wave tiles)
  const bool mask = (p.flags & F_MASK) != 0 && tile_masked, accum = (p.flags & F_ACCUM) != 0;
  const bool stats = EPI == EPI_DGRAD && (p.flags & F_STATS) != 0 && tile_masked;
  const bool resid = EPI == EPI_BIAS && (p.flags & F_RESID) != 0;
  const unsigned dseed = (EPI == EPI_BIAS && (p.flags & F_DROPOUT) != 0) ? epi_seed(p) : 0u;
  const bool need_z = (EPI == EPI_DGRAD && (mask || stats)) || EPI == EPI_GATE || resid;
  float4 s1 = zero4(), s2 = zero4();
  const bool pool = EPI == EPI_GATE && (p.flags & F_POOL) != 0;
  float pmx[4] = {-1.f, -1.f, -1.f, -1.f}, psm[4] = {0.f, 0.f, 0.f, 0.f};     // outputs are >= 0
  int pix[4] = {0, 0, 0, 0};
  // The epilogue operands (z for the ReLU mask / statistics / gate, the old C when
  // accumulating) are fetched in ONE batch per 32-row block, branch-free (rows and columns
  // clamped into the tile's valid range; only the store is predicated), so a block exposes a
  // single memory latency.  With the loads inside the per-row bounds and flag branches the
  // compiler emitted load -> s_waitcnt vmcnt(0) -> load -> wait -> store for each of the 32
  // row groups: ~100 serialized round trips per tile.
  const int col4c = col4 < p.N ? col4 : (p.N - 4);
  // load bases: a wave tile entirely below the matrix reads (and discards) row 0 instead
  const char* __restrict__ Cl = mrows > 0 ? Cb : reinterpret_cast<const char*>(p.C);
  const char* __restrict__ El = mrows > 0 ? Eb : reinterpret_cast<const char*>(p.E1);
  const bool acc_old = EPI == EPI_DGRAD && accum;
  // Batches of 4 row groups (two per 32-row block), software-pipelined one deep: batch b+1's
  // operand loads are issued before batch b is processed.  (Measured against the unpipelined
  // order on one box: fusion dgrad 49.6 vs 50.0 ms - the wave sharing the SIMD already covers
  // most of the latency.)
  constexpr int NB = 2 * MT;
  float4 zz[2][4], oo[2][4];
  auto issue = [&](int b, float4 (&z)[4], float4 (&o)[4]) {
    // pin each batch's loads to its place (the operand pointers are read-only/restrict, so the
    // bases are laundered through an asm statement; loads hoisted further up spill accumulators)
    const char* Em = El; const char* Cm = Cl;
    asm volatile("" : "+s"(Em), "+s"(Cm) : : "memory");
    int lrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      z[i] = zero4(); o[i] = zero4();
      int r = (b >> 1) * 32 + ((b & 1) * 4 + i) * 4 + rr;
      r = r < mrows ? r : mrows - 1;
      lrc[i] = r < 0 ? 0 : r;
    }
    if (need_z) {        // wave-uniform: one batch of loads
      if (rowvec) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float s = *reinterpret_cast<const float*>(Em + (size_t)(lrc[i] * lde1) * 4);
          z[i] = make_float4(s, s, s, s);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) z[i] = ld4e<C16>(Em, lrc[i] * lde1 + col4c);
      }
    }
    if (acc_old) {
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = ld4e<C16>(Cm, lrc[i] * ldc + col4c);
    }
  };
  issue(0, zz[0], oo[0]);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int mt = b >> 1, hb = b & 1;
    if (b + 1 < NB) issue(b + 1, zz[(b + 1) & 1], oo[(b + 1) & 1]);
    // accumulator block -> scratch (column layout: conflict-free 128-B rows)
    if (hb == 0) epi_block_to_scratch(acc, mt, scratch, lane);
    // scratch -> row-major float4 per lane
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int it = hb * 4 + i;
      const int lrow = it * 4 + rr;
      const int lr = mt * 32 + lrow;
      const bool ok = lr < mrows && c4ok;
      float4 v = *reinterpret_cast<const float4*>(scratch + lrow * EPI_LDW + c4);
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      const float4 z = zz[b & 1][i];
      const float4 o = oo[b & 1][i];
      if (EPI == EPI_DGRAD) {
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        if (mask) {
          v.x = fmaf(z.x, es4.x, et4.x) > 0.f ? v.x : 0.f;
          v.y = fmaf(z.y, es4.y, et4.y) > 0.f ? v.y : 0.f;
          v.z = fmaf(z.z, es4.z, et4.z) > 0.f ? v.z : 0.f;
          v.w = fmaf(z.w, es4.w, et4.w) > 0.f ? v.w : 0.f;
        }
        if (C16) {      // the statistics describe the values the consumers will read back
          v.x = bf16_round(v.x); v.y = bf16_round(v.y); v.z = bf16_round(v.z); v.w = bf16_round(v.w);
        }
        const float4 q = ok ? v : zero4();
        s1.x += q.x; s1.y += q.y; s1.z += q.z; s1.w += q.w;
        s2.x = fmaf(q.x, z.x, s2.x); s2.y = fmaf(q.y, z.y, s2.y);
        s2.z = fmaf(q.z, z.z, s2.z); s2.w = fmaf(q.w, z.w, s2.w);
      } else if (EPI == EPI_GATE) {
        // F = relu(zf*s+t) * m,  m = 0.5 + 0.5*sigmoid(acc + b)     (src/model.py:51,54-55)
        float4 m;
        m.x = 0.5f + 0.5f / (1.f + __expf(-v.x)); m.y = 0.5f + 0.5f / (1.f + __expf(-v.y));
        m.z = 0.5f + 0.5f / (1.f + __expf(-v.z)); m.w = 0.5f + 0.5f / (1.f + __expf(-v.w));
        v.x = fmaxf(fmaf(z.x, es4.x, et4.x), 0.f) * m.x; v.y = fmaxf(fmaf(z.y, es4.y, et4.y), 0.f) * m.y;
        v.z = fmaxf(fmaf(z.z, es4.z, et4.z), 0.f) * m.z; v.w = fmaxf(fmaf(z.w, es4.w, et4.w), 0.f) * m.w;
        if (ok && (p.flags & F_STORE_GATE) != 0) st4e<C16>(C2b, lr * ldc2 + col4, m);
        if (pool && ok) {      // a lane walks its rows in increasing order: strict > keeps the first maximum
          if (C16) { v.x = bf16_round(v.x); v.y = bf16_round(v.y); v.z = bf16_round(v.z); v.w = bf16_round(v.w); }
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (vv[e] > pmx[e]) { pmx[e] = vv[e]; pix[e] = lr; }
            psm[e] += vv[e];
          }
        }
      } else {
        if (resid) { v.x += z.x; v.y += z.y; v.z += z.z; v.w += z.w; }
        if ((p.flags & F_RELU_OUT) != 0) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (EPI == EPI_BIAS && (p.flags & F_DROPOUT) != 0) {
          const unsigned rw = (unsigned)(rbase + lr), c0 = (unsigned)col4;
          v.x = epi_keep(dseed, rw, c0, p.drop_thresh) ? v.x * p.drop_scale : 0.f;
          v.y = epi_keep(dseed, rw, c0 + 1, p.drop_thresh) ? v.y * p.drop_scale : 0.f;
          v.z = epi_keep(dseed, rw, c0 + 2, p.drop_thresh) ? v.z * p.drop_scale : 0.f;
          v.w = epi_keep(dseed, rw, c0 + 3, p.drop_thresh) ? v.w * p.drop_scale : 0.f;
        }
      }
      if (ok) st4e<C16>(Cb, lr * ldc + col4, v);
    }
    // keep the running column sums here: left alone, the compiler sinks all 64 accumulation
    // steps into the F_STATS branch below and carries every v and z there (32 spilled VGPRs,
    // 9 GB of scratch writes per fusion-dgrad launch at B=4096)
    if (EPI == EPI_DGRAD)
      asm volatile("" : "+v"(s1.x), "+v"(s1.y), "+v"(s1.z), "+v"(s1.w), "+v"(s2.x), "+v"(s2.y), "+v"(s2.z), "+v"(s2.w));
  }
  if (pool) {
    // combine the 4 row groups (lane bits 4, 5): larger value wins, ties go to the smaller row
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        const float ov = __shfl_xor(pmx[e], o);
        const int oi = __shfl_xor(pix[e], o);
        psm[e] += __shfl_xor(psm[e], o);
        if (ov > pmx[e] || (ov == pmx[e] && oi < pix[e])) { pmx[e] = ov; pix[e] = oi; }
      }
    if (lane < 16 && c4ok) {
      *reinterpret_cast<float4*>(p.ws_a + (size_t)rb * p.N + col4) = make_float4(pmx[0], pmx[1], pmx[2], pmx[3]);
      *reinterpret_cast<float4*>(p.ws_b + (size_t)rb * p.N + col4) = make_float4(psm[0], psm[1], psm[2], psm[3]);
      *reinterpret_cast<int4*>(p.ws_c + (size_t)rb * p.N + col4) = make_int4(pix[0], pix[1], pix[2], pix[3]);
    }
  }
  if (stats) {
    // column sums: reduce over the 4 row groups (lane bits 4,5)
    s1.x += __shfl_xor(s1.x, 16); s1.y += __shfl_xor(s1.y, 16); s1.z += __shfl_xor(s1.z, 16); s1.w += __shfl_xor(s1.w, 16);
    s2.x += __shfl_xor(s2.x, 16); s2.y += __shfl_xor(s2.y, 16); s2.z += __shfl_xor(s2.z, 16); s2.w += __shfl_xor(s2.w, 16);
    s1.x += __shfl_xor(s1.x, 32); s1.y += __shfl_xor(s1.y, 32); s1.z += __shfl_xor(s1.z, 32); s1.w += __shfl_xor(s1.w, 32);
    s2.x += __shfl_xor(s2.x, 32); s2.y += __shfl_xor(s2.y, 32); s2.z += __shfl_xor(s2.z, 32); s2.w += __shfl_xor(s2.w, 32);
    if (lane < 16 && c4ok) {
      *reinterpret_cast<float4*>(p.ws_a + (size_t)rb * p.N + col4) = s1;
      *reinterpret_cast<float4*>(p.ws_b + (size_t)rb * p.N + col4) = s2;
    }
  }
}

template <int EPI, int MT, class ACC, bool C16 = false>
__device__ __forceinline__ void nt_epilogue_vec(ACC& acc, const NTParams& p, int rbase_,
                                                int cbase, int rb, int lane, float* scratch,
                                                bool bias_done = false) {     // bias already in acc
  if constexpr (!C16) {
    nt_epilogue_vec_r2<EPI, MT, ACC, false>(acc, p, rbase_, cbase, rb, lane, scratch, bias_done);
    return;
  }
  const int rbase = __builtin_amdgcn_readfirstlane(rbase_);
  const float* biasp = bias_done ? nullptr : p.bias;
  constexpr int ES = C16 ? 2 : 4;
  const bool rowvec = EPI == EPI_DGRAD && (p.flags & F_E1_ROWVEC) != 0;
  const int ees = (C16 && !rowvec) ? 2 : 4;
  char* __restrict__ Cb = reinterpret_cast<char*>(p.C) + (size_t)rbase * p.ldc * ES;
  const char* __restrict__ Eb = p.E1 != nullptr ? reinterpret_cast<const char*>(p.E1) + (size_t)rbase * p.lde1 * ees : nullptr;
  const int ldc = (int)p.ldc, lde1 = (int)p.lde1;
  const int mrows = p.M - rbase;
  const int rr = lane >> 4, c4 = (lane & 15) * 4;     // row-major phase: 4 rows x 16 float4
  const int col4 = cbase + c4;
  const bool c4ok = col4 < p.N;
  float4 bias4 = zero4(), es4 = zero4(), et4 = zero4();
  if (c4ok) {
    if (biasp != nullptr) bias4 = ldg4(biasp + col4);
    if ((EPI == EPI_DGRAD && (p.flags & F_MASK) != 0) || EPI == EPI_GATE) {
      es4 = ldg4(p.es + col4);
      et4 = ldg4(p.et + col4);
    }
  }
  char* __restrict__ C2b = (EPI == EPI_GATE && p.C2 != nullptr) ? reinterpret_cast<char*>(p.C2) + (size_t)rbase * p.ldc2 * ES : nullptr;
  const int ldc2 = (int)p.ldc2;

  // statistics straight from the accumulators (column on the lane): sum, then centred M2
  if (EPI == EPI_BIAS_STATS) epi_col_stats(acc, p, cbase, mrows, rb, lane, biasp);

  const bool tile_masked = EPI != EPI_DGRAD || cbase >= p.mask_col0;      // wave-uniform (64-column wave tiles)
  const bool mask = (p.flags & F_MASK) != 0 && tile_masked, accum = (p.flags & F_ACCUM) != 0;
  const bool stats = EPI == EPI_DGRAD && (p.flags & F_STATS) != 0 && tile_masked;
  const bool resid = EPI == EPI_BIAS && (p.flags & F_RESID) != 0;
  const unsigned dseed = (EPI == EPI_BIAS && (p.flags & F_DROPOUT) != 0) ? epi_seed(p) : 0u;
  const bool need_z = (EPI == EPI_DGRAD && (mask || stats)) || EPI == EPI_GATE || resid;
  float4 s1 = zero4(), s2 = zero4();
  const bool pool = EPI == EPI_GATE && (p.flags & F_POOL) != 0;
  float pmx[4] = {-1.f, -1.f, -1.f, -1.f}, psm[4] = {0.f, 0.f, 0.f, 0.f};     // outputs are >= 0
  int pix[4] = {0, 0, 0, 0};
  // The epilogue operands (z for the ReLU mask / statistics / gate, the old C when accumulating) are fetched in
  // ONE batch per block of rows, software-pipelined one batch deep.  [r03] Every access goes through a BUFFER
  // resource that spans exactly this wave tile's valid rows: a load beyond them returns 0 and a store is dropped,
  // so a row group needs no row clamp, no exec-mask branch around its store and no 64-bit address arithmetic (a
  // lane whose columns lie beyond N carries an out-of-range offset), and the ReLU mask is applied through a
  // threshold (-inf when the launch has no mask) instead of a flag branch.  In-kernel stamps of the bf16 fusion
  // dgrad (scripts/diag_b16d_stamps.py) had shown the epilogue at 27-30 k of a tile's 86 k cycles with neither
  // deeper operand batches (PRH_EPI_RB16) nor staggered workgroups (F_STAGGER) moving it: it is bound by its
  // own instruction stream (~90 instructions per row group, a third of them branches, s_nop and clamps).
  const bool acc_old = EPI == EPI_DGRAD && accum;
  const int vrows = mrows < 0 ? 0 : (mrows > 32 * MT ? 32 * MT : mrows);
  auto mkrs = [&](const void* base, long ld, int es_, int width) {
    const long ext = (base != nullptr && vrows > 0) ? ((long)(vrows - 1) * ld + width) * es_ : 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)ext, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsC = mkrs(Cb, ldc, ES, p.N);
  const __amdgpu_buffer_rsrc_t rsE = mkrs(Eb, lde1, ees, rowvec ? 1 : p.N);
  const __amdgpu_buffer_rsrc_t rsC2 = mkrs(C2b, ldc2, ES, p.N);
  constexpr int OOB = 0x40000000;                      // beyond every extent, no 32-bit overflow with a row offset added
  const int vC = c4ok ? col4 * ES : OOB;               // byte offset of the lane's 4 columns in a row of C / C2
  const int vE = rowvec ? 0 : (c4ok ? col4 * ees : OOB);
  const int rowC = ldc * ES, rowE = lde1 * ees, rowC2 = ldc2 * ES;
  const float mthr = mask ? 0.f : -__builtin_huge_valf();
  constexpr int RB = C16 ? PRH_EPI_RB16 : 4;           // row groups (4 rows each) per batch
  constexpr int NB = 8 * MT / RB;
  typedef typename std::conditional<C16, uint2, float4>::type EV;
  EV zz[2][RB], oo[2][RB];
  auto ev_zero = [] { EV e; if constexpr (C16) e = make_uint2(0u, 0u); else e = zero4(); return e; };
  auto ev_load = [](const __amdgpu_buffer_rsrc_t& rs, int off) {
    EV e;
    if constexpr (C16) e = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
    else e = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
    return e;
  };
  auto ev_f4 = [](const EV& e) {
    if constexpr (C16) return make_float4(bf16_lo(e.x), bf16_hi(e.x), bf16_lo(e.y), bf16_hi(e.y));
    else return e;
  };
  auto ev_store = [](const __amdgpu_buffer_rsrc_t& rs, int off, const float4& v) {
    if constexpr (C16) {
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i_t, make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w))), rs, off, 0, 0);
    } else {
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_t, v), rs, off, 0, 0);
    }
  };
  auto issue = [&](int b, EV (&z)[RB], EV (&o)[RB]) {
#pragma unroll
    for (int i = 0; i < RB; ++i) { z[i] = ev_zero(); o[i] = ev_zero(); }
    if (need_z) {        // wave-uniform: one batch of loads
      if (rowvec) {
        if constexpr (!C16) {
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            const float s_ = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsE, ((b * RB + i) * 4 + rr) * rowE, 0, 0));
            z[i] = make_float4(s_, s_, s_, s_);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < RB; ++i) z[i] = ev_load(rsE, ((b * RB + i) * 4 + rr) * rowE + vE);
      }
    }
    if (acc_old) {
#pragma unroll
      for (int i = 0; i < RB; ++i) o[i] = ev_load(rsC, ((b * RB + i) * 4 + rr) * rowC + vC);
    }
  };
  issue(0, zz[0], oo[0]);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (b + 1 < NB) {
      __builtin_amdgcn_sched_barrier(0);               // the next batch's loads stay in front of this batch's work
      issue(b + 1, zz[(b + 1) & 1], oo[(b + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // scratch -> row-major float4 per lane
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int g = b * RB + i;
      const int mt = g >> 3, it = g & 7;
      // accumulator block -> scratch (column layout: conflict-free 128-B rows)
      if (it == 0) epi_block_to_scratch(acc, mt, scratch, lane);
      const int lrow = it * 4 + rr;
      const int lr = mt * 32 + lrow;
      const bool ok = lr < mrows && c4ok;
      float4 v = *reinterpret_cast<const float4*>(scratch + lrow * EPI_LDW + c4);
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      const float4 z = ev_f4(zz[b & 1][i]);
      const float4 o = ev_f4(oo[b & 1][i]);
      if (EPI == EPI_DGRAD) {
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        v.x = fmaf(z.x, es4.x, et4.x) > mthr ? v.x : 0.f;      // es = et = 0 and mthr = -inf without F_MASK
        v.y = fmaf(z.y, es4.y, et4.y) > mthr ? v.y : 0.f;
        v.z = fmaf(z.z, es4.z, et4.z) > mthr ? v.z : 0.f;
        v.w = fmaf(z.w, es4.w, et4.w) > mthr ? v.w : 0.f;
        if (C16) {      // the statistics describe the values the consumers will read back
          v.x = bf16_round(v.x); v.y = bf16_round(v.y); v.z = bf16_round(v.z); v.w = bf16_round(v.w);
        }
        const float4 q = ok ? v : zero4();
        s1.x += q.x; s1.y += q.y; s1.z += q.z; s1.w += q.w;
        s2.x = fmaf(q.x, z.x, s2.x); s2.y = fmaf(q.y, z.y, s2.y);
        s2.z = fmaf(q.z, z.z, s2.z); s2.w = fmaf(q.w, z.w, s2.w);
      } else if (EPI == EPI_GATE) {
        // F = relu(zf*s+t) * m,  m = 0.5 + 0.5*sigmoid(acc + b)     (src/model.py:51,54-55)
        float4 m;
        m.x = 0.5f + 0.5f / (1.f + __expf(-v.x)); m.y = 0.5f + 0.5f / (1.f + __expf(-v.y));
        m.z = 0.5f + 0.5f / (1.f + __expf(-v.z)); m.w = 0.5f + 0.5f / (1.f + __expf(-v.w));
        v.x = fmaxf(fmaf(z.x, es4.x, et4.x), 0.f) * m.x; v.y = fmaxf(fmaf(z.y, es4.y, et4.y), 0.f) * m.y;
        v.z = fmaxf(fmaf(z.z, es4.z, et4.z), 0.f) * m.z; v.w = fmaxf(fmaf(z.w, es4.w, et4.w), 0.f) * m.w;
        if ((p.flags & F_STORE_GATE) != 0) ev_store(rsC2, lr * rowC2 + vC, m);
        if (pool && ok) {      // a lane walks its rows in increasing order: strict > keeps the first maximum
          if (C16) { v.x = bf16_round(v.x); v.y = bf16_round(v.y); v.z = bf16_round(v.z); v.w = bf16_round(v.w); }
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (vv[e] > pmx[e]) { pmx[e] = vv[e]; pix[e] = lr; }
            psm[e] += vv[e];
          }
        }
      } else {
        if (resid) { v.x += z.x; v.y += z.y; v.z += z.z; v.w += z.w; }
        if ((p.flags & F_RELU_OUT) != 0) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        if (EPI == EPI_BIAS && (p.flags & F_DROPOUT) != 0) {
          const unsigned rw = (unsigned)(rbase + lr), c0 = (unsigned)col4;
          v.x = epi_keep(dseed, rw, c0, p.drop_thresh) ? v.x * p.drop_scale : 0.f;
          v.y = epi_keep(dseed, rw, c0 + 1, p.drop_thresh) ? v.y * p.drop_scale : 0.f;
          v.z = epi_keep(dseed, rw, c0 + 2, p.drop_thresh) ? v.z * p.drop_scale : 0.f;
          v.w = epi_keep(dseed, rw, c0 + 3, p.drop_thresh) ? v.w * p.drop_scale : 0.f;
        }
      }
      ev_store(rsC, lr * rowC + vC, v);
    }
    // keep the running column sums here: left alone, the compiler sinks all 64 accumulation
    // steps into the F_STATS branch below and carries every v and z there (32 spilled VGPRs,
    // 9 GB of scratch writes per fusion-dgrad launch at B=4096)
    if (EPI == EPI_DGRAD)
      asm volatile("" : "+v"(s1.x), "+v"(s1.y), "+v"(s1.z), "+v"(s1.w), "+v"(s2.x), "+v"(s2.y), "+v"(s2.z), "+v"(s2.w));
  }
  if (pool) {
    // combine the 4 row groups (lane bits 4, 5): larger value wins, ties go to the smaller row
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        const float ov = __shfl_xor(pmx[e], o);
        const int oi = __shfl_xor(pix[e], o);
        psm[e] += __shfl_xor(psm[e], o);
        if (ov > pmx[e] || (ov == pmx[e] && oi < pix[e])) { pmx[e] = ov; pix[e] = oi; }
      }
    if (lane < 16 && c4ok) {
      *reinterpret_cast<float4*>(p.ws_a + (size_t)rb * p.N + col4) = make_float4(pmx[0], pmx[1], pmx[2], pmx[3]);
      *reinterpret_cast<float4*>(p.ws_b + (size_t)rb * p.N + col4) = make_float4(psm[0], psm[1], psm[2], psm[3]);
      *reinterpret_cast<int4*>(p.ws_c + (size_t)rb * p.N + col4) = make_int4(pix[0], pix[1], pix[2], pix[3]);
    }
  }
  if (stats) {
    // column sums: reduce over the 4 row groups (lane bits 4,5)
    s1.x += __shfl_xor(s1.x, 16); s1.y += __shfl_xor(s1.y, 16); s1.z += __shfl_xor(s1.z, 16); s1.w += __shfl_xor(s1.w, 16);
    s2.x += __shfl_xor(s2.x, 16); s2.y += __shfl_xor(s2.y, 16); s2.z += __shfl_xor(s2.z, 16); s2.w += __shfl_xor(s2.w, 16);
    s1.x += __shfl_xor(s1.x, 32); s1.y += __shfl_xor(s1.y, 32); s1.z += __shfl_xor(s1.z, 32); s1.w += __shfl_xor(s1.w, 32);
    s2.x += __shfl_xor(s2.x, 32); s2.y += __shfl_xor(s2.y, 32); s2.z += __shfl_xor(s2.z, 32); s2.w += __shfl_xor(s2.w, 32);
    if (lane < 16 && c4ok) {
      *reinterpret_cast<float4*>(p.ws_a + (size_t)rb * p.N + col4) = s1;
      *reinterpret_cast<float4*>(p.ws_b + (size_t)rb * p.N + col4) = s2;
    }
  }
}

// ---------------------------------------------------------------------------------------
// [r03] Direct epilogue for TRANSPOSED 16x16 MFMA blocks.  With the operands of v_mfma_f32_16x16x32_f16
// swapped (weights as the A operand, activations as B) the block comes out as D'[n][m]: a lane holds
// FOUR CONSECUTIVE OUTPUT COLUMNS of one row,
//     acc[i][j][r] = C[row = 16 i + (lane & 15)][col = 16 j + 4 (lane >> 4) + r],
// i.e. 16 contiguous bytes of the row-major fp32 output - the layout the stores (and every epilogue
// operand read: z for the ReLU mask, the old gradient, the residual, zf and the gate) want anyway.  No
// LDS round trip: the wave-private scratch transposes of nt_epilogue_vec (32 ds_write_b32 + 8
// ds_read_b128 per 32-row block) are gone, and with them a third of the epilogue's instructions; a tile
// of 64 columns is walked as two halves of 32 so that the two 64-byte pieces a row's four lanes write
// per block pair complete one 128-byte line back to back.  Column statistics (a lane's 4 columns x the
// 16 rows of its lane group) are reduced over the 16 lanes by DPP-friendly xor shuffles (1, 2, 4, 8).
// Same arithmetic and the same partial-array formats as nt_epilogue_vec.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float4 shfl_xor4(float4 v, int o) {
  return make_float4(__shfl_xor(v.x, o), __shfl_xor(v.y, o), __shfl_xor(v.z, o), __shfl_xor(v.w, o));
}
__device__ __forceinline__ float4 red16_add(float4 v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) { const float4 t = shfl_xor4(v, o); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
  return v;
}
__device__ __forceinline__ float4 red16_max(float4 v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) { const float4 t = shfl_xor4(v, o); v.x = fmaxf(v.x, t.x); v.y = fmaxf(v.y, t.y); v.z = fmaxf(v.z, t.z); v.w = fmaxf(v.w, t.w); }
  return v;
}
__device__ __forceinline__ float4 red16_min(float4 v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) { const float4 t = shfl_xor4(v, o); v.x = fminf(v.x, t.x); v.y = fminf(v.y, t.y); v.z = fminf(v.z, t.z); v.w = fminf(v.w, t.w); }
  return v;
}

// per-column sum and centred M2 (and optionally max / min) of acc + bias over the wave tile's valid rows
__device__ __forceinline__ void epi_col_stats_t(const f32x4 (&acc)[8][4], const NTParams& p, int cbase, int mrows, int rb,
                                                int lane, const float* biasp) {
  const int l15 = lane & 15, q = lane >> 4;
  const int nrows = mrows < 0 ? 0 : (mrows > 128 ? 128 : mrows);
  const float inv_n = nrows > 0 ? 1.f / (float)nrows : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col4 = cbase + j * 16 + 4 * q;
    const bool cok = col4 < p.N;
    const float4 b4 = (biasp != nullptr && cok) ? ldg4(biasp + col4) : zero4();
    float4 s = zero4();
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i * 16 + l15 < mrows) { s.x += acc[i][j][0] + b4.x; s.y += acc[i][j][1] + b4.y; s.z += acc[i][j][2] + b4.z; s.w += acc[i][j][3] + b4.w; }
    s = red16_add(s);
    const float4 mean = make_float4(s.x * inv_n, s.y * inv_n, s.z * inv_n, s.w * inv_n);
    float4 m2 = zero4();
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i * 16 + l15 < mrows) {
        const float dx = acc[i][j][0] + b4.x - mean.x, dy = acc[i][j][1] + b4.y - mean.y;
        const float dz = acc[i][j][2] + b4.z - mean.z, dw = acc[i][j][3] + b4.w - mean.w;
        m2.x = fmaf(dx, dx, m2.x); m2.y = fmaf(dy, dy, m2.y); m2.z = fmaf(dz, dz, m2.z); m2.w = fmaf(dw, dw, m2.w);
      }
    m2 = red16_add(m2);
    if (l15 == 0 && cok) {
      *reinterpret_cast<float4*>(p.ws_a + (size_t)rb * p.N + col4) = s;
      *reinterpret_cast<float4*>(p.ws_b + (size_t)rb * p.N + col4) = m2;
    }
    if (p.ws_c != nullptr) {
      float4 mx = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f), mn = make_float4(3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f);
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i * 16 + l15 < mrows) {
          const float a0 = acc[i][j][0] + b4.x, a1 = acc[i][j][1] + b4.y, a2 = acc[i][j][2] + b4.z, a3 = acc[i][j][3] + b4.w;
          mx.x = fmaxf(mx.x, a0); mx.y = fmaxf(mx.y, a1); mx.z = fmaxf(mx.z, a2); mx.w = fmaxf(mx.w, a3);
          mn.x = fminf(mn.x, a0); mn.y = fminf(mn.y, a1); mn.z = fminf(mn.z, a2); mn.w = fminf(mn.w, a3);
        }
      mx = red16_max(mx); mn = red16_min(mn);
      if (l15 == 0 && cok) {
        *reinterpret_cast<float4*>(p.ws_c + (size_t)rb * p.N + col4) = mx;
        *reinterpret_cast<float4*>(p.ws_d + (size_t)rb * p.N + col4) = mn;
      }
    }
  }
}

template <int EPI>
__device__ __forceinline__ void nt_epilogue_t(f32x4 (&acc)[8][4], const NTParams& p, int rbase_, int cbase, int rb, int lane) {
  const int rbase = __builtin_amdgcn_readfirstlane(rbase_);
  const int l15 = lane & 15, q = lane >> 4;
  const int mrows = p.M - rbase;
  const int ldc = (int)p.ldc, lde1 = (int)p.lde1, ldc2 = (int)p.ldc2;
  const bool rowvec = EPI == EPI_DGRAD && (p.flags & F_E1_ROWVEC) != 0;
  float* __restrict__ Cb = p.C + (size_t)rbase * p.ldc;
  const float* __restrict__ Eb = p.E1 != nullptr ? p.E1 + (size_t)rbase * p.lde1 : nullptr;
  float* __restrict__ C2b = (EPI == EPI_GATE && p.C2 != nullptr) ? p.C2 + (size_t)rbase * p.ldc2 : nullptr;
  // load bases: a wave tile entirely below the matrix reads (and discards) row 0 instead
  const float* __restrict__ Cl = mrows > 0 ? Cb : p.C;
  const float* __restrict__ El = mrows > 0 ? Eb : p.E1;

  if (EPI == EPI_BIAS_STATS) epi_col_stats_t(acc, p, cbase, mrows, rb, lane, p.bias);

  const bool tile_masked = EPI != EPI_DGRAD || cbase >= p.mask_col0;      // wave-uniform
  const bool mask = (p.flags & F_MASK) != 0 && tile_masked, accum = (p.flags & F_ACCUM) != 0;
  const bool stats = EPI == EPI_DGRAD && (p.flags & F_STATS) != 0 && tile_masked;
  const bool resid = EPI == EPI_BIAS && (p.flags & F_RESID) != 0;
  const bool need_z = (EPI == EPI_DGRAD && (mask || stats)) || EPI == EPI_GATE || resid;
  const bool acc_old = EPI == EPI_DGRAD && accum;
  const bool pool = EPI == EPI_GATE && (p.flags & F_POOL) != 0;
  const unsigned dseed = (EPI == EPI_BIAS && (p.flags & F_DROPOUT) != 0) ? epi_seed(p) : 0u;

#pragma unroll
  for (int jh = 0; jh < 2; ++jh) {
    int col4[2]; bool cok[2];
    float4 bias4[2], es4[2], et4[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      col4[jj] = cbase + (2 * jh + jj) * 16 + 4 * q;
      cok[jj] = col4[jj] < p.N;
      bias4[jj] = (p.bias != nullptr && cok[jj]) ? ldg4(p.bias + col4[jj]) : zero4();
      es4[jj] = zero4(); et4[jj] = zero4();
      if (cok[jj] && ((EPI == EPI_DGRAD && mask) || EPI == EPI_GATE)) { es4[jj] = ldg4(p.es + col4[jj]); et4[jj] = ldg4(p.et + col4[jj]); }
    }
    float4 s1[2] = {zero4(), zero4()}, s2[2] = {zero4(), zero4()};
    float pmx[2][4], psm[2][4]; int pix[2][4];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) { pmx[jj][e] = -1.f; psm[jj][e] = 0.f; pix[jj][e] = 0; }      // outputs are >= 0

    // epilogue operands of row block i + 1 are requested before block i is processed (rows and columns
    // clamped into the valid range, branch-free; only the stores are predicated)
    float4 zz[2][2], oo[2][2];
    auto issue = [&](int i, float4 (&z)[2], float4 (&o)[2]) {
      const float* Em = El; const float* Cm = Cl;
      asm volatile("" : "+s"(Em), "+s"(Cm) : : "memory");
      int r = i * 16 + l15;
      r = r < mrows ? r : mrows - 1;
      r = r < 0 ? 0 : r;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int cc = cok[jj] ? col4[jj] : (p.N - 4);
        z[jj] = zero4(); o[jj] = zero4();
        if (need_z) {
          if (rowvec) { const float sv = Em[(size_t)r * lde1]; z[jj] = make_float4(sv, sv, sv, sv); }
          else z[jj] = ldg4(Em + (size_t)r * lde1 + cc);
        }
        if (acc_old) o[jj] = ldg4(Cm + (size_t)r * ldc + cc);
      }
    };
    issue(0, zz[0], oo[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i + 1 < 8) issue(i + 1, zz[(i + 1) & 1], oo[(i + 1) & 1]);
      const int lr = i * 16 + l15;
      const bool rok = lr < mrows;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * jh + jj;
        const bool ok = rok && cok[jj];
        float4 v = make_float4(acc[i][j][0] + bias4[jj].x, acc[i][j][1] + bias4[jj].y, acc[i][j][2] + bias4[jj].z,
                               acc[i][j][3] + bias4[jj].w);
        const float4 z = zz[i & 1][jj];
        const float4 o = oo[i & 1][jj];
        if (EPI == EPI_DGRAD) {
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          if (mask) {
            v.x = fmaf(z.x, es4[jj].x, et4[jj].x) > 0.f ? v.x : 0.f;
            v.y = fmaf(z.y, es4[jj].y, et4[jj].y) > 0.f ? v.y : 0.f;
            v.z = fmaf(z.z, es4[jj].z, et4[jj].z) > 0.f ? v.z : 0.f;
            v.w = fmaf(z.w, es4[jj].w, et4[jj].w) > 0.f ? v.w : 0.f;
          }
          const float4 qv = ok ? v : zero4();
          s1[jj].x += qv.x; s1[jj].y += qv.y; s1[jj].z += qv.z; s1[jj].w += qv.w;
          s2[jj].x = fmaf(qv.x, z.x, s2[jj].x); s2[jj].y = fmaf(qv.y, z.y, s2[jj].y);
          s2[jj].z = fmaf(qv.z, z.z, s2[jj].z); s2[jj].w = fmaf(qv.w, z.w, s2[jj].w);
        } else if (EPI == EPI_GATE) {
          // F = relu(zf*s+t) * m,  m = 0.5 + 0.5*sigmoid(acc + b)     (src/model.py:51,54-55)
          float4 m;
          m.x = 0.5f + 0.5f / (1.f + __expf(-v.x)); m.y = 0.5f + 0.5f / (1.f + __expf(-v.y));
          m.z = 0.5f + 0.5f / (1.f + __expf(-v.z)); m.w = 0.5f + 0.5f / (1.f + __expf(-v.w));
          v.x = fmaxf(fmaf(z.x, es4[jj].x, et4[jj].x), 0.f) * m.x; v.y = fmaxf(fmaf(z.y, es4[jj].y, et4[jj].y), 0.f) * m.y;
          v.z = fmaxf(fmaf(z.z, es4[jj].z, et4[jj].z), 0.f) * m.z; v.w = fmaxf(fmaf(z.w, es4[jj].w, et4[jj].w), 0.f) * m.w;
          if (ok && (p.flags & F_STORE_GATE) != 0) *reinterpret_cast<float4*>(C2b + (size_t)lr * ldc2 + col4[jj]) = m;
          if (pool && ok) {      // a lane walks its rows in increasing order: strict > keeps the first maximum
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (vv[e] > pmx[jj][e]) { pmx[jj][e] = vv[e]; pix[jj][e] = lr; }
              psm[jj][e] += vv[e];
            }
          }
        } else {
          if (resid) { v.x += z.x; v.y += z.y; v.z += z.z; v.w += z.w; }
          if ((p.flags & F_RELU_OUT) != 0) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          }
          if (EPI == EPI_BIAS && (p.flags & F_DROPOUT) != 0) {
            const unsigned rw = (unsigned)(rbase + lr), c0 = (unsigned)col4[jj];
            v.x = epi_keep(dseed, rw, c0, p.drop_thresh) ? v.x * p.drop_scale : 0.f;
            v.y = epi_keep(dseed, rw, c0 + 1, p.drop_thresh) ? v.y * p.drop_scale : 0.f;
            v.z = epi_keep(dseed, rw, c0 + 2, p.drop_thresh) ? v.z * p.drop_scale : 0.f;
            v.w = epi_keep(dseed, rw, c0 + 3, p.drop_thresh) ? v.w * p.drop_scale : 0.f;
          }
        }
        if (ok) *reinterpret_cast<float4*>(Cb + (size_t)lr * ldc + col4[jj]) = v;
      }
      // keep the running column sums here (left alone, the compiler sinks the accumulation steps into the
      // `stats` branch below and carries every v and z there: spills, as in nt_epilogue_vec)
      if (EPI == EPI_DGRAD)
        asm volatile("" : "+v"(s1[0].x), "+v"(s1[0].y), "+v"(s1[0].z), "+v"(s1[0].w), "+v"(s1[1].x), "+v"(s1[1].y), "+v"(s1[1].z),
                     "+v"(s1[1].w), "+v"(s2[0].x), "+v"(s2[0].y), "+v"(s2[0].z), "+v"(s2[0].w), "+v"(s2[1].x), "+v"(s2[1].y),
                     "+v"(s2[1].z), "+v"(s2[1].w));
    }
    if (pool) {
      // combine the 16 lanes of a group (rows 16 i + l15): larger value wins, ties go to the smaller row
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
            const float ov = __shfl_xor(pmx[jj][e], o);
            const int oi = __shfl_xor(pix[jj][e], o);
            psm[jj][e] += __shfl_xor(psm[jj][e], o);
            if (ov > pmx[jj][e] || (ov == pmx[jj][e] && oi < pix[jj][e])) { pmx[jj][e] = ov; pix[jj][e] = oi; }
          }
        if (l15 == 0 && cok[jj]) {
          *reinterpret_cast<float4*>(p.ws_a + (size_t)rb * p.N + col4[jj]) = make_float4(pmx[jj][0], pmx[jj][1], pmx[jj][2], pmx[jj][3]);
          *reinterpret_cast<float4*>(p.ws_b + (size_t)rb * p.N + col4[jj]) = make_float4(psm[jj][0], psm[jj][1], psm[jj][2], psm[jj][3]);
          *reinterpret_cast<int4*>(p.ws_c + (size_t)rb * p.N + col4[jj]) = make_int4(pix[jj][0], pix[jj][1], pix[jj][2], pix[jj][3]);
        }
      }
    }
    if (stats) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const float4 a = red16_add(s1[jj]), b = red16_add(s2[jj]);
        if (l15 == 0 && cok[jj]) {
          *reinterpret_cast<float4*>(p.ws_a + (size_t)rb * p.N + col4[jj]) = a;
          *reinterpret_cast<float4*>(p.ws_b + (size_t)rb * p.N + col4[jj]) = b;
        }
      }
    }
  }
}

// NARROW: N <= 64 (one column tile).  The 2x2 wave grid then covers 128 x 64 with 64 x 32 wave
// tiles instead of spending half of every MFMA on columns that do not exist (the intensity
// gate's dgrad, K=1024 N=64, ran at 80 % of the fp32 matrix peak with half of it wasted).
template <int PRO, int EPI, bool NARROW = false>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const NTParams p) {
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * BK];
  float* As = smem;
  float* Ws = smem + BM * BK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  constexpr int NTW = NARROW ? 1 : 2;          // 32-column blocks per wave
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * (NARROW ? 32 : 64);
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = vb % p.tiles_n, tile_m = vb / p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int sr = tid >> 3;         // staging row 0..31 (+32*j)
  const int sk = (tid & 7) * 4;    // staging k offset inside the k-tile

  f32x16 acc[2][NTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[4], ra2[4], rw[4];
  const int nk = (p.K + BK - 1) / BK;

  auto load_tile = [&](int kt) {
    const int k = kt * BK + sk;
    const bool kok = k < p.K;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = m0 + sr + 32 * j;
      const bool ok = kok && row < p.M;
      if (PRO == PRO_GATE1) {
        ra[j] = zero4();
        if (ok) ra[j].x = p.A[(size_t)row * p.lda];
      } else {
        ra[j] = ok ? ldg4(p.A + (size_t)row * p.lda + k) : zero4();
        if (PRO == PRO_BNBWD) ra2[j] = ok ? ldg4(p.A2 + (size_t)row * p.lda2 + k) : zero4();
      }
      const int n = n0 + sr + 32 * j;
      if (!NARROW || j < 2) rw[j] = (kok && n < p.N) ? ldg4(p.W + (size_t)n * p.ldw + k) : zero4();
    }
  };
  auto store_tile = [&](int kt) {
    const int k = kt * BK + sk;
    const bool kok = k < p.K;
    float4 ka = zero4(), kb = zero4(), kc = zero4();
    if (PRO != PRO_NONE && kok) {
      ka = ldg4(p.pa + k);
      kb = ldg4(p.pb + k);
      if (PRO == PRO_BNBWD) kc = ldg4(p.pc + k);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = sr + 32 * j;
      const bool ok = kok && (m0 + r) < p.M;
      float4 v = ok ? pro_apply<PRO>(ra[j], ra2[j], ka, kb, kc) : zero4();
      *reinterpret_cast<float4*>(As + lds_off(r, sk >> 2)) = v;
      if (!NARROW || j < 2) *reinterpret_cast<float4*>(Ws + lds_off(r, sk >> 2)) = rw[j];
    }
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 a4[2], b4[NTW];
#pragma unroll
      for (int t = 0; t < 2; ++t)
        a4[t] = *reinterpret_cast<const float4*>(As + lds_off(wm + t * 32 + l31, g * 2 + half));
#pragma unroll
      for (int t = 0; t < NTW; ++t)
        b4[t] = *reinterpret_cast<const float4*>(Ws + lds_off(wn + t * 32 + l31, g * 2 + half));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].x, b4[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].y, b4[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].z, b4[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i].w, b4[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile(kt + 1);
      __syncthreads();
    }
  }

  nt_epilogue<EPI, 2, NTW>(acc, p, m0 + wm, n0 + wn, tile_m * 2 + (wave >> 1), lane);
}

// ---------------------------------------------------------------------------------------
// wgrad core: C[Mo,Ni] (+)= sum over rows p in this split of proA(A)[p,Mo] * proB(B)[p,Ni]
// ---------------------------------------------------------------------------------------
struct TNParams {
  const float* A;  long lda;     // [P,Mo]  (dy)
  const float* A2; long lda2;    // PRO_BNBWD: z
  const float* B;  long ldb;     // [P,Ni]  (z_prev / x / intensity column for GATE1)
  int P, Mo, Ni;
  const float* pa; const float* pb; const float* pc;   // A prologue over Mo
  const float* qa; const float* qb;                    // B prologue over Ni
  float* slab;        // [splits][Mo][Ni]
  float* colsum;      // [splits][Mo] column sums of proA(A), or null
  int splits; int rows_per_split;
  int tiles_m, tiles_n;
  const float* amaxA;            // fp16-plane core: largest |proA(A)|, |proB(B)| (device; null:
  const float* amaxB;            //   the launch measures them)
  int* pace;                     // transposed-read core: per-split progress counters (zeroed), or null
  int skew;                      // diagnostic (PRH_TN_SKEW): odd tiles start this many ~2 us naps late
};

// NARROW: Ni <= 64 (one column tile): 64 x 32 wave tiles, as in gemm_nt_kernel
template <int PROA, int PROB, bool NARROW = false>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TNParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * 128];
  float* As = smem;              // [BK][128]  (k = point rows, m contiguous)
  float* Bs = smem + BK * 128;   // [BK][128]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  constexpr int NTW = NARROW ? 1 : 2;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * (NARROW ? 32 : 64);
  // XCD-aware id: the tiles of one row split run on one XCD, so the split's rows of A and B
  // are fetched into that L2 once and shared by its tiles_m x tiles_n blocks
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = b % p.tiles_n; b /= p.tiles_n;
  const int tile_m = b % p.tiles_m; b /= p.tiles_m;
  const int split = b;
  const int m0 = tile_m * 128, n0 = tile_n * 128;
  const int p_begin = split * p.rows_per_split;
  int p_end = p_begin + p.rows_per_split;
  if (p_end > p.P) p_end = p.P;

  const int sc = (tid & 31) * 4;   // staging column (float4)
  const int sr = tid >> 5;         // staging row 0..7 (+8*j)

  f32x16 acc[2][NTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool aok = (m0 + sc) < p.Mo, bok = (n0 + sc) < p.Ni;
  float4 ka = zero4(), kb = zero4(), kc = zero4(), qa = zero4(), qb = zero4();
  if (PROA != PRO_NONE && aok) {
    ka = ldg4(p.pa + m0 + sc);
    kb = ldg4(p.pb + m0 + sc);
    if (PROA == PRO_BNBWD) kc = ldg4(p.pc + m0 + sc);
  }
  if (PROB != PRO_NONE && bok) {
    qa = ldg4(p.qa + n0 + sc);
    qb = ldg4(p.qb + n0 + sc);
  }
  const bool do_colsum = (p.colsum != nullptr) && tile_n == 0;
  float4 csum = zero4();

  float4 ra[4], ra2[4], rb[4];
  const int nk = (p_end - p_begin + BK - 1) / BK;

  auto load_tile = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = p_begin + kt * BK + sr + 8 * j;
      const bool rok = row < p_end;
      ra[j] = (rok && aok) ? ldg4(p.A + (size_t)row * p.lda + m0 + sc) : zero4();
      if (PROA == PRO_BNBWD) ra2[j] = (rok && aok) ? ldg4(p.A2 + (size_t)row * p.lda2 + m0 + sc) : zero4();
      if (PROB == PRO_GATE1) {
        rb[j] = zero4();
        if (rok && bok) rb[j].x = p.B[(size_t)row * p.ldb];
      } else {
        rb[j] = (rok && bok) ? ldg4(p.B + (size_t)row * p.ldb + n0 + sc) : zero4();
      }
    }
  };
  auto store_tile = [&](int kt) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = sr + 8 * j;
      const bool rok = (p_begin + kt * BK + r) < p_end;
      float4 va = (rok && aok) ? pro_apply<PROA>(ra[j], ra2[j], ka, kb, kc) : zero4();
      float4 vb = (rok && bok) ? pro_apply<PROB>(rb[j], rb[j], qa, qb, qb) : zero4();
      csum.x += va.x; csum.y += va.y; csum.z += va.z; csum.w += va.w;
      *reinterpret_cast<float4*>(As + r * 128 + sc) = va;
      *reinterpret_cast<float4*>(Bs + r * 128 + sc) = vb;
    }
  };

  if (nk > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[2], bb[NTW];
#pragma unroll
      for (int t = 0; t < 2; ++t) a[t] = As[(kk * 2 + half) * 128 + wm + t * 32 + l31];
#pragma unroll
      for (int t = 0; t < NTW; ++t) bb[t] = Bs[(kk * 2 + half) * 128 + wn + t * 32 + l31];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile(kt + 1);
      __syncthreads();
    }
  }

  float* out = p.slab + (size_t)split * p.Mo * p.Ni;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int col = n0 + wn + nt * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + mt * 32 + crow(r, half);
        if (row < p.Mo && col < p.Ni) out[(size_t)row * p.Ni + col] = acc[mt][nt][r];
      }
    }

  if (do_colsum) {
    __syncthreads();
    float* red = smem;   // [8][128]
    *reinterpret_cast<float4*>(red + sr * 128 + sc) = csum;
    __syncthreads();
    if (tid < 128) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += red[r * 128 + tid];
      if (m0 + tid < p.Mo) p.colsum[(size_t)split * p.Mo + m0 + tid] = s;
    }
  }
}

}  // namespace prh
