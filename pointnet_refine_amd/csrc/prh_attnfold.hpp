// Inference-only cross-attention with the key / value PROJECTIONS FOLDED IN (VERDICT
// r01 item 5; src/model.py:119-128 in eval mode).  The reference projects the memory to keys and
// values per layer (two [B N, 256] x [256, 256] GEMMs) and attends over the results.  By
// associativity the same numbers come out of attending over the RAW rows, which are the same for
// all six layers and all eight heads:
//   S_h = Q_h K_h^T = Q_h (X Wk_h^T + 1 bk_h^T)^T = (Q_h Wk_h) X^T + (Q_h bk_h) 1^T
//         - the second term is constant along the keys and drops out of the softmax;
//   O_h = P_h V_h = P_h (Y Wv_h^T + 1 bv_h^T) = (P_h Y) Wv_h^T + bv_h        (rows of P_h sum to 1)
// with X = memory + pos and Y = memory.  So: no k_all / v_all buffers (12.9 GB per 2048 segments
// in bf16), no projection GEMMs (14 of the 65 ms of a config-5 forward, bound by writing those
// buffers), and the attention products become 256 deep instead of 32 - on a kernel that was bound
// by vector-instruction issue, not by the matrix pipe.
//
// One workgroup = one segment, its 8 waves = the 8 heads.  A tile of 32 keys of X and of Y
// (bf16, [32][256], LDS-DMA, three stages, one barrier per tile) is shared by the 8 heads.  Per head:
//   prologue  Q'^T[c][q] = sum_d Wk[h 32 + d][c] Q_h[q][d]       (8 blocks of 32 channels; kept as
//             bf16 B-operand fragments: 64 registers)
//   per tile  S^T[key][q] = sum_c X[key][c] Q'^T[c][q]            16 MFMAs 32x32x16
//             online softmax (no dropout in eval mode)
//             PY^T[c][q] += sum_key Y^T[c][key] P^T[key][q]        16 MFMAs, Y^T by transposed LDS reads
//   epilogue  O^T[d][q] = (sum_c Wv[h 32 + d][c] PY^T[c][q]) / l[q] + bv
// An accumulator tile is used as the B operand of the next product (prh_attn16.hpp): its registers
// 8s..8s+7 hold rows 16s + 8(j>>2) + 4h2 + (j&3).  The X rows are therefore stored with bits 2 and
// 3 of the channel index swapped inside every group of 16 (cast_perm_b16_kernel), so that the
// eight channels a lane needs for k-step s are one 16-B chunk; with that storage the rows of PY^T
// come out such that the Wv fragments are read in natural order.  bf16 operands, fp32 accumulation
// (BASELINE config 5: 5e-2).
#pragma once
#include "prh_attn16.hpp"
#include "prh_small.hpp"

namespace prh {

typedef unsigned short u16_t;
// LDS tiles of 32 keys x 256 bf16 channels.  Rows are PADDED rather than XOR-swizzled, so that every
// fragment address is one per-lane base plus a compile-time offset (the XOR form cost ~100 address
// instructions per tile and still left 2-way conflicts on the transposed reads: 0.32 conflict
// cycles per active LDS cycle, measured):
//   X (row-on-the-lane 16-B reads: 16 lanes = 16 rows, same chunk):  528-B rows -> bank group 4 (row + chunk)
//   Y (transposed 8-B reads: 4 rows x 4 chunks per 32 lanes):         576-B rows -> bank group 4 (4 row + chunk)
// A padded row is not contiguous with its neighbour, so one LDS-DMA instruction carries one row
// (lanes 0..31, 512 B).
constexpr int AF_XROW = 528, AF_YROW = 576;
constexpr int AF_XTILE = 32 * AF_XROW, AF_YTILE = 32 * AF_YROW;
constexpr int AF_STAGE = AF_XTILE + AF_YTILE;            // 35,328 B
constexpr int AF_LDS = 3 * AF_STAGE;                     // three stages, one barrier per tile

// fp32 [rows][256] (ld) -> bf16 [rows][256], channel p of the output = channel swap23(p) of the input
__global__ __launch_bounds__(256) void cast_perm_b16_kernel(const float* __restrict__ src, long ld,
                                                            u16_t* __restrict__ dst, size_t rows) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per 16-channel group
  if (i >= rows * 16) return;
  const size_t r = i >> 4;
  const int g = (int)(i & 15);
  const float* s = src + r * ld + g * 16;
  const float4 a = ldg4(s), b = ldg4(s + 4), c = ldg4(s + 8), d = ldg4(s + 12);
  uint4 lo = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(c.x, c.y), pack_bf16x2(c.z, c.w));
  uint4 hi = make_uint4(pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w), pack_bf16x2(d.x, d.y), pack_bf16x2(d.z, d.w));
  uint4* o = reinterpret_cast<uint4*>(dst + r * 256 + g * 16);
  o[0] = lo;      // positions 0..7  = channels 0..3, 8..11
  o[1] = hi;      // positions 8..15 = channels 4..7, 12..15
}

__device__ __forceinline__ bf16x8 af_pack(const float (&f)[8]) {
  const uint4 u = make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
  return __builtin_bit_cast(bf16x8, u);
}

// The two row images the folded attention reads, straight from their sources (inference): for every
// context point  pos = relu(xyz W0^T + b0) W2^T + b2  (PositionalEncoding, src/model.py:64-75),
//   x16 = bf16(memory + pos),  y16 = bf16(memory)        (channel order of cast_perm_b16_kernel).
// Replaces pos_hidden + the 256 x 256 Linear with its residual epilogue + two casts: the hidden
// layer never exists in memory (each lane generates the 8 values of its point a k-step needs),
// memory + pos never exists in fp32, and memory is read once: 2 KB per point instead of 7.
// Persistent workgroups of 8 waves; W2 is converted to bf16 into LDS once per workgroup
// ([256 n][256 k], rows padded to 528 B); a wave owns 32 points per pass: 16 k-steps x 8 column
// blocks of v_mfma_f32_32x32x16_bf16 with the point on the lane, then the epilogue reads memory and
// writes both images in 16-B runs.
constexpr int PM_WROW = 528;                               // bytes per W2 row in LDS
constexpr int PM_LDS = 256 * PM_WROW + 256 * 16;           // + (w0x, w0y, w0z, b0) per hidden unit

__global__ __launch_bounds__(512, 2) void posmem_images_kernel(const float* __restrict__ xyz, long ldx,
                                                               const float* __restrict__ w0, const float* __restrict__ b0,
                                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                                               const float* __restrict__ mem, long ldm, long P,
                                                               u16_t* __restrict__ x16, u16_t* __restrict__ y16) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h2 = lane >> 5, l31 = lane & 31;
  float4* tab = reinterpret_cast<float4*>(smem + 256 * PM_WROW);
  // W2 [n][k] fp32 -> bf16 rows in LDS; the first-layer table
  for (int i = tid; i < 256 * 32; i += 512) {              // item = (row n, chunk of 8 k)
    const int n = i >> 5, c = i & 31;
    const float* q = w2 + (size_t)n * 256 + c * 8;
    const float4 a = ldg4(q), b = ldg4(q + 4);
    *reinterpret_cast<uint4*>(smem + n * PM_WROW + c * 16) =
        make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
  }
  if (tid < 256) tab[tid] = make_float4(w0[tid * 3], w0[tid * 3 + 1], w0[tid * 3 + 2], b0 != nullptr ? b0[tid] : 0.f);
  __syncthreads();

  const long tiles = (P + 255) / 256;
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long p0 = t * 256 + wave * 32;
    if (p0 >= P) continue;                                  // (no barrier below: a wave may skip)
    const long pr = p0 + l31 < P ? p0 + l31 : P - 1;
    const float px = xyz[pr * ldx], py_ = xyz[pr * ldx + 1], pz = xyz[pr * ldx + 2];
    f32x16 acc[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) acc[nb] = a16_zero();
#pragma unroll 2
    for (int s = 0; s < 16; ++s) {
      // A[row = point][k = 16 s + 8 h2 + j]: the hidden layer of this lane's point
      float hv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 c = tab[16 * s + 8 * h2 + j];
        hv[j] = fmaxf(fmaf(pz, c.z, fmaf(py_, c.y, fmaf(px, c.x, c.w))), 0.f);
      }
      const bf16x8 hf = af_pack(hv);
      const char* wr = smem + l31 * PM_WROW + (16 * s + 8 * h2) * 2;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        // pos^T[n][point] = sum_k W2[n][k] h[point][k]: A = W2 rows (lane n = 32 nb + l31), B = the hidden
        // layer with the POINT on the lane - so that a lane ends up with 16-B runs of its own row
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wr + nb * 32 * PM_WROW);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, hf, acc[nb], 0, 0, 0);
      }
    }
    // D[row = n = 32 nb + crow(r, h2)][col = point]: registers 8q'..8q'+7 (q' = 0, 1) are channels
    // 16 q' + 4 h2 + {0..3} and 16 q' + 8 + 4 h2 + {0..3} = image positions 16 q' + 8 h2 + {0..7}: one
    // 16-B store per image, two float4 loads of memory
    const long pt = p0 + l31;
    if (pt < P) {
      const float* mrow = mem + pt * ldm + 4 * h2;
      u16_t* xrow = x16 + pt * 256 + 8 * h2;
      u16_t* yrow = y16 + pt * 256 + 8 * h2;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int n0 = nb * 32 + 16 * qq;
          const float4 m0 = ldg4(mrow + n0), m1 = ldg4(mrow + n0 + 8);
          float4 c0 = zero4(), c1 = zero4();
          if (b2 != nullptr) { c0 = ldg4(b2 + n0 + 4 * h2); c1 = ldg4(b2 + n0 + 8 + 4 * h2); }
          const f32x16& a = acc[nb];
          const int r0 = 8 * qq;
          const uint4 xo = make_uint4(pack_bf16x2(m0.x + a[r0] + c0.x, m0.y + a[r0 + 1] + c0.y),
                                      pack_bf16x2(m0.z + a[r0 + 2] + c0.z, m0.w + a[r0 + 3] + c0.w),
                                      pack_bf16x2(m1.x + a[r0 + 4] + c1.x, m1.y + a[r0 + 5] + c1.y),
                                      pack_bf16x2(m1.z + a[r0 + 6] + c1.z, m1.w + a[r0 + 7] + c1.w));
          const uint4 yo = make_uint4(pack_bf16x2(m0.x, m0.y), pack_bf16x2(m0.z, m0.w), pack_bf16x2(m1.x, m1.y),
                                      pack_bf16x2(m1.z, m1.w));
          *reinterpret_cast<uint4*>(xrow + n0) = xo;
          *reinterpret_cast<uint4*>(yrow + n0) = yo;
        }
    }
  }
}

struct AttnFoldParams {
  const float* q; long ldq;          // [B*M, H*32] projected queries
  const u16_t* x16; const u16_t* y16;   // [B*N, 256] bf16, channel-permuted: memory + pos, memory
  const float* wk; long ldwk;        // [H*32, 256] key rows of the layer's in_proj_weight
  const float* wv; long ldwv;        // [H*32, 256] value rows
  const float* bv;                   // [H*32]
  float* o; long ldo;                // [B*M, H*32]
  int B, M, N, H;
  float scale;
};

__global__ __launch_bounds__(512, 2) void attn_fold_fwd_kernel(const AttnFoldParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, h = threadIdx.x >> 6;
  const int h2 = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.x;
  const unsigned lds0 = (unsigned)(unsigned long)((__attribute__((address_space(3))) char*)smem);

  // key tiles: 64 rows per stage (32 of X, 32 of Y), 8 per wave, one row per DMA instruction
  auto dma = [&](int k0, int stage) {
    if (lane < 32) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int pc = __builtin_amdgcn_readfirstlane(h * 8 + i);
        const int row = pc & 31;
        int key = k0 + row;
        key = key < p.N ? key : p.N - 1;
        const u16_t* src = ((pc >> 5) ? p.y16 : p.x16) + ((size_t)b * p.N + key) * 256 + (lane << 3);
        glds16(src, lds0 + stage * AF_STAGE + ((pc >> 5) ? AF_XTILE + row * AF_YROW : row * AF_XROW));
      }
    }
  };
  dma(0, 0);
  dma(32, 1);       // (keys beyond N are clamped: a harmless copy when N <= 32)

  // ---- prologue: Q'^T blocks as B fragments
  bf16x8 qf[8][2];
  {
    float xq[16];
    a16_load_rows(p.q, p.ldq, (long)b * p.M, p.M, h * 32, lane, xq);
#pragma unroll
    for (int t = 0; t < 16; ++t) xq[t] *= p.scale * 1.4426950408889634f;      // scores in units of log2 e
    float t0[8], t1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { t0[j] = xq[j]; t1[j] = xq[8 + j]; }
    const bf16x8 qb0 = af_pack(t0), qb1 = af_pack(t1);
    const float* wkh = p.wk + (size_t)(h * 32 + 8 * h2) * p.ldwk + l31;
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      float w0[8], w1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        w0[j] = wkh[(size_t)j * p.ldwk + cb * 32];
        w1[j] = wkh[(size_t)(16 + j) * p.ldwk + cb * 32];
      }
      f32x16 acc = a16_zero();
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af_pack(w0), qb0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af_pack(w1), qb1, acc, 0, 0, 0);
      bf16x8 f0[1], f1[1];
      a16_acc_frag<1>(acc, 0, f0);
      a16_acc_frag<1>(acc, 1, f1);
      qf[cb][0] = f0[0];
      qf[cb][1] = f1[0];
    }
  }

  f32x16 py[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) py[cb] = a16_zero();
  float m_run = -INFINITY, l_run = 0.f;

  const int g = lane >> 4, jj = lane & 15;
  // three stages: tile kt is multiplied while tiles kt+1 and kt+2 are in flight; the barrier that
  // publishes tile kt also says every wave is done with tile kt-1, whose stage takes tile kt+2
  int stage = 0;
  for (int k0 = 0; k0 < p.N; k0 += 32) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // this wave's rows of tile kt (tile kt+1 may fly)
    __syncthreads();
    dma(k0 + 64, stage == 0 ? 2 : stage - 1);
    const char* xt = smem + stage * AF_STAGE;
    const char* yt = xt + AF_XTILE;
    // S^T[key][q]
    f32x16 s = a16_zero();
    {
      const char* xr = xt + l31 * AF_XROW + h2 * 16;
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(xr + cb * 64 + st * 32);
          s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[cb][st], s, 0, 0, 0);
        }
    }
    // scores arrive in units of log2 e (folded into the query scale): exp2 is one instruction
    if (k0 + 32 > p.N) {      // last, partial tile only
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = (k0 + crow(r, h2) >= p.N) ? -INFINITY : s[r];
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(s[r] - m_new);
      psum += e;
      s[r] = e;
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    if (__any(alpha != 1.f)) {
#pragma unroll
      for (int cb = 0; cb < 8; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) py[cb][r] *= alpha;
    }
    // PY^T[c][q] += Y^T P^T
    bf16x8 pf[2][1];
    a16_acc_frag<1>(s, 0, pf[0]);
    a16_acc_frag<1>(s, 1, pf[1]);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      // lane -> (key row 16 st + 4 (g >> 1) + (jj >> 2), four channels at 16 (g & 1) + 4 (jj & 3)) of a 32-channel block
      const char* yr0 = yt + (16 * st + 4 * (g >> 1) + (jj >> 2)) * AF_YROW + (16 * (g & 1) + 4 * (jj & 3)) * 2;
#pragma unroll
      for (int cb = 0; cb < 8; ++cb) {
        const fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(yr0 + cb * 64));
        const fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (__attribute__((address_space(3))) fp16x4*)(yr0 + cb * 64 + 8 * AF_YROW));
        struct Pair { fp16x4 a, b; } pr = {lo, hi};
        const bf16x8 ya = __builtin_bit_cast(bf16x8, pr);
        py[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya, pf[st][0], py[cb], 0, 0, 0);
      }
    }
    stage = stage == 2 ? 0 : stage + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // no copy may land after the workgroup is gone

  // ---- epilogue: O^T[d][q] = Wv_h PY^T / l + bv
  const float inv = 1.f / l_run;
  f32x16 oacc = a16_zero();
  const float* wvh = p.wv + (size_t)(h * 32 + l31) * p.ldwv + 8 * h2;
#pragma unroll
  for (int cb = 0; cb < 8; ++cb)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      float w[8];
      const float4 wa = ldg4(wvh + cb * 32 + 16 * st), wb = ldg4(wvh + cb * 32 + 16 * st + 4);
      w[0] = wa.x; w[1] = wa.y; w[2] = wa.z; w[3] = wa.w; w[4] = wb.x; w[5] = wb.y; w[6] = wb.z; w[7] = wb.w;
      bf16x8 bfr[1];
      a16_acc_frag<1>(py[cb], st, bfr);
      oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af_pack(w), bfr[0], oacc, 0, 0, 0);
    }
  if (l31 < p.M) {
    float* op = p.o + (size_t)((long)b * p.M + l31) * p.ldo + h * 32;
    const float* bvh = p.bv + h * 32;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 bb = ldg4(bvh + 8 * gq + 4 * h2);
      *reinterpret_cast<float4*>(op + 8 * gq + 4 * h2) =
          make_float4(fmaf(oacc[4 * gq], inv, bb.x), fmaf(oacc[4 * gq + 1], inv, bb.y), fmaf(oacc[4 * gq + 2], inv, bb.z),
                      fmaf(oacc[4 * gq + 3], inv, bb.w));
    }
  }
}

}  // namespace prh
