// Small kernels around the GEMM cores: BatchNorm statistics finalisation (forward and
// backward), eval-mode BN coefficients, dual pooling, the gate/BN-ReLU combine backward,
// weight transposes, slab reductions, column padding.  All fp32 storage; cross-tile
// reductions are combined in fp64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace prh {

// ---------------------------------------------------------------------------------------
// BatchNorm statistics, two stages.  The GEMM epilogues leave per-wave-tile partials
// ws_a/ws_b [R][ld] (R = tens of thousands at B=4096).  Stage 1 compresses them into
// BN_SLICES slices of doubles with enough workgroups to use the chip
// (grid = ceil(N/32) x BN_SLICES, block 256 = 32 columns x 8 row lanes); stage 2 (one thread per
// column) combines the slices and writes the results.
//   forward : ws_a = tile sum, ws_b = tile M2 about the TILE mean (Chan et al. pairwise form,
//             so no E[x^2]-E[x]^2 cancellation); slice record = (n, sum, M2 about slice mean)
//   backward: ws_a = sum dy, ws_b = sum dy*z; slice record = (sum_a, sum_b, -)
// ---------------------------------------------------------------------------------------
constexpr int BN_SLICES = 32;

__global__ __launch_bounds__(256) void bn_stage1_kernel(const float* __restrict__ ws_a,
                                                        const float* __restrict__ ws_b, int R,
                                                        long ld, int N, int tile_rows, int P,
                                                        int forward, double* __restrict__ out) {
  __shared__ double red[3][8][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c, sl = blockIdx.y;
  const int per = (R + BN_SLICES - 1) / BN_SLICES;
  const int r0 = sl * per, r1 = (r0 + per < R) ? r0 + per : R;
  const bool ok = col < N;
  double a = 0.0, b = 0.0, n = 0.0;
  if (ok)
    for (int i = r0 + g; i < r1; i += 8) {
      a += (double)ws_a[(size_t)i * ld + col];
      if (forward) {
        int ni = P - i * tile_rows;
        ni = ni > tile_rows ? tile_rows : (ni < 0 ? 0 : ni);
        n += (double)ni;
      } else {
        b += (double)ws_b[(size_t)i * ld + col];
      }
    }
  red[0][g][c] = a; red[1][g][c] = b; red[2][g][c] = n;
  __syncthreads();
  double A = 0.0, Bs = 0.0, Nn = 0.0;
  for (int j = 0; j < 8; ++j) { A += red[0][j][c]; Bs += red[1][j][c]; Nn += red[2][j][c]; }
  if (forward) {
    const double mu = Nn > 0.0 ? A / Nn : 0.0;
    double m2 = 0.0;
    if (ok)
      for (int i = r0 + g; i < r1; i += 8) {
        int ni = P - i * tile_rows;
        ni = ni > tile_rows ? tile_rows : ni;
        if (ni <= 0) continue;
        const double d = (double)ws_a[(size_t)i * ld + col] / (double)ni - mu;
        m2 += (double)ws_b[(size_t)i * ld + col] + d * d * (double)ni;
      }
    __syncthreads();
    red[1][g][c] = m2;
    __syncthreads();
    Bs = 0.0;
    for (int j = 0; j < 8; ++j) Bs += red[1][j][c];
  }
  if (g == 0 && ok) {
    double* o = out + ((size_t)sl * 3) * N;
    o[col] = forward ? Nn : A;
    o[(size_t)N + col] = forward ? A : Bs;
    o[(size_t)2 * N + col] = forward ? Bs : 0.0;
  }
}

// forward stage 2: combine slice records (n, sum, M2) -> mean, rstd, scale, shift and the
// running-stat update of nn.BatchNorm1d (momentum, UNBIASED variance)
__global__ void bn_fwd_finalize_kernel(const double* __restrict__ sl, int P, int N,
                                       const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* running_mean,
                                       float* running_var, int64_t* nbt, float momentum, float eps,
                                       float* mean_out, float* rstd_out, float* scale_out,
                                       float* shift_out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  // atomic: a plain uniform read-modify-write is a scalar load, which may be served from a
  // scalar-cache line that predates the previous launch's increment
  if (nbt != nullptr && col == 0) atomicAdd(reinterpret_cast<unsigned long long*>(nbt), 1ull);
  if (col >= N) return;
  double n = 0.0, mean = 0.0, m2 = 0.0;
  for (int s = 0; s < BN_SLICES; ++s) {
    const double* r = sl + ((size_t)s * 3) * N;
    const double ns = r[col];
    if (ns <= 0.0) continue;
    const double ms = r[(size_t)N + col] / ns, d = ms - mean, nt = n + ns;
    m2 += r[(size_t)2 * N + col] + d * d * n * ns / nt;
    mean += d * ns / nt;
    n = nt;
  }
  const double var = m2 / (double)P;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  mean_out[col] = (float)mean;
  rstd_out[col] = (float)rstd;
  scale_out[col] = (float)((double)gamma[col] * rstd);
  shift_out[col] = (float)((double)beta[col] - mean * (double)gamma[col] * rstd);
  if (running_mean != nullptr) {
    const double unb = P > 1 ? m2 / (double)(P - 1) : var;
    running_mean[col] = (float)((1.0 - momentum) * (double)running_mean[col] + momentum * mean);
    running_var[col] = (float)((1.0 - momentum) * (double)running_var[col] + momentum * unb);
  }
}

// Largest value of relu(z*scale + shift) from the per-row-block column maxima / minima the
// statistics epilogue recorded (the activation is convex in z, so its maximum over a set of
// rows sits at that set's largest or smallest z): the exact operand maximum of the next GEMM,
// without a pass over the activations.  grid (ceil(N/32), BN_SLICES); part[] -> absmax_final.
__global__ __launch_bounds__(256) void act_amax_kernel(const float* __restrict__ ws_c,
                                                       const float* __restrict__ ws_d, int R, long ld, int N,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       float* __restrict__ part) {
  __shared__ float red[256];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c, sl = blockIdx.y;
  const int per = (R + BN_SLICES - 1) / BN_SLICES;
  const int r0 = sl * per, r1 = (r0 + per < R) ? r0 + per : R;
  float mx = -3.0e38f, mn = 3.0e38f;
  if (col < N)
    for (int i = r0 + g; i < r1; i += 8) {
      mx = fmaxf(mx, ws_c[(size_t)i * ld + col]);
      mn = fminf(mn, ws_d[(size_t)i * ld + col]);
    }
  float a = 0.f;
  if (col < N && mx >= mn) {
    const float sc = scale[col], sh = shift[col];
    a = fmaxf(fmaxf(fmaf(mx, sc, sh), fmaf(mn, sc, sh)), 0.f);
  }
  red[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}
// out[0] = max(v[0..n))   (operand maximum of the fusion conv = maximum over the five blocks)
__global__ void max_of_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  float m = 0.f;
  for (int i = 0; i < n; ++i) m = fmaxf(m, v[i]);
  *out = m;
}

// eval-mode coefficients from running statistics
__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm,
                                      const float* rv, float eps, int N, float* mean_out,
                                      float* rstd_out, float* scale_out, float* shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  const float rstd = 1.0f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * rstd;
  mean_out[c] = rm[c];
  rstd_out[c] = rstd;
  scale_out[c] = sc;
  shift_out[c] = beta[c] - rm[c] * sc;
}

// backward stage 2: Sdy, Sdyz ->
//   dgamma = (Sdyz - mean*Sdy)*rstd, dbeta = Sdy
//   coefficients of dz = ca*dy + cb*z + cc  (train: full BN backward; eval: ca=scale, 0, 0)
//   db (bias of the conv ahead of the BN) = sum dz = ca*Sdy + cb*P*mean + cc*P
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sl, int P, int N,
                                       const float* __restrict__ gamma,
                                       const float* __restrict__ mean,
                                       const float* __restrict__ rstd, int training, float* ca,
                                       float* cb, float* cc, float* dgamma, float* dbeta,
                                       float* dbias) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= N) return;
  double Sdy = 0.0, Sdyz = 0.0;
  for (int s = 0; s < BN_SLICES; ++s) {
    const double* r = sl + ((size_t)s * 3) * N;
    Sdy += r[col];
    Sdyz += r[(size_t)N + col];
  }
  const double mu = mean[col], rs = rstd[col], gm = gamma[col];
  const double dg = (Sdyz - mu * Sdy) * rs;
  if (dgamma != nullptr) dgamma[col] = (float)dg;
  if (dbeta != nullptr) dbeta[col] = (float)Sdy;
  double a = gm * rs, b = 0.0, cst = 0.0;
  if (training) {
    const double m1 = Sdy / (double)P, m2 = dg / (double)P;
    b = -a * m2 * rs;
    cst = a * (m2 * mu * rs - m1);
  }
  ca[col] = (float)a;
  cb[col] = (float)b;
  cc[col] = (float)cst;
  if (dbias != nullptr) dbias[col] = (float)(a * Sdy + b * (double)P * mu + cst * (double)P);
}

// plain reduction of two partial arrays over tiles: out_a[c] = sum_i ws_a[i][c] etc.
__global__ __launch_bounds__(1024) void partials_reduce_kernel(
    const float* __restrict__ ws_a, const float* __restrict__ ws_b, int R2, int N, float* out_a,
    float* out_b) {
  __shared__ double red[2][32][33];
  const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  const bool ok = col < N;
  double s1 = 0.0, s2 = 0.0;
  if (ok)
    for (int i = g; i < R2; i += 32) {
      s1 += (double)ws_a[(size_t)i * N + col];
      if (ws_b != nullptr) s2 += (double)ws_b[(size_t)i * N + col];
    }
  red[0][g][c] = s1;
  red[1][g][c] = s2;
  __syncthreads();
  if (g == 0 && ok) {
    double a = 0.0, b = 0.0;
    for (int j = 0; j < 32; ++j) {
      a += red[0][j][c];
      b += red[1][j][c];
    }
    if (out_a != nullptr) out_a[col] = (float)a;
    if (out_b != nullptr) out_b[col] = (float)b;
  }
}

// ---------------------------------------------------------------------------------------
// Dual pooling over the N points of each segment, src/model.py:58-60.
// grid = (ceil(C/64), B), block 256 = 64 columns x 4 point groups; coalesced 256-B rows.
// Ties go to the FIRST point (torch.max semantics).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ F, int N, int C,
                                                   float* gfeat, int32_t* argmax) {
  __shared__ float smax[4][64];
  __shared__ int sidx[4][64];
  __shared__ float ssum[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c, b = blockIdx.y;
  float mx = -INFINITY, sum = 0.f;
  int ix = 0;
  if (col < C) {
    const float* base = F + (size_t)b * N * C + col;
    for (int n = g; n < N; n += 4) {
      const float v = base[(size_t)n * C];
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

// Dual pooling from the per-tile partials the gate epilogue wrote (F_POOL): a segment of N
// points = N / 128 consecutive 128-row wave tiles (N % 128 == 0).  First-index ties: tiles in
// order, strict >.  grid = (ceil(C/256), B).
__global__ __launch_bounds__(256) void pool_tiles_kernel(const float* __restrict__ pmax,
                                                         const float* __restrict__ psum,
                                                         const int* __restrict__ pidx, int tiles_per_seg,
                                                         int N, int C, float* gfeat, int32_t* argmax) {
  const int col = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (col >= C) return;
  float mx = -INFINITY, sum = 0.f;
  int ix = 0;
  for (int t = 0; t < tiles_per_seg; ++t) {
    const size_t o = ((size_t)b * tiles_per_seg + t) * C + col;
    const float v = pmax[o];
    if (v > mx) { mx = v; ix = t * 128 + pidx[o]; }
    sum += psum[o];
  }
  gfeat[(size_t)b * 2 * C + col] = mx;
  gfeat[(size_t)b * 2 * C + C + col] = sum / (float)N;
  if (argmax != nullptr) argmax[(size_t)b * C + col] = ix;
}

// ---------------------------------------------------------------------------------------
// Backward of  F = relu(zf*s+t) * m,  gfeat = [max_n F | mean_n F]   (src/model.py:51-60)
//   dF_total = dF (may be null) + d_mean/N + [n == argmax] * d_max
//   dy  = dF_total * m * [relu>0]           -> written to dy_out (may alias dF)
//   dG  = dF_total * r * 0.5*sig*(1-sig), sig = 2m-1   -> written over `gate` in place
// plus the BN-backward partials (sum dy, sum dy*zf) per 64-row tile.
// grid = (ceil(P/64), ceil(C/64)), block 256: thread = (col, 16-row group)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void combine_bwd_kernel(
    const float* dF, const float* __restrict__ d_gfeat, const int32_t* __restrict__ argmax,
    const float* __restrict__ zf, float* gate, const float* __restrict__ s,
    const float* __restrict__ t, int P, int N, int C, float* dy_out, float* ws_a, float* ws_b) {
  __shared__ float r1[4][64], r2[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + c;
  const int row0 = blockIdx.x * 64 + g * 16;
  float s1 = 0.f, s2 = 0.f;
  if (col < C) {
    const float sc = s[col], sh = t[col];
    for (int i = 0; i < 16; ++i) {
      const int row = row0 + i;
      if (row >= P) break;
      const size_t off = (size_t)row * C + col;
      float d = dF != nullptr ? dF[off] : 0.f;
      if (d_gfeat != nullptr) {
        const int b = row / N, n = row - b * N;
        d += d_gfeat[(size_t)b * 2 * C + C + col] / (float)N;
        if (argmax[(size_t)b * C + col] == n) d += d_gfeat[(size_t)b * 2 * C + col];
      }
      const float z = zf[off], m = gate[off];
      const float pre = fmaf(z, sc, sh);
      const float r = fmaxf(pre, 0.f);
      const float sig = 2.f * m - 1.f;
      const float dy = pre > 0.f ? d * m : 0.f;
      gate[off] = d * r * 0.5f * sig * (1.f - sig);
      dy_out[off] = dy;
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

// Generic BN(+ReLU) apply  y = [relu](z*s+t)  and its backward-side twin
//   dy = dh * [z*s+t > 0]  with per-64-row partial sums (sum dy, sum dy*z).
// Used for the last layer of an MLP stack (point_mlp) where no GEMM consumes z.
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, long ldz,
                                                       const float* __restrict__ s,
                                                       const float* __restrict__ t, int P, int C,
                                                       int relu, float* y, long ldy) {
  const int col = blockIdx.y * 64 + (threadIdx.x & 63);
  const int row0 = blockIdx.x * 64 + (threadIdx.x >> 6) * 16;
  if (col >= C) return;
  const float sc = s[col], sh = t[col];
  for (int i = 0; i < 16; ++i) {
    const int row = row0 + i;
    if (row >= P) break;
    float v = fmaf(z[(size_t)row * ldz + col], sc, sh);
    if (relu) v = fmaxf(v, 0.f);
    y[(size_t)row * ldy + col] = v;
  }
}

__global__ __launch_bounds__(256) void bn_dy_stats_kernel(
    const float* __restrict__ dh, long lddh, const float* __restrict__ z, long ldz,
    const float* __restrict__ s, const float* __restrict__ t, int P, int C, int relu,
    float* dy, long lddy, float* ws_a, float* ws_b) {
  __shared__ float r1[4][64], r2[4][64];
  const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 + c;
  const int row0 = blockIdx.x * 64 + g * 16;
  float s1 = 0.f, s2 = 0.f;
  if (col < C) {
    const float sc = s[col], sh = t[col];
    for (int i = 0; i < 16; ++i) {
      const int row = row0 + i;
      if (row >= P) break;
      const float zz = z[(size_t)row * ldz + col];
      float d = dh[(size_t)row * lddh + col];
      if (relu && !(fmaf(zz, sc, sh) > 0.f)) d = 0.f;
      dy[(size_t)row * lddy + col] = d;
      s1 += d;
      s2 = fmaf(d, zz, s2);
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

// out[c][r] = in[r][c], in is [R][C] (weights only: a few MB per step)
__global__ void transpose_kernel(const float* __restrict__ in, int R, int C, long ldin,
                                 float* out, long ldout) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < C) ? in[(size_t)r * ldin + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, r = r0 + tx;
    if (c < C && r < R) out[(size_t)c * ldout + r] = tile[tx][j];
  }
}

// out[i] = sum_s slab[s][i]  (out may have a leading dimension: [rows][cols] -> ld)
__global__ void slab_reduce_kernel(const float* __restrict__ slab, int S, int rows, int cols,
                                   float* out, long ldout) {
  const size_t len = (size_t)rows * cols;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  double acc = 0.0;
  for (int s = 0; s < S; ++s) acc += (double)slab[(size_t)s * len + i];
  const size_t r = i / cols, c = i - r * cols;
  out[r * ldout + c] = (float)acc;
}

// the weight-gradient slabs and the bias-gradient (column-sum) slabs of one wgrad in one launch
__global__ void slab_reduce2_kernel(const float* __restrict__ slab, int S, int rows, int cols, float* out,
                                    long ldout, const float* __restrict__ cslab, int ccols, float* cout) {
  const size_t len = (size_t)rows * cols;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < len) {
    double acc = 0.0;
    for (int s = 0; s < S; ++s) acc += (double)slab[(size_t)s * len + i];
    const size_t r = i / cols, c = i - r * cols;
    out[r * ldout + c] = (float)acc;
  } else if (i < len + (size_t)ccols) {
    const size_t j = i - len;
    double acc = 0.0;
    for (int s = 0; s < S; ++s) acc += (double)cslab[(size_t)s * ccols + j];
    cout[j] = (float)acc;
  }
}

// out = y > 0 ? dy : 0 over n floats (n % 4 == 0), per-block max|out| into part[blockIdx.x]:
// the ReLU backward of a Linear with a fused ReLU, and the operand maximum the backward GEMMs
// need, in one pass (it was a compare, a select and a maximum pass)
__global__ __launch_bounds__(256) void relu_mask_amax_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                             float* __restrict__ out, size_t n4, float scale,
                                                             float* __restrict__ part) {
  __shared__ float red[4];
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 g = reinterpret_cast<const float4*>(dy)[i];
    const float4 v = reinterpret_cast<const float4*>(y)[i];
    float4 o;
    o.x = v.x > 0.f ? g.x * scale : 0.f; o.y = v.y > 0.f ? g.y * scale : 0.f;
    o.z = v.z > 0.f ? g.z * scale : 0.f; o.w = v.w > 0.f ? g.w * scale : 0.f;
    reinterpret_cast<float4*>(out)[i] = o;
    m = fmaxf(m, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// dst[r][0..cd) = src[r][0..cs) zero-padded (cd >= cs); or truncation when cd < cs
__global__ void copy_cols_kernel(const float* __restrict__ src, long lds_, int cs, float* dst,
                                 long ldd, int cd, size_t rows) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * (size_t)cd) return;
  const size_t r = i / cd;
  const int c = (int)(i - r * cd);
  dst[r * ldd + c] = c < cs ? src[r * lds_ + c] : 0.f;
}

// d_ctx[p][3] += sum_c dU[p][c]*w1[c]   (gate hidden layer back to the intensity channel)
__global__ __launch_bounds__(256) void gate1_dctx_kernel(const float* __restrict__ dU, int P,
                                                         int H, const float* __restrict__ w1,
                                                         float* d_ctx, long ldc) {
  const int lane = threadIdx.x & 63;
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (size_t)P) return;
  float s = 0.f;
  for (int c = lane; c < H; c += 64) s = fmaf(dU[row * H + c], w1[c], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) d_ctx[row * ldc + 3] += s;
}

__global__ void fill_kernel(float* p, size_t n, float v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------------------
// Row f3.  Deep-supervision L1 loss, forward and backward in one pass
// (train.py:63-68, train_dist.py:180-186: (1/L) sum_l mean|pred_l - target| with nn.L1Loss):
//   part[block] = sum over the block's elements and all L layers of |pred - target| / denom
//   d_pred      = sign(pred - target) / denom          (sign(0) = 0, as torch's L1 backward)
// `denom` is passed in (L * elements of the FULL batch) so a micro-batched caller can add
// chunk losses.  Two-stage deterministic reduction (l1_final_kernel), no atomics.
// ---------------------------------------------------------------------------------------
// The geometry metrics the reference logs every step (train_dist.py:190-203, train.py:74-87)
// ride on the same pass when GEO is set (elements are xyz triples): part[nb + b] = sum over
// points of |target| (initial error), part[2 nb + b] = sum of |pred_last - target| (refined).
template <bool GEO>
__global__ __launch_bounds__(256) void l1_deep_kernel(const float* __restrict__ pred,
                                                      const float* __restrict__ target, int L, long R,
                                                      float inv_denom, float* __restrict__ d_pred,
                                                      float* __restrict__ part) {
  __shared__ float red[3][4];
  float s = 0.f, e0 = 0.f, e1 = 0.f;
  const long stride = (long)gridDim.x * 256;
  if (!GEO) {
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < R; r += stride) {
      const float t = target[r];
      for (int l = 0; l < L; ++l) {
        const float d = pred[(size_t)l * R + r] - t;
        s += fabsf(d);
        if (d_pred != nullptr) d_pred[(size_t)l * R + r] = d > 0.f ? inv_denom : (d < 0.f ? -inv_denom : 0.f);
      }
    }
  } else {
    const long npts = R / 3;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npts; q += stride) {
      float t[3], dl[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c) t[c] = target[q * 3 + c];
      for (int l = 0; l < L; ++l)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const size_t i = (size_t)l * R + q * 3 + c;
          const float d = pred[i] - t[c];
          s += fabsf(d);
          dl[c] = d;                           // the last layer's difference survives the loop
          if (d_pred != nullptr) d_pred[i] = d > 0.f ? inv_denom : (d < 0.f ? -inv_denom : 0.f);
        }
      e0 += sqrtf(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
      e1 += sqrtf(dl[0] * dl[0] + dl[1] * dl[1] + dl[2] * dl[2]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    if (GEO) { e0 += __shfl_xor(e0, o); e1 += __shfl_xor(e1, o); }
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = e0; red[2][threadIdx.x >> 6] = e1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    if (GEO) {
      part[gridDim.x + blockIdx.x] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
      part[2 * gridDim.x + blockIdx.x] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    }
  }
}
__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ part, int n, float inv_denom,
                                                       int accumulate, float* __restrict__ loss) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = (float)(red[0] * (double)inv_denom);
    *loss = accumulate ? *loss + v : v;
  }
}

// torch.optim.Adam (no amsgrad, no maximize; train.py:40, train_dist.py:150) over flat buffers:
//   g' = g + wd*p;  m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   float step_size, float b1, float b2, float eps,
                                                   float wd, float inv_bc2_sqrt) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  if (i + 4 <= n) {
    float4 P = *reinterpret_cast<float4*>(p + i), M = *reinterpret_cast<float4*>(m + i);
    float4 V = *reinterpret_cast<float4*>(v + i);
    const float4 G = *reinterpret_cast<const float4*>(g + i);
    float* pp = &P.x; float* mm = &M.x; float* vv = &V.x; const float* gg = &G.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gr = fmaf(wd, pp[k], gg[k]);
      mm[k] = fmaf(b1, mm[k], (1.f - b1) * gr);
      vv[k] = fmaf(b2, vv[k], (1.f - b2) * gr * gr);
      pp[k] -= step_size * (mm[k] / (sqrtf(vv[k]) * inv_bc2_sqrt + eps));
    }
    *reinterpret_cast<float4*>(p + i) = P;
    *reinterpret_cast<float4*>(m + i) = M;
    *reinterpret_cast<float4*>(v + i) = V;
  } else {
    for (long j = i; j < n; ++j) {
      const float gr = fmaf(wd, p[j], g[j]);
      m[j] = fmaf(b1, m[j], (1.f - b1) * gr);
      v[j] = fmaf(b2, v[j], (1.f - b2) * gr * gr);
      p[j] -= step_size * (m[j] / (sqrtf(v[j]) * inv_bc2_sqrt + eps));
    }
  }
}

// ---------------------------------------------------------------------------------------
// Row f1 (query side of the decoder layer): y = LayerNorm(x + dropout(r)) in one pass, forward
// and backward (src/model.py:117,128,133: tgt = norm(tgt + dropout(tgt2)), post-norm DETR
// layer; nn.LayerNorm eps 1e-5, biased variance).  One wave per row of C = 256 channels
// (4 per lane), reductions by shuffles; the dropout decision is a counter hash of
// (seed, row, channel) so the backward regenerates it (no mask tensor).
//   backward: dh = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
//             dx = dh,  dr = dh * keep/(1-p),  dgamma / dbeta partials per block.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool ln_keep(unsigned seed, unsigned row, unsigned col, unsigned thresh) {
  unsigned x = seed ^ (row * 0x9E3779B1u) ^ (col * 0x85EBCA77u);
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x >= thresh;
}
// host seed mixed with the optional device word (see effective_seed in prh_attn.hpp)
__device__ __forceinline__ unsigned ln_seed(unsigned seed, const unsigned* src) {
  if (src == nullptr) return seed;
  const unsigned w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return seed ^ (__builtin_amdgcn_readfirstlane(w) * 0x9E3779B9u);
}
constexpr int LN_C = 256;
__global__ __launch_bounds__(256) void add_dropout_ln_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ gamma,
    const float* __restrict__ beta, long rows, float eps, unsigned seed_host, const unsigned* seed_src,
    unsigned thresh, float keep_scale, float* __restrict__ y, float* __restrict__ mean_out,
    float* __restrict__ rstd_out) {
  const unsigned seed = ln_seed(seed_host, seed_src);
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int c = lane * 4;
  float4 h = *reinterpret_cast<const float4*>(x + row * LN_C + c);
  const float4 rv = *reinterpret_cast<const float4*>(r + row * LN_C + c);
  float* hp = &h.x; const float* rp = &rv.x;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float t = rp[k];
    if (thresh != 0u) t = ln_keep(seed, (unsigned)row, (unsigned)(c + k), thresh) ? t * keep_scale : 0.f;
    hp[k] += t;
  }
  float s = (h.x + h.y) + (h.z + h.w);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s * (1.f / LN_C);
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const float d = hp[k] - mu; v = fmaf(d, d, v); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const float rstd = rsqrtf(v * (1.f / LN_C) + eps);
  const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
  float4 o4;
  o4.x = fmaf((h.x - mu) * rstd, g.x, b.x); o4.y = fmaf((h.y - mu) * rstd, g.y, b.y);
  o4.z = fmaf((h.z - mu) * rstd, g.z, b.z); o4.w = fmaf((h.w - mu) * rstd, g.w, b.w);
  *reinterpret_cast<float4*>(y + row * LN_C + c) = o4;
  if (lane == 0 && mean_out != nullptr) { mean_out[row] = mu; rstd_out[row] = rstd; }
}

// grid = LN_BWD_BLOCKS blocks of 4 waves; a wave walks rows wave, wave + 4*grid, ...;
// part_g / part_b [grid][256]: the block's sums of dy*xhat and dy per channel
constexpr int LN_BWD_BLOCKS = 512;
__global__ __launch_bounds__(256) void add_dropout_ln_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ r,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd, long rows,
    unsigned seed_host, const unsigned* seed_src, unsigned thresh, float keep_scale, float* __restrict__ dx,
    float* __restrict__ dr, float* __restrict__ part_g, float* __restrict__ part_b) {
  __shared__ float sg[4][LN_C], sb[4][LN_C];
  const unsigned seed = ln_seed(seed_host, seed_src);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane * 4;
  const float4 g4 = *reinterpret_cast<const float4*>(gamma + c);
  const float* gp = &g4.x;
  float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float4 d4 = *reinterpret_cast<const float4*>(dy + row * LN_C + c);
    const float4 x4 = *reinterpret_cast<const float4*>(x + row * LN_C + c);
    const float4 r4 = *reinterpret_cast<const float4*>(r + row * LN_C + c);
    const float mu = mean[row], rs = rstd[row];
    const float* dp = &d4.x; const float* xp = &x4.x; const float* rp = &r4.x;
    float xh[4], gg[4], kf[4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      kf[k] = 1.f;
      if (thresh != 0u) kf[k] = ln_keep(seed, (unsigned)row, (unsigned)(c + k), thresh) ? keep_scale : 0.f;
      xh[k] = (xp[k] + rp[k] * kf[k] - mu) * rs;
      gg[k] = dp[k] * gp[k];
      s1 += gg[k];
      s2 = fmaf(gg[k], xh[k], s2);
      ag[k] = fmaf(dp[k], xh[k], ag[k]);
      ab[k] += dp[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    const float m1 = s1 * (1.f / LN_C), m2 = s2 * (1.f / LN_C);
    float4 ox, orr;
    float* oxp = &ox.x; float* orp = &orr.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dh = rs * (gg[k] - m1 - xh[k] * m2);
      oxp[k] = dh;
      orp[k] = dh * kf[k];
    }
    *reinterpret_cast<float4*>(dx + row * LN_C + c) = ox;
    *reinterpret_cast<float4*>(dr + row * LN_C + c) = orr;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { sg[wave][c + k] = ag[k]; sb[wave][c + k] = ab[k]; }
  __syncthreads();
  const int t = threadIdx.x;
  part_g[(size_t)blockIdx.x * LN_C + t] = (sg[0][t] + sg[1][t]) + (sg[2][t] + sg[3][t]);
  part_b[(size_t)blockIdx.x * LN_C + t] = (sb[0][t] + sb[1][t]) + (sb[2][t] + sb[3][t]);
}
// dgamma[c] = sum_blocks part_g[b][c], dbeta likewise (fp64 accumulation).  grid = 4 blocks of
// 64 channels; a block's 256 threads = 64 channels x 4 interleaved slices of the partial rows.
__global__ __launch_bounds__(256) void ln_param_grad_kernel(const float* __restrict__ part_g,
                                                            const float* __restrict__ part_b, int nb,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double ra[4][64], rb[4][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double a = 0.0, b = 0.0;
  for (int i = sl; i < nb; i += 4) { a += (double)part_g[(size_t)i * LN_C + c]; b += (double)part_b[(size_t)i * LN_C + c]; }
  ra[sl][cl] = a; rb[sl][cl] = b;
  __syncthreads();
  if (sl == 0) {
    dgamma[c] = (float)((ra[0][cl] + ra[1][cl]) + (ra[2][cl] + ra[3][cl]));
    dbeta[c] = (float)((rb[0][cl] + rb[1][cl]) + (rb[2][cl] + rb[3][cl]));
  }
}

// ---------------------------------------------------------------------------------------
// First layer of the positional-encoding MLP (src/model.py:64-75: Linear(3, H) + ReLU) as an
// elementwise kernel: h[r, c] = relu(b0[c] + sum_j xyz[r*ld + j] * w0[c*3 + j]).  A 3-deep
// "GEMM" is pure HBM work - 12 B read and 4*H B written per point - and the points are read
// in place from the (B, N, C) context rows (ld = C), not from a sliced copy.
// Thread -> 4 consecutive columns; the block covers 256 / (H/4) rows per pass.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pos_hidden_fwd_kernel(const float* __restrict__ xyz, long ld,
                                                             const float* __restrict__ w0,
                                                             const float* __restrict__ b0,
                                                             float* __restrict__ h, long P, int H) {
  const int cg = H >> 2, tc = threadIdx.x % cg, tr = threadIdx.x / cg, rpp = 256 / cg;
  float w[4][3], b[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    b[c] = b0 != nullptr ? b0[4 * tc + c] : 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) w[c][j] = w0[(4 * tc + c) * 3 + j];
  }
  for (long r = (long)blockIdx.x * rpp + tr; r < P; r += (long)gridDim.x * rpp) {
    const float x0 = xyz[r * ld], x1 = xyz[r * ld + 1], x2 = xyz[r * ld + 2];
    float4 o;
    o.x = fmaxf(fmaf(x2, w[0][2], fmaf(x1, w[0][1], fmaf(x0, w[0][0], b[0]))), 0.f);
    o.y = fmaxf(fmaf(x2, w[1][2], fmaf(x1, w[1][1], fmaf(x0, w[1][0], b[1]))), 0.f);
    o.z = fmaxf(fmaf(x2, w[2][2], fmaf(x1, w[2][1], fmaf(x0, w[2][0], b[2]))), 0.f);
    o.w = fmaxf(fmaf(x2, w[3][2], fmaf(x1, w[3][1], fmaf(x0, w[3][0], b[3]))), 0.f);
    *reinterpret_cast<float4*>(h + r * H + 4 * tc) = o;
  }
}

// Its backward: dz = dh * (h > 0); part[block][c*4 + j] = sum_r dz[r,c] * xyz[r,j] (j < 3) and
// sum_r dz[r,c] (j == 3) over the block's rows.  pos_hidden_final_kernel adds the blocks up
// (two stages, no atomics: the result does not depend on the launch order).
// DX: also dxyz[r, j] = sum_c dz[r,c] * w0[c*3 + j] (the decoder's query positions carry
// gradients, src/model.py:209-231 "no detach"): the row's H/4 threads share a wave (H <= 256)
// and add their partial sums with shuffles.
template <bool DX>
__global__ __launch_bounds__(256) void pos_hidden_bwd_kernel(const float* __restrict__ xyz, long ld,
                                                             const float* __restrict__ h,
                                                             const float* __restrict__ dh,
                                                             float* __restrict__ part, long P, int H,
                                                             const float* __restrict__ w0,
                                                             float* __restrict__ dxyz) {
  __shared__ float red[256 * 17];
  const int cg = H >> 2, tc = threadIdx.x % cg, tr = threadIdx.x / cg, rpp = 256 / cg;
  float w[4][3];
  if (DX) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 3; ++j) w[c][j] = w0[(4 * tc + c) * 3 + j];
  }
  float a[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) a[c][j] = 0.f;
  for (long r = (long)blockIdx.x * rpp + tr; r < P; r += (long)gridDim.x * rpp) {
    const float x0 = xyz[r * ld], x1 = xyz[r * ld + 1], x2 = xyz[r * ld + 2];
    const float4 hv = *reinterpret_cast<const float4*>(h + r * H + 4 * tc);
    const float4 g = *reinterpret_cast<const float4*>(dh + r * H + 4 * tc);
    const float dz[4] = {hv.x > 0.f ? g.x : 0.f, hv.y > 0.f ? g.y : 0.f, hv.z > 0.f ? g.z : 0.f,
                         hv.w > 0.f ? g.w : 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a[c][0] = fmaf(dz[c], x0, a[c][0]); a[c][1] = fmaf(dz[c], x1, a[c][1]);
      a[c][2] = fmaf(dz[c], x2, a[c][2]); a[c][3] += dz[c];
    }
    if (DX) {
      float t[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        t[j] = dz[0] * w[0][j] + dz[1] * w[1][j] + dz[2] * w[2][j] + dz[3] * w[3][j];
        for (int o = cg >> 1; o > 0; o >>= 1) t[j] += __shfl_xor(t[j], o);
      }
      if (tc == 0) { dxyz[r * 3] = t[0]; dxyz[r * 3 + 1] = t[1]; dxyz[r * 3 + 2] = t[2]; }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[threadIdx.x * 17 + c * 4 + j] = a[c][j];
  __syncthreads();
  // thread t < 4*H sums element t (= column*4 + j) over the rpp row groups
  for (int e = threadIdx.x; e < 4 * H; e += 256) {
    const int col = e >> 2, j = e & 3, tcc = col >> 2, c = col & 3;
    float s = 0.f;
    for (int g = 0; g < rpp; ++g) s += red[(g * cg + tcc) * 17 + c * 4 + j];
    part[(size_t)blockIdx.x * 4 * H + e] = s;
  }
}

__global__ __launch_bounds__(256) void pos_hidden_final_kernel(const float* __restrict__ part, int nblk,
                                                               int H, float* __restrict__ dw0,
                                                               float* __restrict__ db0) {
  // 16 elements x 16 slices of the block list per workgroup, then an LDS tree over the slices
  __shared__ float red[256];
  const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float s = 0.f;
  if (e < 4 * H)
    for (int b = sl; b < nblk; b += 16) s += part[(size_t)b * 4 * H + e];
  red[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && e < 4 * H) {
#pragma unroll
    for (int g = 1; g < 16; ++g) s += red[g * 16 + el];
    const int col = e >> 2, j = e & 3;
    if (j < 3) { if (dw0 != nullptr) dw0[col * 3 + j] = s; }
    else if (db0 != nullptr) db0[col] = s;
  }
}

// ---------------------------------------------------------------------------------------
// nn.Linear with a handful of outputs (the regression heads' Linear(128, 3),
// src/model.py:162-166): y[r, j] = b[j] + sum_k x[r,k] w[j,k], n <= 4.  HBM work (one read
// of x), which the library GEMM served with a 16x32 macro-tile at 0.3 ms per 65536 rows.
// K/4 lanes (a power of two <= 64) share a row: float4 slices of x against register-resident
// slices of the weight rows, then a shuffle tree.
// ---------------------------------------------------------------------------------------
template <int NOUT>
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ b,
                                                               float* __restrict__ y, long rows, int K) {
  const int lpr = K >> 2, tl = threadIdx.x % lpr, tr = threadIdx.x / lpr, rpp = 256 / lpr;
  float4 wr[NOUT];
#pragma unroll
  for (int j = 0; j < NOUT; ++j) wr[j] = ldg4(w + (size_t)j * K + 4 * tl);
  for (long r0 = (long)blockIdx.x * rpp; r0 < rows; r0 += (long)gridDim.x * rpp) {
    const long r = r0 + tr;
    const bool ok = r < rows;                       // whole wave stays in the shuffles
    const float4 xv = ok ? ldg4(x + r * K + 4 * tl) : zero4();
    float t[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
      t[j] = xv.x * wr[j].x + xv.y * wr[j].y + xv.z * wr[j].z + xv.w * wr[j].w;
      for (int o = lpr >> 1; o > 0; o >>= 1) t[j] += __shfl_xor(t[j], o);
    }
    if (ok && tl == 0) {
#pragma unroll
      for (int j = 0; j < NOUT; ++j) y[r * NOUT + j] = t[j] + (b != nullptr ? b[j] : 0.f);
    }
  }
}

// backward: dx[r,k] = sum_j dy[r,j] w[j,k] (optional);  part[block][j*K + k] = sum_r dy[r,j] x[r,k]
// and part[block][NOUT*K + j] = sum_r dy[r,j] over the block's rows (linear_small_final_kernel
// adds the blocks: two stages, no atomics).
template <int NOUT>
__global__ __launch_bounds__(256) void linear_small_bwd_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ dy,
                                                               float* __restrict__ dx,
                                                               float* __restrict__ part, long rows, int K) {
  __shared__ float red[256 * (4 * NOUT + 1)];
  const int lpr = K >> 2, tl = threadIdx.x % lpr, tr = threadIdx.x / lpr, rpp = 256 / lpr;
  constexpr int RS = 4 * NOUT + 1;
  float4 wr[NOUT], acc[NOUT];
  float sb[NOUT];
#pragma unroll
  for (int j = 0; j < NOUT; ++j) { wr[j] = ldg4(w + (size_t)j * K + 4 * tl); acc[j] = zero4(); sb[j] = 0.f; }
  for (long r = (long)blockIdx.x * rpp + tr; r < rows; r += (long)gridDim.x * rpp) {
    const float4 xv = ldg4(x + r * K + 4 * tl);
    float4 d = zero4();
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
      const float g = dy[r * NOUT + j];
      d.x = fmaf(g, wr[j].x, d.x); d.y = fmaf(g, wr[j].y, d.y); d.z = fmaf(g, wr[j].z, d.z); d.w = fmaf(g, wr[j].w, d.w);
      acc[j].x = fmaf(g, xv.x, acc[j].x); acc[j].y = fmaf(g, xv.y, acc[j].y);
      acc[j].z = fmaf(g, xv.z, acc[j].z); acc[j].w = fmaf(g, xv.w, acc[j].w);
      sb[j] += g;
    }
    if (dx != nullptr) *reinterpret_cast<float4*>(dx + r * K + 4 * tl) = d;
  }
#pragma unroll
  for (int j = 0; j < NOUT; ++j) {
    red[threadIdx.x * RS + 4 * j] = acc[j].x; red[threadIdx.x * RS + 4 * j + 1] = acc[j].y;
    red[threadIdx.x * RS + 4 * j + 2] = acc[j].z; red[threadIdx.x * RS + 4 * j + 3] = acc[j].w;
  }
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * (NOUT * K + NOUT);
  for (int e = threadIdx.x; e < NOUT * K; e += 256) {        // e = j*K + k
    const int j = e / K, k = e - j * K, tlk = k >> 2, c = k & 3;
    float s = 0.f;
    for (int g = 0; g < rpp; ++g) s += red[(g * lpr + tlk) * RS + 4 * j + c];
    out[e] = s;
  }
  __syncthreads();
  // bias sums: every lane of a row group holds the same sb[] - take lane 0 of each group
  if (tl == 0) {
#pragma unroll
    for (int j = 0; j < NOUT; ++j) red[tr * NOUT + j] = sb[j];
  }
  __syncthreads();
  if (threadIdx.x < NOUT) {
    float s = 0.f;
    for (int g = 0; g < rpp; ++g) s += red[g * NOUT + threadIdx.x];
    out[NOUT * K + threadIdx.x] = s;
  }
}

// out[e] = sum over blocks of part[block][e], e < n_el (16 elements x 16 block slices per workgroup)
__global__ __launch_bounds__(256) void partial_rows_final_kernel(const float* __restrict__ part, int nblk,
                                                                 int n_el, float* __restrict__ out0, int n0,
                                                                 float* __restrict__ out1) {
  __shared__ float red[256];
  const int el = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  float s = 0.f;
  if (e < n_el)
    for (int b = sl; b < nblk; b += 16) s += part[(size_t)b * n_el + e];
  red[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && e < n_el) {
#pragma unroll
    for (int g = 1; g < 16; ++g) s += red[g * 16 + el];
    if (e < n0) { if (out0 != nullptr) out0[e] = s; }
    else if (out1 != nullptr) out1[e - n0] = s;
  }
}

// diagnostic behind prh_test_xcc_map: where the dispatcher put each workgroup
__global__ __launch_bounds__(512) void xcc_probe_kernel(int* out) {
  extern __shared__ char probe_lds[];
  if (threadIdx.x == 0) {
    probe_lds[0] = 0;
    out[2 * blockIdx.x] = (int)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // XCC_ID[3:0]
    out[2 * blockIdx.x + 1] = (int)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));  // HW_ID
  }
  for (int i = 0; i < 400; ++i) __builtin_amdgcn_s_sleep(64);   // stay resident ~0.1 ms
}

}  // namespace prh
