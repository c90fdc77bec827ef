// Per-line context builder on the GPU: for every polyline of a scene,
// crop the point cloud to a tube around the line, weight the candidates by distance and
// intensity, draw N of them and centre the block on the line - the hot loop of the reference's
// whole-scene inference and of every dataset item:
//   crop     : distance to the polyline densified to 200 points < radius
//              (src/dataset.py:210-222, inference_whole_scene.py:104-110; KDTree there, brute
//               force here: L x P x 200 distance evaluations are ~1e11 fp32 ops per scene)
//   weights  : exp(-d/decay) * (0.5 + (I-Imin)/(Imax-Imin+1e-6)), d to the nearest of the line's
//              32 points, 0.5 for a constant intensity (src/dataset.py:93-111)
//   sampling : K = 0 -> zeros; K <= N -> N uniform draws with replacement; K > N -> N draws
//              without replacement proportional to the weights (uniform if they sum to < 1e-6)
//              (src/dataset.py:86-91,115-130).  numpy.random.choice is replaced by counter-based
//              hashing of (seed, line, point): Gumbel-top-k, i.e. the same Plackett-Luce law as
//              sequential weighted draws without replacement - parity is distributional.
//   centring : xyz minus the mean of the line's points, raw intensity (src/dataset.py:229-234).
// Everything is deterministic for a given seed: candidates are compacted in cloud order by a
// two-pass count / scan / fill (no atomics), the top-N keys by an exact radix select, ties in
// cloud order.  HBM-light, VALU-bound: one thread per (point, line) in the crop passes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace prh {

constexpr int CTX_MAX_DENSE = 256;    // polyline samples held in LDS by the crop kernels
constexpr int CTX_MAX_LINE = 64;

__device__ __forceinline__ uint64_t ctx_mix(uint64_t x) {      // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ uint64_t ctx_hash(uint64_t seed, unsigned line, unsigned idx) {
  return ctx_mix(ctx_mix(seed ^ ((uint64_t)line << 32 | idx)));
}
// uniform in (0,1), 24 bits
__device__ __forceinline__ float ctx_u01(uint64_t h) { return ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f); }
// order-preserving map float -> uint32
__device__ __forceinline__ unsigned ctx_sortable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float ctx_min_d2(float x, float y, float z, const float* pts, int n) {
  float best = 3.0e38f;
  for (int j = 0; j < n; ++j) {
    const float dx = x - pts[3 * j], dy = y - pts[3 * j + 1], dz = z - pts[3 * j + 2];
    best = fminf(best, fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
  }
  return best;
}

// pass 0: bounding box of each line's tube (polyline samples +- radius)
__global__ __launch_bounds__(64) void ctx_bbox_kernel(const float* __restrict__ dense, int nd, float radius,
                                                      float* __restrict__ box) {
  const int line = blockIdx.x, lane = threadIdx.x;
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int j = lane; j < nd; j += 64)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = dense[((size_t)line * nd + j) * 3 + c];
      lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v);
    }
#pragma unroll
  for (int c = 0; c < 3; ++c)
    for (int o = 32; o > 0; o >>= 1) {
      lo[c] = fminf(lo[c], __shfl_xor(lo[c], o));
      hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o));
    }
  const float r = radius * 1.0001f + 1e-6f;
  if (lane < 3) { box[line * 6 + lane] = lo[lane] - r; box[line * 6 + 3 + lane] = hi[lane] + r; }
}

// pass 1 (FILL = false): per (256-point block, line) number of points inside the tube;
// pass 3 (FILL = true): the same test again, points written to cand[line][offset + rank]
template <bool FILL>
__global__ __launch_bounds__(256) void ctx_crop_kernel(const float* __restrict__ cloud, int npts,
                                                       const float* __restrict__ dense, int nd,
                                                       const float* __restrict__ box, float r2, int nblk,
                                                       int* __restrict__ blkcnt,
                                                       const int* __restrict__ blkoff,
                                                       int* __restrict__ cand, int max_cand) {
  __shared__ float pts[3 * CTX_MAX_DENSE];
  __shared__ int wcnt[4];
  const int line = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int p = blk * 256 + tid;
  // the tube's bounding box first: most (point, line) pairs - and most whole blocks - end here
  const float* bx = box + line * 6;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  bool near = false;
  if (p < npts) {
    v = *reinterpret_cast<const float4*>(cloud + (size_t)p * 4);
    near = v.x >= bx[0] && v.x <= bx[3] && v.y >= bx[1] && v.y <= bx[4] && v.z >= bx[2] && v.z <= bx[5];
  }
  if (!__syncthreads_or(near)) {
    if (!FILL && tid == 0) blkcnt[(size_t)line * nblk + blk] = 0;
    return;
  }
  for (int i = tid; i < 3 * nd; i += 256) pts[i] = dense[(size_t)line * nd * 3 + i];
  __syncthreads();
  bool in = false;
  if (near) in = ctx_min_d2(v.x, v.y, v.z, pts, nd) < r2;
  const unsigned long long bal = __ballot(in);
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 0) wcnt[wave] = __popcll(bal);
  __syncthreads();
  if (!FILL) {
    if (tid == 0) blkcnt[(size_t)line * nblk + blk] = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    return;
  }
  if (in) {
    int pos = blkoff[(size_t)line * nblk + blk] + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += wcnt[w];
    if (pos < max_cand) cand[(size_t)line * max_cand + pos] = p;
  }
}

// pass 2: exclusive scan of a line's block counts; counts[line] = total inside the tube
__global__ __launch_bounds__(256) void ctx_scan_kernel(const int* __restrict__ blkcnt, int nblk,
                                                       int* __restrict__ blkoff, int* __restrict__ counts) {
  __shared__ int part[256];
  const int line = blockIdx.x, tid = threadIdx.x;
  const int per = (nblk + 255) / 256;
  const int b0 = tid * per, b1 = min(b0 + per, nblk);
  int s = 0;
  for (int b = b0; b < b1; ++b) s += blkcnt[(size_t)line * nblk + b];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { const int t = part[i]; part[i] = run; run += t; }
    counts[line] = run;
  }
  __syncthreads();
  int run = part[tid];
  for (int b = b0; b < b1; ++b) {
    blkoff[(size_t)line * nblk + b] = run;
    run += blkcnt[(size_t)line * nblk + b];
  }
}

// pass 4: one workgroup per line - weights, keys, exact top-N, gather + centre
__global__ __launch_bounds__(256) void ctx_select_kernel(
    const float* __restrict__ cloud, const float* __restrict__ linepts, int m, const int* __restrict__ counts,
    const int* __restrict__ cand, int max_cand, float decay, int N, uint64_t seed, unsigned* __restrict__ keys,
    float* __restrict__ out, float* __restrict__ dbg_w) {
  __shared__ float lp[3 * CTX_MAX_LINE];
  __shared__ float red[256];
  __shared__ float red2[256];
  __shared__ int hist[256];
  __shared__ int sh_i[4];
  __shared__ float sh_f[4];
  const int line = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < 3 * m; i += 256) lp[i] = linepts[(size_t)line * m * 3 + i];
  __syncthreads();
  if (tid < 3) {
    float s = 0.f;
    for (int j = 0; j < m; ++j) s += lp[3 * j + tid];
    sh_f[tid] = s / (float)m;
  }
  __syncthreads();
  const float cx = sh_f[0], cy = sh_f[1], cz = sh_f[2];
  const int K = min(counts[line], max_cand);
  const int* cl = cand + (size_t)line * max_cand;
  float* o = out + (size_t)line * N * 4;
  if (K <= N) {        // zeros, or uniform draws with replacement
    for (int i = tid; i < N; i += 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (K > 0) {
        const int j = (int)(ctx_hash(seed, (unsigned)line, 0x40000000u + (unsigned)i) % (uint64_t)K);
        v = *reinterpret_cast<const float4*>(cloud + (size_t)cl[j] * 4);
      }
      *reinterpret_cast<float4*>(o + (size_t)i * 4) = make_float4(v.x - cx, v.y - cy, v.z - cz, v.w);
    }
    return;
  }
  // intensity range of the candidates
  float lo = 3.0e38f, hi = -3.0e38f;
  for (int j = tid; j < K; j += 256) {
    const float I = cloud[(size_t)cl[j] * 4 + 3];
    lo = fminf(lo, I); hi = fmaxf(hi, I);
  }
  red[tid] = lo; red2[tid] = hi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) { red[tid] = fminf(red[tid], red[tid + s]); red2[tid] = fmaxf(red2[tid], red2[tid + s]); }
    __syncthreads();
  }
  lo = red[0]; hi = red2[0];
  __syncthreads();
  const bool flat = !(hi > lo);
  const float inv = 1.f / (hi - lo + 1e-6f);
  // weights (kept in the key buffer as floats for now) and their sum
  float* wbuf = reinterpret_cast<float*>(keys + (size_t)line * max_cand);
  float wsum = 0.f;
  for (int j = tid; j < K; j += 256) {
    const float4 v = *reinterpret_cast<const float4*>(cloud + (size_t)cl[j] * 4);
    const float d = sqrtf(ctx_min_d2(v.x, v.y, v.z, lp, m));
    const float w = expf(-d / decay) * (0.5f + (flat ? 0.5f : (v.w - lo) * inv));
    wbuf[j] = w;
    if (dbg_w != nullptr) dbg_w[(size_t)line * max_cand + j] = w;
    wsum += w;
  }
  red[tid] = wsum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const bool uniform = red[0] < 1e-6f;
  __syncthreads();
  // Gumbel-top-k keys: log w + G,  G = -log(-log u)   (uniform weights: G alone)
  unsigned* kb = keys + (size_t)line * max_cand;
  for (int j = tid; j < K; j += 256) {
    const float u = ctx_u01(ctx_hash(seed, (unsigned)line, (unsigned)cl[j]));
    const float g = -logf(-logf(u));
    const float w = wbuf[j];
    kb[j] = ctx_sortable(uniform ? g : (w > 0.f ? logf(w) + g : -3.0e38f));
  }
  __syncthreads();
  // exact radix select of the N-th largest key: 4 passes of 8 bits, most significant first
  unsigned prefix = 0, pmask = 0;
  int need = N;                        // how many keys still to take from the current bucket
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    for (int j = tid; j < K; j += 256) {
      const unsigned k = kb[j];
      if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int b = 255, acc = 0;
      while (b > 0 && acc + hist[b] < need) { acc += hist[b]; --b; }
      sh_i[0] = b; sh_i[1] = need - acc;
    }
    __syncthreads();
    prefix |= (unsigned)sh_i[0] << shift;
    pmask |= 255u << shift;
    need = sh_i[1];
    __syncthreads();
  }
  const unsigned T = prefix;           // N-th largest key; `need` keys equal to T are taken
  // ordered compaction in candidate (= cloud) order
  int base = 0, eq_seen = 0;
  for (int j0 = 0; j0 < K; j0 += 256) {
    const int j = j0 + tid;
    const unsigned k = j < K ? kb[j] : 0u;
    const bool gt = j < K && k > T, eq = j < K && k == T;
    // rank of this thread among the block's gt / eq flags
    const unsigned long long bg = __ballot(gt), be = __ballot(eq);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) { hist[wave] = __popcll(bg); hist[4 + wave] = __popcll(be); }
    __syncthreads();
    int eq_before = eq_seen + __popcll(be & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) eq_before += hist[4 + w];
    const bool take = gt || (eq && eq_before < need);
    const int eq_tot = hist[4] + hist[5] + hist[6] + hist[7];
    __syncthreads();
    const unsigned long long bt = __ballot(take);
    if (lane == 0) hist[8 + wave] = __popcll(bt);
    __syncthreads();
    if (take) {
      int pos = base + __popcll(bt & ((1ull << lane) - 1ull));
      for (int w = 0; w < wave; ++w) pos += hist[8 + w];
      const float4 v = *reinterpret_cast<const float4*>(cloud + (size_t)cl[j] * 4);
      if (pos < N) *reinterpret_cast<float4*>(o + (size_t)pos * 4) = make_float4(v.x - cx, v.y - cy, v.z - cz, v.w);
    }
    base += hist[8] + hist[9] + hist[10] + hist[11];
    eq_seen += eq_tot;
    __syncthreads();
  }
}

}  // namespace prh
