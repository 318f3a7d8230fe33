// In-batch InfoNCE "theme" contrastive loss, forward and backward, fused
// (ContrastiveRankingTrainer._compute_contrastive_loss, xnrs/training.py:433-472).
//
//   e_i   = x_i / max(||x_i||, 1e-12)                               (F.normalize, :445)
//   s_ij  = <e_i, e_j> / T
//   rows with no positive (same label, j != i) are skipped           (:463-464)
//   L_i   = -log( sum_{pos} exp(s_ij) / (sum_{j != i} exp(s_ij) + 1e-12) )      (:465-469)
//   loss  = sum_i L_i / (count + 1e-8)                               (:471-472)
//
// The reference walks the rows in a Python loop; here one workgroup owns one row, the B x B similarity
// matrix is never materialised, and the backward recomputes s_ij from the normalised embeddings.
// Everything is summed in a fixed order (no atomics): bitwise reproducible.
#include "kernels.h"

namespace xnrs {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// en[i,:] = x[i,:] / max(||x_i||, eps); inv[i] = 1 / max(||x_i||, eps): one wave per row
__global__ __launch_bounds__(256) void infonce_normalize_kernel(const float* x, float* en, float* inv, int64_t B, int E) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B) return;
  const int lane = threadIdx.x & 63;
  float ss = 0.f;
  for (int k = lane; k < E; k += 64) ss = fmaf(x[i * E + k], x[i * E + k], ss);
  ss = wsum(ss);
  const float r = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  for (int k = lane; k < E; k += 64) en[i * E + k] = x[i * E + k] * r;
  if (lane == 0) inv[i] = r;
}

// one workgroup per row i: num_i, den_i, L_i
__global__ __launch_bounds__(256) void infonce_row_kernel(const float* en, const int64_t* lab, float* num, float* den, float* li,
                                                           int64_t B, int E, float inv_t) {
  __shared__ float s_e[1024];
  __shared__ float s_red[8];
  const int64_t i = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < E; k += 256) s_e[k] = en[i * E + k];
  __syncthreads();
  const int64_t li_lab = lab[i];
  float pn = 0.f, pd = 0.f;
  for (int64_t j = wave; j < B; j += 4) {
    float d = 0.f;
    for (int k = lane; k < E; k += 64) d = fmaf(s_e[k], en[j * E + k], d);
    d = wsum(d);
    if (j != i) {
      const float ex = expf(d * inv_t);
      pd += ex;
      if (lab[j] == li_lab) pn += ex;
    }
  }
  if (lane == 0) {
    s_red[wave] = pn;
    s_red[4 + wave] = pd;
  }
  __syncthreads();
  if (tid == 0) {
    const float n = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    const float d = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    num[i] = n;
    den[i] = d;
    li[i] = n > 0.f ? -logf(n / (d + 1e-12f)) : 0.f;  // n == 0 <=> the row has no positive
  }
}

// loss = sum_i L_i / (count + 1e-8); single workgroup, ordered
__global__ __launch_bounds__(256) void infonce_final_kernel(const float* li, const float* num, float* loss, float* scale, int64_t B) {
  __shared__ float s_l[256];
  __shared__ float s_c[256];
  float l = 0.f, c = 0.f;
  for (int64_t i = threadIdx.x; i < B; i += 256) {
    l += li[i];
    c += num[i] > 0.f ? 1.f : 0.f;
  }
  s_l[threadIdx.x] = l;
  s_c[threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    float L = 0.f, Cn = 0.f;
    for (int t = 0; t < 256; ++t) {
      L += s_l[t];
      Cn += s_c[t];
    }
    loss[0] = L / (Cn + 1e-8f);
    scale[0] = 1.f / (Cn + 1e-8f);
  }
}

// backward, one workgroup per row k:
//   G_kj = w_kj + w_jk,  w_ij = [row i counted] * ( -[pos_ij] ex_ij / num_i + ex_ij / (den_i + eps) ) / T   (j != i)
//   de_k = g * scale * sum_j G_kj e_j ;  dx_k = (de_k - <de_k, e_k> e_k) * inv_k
__global__ __launch_bounds__(256) void infonce_bwd_kernel(const float* en, const float* inv, const int64_t* lab, const float* num,
                                                           const float* den, const float* scale, const float* gout, float* dx,
                                                           int64_t B, int E, float inv_t) {
  __shared__ float s_e[1024];
  __shared__ float s_acc[4][1024];
  __shared__ float s_red[4];
  const int64_t kk = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < E; c += 256) s_e[c] = en[kk * E + c];
  for (int c = lane; c < E; c += 64) s_acc[wave][c] = 0.f;
  __syncthreads();
  const int64_t klab = lab[kk];
  const float nk = num[kk], dk = den[kk] + 1e-12f;
  for (int64_t j = wave; j < B; j += 4) {
    if (j == kk) continue;  // wave-uniform
    float d = 0.f;
    for (int c = lane; c < E; c += 64) d = fmaf(s_e[c], en[j * E + c], d);
    d = wsum(d);
    const float ex = expf(d * inv_t);
    const bool pos = lab[j] == klab;
    float g = 0.f;
    if (nk > 0.f) g += (pos ? -ex / nk : 0.f) + ex / dk;                    // w_kj
    const float nj = num[j];
    if (nj > 0.f) g += (pos ? -ex / nj : 0.f) + ex / (den[j] + 1e-12f);     // w_jk
    g *= inv_t;
    for (int c = lane; c < E; c += 64) s_acc[wave][c] = fmaf(g, en[j * E + c], s_acc[wave][c]);
  }
  __syncthreads();
  const float gs = gout[0] * scale[0];
  float part = 0.f;
  for (int c = tid; c < E; c += 256) {
    const float de = gs * ((s_acc[0][c] + s_acc[1][c]) + (s_acc[2][c] + s_acc[3][c]));
    s_acc[0][c] = de;
    part = fmaf(de, s_e[c], part);
  }
  part = wsum(part);
  if (lane == 0) s_red[wave] = part;
  __syncthreads();
  const float dot = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  const float r = inv[kk];
  for (int c = tid; c < E; c += 256) dx[kk * E + c] = (s_acc[0][c] - dot * s_e[c]) * r;
}

hipError_t launch_infonce_fwd(const float* x, const int64_t* lab, int64_t B, int E, float temperature, float* loss, float* ws,
                              hipStream_t stream) {
  if (E > 1024) return hipErrorInvalidValue;
  float* en = ws;
  float* inv = en + B * E;
  float* num = inv + B;
  float* den = num + B;
  float* li = den + B;
  float* scale = li + B;
  hipLaunchKernelGGL(infonce_normalize_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream, x, en, inv, B, E);
  hipLaunchKernelGGL(infonce_row_kernel, dim3((unsigned)B), dim3(256), 0, stream, en, lab, num, den, li, B, E, 1.f / temperature);
  hipLaunchKernelGGL(infonce_final_kernel, dim3(1), dim3(256), 0, stream, li, num, loss, scale, B);
  return hipGetLastError();
}

hipError_t launch_infonce_bwd(const int64_t* lab, int64_t B, int E, float temperature, const float* ws, const float* gout,
                              float* dx, hipStream_t stream) {
  if (E > 1024) return hipErrorInvalidValue;
  const float* en = ws;
  const float* inv = en + B * E;
  const float* num = inv + B;
  const float* den = num + B;
  const float* scale = den + 2 * B;
  hipLaunchKernelGGL(infonce_bwd_kernel, dim3((unsigned)B), dim3(256), 0, stream, en, inv, lab, num, den, scale, gout, dx, B, E,
                     1.f / temperature);
  return hipGetLastError();
}

}  // namespace xnrs
