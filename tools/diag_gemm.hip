// Diagnostic build of the fp32 GEMM (segment cycle stamps, XNRS_GEMM_DIAG): NOT part of the product.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DXNRS_GEMM_DIAG -I xnrs_amd/csrc tools/diag_gemm.hip -o gpurun_bin/diag_gemm
#include <cstdio>
#include <vector>
#include <algorithm>

#include "../xnrs_amd/csrc/gemm_f32.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int64_t M = 65500; const int N = 2304, K = 768;
  float *A, *W, *C; unsigned long long* diag;
  CK(hipMalloc(&A, M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&C, M * (size_t)N * 4));
  std::vector<float> h(M * K);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  CK(hipMemcpy(A, h.data(), M * K * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
  const int64_t wgs = ((M + 127) / 128) * ((N + 127) / 128);
  CK(hipMalloc(&diag, wgs * 4 * 5 * 8));
  CK(hipMemset(diag, 0, wgs * 4 * 5 * 8));
  xnrs::GemmArgs g{};
  g.A = A; g.lda = K; g.W[0] = W; g.nseg = 1; g.Nseg = N; g.ldw = K; g.C = C; g.ldc = N; g.M = M; g.K = K;
  g.diag = diag;
  setenv("XNRS_GEMM_PIPE", argc > 1 ? argv[1] : "3", 1);
  setenv("XNRS_GEMM_BK", argc > 2 ? argv[2] : "32", 1);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) CK(xnrs::launch_gemm_f32(g, 0));
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 10; ++i) CK(xnrs::launch_gemm_f32(g, 0));
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("diag build: %.1f us/launch = %.1f TF (stamps perturb the kernel)\n", ms * 100, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
  std::vector<unsigned long long> d(wgs * 4 * 5);
  CK(hipMemcpy(d.data(), diag, d.size() * 8, hipMemcpyDeviceToHost));
  const char* names[5] = {"S0 frag+16mfma", "S1 barrier", "S2 vmcnt+ds_write", "S3 gload issue", "S4 48mfma+frags"};
  const int nk = K / 32;
  double tot = 0;
  for (int s = 0; s < 5; ++s) {
    std::vector<double> v;
    for (int64_t w = 0; w < wgs * 4; ++w) v.push_back((double)d[w * 5 + s] / nk);
    std::sort(v.begin(), v.end());
    double mean = 0; for (double x : v) mean += x; mean /= v.size();
    tot += mean;
    printf("%-20s per-iteration cycles: mean %8.0f  p10 %8.0f  median %8.0f  p90 %8.0f\n", names[s], mean, v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
  }
  printf("sum of means %.0f cycles per iteration (64 MFMAs = 4096 cycles of matrix pipe)\n", tot);
  return 0;
}
