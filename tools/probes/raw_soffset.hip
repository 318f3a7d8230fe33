// Probe: is the scalar offset part of a raw buffer load's bounds check on gfx950?  (voffset < num_records but
// voffset + soffset >= num_records: zeros or data?)  The memory behind num_records is allocated, so either answer is safe.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, float* out, int num_records, int soff) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, num_records, 0x00020000);
  const int t = threadIdx.x;
  f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, t * 16, soff, 0));
  reinterpret_cast<f32x4*>(out)[t] = v;
}
int main() {
  const int n = 1 << 16;
  float *A, *dout;
  hipMalloc(&A, n * 4); hipMalloc(&dout, 64 * 16);
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = 1.f + i;
  hipMemcpy(A, h.data(), n * 4, hipMemcpyHostToDevice);
  const int num_records = 4096;       // bytes: floats 0..1023 are in range
  const int soff = 3584;              // lanes 0..31: voffset + soff < 4096 (in range); lanes 32..63: beyond
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, dout, num_records, soff);
  hipDeviceSynchronize();
  std::vector<float> out(256);
  hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
  int in_ok = 0, out_zero = 0, out_data = 0;
  for (int t = 0; t < 64; ++t) {
    const float want = 1.f + (t * 16 + soff) / 4;
    if (t < 32) in_ok += out[t * 4] == want;
    else { out_zero += out[t * 4] == 0.f; out_data += out[t * 4] == want; }
  }
  printf("raw buffer load, soffset in the bounds check: in-range lanes correct %d/32; beyond-range lanes: zeros %d/32, data %d/32\n", in_ok, out_zero, out_data);
  return 0;
}
