// Probe: does a STRUCTURED buffer load (idxen: address = base + index * stride + offset) reach rows of a table far beyond
// the 4-GB reach of a raw buffer offset, and do index >= num_records / flagged lanes return zeros?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 llvm_struct_buffer_load_v4f32(i32x4 rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.buffer.load.v4f32");

__global__ void k(const float* A, const int* rows, float* out, int n_rows, int stride_bytes, int koff) {
  const uint64_t b = reinterpret_cast<uint64_t>(A);
  i32x4 rs;
  rs[0] = (int)(uint32_t)b;
  rs[1] = (int)((uint32_t)(b >> 32) & 0xffffu) | (stride_bytes << 16);
  rs[2] = n_rows;
  rs[3] = 0x00020000;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int idx = rows[t];
  f32x4 v = llvm_struct_buffer_load_v4f32(rs, idx, 16 * (t & 3), koff, 0);
  reinterpret_cast<f32x4*>(out)[t] = v;
}

int main() {
  const int64_t n_rows = 2000000;  // x 3072 B = 6.1 GB
  const int ld = 768;
  float* A;
  if (hipMalloc(&A, n_rows * ld * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  // fill: A[r][c] = r * 0.5 + c   (computed on the host for the probed rows only)
  hipMemset(A, 0, n_rows * ld * 4);
  const int T = 256;
  std::vector<int> rows(T);
  for (int t = 0; t < T; ++t) rows[t] = (int)((int64_t)t * 7811 % n_rows);
  rows[5] = (int)n_rows - 1; rows[6] = (int)n_rows; rows[7] = 0x40000000 | 17; rows[8] = 1500000;
  std::vector<float> line(ld);
  for (int t = 0; t < T; ++t) {
    if (rows[t] < 0 || rows[t] >= n_rows) continue;
    for (int c = 0; c < ld; ++c) line[c] = rows[t] * 0.5f + c;
    hipMemcpy(A + (int64_t)rows[t] * ld, line.data(), ld * 4, hipMemcpyHostToDevice);
  }
  int* drows; float* dout;
  hipMalloc(&drows, T * 4); hipMalloc(&dout, T * 16);
  hipMemcpy(drows, rows.data(), T * 4, hipMemcpyHostToDevice);
  const int koff = 4 * 100;  // floats 100.. of the row (soffset, bytes)
  hipLaunchKernelGGL(k, dim3(1), dim3(T), 0, 0, A, drows, dout, (int)n_rows, ld * 4, koff);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  std::vector<float> out(T * 4);
  hipMemcpy(out.data(), dout, T * 16, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < T; ++t) {
    const bool oob = rows[t] < 0 || rows[t] >= n_rows;
    for (int e = 0; e < 4; ++e) {
      const float want = oob ? 0.f : rows[t] * 0.5f + (100 + 4 * (t & 3) + e);
      if (out[t * 4 + e] != want) { if (bad < 8) printf("t=%d row=%d e=%d got %g want %g\n", t, rows[t], e, out[t * 4 + e], want); ++bad; }
    }
  }
  printf("struct buffer probe: %d mismatches of %d (rows up to %.1f GB into the table; OOB lanes -> 0)\n", bad, T * 4, n_rows * ld * 4 / 1e9);
  return bad != 0;
}
