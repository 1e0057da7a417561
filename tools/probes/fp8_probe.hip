#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// A [16][128] e4m3 bytes (row-major, K contiguous), B [16][128] likewise (B^T rows): C[i][j] = sum_k A[i][k] B[j][k]
__global__ void probe16(const uint8_t* A, const uint8_t* B, float* C) {
  const int lane = threadIdx.x;
  i32x8 a = *(const i32x8*)(A + (lane & 15) * 128 + (lane >> 4) * 32);
  i32x8 b = *(const i32x8*)(B + (lane & 15) * 128 + (lane >> 4) * 32);
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int i = 0; i < 4; ++i) C[(4 * (lane >> 4) + i) * 16 + (lane & 15)] = c[i];   // row = 4*(lane>>4)+i (A row), col = lane&15 (B row)
}
__global__ void probe32(const uint8_t* A, const uint8_t* B, float* C) {   // 32x32x64: lane l: row l%32, k = 32*(l/32) + 0..31
  const int lane = threadIdx.x;
  i32x8 a = *(const i32x8*)(A + (lane & 31) * 64 + (lane >> 5) * 32);
  i32x8 b = *(const i32x8*)(B + (lane & 31) * 64 + (lane >> 5) * 32);
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = c[i];
}
static uint8_t e4m3(int v) {  // small ints -4..4 exactly
  static const uint8_t pos[5] = {0x00, 0x38, 0x40, 0x44, 0x48};   // 0, 1, 2, 3, 4
  return v >= 0 ? pos[v] : (uint8_t)(0x80 | pos[-v]);
}
int main() {
  uint8_t hA[32 * 128], hB[32 * 128]; int iA[32 * 128], iB[32 * 128];
  unsigned s = 12345;
  for (int i = 0; i < 32 * 128; ++i) { s = s * 1664525u + 1013904223u; iA[i] = (int)((s >> 16) % 9) - 4; s = s * 1664525u + 1013904223u; iB[i] = (int)((s >> 16) % 9) - 4; hA[i] = e4m3(iA[i]); hB[i] = e4m3(iB[i]); }
  uint8_t *dA, *dB; float* dC; float hC[32 * 32];
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, 16 * 16 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int r = 0; for (int k = 0; k < 128; ++k) r += iA[i * 128 + k] * iB[j * 128 + k]; if ((float)r != hC[i * 16 + j]) ++bad; }
  printf("16x16x128 fp8: %d / 256 wrong (C[0][0..3] = %g %g %g %g)\n", bad, hC[0], hC[1], hC[2], hC[3]);
  hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, 32 * 32 * 4, hipMemcpyDeviceToHost);
  bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int r = 0; for (int k = 0; k < 64; ++k) r += iA[i * 64 + k] * iB[j * 64 + k]; if ((float)r != hC[i * 32 + j]) ++bad; }
  printf("32x32x64 fp8: %d / 1024 wrong\n", bad);
  return 0;
}
