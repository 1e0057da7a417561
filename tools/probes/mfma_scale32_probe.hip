// Dev probe: operand maps of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands, by exact integer data.
// Found (tools/probes/mfma_scale32_map.hip): lane (r = l & 31, h = l >> 5) holds A[row r][k] and B[k][col r] for k = 32 (j / 16) + 16 h + j % 16
// in byte j = 0..31 of its 8 VGPRs -- 16 bytes of each of the two 32-deep MX blocks -- and the E8M0 byte selected by op_sel from the
// scale VGPR of lane (r, h) scales MX block h (k = 32 h .. 32 h + 31) of row / column r; C/D as every 32x32 form.
// The probe runs D = A.B with per-(row, k block) power-of-two scales and random small integers, and compares with the host.
// build: hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o mfma_scale32_probe mfma_scale32_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ inline unsigned char to_e4m3(int v) {   // exact for |v| <= 16 integers
  if (v == 0) return 0;
  unsigned char s = v < 0 ? 0x80 : 0;
  int a = v < 0 ? -v : v;
  int e = 0;
  while ((a >> (e + 1)) != 0) ++e;        // floor(log2 a)
  int mant = ((a << 3) >> e) & 7;         // 3 mantissa bits (exact when a < 16 or a == 16)
  return s | (unsigned char)(((e + 7) << 3) | mant);
}

template <int OPA, int OPB>
__global__ void k(const unsigned char* A, const unsigned char* B, const unsigned* sa, const unsigned* sb, float* D) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  i32x8 a, b;
  for (int w = 0; w < 8; ++w) {
    const int k0 = 32 * (w / 4) + 16 * h + 4 * (w % 4);      // H2: byte j of lane half h is k = 32 (j / 16) + 16 h + j % 16
    a[w] = *(const int*)(A + r * 64 + k0);
    unsigned x = 0;
    for (int j = 0; j < 4; ++j) x |= (unsigned)B[(k0 + j) * 32 + r] << (8 * j);   // B[k][r]
    b[w] = (int)x;
  }
  f32x16 c;
  for (int e = 0; e < 16; ++e) c[e] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, OPA, (int)sa[l], OPB, (int)sb[l]);
  for (int e = 0; e < 16; ++e) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = c[e];
}

int main() {
  unsigned char hA[32 * 64], hB[64 * 32];
  int iA[32 * 64], iB[64 * 32], eA[32][2], eB[32][2];
  srand(5);
  for (int i = 0; i < 32 * 64; ++i) { iA[i] = rand() % 17 - 8; hA[i] = to_e4m3(iA[i]); iB[i] = rand() % 13 - 6; hB[i] = to_e4m3(iB[i]); }
  for (int r = 0; r < 32; ++r) for (int h = 0; h < 2; ++h) { eA[r][h] = rand() % 5 - 2; eB[r][h] = rand() % 7 - 3; }
  unsigned char *dA, *dB; unsigned *dsa, *dsb; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  float ref[32][32];
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double s = 0;
    for (int kk = 0; kk < 64; ++kk) s += (double)iA[i * 64 + kk] * ldexp(1.0, eA[i][kk / 32]) * iB[kk * 32 + j] * ldexp(1.0, eB[j][kk / 32]);
    ref[i][j] = (float)s;
  }
  for (int sel = 0; sel < 4; ++sel) {
    unsigned hsa[64], hsb[64];
    for (int l = 0; l < 64; ++l) {
      const int r = l & 31, h = l >> 5;
      // the selected byte carries the scale, every other byte a wrong one
      hsa[l] = 0x31313131u; hsb[l] = 0x99999999u;
      hsa[l] = (hsa[l] & ~(0xFFu << (8 * sel))) | ((unsigned)(eA[r][h] + 127) << (8 * sel));
      hsb[l] = (hsb[l] & ~(0xFFu << (8 * ((sel + 1) & 3)))) | ((unsigned)(eB[r][h] + 127) << (8 * ((sel + 1) & 3)));
    }
    hipMemcpy(dsa, hsa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
    switch (sel) {
      case 0: hipLaunchKernelGGL((k<0, 1>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD); break;
      case 1: hipLaunchKernelGGL((k<1, 2>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD); break;
      case 2: hipLaunchKernelGGL((k<2, 3>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD); break;
      default: hipLaunchKernelGGL((k<3, 0>), dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD); break;
    }
    float hD[32 * 32];
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double md = 0; int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double d = fabs(hD[i * 32 + j] - ref[i][j]); if (d > md) md = d; if (d != 0) ++bad; }
    printf("op_sel A=%d B=%d: max |D - ref| = %g, %d of 1024 differ   (D[0][0..3] = %g %g %g %g, ref %g %g %g %g)\n", sel, (sel + 1) & 3, md, bad,
           hD[0], hD[1], hD[2], hD[3], ref[0][0], ref[0][1], ref[0][2], ref[0][3]);
  }
  return 0;
}
