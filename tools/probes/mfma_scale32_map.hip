// Dev probe: which (lane half, byte) of the B operand of v_mfma_scale_f32_32x32x64_f8f6f4 meets which (lane half, byte) of the A
// operand, and whose scale byte applies.  One-hot operands (e4m3 1.0 = 0x38), unit scales unless stated.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void pair_map(float* tab) {        // tab[(hA*32+jA)*64 + hB*32+jB] = D[0][0]
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  for (int pa = 0; pa < 64; ++pa)
    for (int pb = 0; pb < 64; ++pb) {
      i32x8 a, b;
      for (int w = 0; w < 8; ++w) { a[w] = 0; b[w] = 0; }
      if (r == 0 && h == pa / 32) a[(pa % 32) / 4] = 0x38 << (8 * (pa % 4));
      if (r == 0 && h == pb / 32) b[(pb % 32) / 4] = 0x38 << (8 * (pb % 4));
      f32x16 c;
      for (int e = 0; e < 16; ++e) c[e] = 0.f;
      c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      if (l == 0) tab[pa * 64 + pb] = c[0];
    }
}
// scale test: A one-hot at (row 0, half hA, byte jA), B its partner; scale_a = 2^1 in lane (row 0, half hs) only, 2^0 elsewhere
__global__ void scale_map(float* out, const int* partner) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  for (int pa = 0; pa < 64; ++pa)
    for (int hs = 0; hs < 2; ++hs)
      for (int side = 0; side < 2; ++side) {
        const int pb = partner[pa];
        i32x8 a, b;
        for (int w = 0; w < 8; ++w) { a[w] = 0; b[w] = 0; }
        if (r == 0 && h == pa / 32) a[(pa % 32) / 4] = 0x38 << (8 * (pa % 4));
        if (r == 0 && h == pb / 32) b[(pb % 32) / 4] = 0x38 << (8 * (pb % 4));
        const int s = (r == 0 && h == hs) ? 0x80 : 0x7F;
        f32x16 c;
        for (int e = 0; e < 16; ++e) c[e] = 0.f;
        c = side == 0 ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, s, 0, 0x7F)
                      : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7F, 0, s);
        if (l == 0) out[(pa * 2 + hs) * 2 + side] = c[0];
      }
}

int main() {
  float* tab; hipMalloc(&tab, 64 * 64 * 4);
  hipLaunchKernelGGL(pair_map, dim3(1), dim3(64), 0, 0, tab);
  static float h[64 * 64];
  hipMemcpy(h, tab, sizeof h, hipMemcpyDeviceToHost);
  int partner[64];
  printf("A (half, byte) -> B (half, byte) partners:\n");
  for (int pa = 0; pa < 64; ++pa) {
    int n = 0, pbb = -1;
    for (int pb = 0; pb < 64; ++pb) if (h[pa * 64 + pb] != 0.f) { ++n; pbb = pb; }
    partner[pa] = pbb;
    printf("  A(%d,%2d) -> B(%d,%2d)%s%s", pa / 32, pa % 32, pbb / 32, pbb % 32, n == 1 ? "" : " [not unique]", pa % 4 == 3 ? "\n" : "");
  }
  int* dp; hipMalloc(&dp, 256); hipMemcpy(dp, partner, 256, hipMemcpyHostToDevice);
  float* so; hipMalloc(&so, 64 * 4 * 4);
  hipLaunchKernelGGL(scale_map, dim3(1), dim3(64), 0, 0, so, dp);
  static float hs[256];
  hipMemcpy(hs, so, sizeof hs, hipMemcpyDeviceToHost);
  printf("D00 with scale 2 in the lane of (row 0, half hs) [A side hs=0, B side hs=0, A side hs=1, B side hs=1]:\n");
  for (int pa = 0; pa < 64; ++pa)
    printf("  A(%d,%2d): %g %g %g %g%s", pa / 32, pa % 32, hs[(pa * 2 + 0) * 2 + 0], hs[(pa * 2 + 0) * 2 + 1], hs[(pa * 2 + 1) * 2 + 0], hs[(pa * 2 + 1) * 2 + 1], pa % 4 == 3 ? "\n" : "");
  return 0;
}
