// Dev probe: what v_cvt_pk_fp8_f32 (OCP e4m3 on gfx950) returns for values beyond 448.
// build: hipcc --offload-arch=gfx950 -O2 -o cvt_fp8_overflow_probe cvt_fp8_overflow_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const float* in, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(in[i], 0.f, 0, false) & 0xFF;
}
int main() {
  float h[] = {447.f, 448.f, 464.f, 480.f, 500.f, 1000.f, 1e9f, INFINITY, -1000.f, NAN, 1e-3f, 1.953125e-3f /* 2^-9 */, 9.765625e-4f /* 2^-10 */, 0.f};
  const int n = sizeof(h) / sizeof(h[0]);
  float* di; unsigned* dout; unsigned ho[32];
  (void)hipMalloc(&di, sizeof h); (void)hipMalloc(&dout, 128);
  (void)hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, n);
  (void)hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%14g -> 0x%02X\n", h[i], ho[i]);
  return 0;
}
