// Dev probe: what does ds_read_b64_tr_b16 deliver?  LDS holds lds[j] = j (u16); lane L reads 8 bytes at
// byte address 8*L, i.e. loaded[L][e] = 4L + e.  Prints result[lane][0..3].
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
  __syncthreads();
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * threadIdx.x));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = r[e];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  return 0;
}
