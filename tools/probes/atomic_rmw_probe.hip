// Dev probe: how fast does the chip add a fp32 value into a 286 MB array (the o-projection's X, 23296 x 3072 fp32)
//   mode 0: load f32x4 + add + store f32x4 (what the gate-residual epilogue does today)
//   mode 1: no-return global_atomic_add_f32, lanes on consecutive dwords (256 contiguous bytes per instruction)
//   mode 2: no-return global_atomic_add_f32, lane stride 16 bytes, 4 instructions per 1 KiB (the f32x4 footprint)
//   mode 3: plain f32x4 store (write only)
//   mode 4: no-return global_atomic_pk_add_bf16 (for scale: 2 values per dword)
// build: hipcc --offload-arch=gfx950 -O3 -o atomic_rmw_probe atomic_rmw_probe.hip ; run: ./atomic_rmw_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* x, long n4, float v) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    if constexpr (MODE == 0) {
      f32x4 a = *(f32x4*)(x + 4 * i);
      a += v;
      *(f32x4*)(x + 4 * i) = a;
    } else if constexpr (MODE == 1) {
      // 4 instructions, each: wave covers 256 contiguous bytes
      const long w = i / 64, l = i % 64;
      float* b = x + w * 256 + l;
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(b + 64 * j), "v"(v) : "memory");
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(x + 4 * i + j), "v"(v) : "memory");
    } else if constexpr (MODE == 3) {
      *(f32x4*)(x + 4 * i) = (f32x4){v, v, v, v};
    } else {
      const long w = i / 64, l = i % 64;
      float* b = x + w * 256 + l;
      const unsigned pk = 0x3c003c00u;
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("global_atomic_pk_add_bf16 %0, %1, off" ::"v"(b + 64 * j), "v"(pk) : "memory");
    }
  }
}

template <int MODE>
void run(float* x, long n, const char* name, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, x, n / 4, 1.0f);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, x, n / 4, 1.0f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-58s grid %5d: %7.1f us per pass, %.2f TB/s of array bytes\n", name, grid, ms / 5 * 1e3, n * 4.0 / (ms / 5 * 1e-3) / 1e12);
}

int main() {
  const long n = 23296L * 3072;
  float* x; hipMalloc(&x, n * 4); hipMemset(x, 0, n * 4);
  for (int grid : {2048, 8192}) {
    run<0>(x, n, "load f32x4 + add + store f32x4", grid);
    run<1>(x, n, "atomic_add_f32 no return, 256 contiguous bytes / instr", grid);
    run<2>(x, n, "atomic_add_f32 no return, 16-byte lane stride", grid);
    run<3>(x, n, "store f32x4 only", grid);
    run<4>(x, n, "atomic_pk_add_bf16 no return, 256 contiguous bytes", grid);
  }
  float h[4]; hipMemcpy(h, x, 16, hipMemcpyDeviceToHost); printf("x[0..3] = %g %g %g %g\n", h[0], h[1], h[2], h[3]);
  return 0;
}
