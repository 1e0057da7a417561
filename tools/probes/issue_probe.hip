// Dev probe: what does one staging instruction cost inside a one-wave-per-SIMD MFMA stream on gfx950?
// 256 workgroups x 256 threads (one wave per SIMD).  Each wave loops: 8 x v_mfma_f32_16x16x32_bf16 (128 MFMA cycles)
// with CNT extra instructions of one kind spread between them; reports shader cycles per iteration (s_memtime).
//   mode 0 none | 1 global_load_lds_dwordx4 (LDS-DMA) | 2 global_load_dwordx4 -> VGPR | 3 ds_write_b128 | 4 ds_read_b128
//   mode 5 buffer_load_dwordx4 -> VGPR | 6 global_load_dwordx4 + ds_write_b128 (register-staged copy)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE, int CNT>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ src, uint64_t* __restrict__ out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * ((lane * 7 + i) % 13)); b[i] = (__bf16)(0.02f * ((lane * 5 + i) % 11)); }
  f32x4 acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const char* g = src + (size_t)blockIdx.x * 262144 + threadIdx.x * 16;
  const uint32_t lds_lane = wave * 1024 + lane * 16;
  i32x4 tmp = {0, 0, 0, 0}, tmp2 = {1, 2, 3, 4};
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const char* gp = g + (it & 31) * 4096;
    const uint32_t lp = (it & 7) * 4096;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // inline asm: with a 512-register budget hipcc's builtin MFMA rotates accumulators between AGPRs and VGPRs
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
      if (j % (8 / CNT) == (8 / CNT) - 1 || CNT == 8) {
        const int e = j / (8 / CNT);
        if constexpr (MODE == 1) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + e * 65536 / 8),
                                           (__attribute__((address_space(3))) void*)(smem + lp + wave * 1024 + e * 32768 / 8), 16, 0, 0);
        } else if constexpr (MODE == 2) {
          asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(tmp) : "v"(gp + e * 8192) : "memory");
        } else if constexpr (MODE == 3) {
          asm volatile("ds_write_b128 %0, %1" ::"v"(lds_lane + lp + e * 32768 / 8), "v"(tmp2) : "memory");
        } else if constexpr (MODE == 4) {
          asm volatile("ds_read_b128 %0, %1" : "+v"(tmp) : "v"(lds_lane + lp + e * 32768 / 8) : "memory");
        } else if constexpr (MODE == 5) {
          asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(tmp) : "v"((uint32_t)(threadIdx.x * 16 + (it & 31) * 4096 + e * 8192)), "s"(src + (size_t)blockIdx.x * 262144) : "memory");
        } else if constexpr (MODE == 6) {
          asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(tmp) : "v"(gp + e * 8192) : "memory");
          asm volatile("ds_write_b128 %0, %1" ::"v"(lds_lane + lp + e * 32768 / 8), "v"(tmp2) : "memory");
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  s += (float)(tmp[0] ^ tmp[1] ^ tmp[2] ^ tmp[3]) * 1e-30f;
  if (s == 123.456f) sink[0] = s + smem[lane];
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE, int CNT>
void run_(const char* name, const char* src, uint64_t* out, float* sink);
template <int MODE, int CNT>
void run(const char* name, const char* src, uint64_t* out, float* sink) { run_<MODE, CNT>(name, src, out, sink); fflush(stdout); }
template <int MODE, int CNT>
void run_(const char* name, const char* src, uint64_t* out, float* sink) {
  const int iters = 4000, nb = 256;
  static uint64_t h[256 * 4];
  auto k = probe<MODE, CNT>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(nb), dim3(256), 65536, 0, src, out, sink, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < nb * 4; ++i) sum += (double)h[i];
  const double cyc = sum / (nb * 4) / iters;
  printf("%-44s x%d per 8 MFMA: %7.1f cyc/iter  (+%6.1f over bare 128; %5.1f per extra)\n", name, CNT, cyc, cyc - 128.0, (cyc - 128.0) / CNT);
}


// GEMM-like mix per iteration (8 MFMA): NREAD ds_read_b128 + staging of STG KiB by (KIND 0) LDS-DMA or (KIND 1)
// global_load_dwordx4 -> VGPR -> ds_write_b128 (written one iteration later).  NW waves per workgroup (4 or 8).
template <int KIND, int STG, int NREAD, int NW>
__global__ __launch_bounds__(NW * 64) void mix(const char* __restrict__ src, uint64_t* __restrict__ out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * ((lane * 7 + i) % 13)); b[i] = (__bf16)(0.02f * ((lane * 5 + i) % 11)); }
  f32x4 acc[8];
  for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const char* g = src + (size_t)blockIdx.x * 262144 + threadIdx.x * 16;
  const uint32_t lds_lane = wave * 1024 + lane * 16;
  i32x4 st[2] = {{1, 2, 3, 4}, {5, 6, 7, 8}};
  i32x4 rd = {0, 0, 0, 0};
  // raw buffer resource over `src`: base, stride 0, num_records = max, gfx9-family dword 3
  i32x4 srd;
  srd[0] = (int)(uintptr_t)src;
  srd[1] = (int)((uintptr_t)src >> 32) & 0xFFFF;
  srd[2] = -1;
  srd[3] = 0x00020000;
  srd[0] = __builtin_amdgcn_readfirstlane(srd[0]);
  srd[1] = __builtin_amdgcn_readfirstlane(srd[1]);
  srd[2] = __builtin_amdgcn_readfirstlane(srd[2]);
  srd[3] = __builtin_amdgcn_readfirstlane(srd[3]);
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const char* gp = g + (it & 15) * 8192;
    const uint32_t lp = (it & 3) * 8192 * (NW / 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
      if (j < NREAD) asm volatile("ds_read_b128 %0, %1" : "+v"(rd) : "v"(lds_lane + 32768 + j * 1024 * NW / 4) : "memory");
      if (STG > 0 && j >= 8 - STG) {
        const int e = j - (8 - STG);
        if constexpr (KIND == 0) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + e * 65536),
                                           (__attribute__((address_space(3))) void*)(smem + lp + wave * 1024 + e * 1024 * NW), 16, 0, 0);
        } else if constexpr (KIND == 2) {          // scalar base + 32-bit VGPR offset, M0 saved / set / restored (gemm.hip, attn.hip)
          unsigned keep;
          const char* sb = src + (size_t)blockIdx.x * 262144 + (it & 15) * 8192 + e * 65536;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"((unsigned)(threadIdx.x * 16)), "s"(sb), "s"(lp + wave * 1024 + e * 1024 * NW) : "memory");
        } else if constexpr (KIND == 3) {          // buffer form: constant voffset, scalar offset per piece, M0 set without save/restore
          asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                       :: "v"((unsigned)(threadIdx.x * 16)), "s"(srd), "s"((unsigned)(blockIdx.x * 262144 + (it & 15) * 8192 + e * 65536)),
                          "s"(lp + wave * 1024 + e * 1024 * NW) : "memory");
        } else {
          asm volatile("ds_write_b128 %0, %1" ::"v"(lds_lane + lp + e * 1024 * NW), "v"(st[e & 1]) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(st[e & 1]) : "v"(gp + e * 65536) : "memory");
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  s += (float)(rd[0] ^ rd[1] ^ st[0][0] ^ st[1][1]) * 1e-30f;
  if (s == 123.456f) sink[0] = s + smem[lane];
  if (lane == 0) out[blockIdx.x * NW + wave] = t1 - t0;
}

// Same mix with the work of 8 x v_mfma_f32_16x16x32_bf16 done by 4 x v_mfma_f32_32x32x16_bf16 (same 128 matrix-pipe cycles, half
// the MFMA issue slots): KIND 2 staging only.
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int STG, int NREAD, int NW>
__global__ __launch_bounds__(NW * 64) void mix32(const char* __restrict__ src, uint64_t* __restrict__ out, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * ((lane * 7 + i) % 13)); b[i] = (__bf16)(0.02f * ((lane * 5 + i) % 11)); }
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const uint32_t lds_lane = wave * 1024 + lane * 16;
  i32x4 rd = {0, 0, 0, 0};
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const uint32_t lp = (it & 3) * 8192 * (NW / 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
#pragma unroll
      for (int r = 2 * j; r < 2 * j + 2; ++r)
        if (r < NREAD) asm volatile("ds_read_b128 %0, %1" : "+v"(rd) : "v"(lds_lane + 32768 + r * 1024 * NW / 4) : "memory");
      if (STG > 0 && j >= 4 - STG) {
        const int e = j - (4 - STG);
        unsigned keep;
        const char* sb = src + (size_t)blockIdx.x * 262144 + (it & 15) * 8192 + e * 65536;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"((unsigned)(threadIdx.x * 16)), "s"(sb), "s"(lp + wave * 1024 + e * 1024 * NW) : "memory");
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][5] + acc[j][10] + acc[j][15];
  s += (float)(rd[0] ^ rd[1]) * 1e-30f;
  if (s == 123.456f) sink[0] = s + smem[lane];
  if (lane == 0) out[blockIdx.x * NW + wave] = t1 - t0;
}

template <int STG, int NREAD, int NW>
void run_mix32(const char* name, const char* src, uint64_t* out, float* sink) {
  const int iters = 4000, nb = 256;
  static uint64_t h[256 * 8];
  auto k = mix32<STG, NREAD, NW>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(nb), dim3(NW * 64), 131072, 0, src, out, sink, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, out, sizeof(uint64_t) * nb * NW, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < nb * NW; ++i) sum += (double)h[i];
  const double cyc = sum / (nb * NW) / iters;
  printf("%d waves/WG  32x32x16: %-26s stage %d KiB + %d ds_read per 4 MFMA: %7.1f cyc/iter/wave -> MFMA pipe use %4.1f %%\n", NW, name, STG, NREAD, cyc,
         100.0 * 128.0 * (NW / 4) / cyc);
  fflush(stdout);
}

template <int KIND, int STG, int NREAD, int NW>
void run_mix(const char* name, const char* src, uint64_t* out, float* sink) {
  const int iters = 4000, nb = 256;
  static uint64_t h[256 * 8];
  auto k = mix<KIND, STG, NREAD, NW>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(nb), dim3(NW * 64), 131072, 0, src, out, sink, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, out, sizeof(uint64_t) * nb * NW, hipMemcpyDeviceToHost);
  double sum = 0;
  for (int i = 0; i < nb * NW; ++i) sum += (double)h[i];
  const double cyc = sum / (nb * NW) / iters;
  printf("%d waves/WG  %-36s stage %d KiB + %d ds_read per 8 MFMA: %7.1f cyc/iter/wave -> MFMA pipe use %4.1f %%\n", NW, name, STG, NREAD, cyc,
         100.0 * 128.0 * (NW / 4) / cyc);
  fflush(stdout);
}

int main() {
  char* src; uint64_t* out; float* sink;
  hipMalloc(&src, (size_t)256 * 262144 + 1048576);
  hipMemset(src, 1, (size_t)256 * 262144 + 1048576);
  hipMalloc(&out, 256 * 8 * 8);
  hipMalloc(&sink, 64);
  run<0, 1>("bare MFMA loop", src, out, sink);
  run<1, 1>("global_load_lds_dwordx4 (LDS-DMA)", src, out, sink);
  run<1, 2>("global_load_lds_dwordx4 (LDS-DMA)", src, out, sink);
  run<1, 4>("global_load_lds_dwordx4 (LDS-DMA)", src, out, sink);
  run<2, 1>("global_load_dwordx4 -> VGPR (64-bit vaddr)", src, out, sink);
  run<2, 2>("global_load_dwordx4 -> VGPR (64-bit vaddr)", src, out, sink);
  run<2, 4>("global_load_dwordx4 -> VGPR (64-bit vaddr)", src, out, sink);
  run<5, 2>("global_load_dwordx4 -> VGPR (saddr + voff)", src, out, sink);
  run<5, 4>("global_load_dwordx4 -> VGPR (saddr + voff)", src, out, sink);
  run<3, 2>("ds_write_b128", src, out, sink);
  run<3, 4>("ds_write_b128", src, out, sink);
  run<4, 2>("ds_read_b128", src, out, sink);
  run<4, 4>("ds_read_b128", src, out, sink);
  run<4, 8>("ds_read_b128", src, out, sink);
  run<6, 2>("global_load_dwordx4 + ds_write_b128", src, out, sink);
  run<6, 4>("global_load_dwordx4 + ds_write_b128", src, out, sink);
  printf("---- GEMM-like mixes (W4 density: 1 KiB staged + 2 reads per 8 MFMA per wave; 8-wave density: 1 KiB + 3 reads)\n");
  run_mix<0, 0, 0, 4>("MFMA only", src, out, sink);
  run_mix<0, 0, 2, 4>("reads only", src, out, sink);
  run_mix<0, 1, 2, 4>("LDS-DMA", src, out, sink);
  run_mix<2, 1, 2, 4>("LDS-DMA asm global form + M0 swap", src, out, sink);
  run_mix<3, 1, 2, 4>("LDS-DMA asm buffer form", src, out, sink);
  run_mix<1, 1, 2, 4>("global_load + ds_write_b128", src, out, sink);
  run_mix<0, 2, 2, 4>("LDS-DMA", src, out, sink);
  run_mix<1, 2, 2, 4>("global_load + ds_write_b128", src, out, sink);
  run_mix<0, 0, 0, 8>("MFMA only", src, out, sink);
  run_mix<0, 0, 3, 8>("reads only", src, out, sink);
  run_mix<0, 1, 3, 8>("LDS-DMA", src, out, sink);
  run_mix<2, 1, 3, 8>("LDS-DMA asm global form + M0 swap", src, out, sink);
  run_mix<3, 1, 3, 8>("LDS-DMA asm buffer form", src, out, sink);
  run_mix<1, 1, 3, 8>("global_load + ds_write_b128", src, out, sink);
  run_mix<0, 2, 3, 8>("LDS-DMA", src, out, sink);
  run_mix<1, 2, 3, 8>("global_load + ds_write_b128", src, out, sink);
  printf("---- the same densities with 32x32x16 MFMAs\n");
  run_mix32<0, 0, 4>("MFMA only", src, out, sink);
  run_mix32<1, 2, 4>("LDS-DMA asm global form", src, out, sink);
  run_mix32<0, 0, 8>("MFMA only", src, out, sink);
  run_mix32<0, 3, 8>("reads only", src, out, sink);
  run_mix32<1, 3, 8>("LDS-DMA asm global form", src, out, sink);
  run_mix32<2, 3, 8>("LDS-DMA asm global form", src, out, sink);
  return 0;
}
