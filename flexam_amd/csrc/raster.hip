// flexam_amd/csrc/raster.hip -- the conditioning rasteriser: tracked 3-D points -> conditioning video frames
// (/root/reference/pipelines.py: fun_visualize_tracking_with_depth :1501-1575, _render_cosine_encoded_frame :1694-1728,
// _visualize_depth_tracking :1763-1820; helpers valid_mask :1200-1212, sort_points_by_depth :1214-1232, draw_rectangle :1234-1253,
// _should_draw_point :1842-1850).
//
// The reference draws, per frame and in Python, every visible point as a filled square with PIL in order of descending depth, so
// that nearer points overwrite farther ones: 125 s per 97-frame clip with a 4-pixel grid of points (28672 per frame), nine times
// the 50 denoising steps.  Drawing order only decides WHICH point a pixel ends up showing -- the one with the smallest depth among
// the squares covering it -- so the frame is a per-pixel minimum:
//   flexam_raster_keys     one thread per (frame, point): 64-bit key (depth as an order-preserving integer << 32 | point index),
//                          atomicMin into every pixel of its square.  Byte / integer work, bound by L2 atomics.
//   flexam_raster_resolve  one thread per pixel: key -> point index -> its colour (per-video table), written as bytes [T, H, W, 3]
//                          and / or as the float [3, T, H, W] planes (value / 255) the VAE encode takes.  HBM-bound.
// One key image serves every video that selects points the same way (the four cosine levels and the depth video differ in colours only).
// Equal depths: lower point index wins (the reference's order among equal depths is numpy's unstable argsort: not reproducible).
#include "common.h"
#include "flexam_hip.h"

namespace {

constexpr unsigned long long RASTER_EMPTY = ~0ull;

// ascending depth -> ascending code; NaN above everything (numpy sorts NaN last: drawn first, overwritten by all); -0 = +0
__device__ __forceinline__ unsigned depth_code(float z) {
  if (z != z) return 0xFFFFFFFFu;
  if (z == 0.f) z = 0.f;
  const unsigned b = __float_as_uint(z);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(256) void raster_keys_kernel(const float* __restrict__ pts, const unsigned char* __restrict__ vis, int64_t total,
                                                          int N, int H, int W, int half, int y_min, const float* __restrict__ mask,
                                                          unsigned long long* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  if (vis && !vis[i]) return;
  const float u = pts[3 * i], v = pts[3 * i + 1], z = pts[3 * i + 2];
  // finite, then astype(int) = truncation towards zero (pipelines.py:1556-1560): trunc(u) in [0, W)  <=>  -1 < u < W
  if (!(u > -1.f && u < (float)W && v > -1.f && v < (float)H)) return;          // also rejects NaN / inf
  const int x = (int)u, y = (int)v;
  if (y < y_min) return;                                                          // the tracking video's frame test is y > 0 (pipelines.py:1211)
  const int64_t t = i / N;
  const unsigned n = (unsigned)(i - t * N);
  unsigned long long* frame = keys + t * H * W;
  if (mask && !(mask[(t * H + y) * W + x] > 0.5f)) return;                      // _should_draw_point: the mask under the square's centre
  const unsigned long long key = ((unsigned long long)depth_code(z) << 32) | n;
  const int y0 = max(y - half, 0), y1 = min(y + half, H - 1), x0 = max(x - half, 0), x1 = min(x + half, W - 1);
  for (int yy = y0; yy <= y1; ++yy)
    for (int xx = x0; xx <= x1; ++xx) atomicMin(frame + (int64_t)yy * W + xx, key);
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(const unsigned long long* __restrict__ keys, const unsigned char* __restrict__ colors,
                                                             int64_t color_frame_stride, unsigned n_colors, int64_t frame_px, int64_t total,
                                                             unsigned char* __restrict__ out_u8, float* __restrict__ out_f32) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const unsigned long long key = keys[i];
  unsigned char r = 0, g = 0, b = 0;
  if (key != RASTER_EMPTY && (unsigned)(key & 0xFFFFFFFFull) < n_colors) {      // an index past the table (keys of another point set) stays black
    const unsigned char* c = colors + (i / frame_px) * color_frame_stride + 3 * (int64_t)(unsigned)(key & 0xFFFFFFFFull);
    r = c[0]; g = c[1]; b = c[2];
  }
  if (out_u8) {
    out_u8[3 * i] = r; out_u8[3 * i + 1] = g; out_u8[3 * i + 2] = b;
  }
  if (out_f32) {                                 // torch: uint8 -> float, / 255.0 (pipelines.py:1660): the correctly rounded quotient
    out_f32[i] = __fdiv_rn((float)r, 255.0f);
    out_f32[total + i] = __fdiv_rn((float)g, 255.0f);
    out_f32[2 * total + i] = __fdiv_rn((float)b, 255.0f);
  }
}

}  // namespace

extern "C" int flexam_raster_keys(const float* points, const unsigned char* visible, int T, int N, int H, int W, int half, int y_min,
                                  const float* mask, unsigned long long* keys, void* stream) {
  FX_REQUIRE(points && keys, FLEXAM_E_ARG, "raster_keys: null pointer");
  FX_REQUIRE(T > 0 && N > 0 && H > 0 && W > 0 && half >= 0 && (y_min == 0 || y_min == 1), FLEXAM_E_SHAPE,
             "raster_keys: T=%d N=%d H=%d W=%d half=%d y_min=%d", T, N, H, W, half, y_min);
  FX_REQUIRE(W < (1 << 24) && H < (1 << 24), FLEXAM_E_SHAPE, "raster_keys: frame sides must be exact in float32");
  const int64_t total = (int64_t)T * N;
  if (hipMemsetAsync(keys, 0xFF, (size_t)T * H * W * sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess)
    return flexam_fail(FLEXAM_E_LAUNCH, "raster_keys: clearing the key image failed");
  hipLaunchKernelGGL(raster_keys_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points, visible, total, N, H, W,
                     half, y_min, mask, keys);
  return flexam_check_launch("flexam_raster_keys");
}

extern "C" int flexam_raster_resolve(const unsigned long long* keys, const unsigned char* colors, int64_t color_frame_stride, int N, int T, int H, int W,
                                     unsigned char* out_u8, float* out_f32, void* stream) {
  FX_REQUIRE(keys && colors && (out_u8 || out_f32), FLEXAM_E_ARG, "raster_resolve: null pointer (keys, colors and at least one output)");
  FX_REQUIRE(T > 0 && H > 0 && W > 0 && N > 0, FLEXAM_E_SHAPE, "raster_resolve: T=%d H=%d W=%d N=%d", T, H, W, N);
  FX_REQUIRE(color_frame_stride == 0 || color_frame_stride == (int64_t)N * 3, FLEXAM_E_SHAPE,
             "raster_resolve: colour tables are [N][3] (stride 0) or [T][N][3] (stride N * 3 = %ld), got stride %ld", (long)N * 3, (long)color_frame_stride);
  const int64_t frame_px = (int64_t)H * W, total = frame_px * T;
  hipLaunchKernelGGL(raster_resolve_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keys, colors,
                     color_frame_stride, (unsigned)N, frame_px, total, out_u8, out_f32);
  return flexam_check_launch("flexam_raster_resolve");
}
