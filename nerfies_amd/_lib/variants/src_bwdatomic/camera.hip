// Camera geometry on the GPU: pixel -> world ray (with the 10-step Newton undistort), world point -> pixel.
//
// Reference behaviour: nerfies/camera.py:225-269 (pixel_to_local_rays, pixels_to_rays), :26-105 (undistort),
// :283-315 (project), :317-321 (get_pixel_centers), nerfies/datasets/core.py:50-75 (camera_to_rays).
// One thread per pixel; every output is an HBM-bound stream (12 B in / 32 B out per pixel), so the [n,3]
// outputs are transposed through LDS and written as one contiguous run per workgroup.
#include <hip/hip_runtime.h>

#include "nrf_internal.h"

namespace nrf {
namespace {

constexpr int CAM_THREADS = 256;

// Newton on (fx, fy) = distort(x, y) - (xd, yd); the update is skipped where |det J| <= 1e-9 and the iteration
// count is fixed at 10 (camera.py:76-105) so the result does not depend on a convergence test.
__device__ __forceinline__ void undistort(const CameraArgs& c, float xd, float yd, float& xo, float& yo) {
  float x = xd, y = yd;
#pragma unroll 1
  for (int it = 0; it < 10; ++it) {
    const float r = x * x + y * y;
    const float d = 1.0f + r * (c.k1 + r * (c.k2 + c.k3 * r));
    const float fx = d * x + 2.0f * c.p1 * x * y + c.p2 * (r + 2.0f * x * x) - xd;
    const float fy = d * y + 2.0f * c.p2 * x * y + c.p1 * (r + 2.0f * y * y) - yd;
    const float d_r = c.k1 + r * (2.0f * c.k2 + 3.0f * c.k3 * r);
    const float d_x = 2.0f * x * d_r, d_y = 2.0f * y * d_r;
    const float fx_x = d + d_x * x + 2.0f * c.p1 * y + 6.0f * c.p2 * x;
    const float fx_y = d_y * x + 2.0f * c.p1 * x + 2.0f * c.p2 * y;
    const float fy_x = d_x * y + 2.0f * c.p2 * y + 2.0f * c.p1 * x;
    const float fy_y = d + d_y * y + 2.0f * c.p2 * x + 6.0f * c.p1 * y;
    const float den = fy_x * fx_y - fx_x * fy_y;
    const bool ok = fabsf(den) > 1e-9f;
    x += ok ? (fx * fy_y - fy * fx_y) / den : 0.0f;
    y += ok ? (fy * fx_x - fx * fy_x) / den : 0.0f;
  }
  xo = x;
  yo = y;
}

// MODE 0: unit ray directions.  MODE 1: points at a given depth along the optical axis (pixels_to_points).
template <int MODE>
__global__ __launch_bounds__(CAM_THREADS) void camera_rays_kernel(CameraArgs c, const float2* __restrict__ pixels,
                                                                  const float* __restrict__ depth, long n,
                                                                  float* __restrict__ origins,
                                                                  float* __restrict__ directions,
                                                                  float2* __restrict__ pixels_out) {
  __shared__ float tile[3 * CAM_THREADS];
  const long base = (long)blockIdx.x * CAM_THREADS;
  const long i = base + threadIdx.x;
  float o[3] = {0.f, 0.f, 0.f};
  if (i < n) {
    float2 px;
    if (pixels) {
      px = pixels[i];
    } else {   // pixel centres of the [H, W] image in row-major order (camera.py:317-321)
      const long row = i / c.width;
      px = make_float2((float)(i - row * c.width) + 0.5f, (float)row + 0.5f);
    }
    if (pixels_out) pixels_out[i] = px;
    float y = (px.y - c.cy) / (c.focal * c.aspect);
    float x = (px.x - c.cx - y * c.skew) / c.focal;
    if (c.distorted) undistort(c, x, y, x, y);
    const float inv = 1.0f / sqrtf(x * x + y * y + 1.0f);
    const float lx = x * inv, ly = y * inv, lz = inv;
    // world = orientation^T * local, renormalised (camera.py:262-267)
    float wx = c.R[0] * lx + c.R[3] * ly + c.R[6] * lz;
    float wy = c.R[1] * lx + c.R[4] * ly + c.R[7] * lz;
    float wz = c.R[2] * lx + c.R[5] * ly + c.R[8] * lz;
    const float inv2 = 1.0f / sqrtf(wx * wx + wy * wy + wz * wz);
    wx *= inv2; wy *= inv2; wz *= inv2;
    if (MODE == 1) {   // rays * depth / cos(angle to the optical axis) + position (camera.py:271-277)
      const float s = depth[i] / (wx * c.R[6] + wy * c.R[7] + wz * c.R[8]);
      wx = wx * s + c.pos[0]; wy = wy * s + c.pos[1]; wz = wz * s + c.pos[2];
    }
    o[0] = wx; o[1] = wy; o[2] = wz;
  }
  tile[3 * threadIdx.x + 0] = o[0];
  tile[3 * threadIdx.x + 1] = o[1];
  tile[3 * threadIdx.x + 2] = o[2];
  __syncthreads();
  const long lim = min((long)3 * CAM_THREADS, 3 * (n - base));
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int j = e * CAM_THREADS + threadIdx.x;
    if (j < lim) {
      directions[3 * base + j] = tile[j];
      if (origins) origins[3 * base + j] = c.pos[j % 3];   // 3*base is a multiple of 3
    }
  }
}

__global__ __launch_bounds__(CAM_THREADS) void camera_project_kernel(CameraArgs c, const float* __restrict__ points, long n,
                                                                     float2* __restrict__ pixels) {
  __shared__ float tile[3 * CAM_THREADS];
  const long base = (long)blockIdx.x * CAM_THREADS;
  const long lim = min((long)3 * CAM_THREADS, 3 * (n - base));
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int j = e * CAM_THREADS + threadIdx.x;
    if (j < lim) tile[j] = points[3 * base + j] - c.pos[j % 3];
  }
  __syncthreads();
  const long i = base + threadIdx.x;
  if (i >= n) return;
  const float tx = tile[3 * threadIdx.x], ty = tile[3 * threadIdx.x + 1], tz = tile[3 * threadIdx.x + 2];
  const float lx = c.R[0] * tx + c.R[1] * ty + c.R[2] * tz;
  const float ly = c.R[3] * tx + c.R[4] * ty + c.R[5] * tz;
  const float lz = c.R[6] * tx + c.R[7] * ty + c.R[8] * tz;
  float x = lx / lz, y = ly / lz;
  const float r2 = x * x + y * y;
  const float dist = 1.0f + r2 * (c.k1 + r2 * (c.k2 + c.k3 * r2));
  const float xy = x * y;
  const float xd = x * dist + 2.0f * c.p1 * xy + c.p2 * (r2 + 2.0f * x * x);
  const float yd = y * dist + 2.0f * c.p2 * xy + c.p1 * (r2 + 2.0f * y * y);
  pixels[i] = make_float2(c.focal * xd + c.skew * yd + c.cx, c.focal * c.aspect * yd + c.cy);
}

}  // namespace

void launch_camera_rays(const CameraArgs& c, const float* pixels, const float* depth, long n, float* origins,
                        float* directions, float* pixels_out, hipStream_t stream) {
  const unsigned blocks = (unsigned)((n + CAM_THREADS - 1) / CAM_THREADS);
  if (depth)
    camera_rays_kernel<1><<<blocks, CAM_THREADS, 0, stream>>>(c, (const float2*)pixels, depth, n, nullptr, directions,
                                                             (float2*)pixels_out);
  else
    camera_rays_kernel<0><<<blocks, CAM_THREADS, 0, stream>>>(c, (const float2*)pixels, nullptr, n, origins, directions,
                                                             (float2*)pixels_out);
}

void launch_camera_project(const CameraArgs& c, const float* points, long n, float* pixels, hipStream_t stream) {
  const unsigned blocks = (unsigned)((n + CAM_THREADS - 1) / CAM_THREADS);
  camera_project_kernel<<<blocks, CAM_THREADS, 0, stream>>>(c, points, n, (float2*)pixels);
}

}  // namespace nrf
