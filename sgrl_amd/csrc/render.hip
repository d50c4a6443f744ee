// render.hip -- ray caster behind BatchedModularVecEnv.get_images() (C ABI: include/sgrl_render.h).  One thread per pixel,
// 16 x 16 pixel workgroups, the image's geom list (<= 64 records) staged in LDS; nearest hit over plane / spheres / capsules,
// Lambert shading from a head light plus ambient, checker pattern on the ground plane.  A visualisation aid: no parity claim.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/sgrl.h"
#include "../../include/sgrl_render.h"

namespace {

constexpr int kMaxGeoms = 64;

__device__ __forceinline__ float dot3f(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

__global__ __launch_bounds__(256) void k_render(const float* __restrict__ geoms, const int32_t* __restrict__ n_geoms, int max_geoms,
                                                const float* __restrict__ cams, int width, int height, uint8_t* rgb) {
  __shared__ float G[kMaxGeoms * SGRL_RENDER_GEOM_FLOATS];
  __shared__ float C[SGRL_RENDER_CAM_FLOATS];
  const int img = blockIdx.z, t = threadIdx.y * 16 + threadIdx.x;
  const int ng = min(n_geoms[img], kMaxGeoms);
  for (int i = t; i < ng * SGRL_RENDER_GEOM_FLOATS; i += 256) G[i] = geoms[((size_t)img * max_geoms) * SGRL_RENDER_GEOM_FLOATS + i];
  if (t < SGRL_RENDER_CAM_FLOATS) C[t] = cams[(size_t)img * SGRL_RENDER_CAM_FLOATS + t];
  __syncthreads();
  const int px = blockIdx.x * 16 + threadIdx.x, py = blockIdx.y * 16 + threadIdx.y;
  if (px >= width || py >= height) return;
  // ray through the pixel centre
  const float th = C[12], aspect = (float)width / (float)height;
  const float u = (2.f * (px + 0.5f) / width - 1.f) * th * aspect, v = (1.f - 2.f * (py + 0.5f) / height) * th;
  float d[3], o[3] = {C[0], C[1], C[2]};
  for (int k = 0; k < 3; k++) d[k] = C[3 + k] + u * C[6 + k] + v * C[9 + k];
  const float dn = rsqrtf(dot3f(d, d));
  for (int k = 0; k < 3; k++) d[k] *= dn;
  float best = 1e30f, nrm[3] = {0.f, 0.f, 1.f}, col[3] = {0.55f, 0.7f, 0.9f};   // sky
  bool hit = false;
  for (int g = 0; g < ng; g++) {
    const float* q = G + g * SGRL_RENDER_GEOM_FLOATS;
    const int type = (int)q[0];
    const float* c = q + 1;
    const float r = q[7], hl = q[8];
    if (type == 0) {                                  // plane z = c.z
      if (d[2] < -1e-6f) {
        const float tt = (c[2] - o[2]) / d[2];
        if (tt > 0.f && tt < best) {
          best = tt; hit = true; nrm[0] = 0.f; nrm[1] = 0.f; nrm[2] = 1.f;
          const float x = o[0] + tt * d[0], y = o[1] + tt * d[1];
          const bool chk = (((int)floorf(x)) + ((int)floorf(y))) & 1;
          const float s = chk ? 0.9f : 0.6f;
          col[0] = q[9] * s; col[1] = q[10] * s; col[2] = q[11] * s;
        }
      }
      continue;
    }
    // sphere (hl = 0) or capsule: segment pa..pb with radius r (closed form of the ray / capsule intersection)
    const float* ax = q + 4;
    float pa[3], ba[3], oa[3];
    for (int k = 0; k < 3; k++) { pa[k] = c[k] - hl * ax[k]; ba[k] = 2.f * hl * ax[k]; oa[k] = o[k] - pa[k]; }
    float tt = -1.f, y = 0.f;
    if (type == 3 && hl > 0.f) {
      const float baba = dot3f(ba, ba), bard = dot3f(ba, d), baoa = dot3f(ba, oa), rdoa = dot3f(d, oa), oaoa = dot3f(oa, oa);
      const float a = baba - bard * bard;
      float b = baba * rdoa - baoa * bard, cc = baba * oaoa - baoa * baoa - r * r * baba;
      float h = b * b - a * cc;
      if (h >= 0.f && a > 1e-12f) {
        const float t0 = (-b - sqrtf(h)) / a;
        y = baoa + t0 * bard;
        if (y > 0.f && y < baba) tt = t0;             // the cylinder body
        else {                                        // one of the caps
          float oc[3];
          for (int k = 0; k < 3; k++) oc[k] = y <= 0.f ? oa[k] : o[k] - (pa[k] + ba[k]);
          b = dot3f(d, oc); cc = dot3f(oc, oc) - r * r; h = b * b - cc;
          if (h > 0.f) tt = -b - sqrtf(h);
          y = y <= 0.f ? 0.f : baba;
        }
      } else if (a <= 1e-12f) {                        // ray parallel to the axis: caps only
        for (int e = 0; e < 2; e++) {
          float oc[3];
          for (int k = 0; k < 3; k++) oc[k] = o[k] - (pa[k] + e * ba[k]);
          const float bb = dot3f(d, oc), c2 = dot3f(oc, oc) - r * r, hh = bb * bb - c2;
          if (hh > 0.f) { const float t1 = -bb - sqrtf(hh); if (t1 > 0.f && (tt < 0.f || t1 < tt)) { tt = t1; y = e ? baba : 0.f; } }
        }
      }
      if (tt > 1e-4f && tt < best) {
        best = tt; hit = true;
        const float baba2 = dot3f(ba, ba), f = fminf(fmaxf(y / baba2, 0.f), 1.f);
        for (int k = 0; k < 3; k++) nrm[k] = (o[k] + tt * d[k] - (pa[k] + f * ba[k])) / r;
        col[0] = q[9]; col[1] = q[10]; col[2] = q[11];
      }
    } else {
      float oc[3] = {o[0] - c[0], o[1] - c[1], o[2] - c[2]};
      const float b = dot3f(d, oc), cc = dot3f(oc, oc) - r * r, h = b * b - cc;
      if (h > 0.f) {
        tt = -b - sqrtf(h);
        if (tt > 1e-4f && tt < best) {
          best = tt; hit = true;
          for (int k = 0; k < 3; k++) nrm[k] = (oc[k] + tt * d[k]) / r;
          col[0] = q[9]; col[1] = q[10]; col[2] = q[11];
        }
      }
    }
  }
  float shade = 1.f;
  if (hit) {                                           // head light + a fixed overhead light + ambient
    const float l1 = fmaxf(-dot3f(nrm, d), 0.f), l2 = fmaxf(nrm[2], 0.f);
    shade = 0.35f + 0.45f * l1 + 0.2f * l2;
  }
  uint8_t* out = rgb + (((size_t)img * height + py) * width + px) * 3;
  for (int k = 0; k < 3; k++) out[k] = (uint8_t)fminf(255.f, fmaxf(0.f, col[k] * shade * 255.f + 0.5f));
}

}  // namespace

extern "C" int sgrl_render(const float* geoms, const int32_t* n_geoms, int max_geoms, const float* cams, int n_img, int width,
                           int height, uint8_t* rgb, void* stream) {
  if (!geoms || !n_geoms || !cams || !rgb || n_img <= 0 || width <= 0 || height <= 0 || max_geoms <= 0 || max_geoms > kMaxGeoms)
    return SGRL_ERR_ARG;
  hipLaunchKernelGGL(k_render, dim3((width + 15) / 16, (height + 15) / 16, n_img), dim3(16, 16), 0, (hipStream_t)stream, geoms, n_geoms,
                     max_geoms, cams, width, height, rgb);
  return hipGetLastError() == hipSuccess ? SGRL_OK : SGRL_ERR_HIP;
}
