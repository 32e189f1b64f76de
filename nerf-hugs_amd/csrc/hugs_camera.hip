// Device-side ray generation and batch assembly: pixel -> camera ray with lens
// undistortion / fisheye / NDC, ray radii, and the per-pixel gathers that build a training batch from
// images resident in HBM.  One thread per pixel; HBM-bound (36 B of camera matrices per ray come from
// L2, 56 B written per ray).  Replaces the reference's host numpy thread (datasets.py:289,447-492).
#include "hugs_common.h"

namespace {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mat3_mul(const float* m, int pitch, V3 v) {
  return {m[0] * v.x + m[1] * v.y + m[2] * v.z, m[pitch] * v.x + m[pitch + 1] * v.y + m[pitch + 2] * v.z,
          m[2 * pitch] * v.x + m[2 * pitch + 1] * v.y + m[2 * pitch + 2] * v.z};
}
__device__ __forceinline__ float norm3(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }

// camera_utils.py:410-495: ten Newton steps on (x, y) -> distorted (xd, yd); a step whose 2x2
// determinant is <= 1e-9 in magnitude is dropped.
__device__ __forceinline__ void undistort(float xd, float yd, const float* k, float& xo, float& yo) {
  const float k1 = k[0], k2 = k[1], k3 = k[2], k4 = k[3], p1 = k[4], p2 = k[5];
  float x = xd, y = yd;
  for (int it = 0; it < 10; ++it) {
    float r = x * x + y * y;
    float d = 1.f + r * (k1 + r * (k2 + r * (k3 + r * k4)));
    float fx = d * x + 2.f * p1 * x * y + p2 * (r + 2.f * x * x) - xd;
    float fy = d * y + 2.f * p2 * x * y + p1 * (r + 2.f * y * y) - yd;
    float d_r = k1 + r * (2.f * k2 + r * (3.f * k3 + r * 4.f * k4));
    float d_x = 2.f * x * d_r, d_y = 2.f * y * d_r;
    float fx_x = d + d_x * x + 2.f * p1 * y + 6.f * p2 * x;
    float fx_y = d_y * x + 2.f * p1 * x + 2.f * p2 * y;
    float fy_x = d_x * y + 2.f * p2 * y + 2.f * p1 * x;
    float fy_y = d + d_y * y + 2.f * p2 * x + 6.f * p1 * y;
    float den = fy_x * fx_y - fx_x * fy_y;
    bool ok = fabsf(den) > 1e-9f;
    x += ok ? (fx * fy_y - fy * fx_y) / den : 0.f;
    y += ok ? (fy * fx_x - fx * fy_x) / den : 0.f;
  }
  xo = x;
  yo = y;
}

// camera_utils.py:32-100 convert_to_ndc (near = 1)
__device__ __forceinline__ void to_ndc(V3 o, V3 d, float xm, float ym, V3& o_ndc, V3& d_ndc) {
  float t = -(1.f + o.z) / d.z;
  o = {o.x + t * d.x, o.y + t * d.y, o.z + t * d.z};
  o_ndc = {xm * o.x / o.z, ym * o.y / o.z, -1.f};
  V3 inf = {xm * d.x / d.z, ym * d.y / d.z, 1.f};
  d_ndc = sub3(inf, o_ndc);
}

__global__ void __launch_bounds__(256)
k_pixels_to_rays(int n, const int* __restrict__ pix_x, const int* __restrict__ pix_y, const int* __restrict__ cam_idx,
                 int ncams, const float* __restrict__ pixtocams, const float* __restrict__ camtoworlds,
                 const float* __restrict__ dist, int dist_per_cam, const float* __restrict__ ndc, int fisheye,
                 const int* __restrict__ widths, const int* __restrict__ heights, float* __restrict__ origins,
                 float* __restrict__ directions, float* __restrict__ viewdirs, float* __restrict__ radii,
                 float* __restrict__ pix_coords) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int cam = cam_idx ? cam_idx[i] : 0;
  if (cam < 0 || cam >= ncams) cam = 0;   // host validates; keep the access in bounds regardless
  const float* p2c = pixtocams + 9 * (ncams == 1 ? 0 : cam);
  const float* c2w = camtoworlds + 12 * (ncams == 1 ? 0 : cam);
  const float* kd = dist ? dist + 6 * (dist_per_cam ? cam : 0) : nullptr;
  int px = pix_x[i], py = pix_y[i];
  V3 dirs[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {   // the pixel centre, its +x and its +y neighbour (ray radii)
    V3 p = {(float)(px + (j == 1)) + .5f, (float)(py + (j == 2)) + .5f, 1.f};
    V3 c = mat3_mul(p2c, 3, p);
    if (kd) {
      undistort(c.x, c.y, kd, c.x, c.y);
      c.z = 1.f;
    }
    if (fisheye) {
      float th = fminf(3.14159265358979323846f, sqrtf(c.x * c.x + c.y * c.y));
      float s = sinf(th) / th;
      c = {c.x * s, c.y * s, cosf(th)};
    }
    c.y = -c.y;   // OpenCV -> OpenGL
    c.z = -c.z;
    dirs[j] = mat3_mul(c2w, 4, c);
  }
  V3 o = {c2w[3], c2w[7], c2w[11]};
  V3 d = dirs[0];
  float inv = 1.f / norm3(d);
  V3 v = {d.x * inv, d.y * inv, d.z * inv};
  float nx, ny;
  if (!ndc) {
    nx = norm3(sub3(dirs[1], d));
    ny = norm3(sub3(dirs[2], d));
  } else {
    float xm = 1.f / ndc[2], ym = 1.f / ndc[5];
    V3 ox, oy, on, dn, tmp;
    to_ndc(o, dirs[1], xm, ym, ox, tmp);
    to_ndc(o, dirs[2], xm, ym, oy, tmp);
    to_ndc(o, d, xm, ym, on, dn);
    o = on;
    d = dn;
    nx = norm3(sub3(ox, on));
    ny = norm3(sub3(oy, on));
  }
  origins[3 * i] = o.x; origins[3 * i + 1] = o.y; origins[3 * i + 2] = o.z;
  directions[3 * i] = d.x; directions[3 * i + 1] = d.y; directions[3 * i + 2] = d.z;
  viewdirs[3 * i] = v.x; viewdirs[3 * i + 1] = v.y; viewdirs[3 * i + 2] = v.z;
  radii[i] = (.5f * (nx + ny)) * 2.f / 3.4641016151377544f;
  if (pix_coords) {   // camera_utils.py:649-652
    pix_coords[2 * i] = ((float)px + .5f) / (float)widths[cam];
    pix_coords[2 * i + 1] = ((float)py + .5f) / (float)heights[cam];
  }
}

// dst[i, :] = src[(offset[cam] + y*width[cam] + x) * C : +C]; u8 sources are divided by 255 in binary32
// (what the loaders do on the host, datasets.py image decode).  per_pixel == 0: src is [ncams, C].
template <bool U8>
__global__ void __launch_bounds__(256)
k_gather_pixels(int n, int C, const int* __restrict__ pix_x, const int* __restrict__ pix_y,
                const int* __restrict__ cam_idx, const int64_t* __restrict__ offsets, const int* __restrict__ widths,
                int per_pixel, const void* __restrict__ src, float* __restrict__ dst) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int cam = cam_idx ? cam_idx[i] : 0;
  int64_t e = per_pixel ? offsets[cam] + (int64_t)pix_y[i] * widths[cam] + pix_x[i] : cam;
  for (int c = 0; c < C; ++c) {
    float v;
    if (U8) v = __fdiv_rn((float)((const uint8_t*)src)[e * C + c], 255.f);
    else v = ((const float*)src)[e * C + c];
    dst[(int64_t)i * C + c] = v;
  }
}

// datasets.py:498-524: patch origin + (dx, dy) * dilation, camera index broadcast over the patch
__global__ void __launch_bounds__(256)
k_expand_patches(int npatch, int ps, int dilation, const int* __restrict__ org_x, const int* __restrict__ org_y,
                 const int* __restrict__ cam_of_patch, int* __restrict__ pix_x, int* __restrict__ pix_y,
                 int* __restrict__ cam_idx) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int per = ps * ps;
  if (i >= npatch * per) return;
  int p = i / per, r = i - p * per;
  pix_x[i] = org_x[p] + (r % ps) * dilation;
  pix_y[i] = org_y[p] + (r / ps) * dilation;
  cam_idx[i] = cam_of_patch[p];
}

}  // namespace

extern "C" int hugs_pixels_to_rays(int n, const int32_t* pix_x, const int32_t* pix_y, const int32_t* cam_idx, int ncams,
                                   const float* pixtocams, const float* camtoworlds, const float* dist,
                                   int dist_per_cam, const float* pixtocam_ndc, int camtype, const int32_t* widths,
                                   const int32_t* heights, float* origins, float* directions, float* viewdirs,
                                   float* radii, float* pix_coords, void* stream) {
  HUGS_REQUIRE(n >= 0 && ncams >= 1, -2, "hugs_pixels_to_rays: n=%d ncams=%d", n, ncams);
  HUGS_REQUIRE(camtype == 0 || camtype == 1, -2, "hugs_pixels_to_rays: camtype %d (0 perspective, 1 fisheye)", camtype);
  HUGS_REQUIRE(pix_x && pix_y && pixtocams && camtoworlds && origins && directions && viewdirs && radii, -2,
               "hugs_pixels_to_rays: null pointer");
  HUGS_REQUIRE(!pix_coords || (widths && heights), -2, "hugs_pixels_to_rays: pix_coords needs widths and heights");
  HUGS_REQUIRE(ncams == 1 || cam_idx, -2, "hugs_pixels_to_rays: %d cameras but no cam_idx", ncams);
  if (n == 0) return 0;
  k_pixels_to_rays<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(
      n, pix_x, pix_y, cam_idx, ncams, pixtocams, camtoworlds, dist, dist_per_cam, pixtocam_ndc, camtype, widths, heights,
      origins, directions, viewdirs, radii, pix_coords);
  HUGS_CHECK_LAUNCH("k_pixels_to_rays");
  return 0;
}

extern "C" int hugs_gather_pixels(int n, int channels, const int32_t* pix_x, const int32_t* pix_y,
                                  const int32_t* cam_idx, const int64_t* offsets, const int32_t* widths, int per_pixel,
                                  int src_u8, const void* src, float* dst, void* stream) {
  HUGS_REQUIRE(n >= 0 && channels >= 1, -2, "hugs_gather_pixels: n=%d channels=%d", n, channels);
  HUGS_REQUIRE(src && dst && (!per_pixel || (pix_x && pix_y && offsets && widths)), -2,
               "hugs_gather_pixels: null pointer");
  if (n == 0) return 0;
  if (src_u8)
    k_gather_pixels<true><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, channels, pix_x, pix_y, cam_idx, offsets,
                                                                           widths, per_pixel, src, dst);
  else
    k_gather_pixels<false><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, channels, pix_x, pix_y, cam_idx,
                                                                            offsets, widths, per_pixel, src, dst);
  HUGS_CHECK_LAUNCH("k_gather_pixels");
  return 0;
}

extern "C" int hugs_expand_patches(int npatch, int patch_size, int dilation, const int32_t* org_x, const int32_t* org_y,
                                   const int32_t* cam_of_patch, int32_t* pix_x, int32_t* pix_y, int32_t* cam_idx,
                                   void* stream) {
  HUGS_REQUIRE(npatch >= 0 && patch_size >= 1 && dilation >= 1, -2, "hugs_expand_patches: npatch=%d size=%d dilation=%d",
               npatch, patch_size, dilation);
  if (npatch == 0) return 0;
  int n = npatch * patch_size * patch_size;
  k_expand_patches<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(npatch, patch_size, dilation, org_x, org_y,
                                                                   cam_of_patch, pix_x, pix_y, cam_idx);
  HUGS_CHECK_LAUNCH("k_expand_patches");
  return 0;
}
