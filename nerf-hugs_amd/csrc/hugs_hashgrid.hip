// Multiresolution hash-grid encoding and degree-4 spherical harmonics for the nerfacto path (SURVEY §8f row 3,
// groundwork): what nerfacto/models/nerfacto.py:693-733,761-770,921-947 obtains from tiny-cuda-nn.  Algorithm as in
// oracle/hashgrid_ref.py (Instant-NGP / tiny-cuda-nn HashGrid; PARITY UNPINNED: tiny-cuda-nn is not available).
// One thread per sample walks the levels: 8 trilinear corners x F features per level gathered from the fp32 table
// (random 8-byte reads: L2 / HBM-latency bound, not a GEMM), the [n_levels*F] row written contiguously.
// Backward scatters with fp32 atomics (as tiny-cuda-nn does with half2 atomics).
#include <type_traits>
#include "hugs_common.h"

#define HG_MAXL 32

struct HgLevels {
  uint32_t off[HG_MAXL + 1];   // first entry of each level (in entries of F floats)
  uint32_t res[HG_MAXL];
  float scale[HG_MAXL];
};

namespace {

// tiny-cuda-nn grid_index: dense levels index x + y res + z res^2, hashed ones x ^ y p1 ^ z p2, both modulo the level's
// entry count.  The modulo is never a division here: a dense level has res^3 <= entries and corner coordinates <= res, so
// the index of a point inside the unit cube is below 2 * entries (one conditional subtraction); a hashed level's count is 2^log2_hashmap_size (a mask) --
// anything else takes the generic remainder.  (Eight 32-bit remainders per sample and level were most of the forward
// kernel's instructions.)
template <int F>
__device__ __forceinline__ uint32_t hg_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res, uint32_t entries, bool dense) {
  if (dense) {
    uint32_t idx = cx + cy * res + cz * res * res;
    if (idx >= entries) { idx -= entries; if (idx >= entries) idx %= entries; }      // (second case: points outside [0,1]^3 only)
    return idx;
  }
  const uint32_t idx = (cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u);
  return (entries & (entries - 1u)) == 0u ? (idx & (entries - 1u)) : idx % entries;
}

// one table entry (F features) as floats; TH: the table is an IEEE-half copy (the fp16 mode: 4 / 8 bytes per corner instead of 8 / 16)
template <int F, bool TH>
__device__ __forceinline__ void hg_entry(const void* __restrict__ table, size_t entry, float (&t)[F]) {
  if (TH) {
    if (F == 2) { const uint32_t u = ((const uint32_t*)table)[entry]; t[0] = h16_to_f((uint16_t)u); t[1] = h16_to_f((uint16_t)(u >> 16)); }
    else {
      const uint2 u = ((const uint2*)table)[entry];
      t[0] = h16_to_f((uint16_t)u.x); t[1] = h16_to_f((uint16_t)(u.x >> 16)); t[F - 2] = h16_to_f((uint16_t)u.y); t[F - 1] = h16_to_f((uint16_t)(u.y >> 16));
    }
  } else {
#pragma unroll
    for (int f = 0; f < F; ++f) t[f] = ((const float*)table)[entry * F + f];
  }
}

// Forward.  Blocks are (group of HG_LG levels, 256 samples) with the level groups OUTERMOST in launch order: at any moment the
// chip works on one or two groups, whose tables (2^19 x 2 halfs = 2 MB per hashed level of the field grid) stay in every XCD's
// 4 MB L2 -- with one thread walking all 16 levels of its sample the waves were spread over all levels at once (32 MB of tables)
// and every fine-level gather was a 64-byte sector from MALL / HBM.  Same-box (scratch/hg_lm.py, fp16 tables): field grid
// 968 -> 670 us, proposal grids 519 -> 466 and 416 -> 293 us; groups of 1 or 2 levels help the field as much but cost the
// proposal grids their contiguous output runs.  A thread writes its group's HG_LG x F outputs as one contiguous run.
#define HG_LG 4
template <int F, int BF16, bool TH>
__global__ void __launch_bounds__(256)
k_hashgrid_fwd(int n, int nblk, int L, HgLevels lv, const float* __restrict__ x, const void* __restrict__ table, int row_pitch,
               void* __restrict__ out) {
  const int g = blockIdx.x / nblk;
  const int i = (blockIdx.x - g * nblk) * 256 + threadIdx.x;
  if (i >= n) return;
  const float px = x[3 * i], py = x[3 * i + 1], pz = x[3 * i + 2];
  float o[HG_LG][F];
#pragma unroll
  for (int q = 0; q < HG_LG; ++q) {
    const int l = g * HG_LG + q;
#pragma unroll
    for (int f = 0; f < F; ++f) o[q][f] = 0.f;
    if (l >= L) continue;
    const uint32_t res = lv.res[l], entries = lv.off[l + 1] - lv.off[l];
    const bool dense = (uint64_t)res * res * res <= entries;
    const float sc = lv.scale[l];
    const float fx = fmaf(px, sc, .5f), fy = fmaf(py, sc, .5f), fz = fmaf(pz, sc, .5f);
    const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
    const float wx = fx - gx, wy = fy - gy, wz = fz - gz;
    const uint32_t cx = (uint32_t)(int)gx, cy = (uint32_t)(int)gy, cz = (uint32_t)(int)gz;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float w = ((c & 1) ? wx : 1.f - wx) * ((c & 2) ? wy : 1.f - wy) * ((c & 4) ? wz : 1.f - wz);
      const uint32_t idx = hg_index<F>(cx + (c & 1), cy + ((c >> 1) & 1), cz + ((c >> 2) & 1), res, entries, dense);
      float t[F];
      hg_entry<F, TH>(table, (size_t)lv.off[l] + idx, t);
#pragma unroll
      for (int f = 0; f < F; ++f) o[q][f] += w * t[f];
    }
  }
  const int nl = min(HG_LG, L - g * HG_LG);
  const size_t e0 = (size_t)i * row_pitch + (size_t)g * HG_LG * F;
  if (BF16 && F == 2 && nl == HG_LG && (row_pitch & 7) == 0) {      // a whole group in a 16-bit format: one 16-byte store
    uint4 u;
    u.x = f2_to_op16(o[0][0], o[0][1], BF16); u.y = f2_to_op16(o[1][0], o[1][1], BF16);
    u.z = f2_to_op16(o[2][0], o[2][1], BF16); u.w = f2_to_op16(o[3][0], o[3][1], BF16);
    *(uint4*)((uint16_t*)out + e0) = u;
    return;
  }
#pragma unroll
  for (int q = 0; q < HG_LG; ++q)
    if (q < nl)
#pragma unroll
      for (int f = 0; f < F; ++f) {
        if (BF16) ((uint16_t*)out)[e0 + q * F + f] = f_to_op16(o[q][f], BF16);
        else ((float*)out)[e0 + q * F + f] = o[q][f];
      }
}

// Table gradient.  Thread = sample, and consecutive samples are consecutive points of one ray: at every level whose
// cells are wider than the sample spacing a wavefront holds RUNS of lanes inside the same cell (8.4 M proposal samples
// share the 4096 cells of level 0).  Scattering each lane's 8 x F products separately is what made this kernel 90 % of
// the nerfacto step (1.7 G fp32 atomics at ~21 G/s, worse where they collide); instead each run is summed inside the
// wave first -- a segmented inclusive scan over the lanes, keyed by the cell -- and only the last lane of a run issues
// atomics.  Levels where no two neighbouring lanes share a cell skip the scan (one ballot).
template <int F, int BF16>
__global__ void __launch_bounds__(256)
k_hashgrid_bwd(int n, int L, HgLevels lv, const float* __restrict__ x, const void* __restrict__ d_out, int row_pitch,
               float* __restrict__ d_table, int l_begin) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = i < n;
  const int ii = live ? i : n - 1;
  const float px = x[3 * ii], py = x[3 * ii + 1], pz = x[3 * ii + 2];
  for (int l = l_begin; l < L; ++l) {
    float g[F];
    bool gnz = false;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      g[f] = !live ? 0.f : BF16 ? op16_to_f(((const uint16_t*)d_out)[(size_t)ii * row_pitch + l * F + f], BF16)
                                : ((const float*)d_out)[(size_t)ii * row_pitch + l * F + f];
      gnz |= g[f] != 0.f;
    }
    // a wave whose 64 samples all have a zero output gradient at this level (outside the box, behind the surface, or -- in the
    // fp16 mode -- flushed below half's subnormals) has nothing to scatter: skip the index arithmetic and the scan
    if (__ballot(gnz) == 0ull) continue;
    const uint32_t res = lv.res[l], entries = lv.off[l + 1] - lv.off[l];
    const bool dense = (uint64_t)res * res * res <= entries;
    const float sc = lv.scale[l];
    const float fx = fmaf(px, sc, .5f), fy = fmaf(py, sc, .5f), fz = fmaf(pz, sc, .5f);
    const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
    const float wx = fx - gx, wy = fy - gy, wz = fz - gz;
    const uint32_t cx = (uint32_t)(int)gx, cy = (uint32_t)(int)gy, cz = (uint32_t)(int)gz;
    float v[8][F];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float w = ((c & 1) ? wx : 1.f - wx) * ((c & 2) ? wy : 1.f - wy) * ((c & 4) ? wz : 1.f - wz);
#pragma unroll
      for (int f = 0; f < F; ++f) v[c][f] = w * g[f];
    }
    // run structure of this wave at this level
    const uint32_t pcx = __shfl_up(cx, 1), pcy = __shfl_up(cy, 1), pcz = __shfl_up(cz, 1);
    const int plive = __shfl_up((int)live, 1);
    int head = (lane == 0) || !plive || pcx != cx || pcy != cy || pcz != cz || !live;
    if (__ballot(!head) != 0ull) {          // some lane continues its neighbour's cell: segmented inclusive scan
      int fl = head;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int tf = __shfl_up(fl, d);
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int f = 0; f < F; ++f) {
            const float tv = __shfl_up(v[c][f], d);
            if (lane >= d && !fl) v[c][f] += tv;
          }
        if (lane >= d) fl |= tf;
      }
    }
    const int nhead = __shfl_down(head, 1);
    // the last lane of a run carries the run's sums; a run whose sums are all zero adds nothing (samples outside the
    // scene box, fully occluded ones: their output gradient is exactly 0 -- a large share of a real batch)
    bool nz = false;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int f = 0; f < F; ++f) nz |= v[c][f] != 0.f;
    const bool tail = live && nz && (lane == 63 || nhead);
    float* tb = d_table + (size_t)lv.off[l] * F;
    if constexpr (F == 2) {
      // The chip retires ~21 G atomic TRANSACTIONS per second, not atomics: neighbouring lanes of one instruction that hit
      // neighbouring floats share a transaction (scratch/atomic_pair.hip: 21 / 42 / 84 / 335 G atomics/s for groups of
      // 1 / 2 / 4 / 16 adjacent floats).  So a run's 16 adds are not issued by its tail lane alone: the four lanes of a quad
      // serve the quad's four items one after the other, lane q adding feature (q & 1) of the corner with x-offset (q >> 1) --
      // 8 adjacent bytes always, 16 when the x-neighbour entry is adjacent (dense levels; hashed levels when cx is even).
      const int q = lane & 3, fq = q & 1, dxq = q >> 1;
      auto serve = [&](auto subc) {
        constexpr int CTL = decltype(subc)::value * 0x55;             // quad_perm: every lane reads lane `sub` of its quad
        const int t_ = __builtin_amdgcn_mov_dpp(tail ? 1 : 0, CTL, 0xf, 0xf, true);
        if (__ballot(t_) == 0ull) return;
        const uint32_t bx = (uint32_t)__builtin_amdgcn_mov_dpp((int)cx, CTL, 0xf, 0xf, true);
        const uint32_t by = (uint32_t)__builtin_amdgcn_mov_dpp((int)cy, CTL, 0xf, 0xf, true);
        const uint32_t bz = (uint32_t)__builtin_amdgcn_mov_dpp((int)cz, CTL, 0xf, 0xf, true);
#pragma unroll
        for (int cb = 0; cb < 8; cb += 2) {
          const float a0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[cb][0]), CTL, 0xf, 0xf, true));
          const float a1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[cb][1]), CTL, 0xf, 0xf, true));
          const float b0 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[cb + 1][0]), CTL, 0xf, 0xf, true));
          const float b1 = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[cb + 1][1]), CTL, 0xf, 0xf, true));
          const float val = dxq ? (fq ? b1 : b0) : (fq ? a1 : a0);
          const uint32_t idx = hg_index<F>(bx + dxq, by + ((cb >> 1) & 1), bz + ((cb >> 2) & 1), res, entries, dense);
          if (t_) atomicAdd(tb + (size_t)idx * 2 + fq, val);
        }
      };
      serve(std::integral_constant<int, 0>{}); serve(std::integral_constant<int, 1>{});
      serve(std::integral_constant<int, 2>{}); serve(std::integral_constant<int, 3>{});
    } else if (tail) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t idx = hg_index<F>(cx + (c & 1), cy + ((c >> 1) & 1), cz + ((c >> 2) & 1), res, entries, dense);
#pragma unroll
        for (int f = 0; f < F; ++f) atomicAdd(tb + (size_t)idx * F + f, v[c][f]);
      }
    }
  }
}

// Coarse levels in LDS.  The coarsest level (16^3 cells: 4913 entries) receives every sample's gradient in a few thousand
// addresses: the most contended -- and, per level, the most expensive -- part of the global scatter.  Here a persistent
// workgroup accumulates the level in a private LDS copy (ds_add_f32; runs of lanes in one cell are summed first, as in
// the global kernel) and adds the copy to the table once at the end: neighbouring lanes, neighbouring floats, so 16 floats per
// atomic transaction.  Round 5: any dense level that fits -- CAP = 12288 floats (48 KiB, 256 threads, three workgroups per CU: level
// 0) or CAP = 39936 floats (156 KiB, 1024 threads, one workgroup per CU: level 1 of the yml's field grid, 23^3 x 2 floats = 95 KiB,
// and of its second proposal grid, 26^3 x 2 = 137 KiB); `l` = the level, its gradient columns are l*F .. l*F + F - 1 of d_out.
#define HG_L0_MAX_FLOATS 12288      // 48 KiB
#define HG_L1_MAX_FLOATS 39936      // 156 KiB
template <int F, int BF16, int CAP, int NT>
__global__ void __launch_bounds__(NT)
k_hashgrid_bwd_lds(int n, int l, HgLevels lv, const float* __restrict__ x, const void* __restrict__ d_out, int row_pitch,
                   float* __restrict__ d_table) {
  __shared__ float acc[CAP];
  const uint32_t res = lv.res[l], entries = lv.off[l + 1] - lv.off[l];
  const int nfl = (int)entries * F;
  for (int e = threadIdx.x; e < nfl; e += NT) acc[e] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float sc = lv.scale[l];
  const int ntile = (n + NT - 1) / NT;
  for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int i = t * NT + threadIdx.x;
    const bool live = i < n;
    const int ii = live ? i : n - 1;
    float g[F];
    bool gnz = false;
#pragma unroll
    for (int f = 0; f < F; ++f) {
      g[f] = !live ? 0.f : BF16 ? op16_to_f(((const uint16_t*)d_out)[(size_t)ii * row_pitch + l * F + f], BF16)
                                : ((const float*)d_out)[(size_t)ii * row_pitch + l * F + f];
      gnz |= g[f] != 0.f;
    }
    if (__ballot(gnz) == 0ull) continue;      // (a wave without gradient at this level: see k_hashgrid_bwd)
    const float fx = fmaf(x[3 * ii], sc, .5f), fy = fmaf(x[3 * ii + 1], sc, .5f), fz = fmaf(x[3 * ii + 2], sc, .5f);
    const float gx = floorf(fx), gy = floorf(fy), gz = floorf(fz);
    const float wx = fx - gx, wy = fy - gy, wz = fz - gz;
    const uint32_t cx = (uint32_t)(int)gx, cy = (uint32_t)(int)gy, cz = (uint32_t)(int)gz;
    float v[8][F];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float w = ((c & 1) ? wx : 1.f - wx) * ((c & 2) ? wy : 1.f - wy) * ((c & 4) ? wz : 1.f - wz);
#pragma unroll
      for (int f = 0; f < F; ++f) v[c][f] = w * g[f];
    }
    const uint32_t pcx = __shfl_up(cx, 1), pcy = __shfl_up(cy, 1), pcz = __shfl_up(cz, 1);
    const int plive = __shfl_up((int)live, 1);
    const int head = (lane == 0) || !plive || pcx != cx || pcy != cy || pcz != cz || !live;
    if (__ballot(!head) != 0ull) {
      int fl = head;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int tf = __shfl_up(fl, d);
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int f = 0; f < F; ++f) {
            const float tv = __shfl_up(v[c][f], d);
            if (lane >= d && !fl) v[c][f] += tv;
          }
        if (lane >= d) fl |= tf;
      }
    }
    const int nhead = __shfl_down(head, 1);
    bool nz = false;
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int f = 0; f < F; ++f) nz |= v[c][f] != 0.f;
    if (live && nz && (lane == 63 || nhead)) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t idx = hg_index<F>(cx + (c & 1), cy + ((c >> 1) & 1), cz + ((c >> 2) & 1), res, entries, true);
#pragma unroll
        for (int f = 0; f < F; ++f) atomicAdd(&acc[idx * F + f], v[c][f]);
      }
    }
  }
  __syncthreads();
  float* tb = d_table + (size_t)lv.off[l] * F;
  for (int e = threadIdx.x; e < nfl; e += NT) {
    const float a = acc[e];
    if (a != 0.f) atomicAdd(tb + e, a);
  }
}

template <int BF16>
__global__ void __launch_bounds__(256)
k_sh4(int n, const float* __restrict__ d01, int row_pitch, int col0, void* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = d01[3 * i] * 2.f - 1.f, y = d01[3 * i + 1] * 2.f - 1.f, z = d01[3 * i + 2] * 2.f - 1.f;
  const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
  float o[16];
  o[0] = 0.28209479177387814f;
  o[1] = -0.48860251190291987f * y;
  o[2] = 0.48860251190291987f * z;
  o[3] = -0.48860251190291987f * x;
  o[4] = 1.0925484305920792f * xy;
  o[5] = -1.0925484305920792f * yz;
  o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
  o[7] = -1.0925484305920792f * xz;
  o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
  o[9] = 0.59004358992664352f * y * (-3.f * x2 + y2);
  o[10] = 2.8906114426405538f * xy * z;
  o[11] = 0.45704579946446572f * y * (1.f - 5.f * z2);
  o[12] = 0.3731763325901154f * z * (5.f * z2 - 3.f);
  o[13] = 0.45704579946446572f * x * (1.f - 5.f * z2);
  o[14] = 1.4453057213202769f * z * (x2 - y2);
  o[15] = 0.59004358992664352f * x * (-x2 + 3.f * y2);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (BF16) ((uint16_t*)out)[(size_t)i * row_pitch + col0 + k] = f_to_op16(o[k], BF16);
    else ((float*)out)[(size_t)i * row_pitch + col0 + k] = o[k];
  }
}

// ---- 2-D grid: HA-NeRF's ImplicitMask in the nerfacto model (nerfacto.py:1036-1047, 1080-1091) encodes the pixel coordinate
// of a RAY (not a sample): resolution^2 dense entries or the two-prime hash, bilinear over 4 corners.  One thread per ray
// writes the whole input row of the mask MLP: [grid features | per-ray transient embedding | zero padding].
template <int F, int BF16>
__global__ void __launch_bounds__(256)
k_hashgrid2d_fwd(int n, int L, HgLevels lv, const float* __restrict__ x, const float* __restrict__ table,
                 const float* __restrict__ extra, int T, int row_pitch, void* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float px = x[2 * i], py = x[2 * i + 1];
  auto put = [&](int col, float v) {
    if (BF16) ((uint16_t*)out)[(size_t)i * row_pitch + col] = f_to_op16(v, BF16);
    else ((float*)out)[(size_t)i * row_pitch + col] = v;
  };
  for (int l = 0; l < L; ++l) {
    const uint32_t res = lv.res[l], entries = lv.off[l + 1] - lv.off[l];
    const bool dense = (uint64_t)res * res <= entries;
    const float sc = lv.scale[l];
    const float fx = fmaf(px, sc, .5f), fy = fmaf(py, sc, .5f);
    const float gx = floorf(fx), gy = floorf(fy);
    const float wx = fx - gx, wy = fy - gy;
    const uint32_t cx = (uint32_t)(int)gx, cy = (uint32_t)(int)gy;
    const float* tb = table + (size_t)lv.off[l] * F;
    float acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float w = ((c & 1) ? wx : 1.f - wx) * ((c & 2) ? wy : 1.f - wy);
      const uint32_t ux = cx + (c & 1), uy = cy + (c >> 1);
      const uint32_t idx = (dense ? ux + uy * res : (ux * 1u) ^ (uy * 2654435761u)) % entries;
#pragma unroll
      for (int f = 0; f < F; ++f) acc[f] += w * tb[(size_t)idx * F + f];
    }
#pragma unroll
    for (int f = 0; f < F; ++f) put(l * F + f, acc[f]);
  }
  for (int t = 0; t < T; ++t) put(L * F + t, extra[(size_t)i * T + t]);
  for (int c = L * F + T; c < row_pitch; ++c) put(c, 0.f);
}

template <int F, int BF16>
__global__ void __launch_bounds__(256)
k_hashgrid2d_bwd(int n, int L, HgLevels lv, const float* __restrict__ x, const void* __restrict__ d_out, int row_pitch,
                 float* __restrict__ d_table) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float px = x[2 * i], py = x[2 * i + 1];
  for (int l = 0; l < L; ++l) {
    const uint32_t res = lv.res[l], entries = lv.off[l + 1] - lv.off[l];
    const bool dense = (uint64_t)res * res <= entries;
    const float sc = lv.scale[l];
    const float fx = fmaf(px, sc, .5f), fy = fmaf(py, sc, .5f);
    const float gx = floorf(fx), gy = floorf(fy);
    const float wx = fx - gx, wy = fy - gy;
    const uint32_t cx = (uint32_t)(int)gx, cy = (uint32_t)(int)gy;
    float* tb = d_table + (size_t)lv.off[l] * F;
    float g[F];
#pragma unroll
    for (int f = 0; f < F; ++f)
      g[f] = BF16 ? op16_to_f(((const uint16_t*)d_out)[(size_t)i * row_pitch + l * F + f], BF16) : ((const float*)d_out)[(size_t)i * row_pitch + l * F + f];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float w = ((c & 1) ? wx : 1.f - wx) * ((c & 2) ? wy : 1.f - wy);
      const uint32_t ux = cx + (c & 1), uy = cy + (c >> 1);
      const uint32_t idx = (dense ? ux + uy * res : (ux * 1u) ^ (uy * 2654435761u)) % entries;
#pragma unroll
      for (int f = 0; f < F; ++f) atomicAdd(tb + (size_t)idx * F + f, w * g[f]);
    }
  }
}

#include "hugs_hashgrid_binned.inc"

int fill_levels(HgLevels& lv, int L, const long long* off, const int* res, const float* scale, const char* who) {
  HUGS_REQUIRE(L >= 1 && L <= HG_MAXL && off && res && scale, -2, "%s: %d levels (1..%d) / null level table", who, L, HG_MAXL);
  for (int l = 0; l <= L; ++l) {
    HUGS_REQUIRE(off[l] >= 0 && off[l] < (1ll << 32) && (l == 0 || off[l] > off[l - 1]), -2, "%s: bad level offsets", who);
    lv.off[l] = (uint32_t)off[l];
  }
  for (int l = 0; l < L; ++l) { lv.res[l] = (uint32_t)res[l]; lv.scale[l] = scale[l]; }
  return 0;
}

}  // namespace

// launch K<F, DT> (or K<F, DT, TH>) for the run-time dtype code dt_ (0 fp32 / 1 bf16 / 2 half); the variadic part is the
// <<<grid, block, lds, stream>>>(arguments...) text
#define HG_BY_DTYPE(dt_, K_, F_, ...)                                                                                     \
  do { if ((dt_) == 0) K_<F_, 0> __VA_ARGS__; else if ((dt_) == 1) K_<F_, 1> __VA_ARGS__; else K_<F_, 2> __VA_ARGS__; } while (0)
#define HG_BY_DTYPE_T(dt_, K_, F_, TH_, ...)                                                                              \
  do { if ((dt_) == 0) K_<F_, 0, TH_> __VA_ARGS__; else if ((dt_) == 1) K_<F_, 1, TH_> __VA_ARGS__; else K_<F_, 2, TH_> __VA_ARGS__; } while (0)

static int hashgrid_fwd_impl(int n, int n_levels, int features, const long long* level_offsets, const int* level_resolutions,
                             const float* level_scales, const float* x01, const void* table, int table_half, int out_bf16,
                             int row_pitch, void* out, void* stream);
extern "C" int hugs_hashgrid_fwd(int n, int n_levels, int features, const long long* level_offsets,
                                 const int* level_resolutions, const float* level_scales, const float* x01,
                                 const float* table, int out_bf16, int row_pitch, void* out, void* stream) {
  return hashgrid_fwd_impl(n, n_levels, features, level_offsets, level_resolutions, level_scales, x01, table, 0, out_bf16, row_pitch, out, stream);
}
/* The same with the table given as an IEEE-half copy (table_dtype 2; 0 = fp32 as above): the fp16 mode's forward gathers
 * 4 / 8 bytes per corner (tiny-cuda-nn keeps half parameters under the reference's enable_amp). */
extern "C" int hugs_hashgrid_fwd_t(int n, int n_levels, int features, const long long* level_offsets,
                                   const int* level_resolutions, const float* level_scales, const float* x01,
                                   const void* table, int table_dtype, int out_dtype, int row_pitch, void* out, void* stream) {
  HUGS_REQUIRE(table_dtype == 0 || table_dtype == 2, -2, "hugs_hashgrid_fwd_t: table dtype %d (0 = fp32, 2 = fp16)", table_dtype);
  return hashgrid_fwd_impl(n, n_levels, features, level_offsets, level_resolutions, level_scales, x01, table, table_dtype == 2, out_dtype, row_pitch, out, stream);
}
static int hashgrid_fwd_impl(int n, int n_levels, int features, const long long* level_offsets, const int* level_resolutions,
                             const float* level_scales, const float* x01, const void* table, int table_half, int out_bf16,
                             int row_pitch, void* out, void* stream) {
  HUGS_REQUIRE(out_bf16 >= 0 && out_bf16 <= 2, -2, "hugs_hashgrid_fwd: output dtype %d", out_bf16);
  HgLevels lv;
  int rc = fill_levels(lv, n_levels, level_offsets, level_resolutions, level_scales, "hugs_hashgrid_fwd");
  if (rc) return rc;
  HUGS_REQUIRE(features == 2 || features == 4, -2, "hugs_hashgrid_fwd: %d features per level (2 or 4)", features);
  HUGS_REQUIRE(row_pitch >= n_levels * features, -2, "hugs_hashgrid_fwd: row pitch %d < %d", row_pitch, n_levels * features);
  if (n <= 0) return 0;
  const int nblk = (n + 255) / 256;
  const dim3 g((unsigned)nblk * (unsigned)((n_levels + HG_LG - 1) / HG_LG)), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (features == 2) {
    if (table_half) HG_BY_DTYPE_T(out_bf16, k_hashgrid_fwd, 2, true, <<<g, b, 0, st>>>(n, nblk, n_levels, lv, x01, table, row_pitch, out));
    else HG_BY_DTYPE_T(out_bf16, k_hashgrid_fwd, 2, false, <<<g, b, 0, st>>>(n, nblk, n_levels, lv, x01, table, row_pitch, out));
  } else {
    if (table_half) HG_BY_DTYPE_T(out_bf16, k_hashgrid_fwd, 4, true, <<<g, b, 0, st>>>(n, nblk, n_levels, lv, x01, table, row_pitch, out));
    else HG_BY_DTYPE_T(out_bf16, k_hashgrid_fwd, 4, false, <<<g, b, 0, st>>>(n, nblk, n_levels, lv, x01, table, row_pitch, out));
  }
  HUGS_CHECK_LAUNCH("k_hashgrid_fwd");
  return 0;
}

static int hashgrid_bwd_impl(int n, int n_levels, int features, const long long* level_offsets, const int* level_resolutions,
                             const float* level_scales, const float* x01, const void* d_out, int d_out_bf16, int row_pitch,
                             float* d_table_accum, void* ws, long long ws_bytes, void* stream);
extern "C" int hugs_hashgrid_bwd(int n, int n_levels, int features, const long long* level_offsets,
                                 const int* level_resolutions, const float* level_scales, const float* x01,
                                 const void* d_out, int d_out_bf16, int row_pitch, float* d_table_accum, void* stream) {
  return hashgrid_bwd_impl(n, n_levels, features, level_offsets, level_resolutions, level_scales, x01, d_out, d_out_bf16, row_pitch,
                           d_table_accum, nullptr, 0, stream);
}
/* The same with a caller-owned workspace (include/hugs.h): the levels behind the LDS-resident ones are then reduced by table slot
 * (hugs_hashgrid_binned.inc) instead of scattered with L2 atomics.  Measured not faster (profiles/r06_cfg5_hashgrid_levels.txt): the
 * Python layer calls it only under HUGS_HG_BINNED=1. */
extern "C" long long hugs_hashgrid_bwd_ws_bytes(int n, int n_levels, int features) {
  if (n <= 0 || n_levels <= 0 || features != 2) return 0;
  return 3ll * HG_TB_MAX * 4 + 12ll * 8 * (long long)n * n_levels;
}
extern "C" int hugs_hashgrid_bwd_ws(int n, int n_levels, int features, const long long* level_offsets,
                                    const int* level_resolutions, const float* level_scales, const float* x01,
                                    const void* d_out, int d_out_bf16, int row_pitch, float* d_table_accum, void* ws, long long ws_bytes,
                                    void* stream) {
  return hashgrid_bwd_impl(n, n_levels, features, level_offsets, level_resolutions, level_scales, x01, d_out, d_out_bf16, row_pitch,
                           d_table_accum, ws, ws_bytes, stream);
}
static int hashgrid_bwd_impl(int n, int n_levels, int features, const long long* level_offsets, const int* level_resolutions,
                             const float* level_scales, const float* x01, const void* d_out, int d_out_bf16, int row_pitch,
                             float* d_table_accum, void* ws, long long ws_bytes, void* stream) {
  HgLevels lv;
  int rc = fill_levels(lv, n_levels, level_offsets, level_resolutions, level_scales, "hugs_hashgrid_bwd");
  if (rc) return rc;
  HUGS_REQUIRE(features == 2 || features == 4, -2, "hugs_hashgrid_bwd: %d features per level (2 or 4)", features);
  if (n <= 0) return 0;
  const dim3 g((n + 255) / 256), b(256);
  hipStream_t st = (hipStream_t)stream;
  // the coarse levels through LDS while they are dense and fit (see k_hashgrid_bwd_lds); the global kernel starts behind them.
  // HUGS_HG_LDS_LEVELS = 0 / 1 / 2 (default 2): none / level 0 only (round 3) / every level that fits 156 KiB (round 5)
  static const int lds_levels = []() { const char* e = getenv("HUGS_HG_LDS_LEVELS"); return e ? atoi(e) : 2; }();
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = 256;
  int l_begin = 0;
  while (l_begin < n_levels && n >= 65536) {
    const unsigned long long r = (unsigned long long)lv.res[l_begin];
    const unsigned e = lv.off[l_begin + 1] - lv.off[l_begin];
    const long long fl = (long long)e * features;
    if (r * r * r > e) break;                                  // hashed: not a private copy's job
    if (fl <= HG_L0_MAX_FLOATS && lds_levels >= 1) {
      const int g0 = (int)(((n + 255) / 256) < 3 * ncu ? ((n + 255) / 256) : 3 * ncu);
#define HG_LDS_SMALL(F_) do { if (d_out_bf16 == 2) k_hashgrid_bwd_lds<F_, 2, HG_L0_MAX_FLOATS, 256><<<g0, 256, 0, st>>>(n, l_begin, lv, x01, d_out, row_pitch, d_table_accum); \
        else if (d_out_bf16) k_hashgrid_bwd_lds<F_, 1, HG_L0_MAX_FLOATS, 256><<<g0, 256, 0, st>>>(n, l_begin, lv, x01, d_out, row_pitch, d_table_accum); \
        else k_hashgrid_bwd_lds<F_, 0, HG_L0_MAX_FLOATS, 256><<<g0, 256, 0, st>>>(n, l_begin, lv, x01, d_out, row_pitch, d_table_accum); } while (0)
      if (features == 2) HG_LDS_SMALL(2); else HG_LDS_SMALL(4);
#undef HG_LDS_SMALL
    } else if (fl <= HG_L1_MAX_FLOATS && lds_levels >= 2) {
      const int g1 = (int)(((n + 1023) / 1024) < ncu ? ((n + 1023) / 1024) : ncu);
#define HG_LDS_BIG(F_) do { if (d_out_bf16 == 2) k_hashgrid_bwd_lds<F_, 2, HG_L1_MAX_FLOATS, 1024><<<g1, 1024, 0, st>>>(n, l_begin, lv, x01, d_out, row_pitch, d_table_accum); \
        else if (d_out_bf16) k_hashgrid_bwd_lds<F_, 1, HG_L1_MAX_FLOATS, 1024><<<g1, 1024, 0, st>>>(n, l_begin, lv, x01, d_out, row_pitch, d_table_accum); \
        else k_hashgrid_bwd_lds<F_, 0, HG_L1_MAX_FLOATS, 1024><<<g1, 1024, 0, st>>>(n, l_begin, lv, x01, d_out, row_pitch, d_table_accum); } while (0)
      if (features == 2) HG_LDS_BIG(2); else HG_LDS_BIG(4);
#undef HG_LDS_BIG
    } else {
      break;
    }
    HUGS_CHECK_LAUNCH("k_hashgrid_bwd_lds");
    ++l_begin;
  }
  if (l_begin >= n_levels) return 0;
  // the remaining levels as a segmented reduction by table slot when the caller brought a workspace (F = 2, batches worth four launches)
  if (ws && features == 2 && n >= 65536) {
    HgBins bn;
    uint32_t tb = 0;
    bool fits = true;
    for (int l = 0; l <= n_levels; ++l) {
      bn.first[l] = tb;
      if (l < n_levels && l >= l_begin) {
        const uint32_t e = lv.off[l + 1] - lv.off[l], nb = (e + HG_BIN - 1) >> HG_BIN_LOG;
        if (nb > HG_NB_MAX) fits = false;
        tb += nb;
      }
    }
    const long long nrec = 8ll * n * (n_levels - l_begin);
    if (fits && tb <= HG_TB_MAX && 3ll * HG_TB_MAX * 4 + 12ll * nrec <= ws_bytes && nrec < (1ll << 32)) {
      uint32_t* counts = (uint32_t*)ws;
      uint32_t* base = counts + HG_TB_MAX;
      uint32_t* cursor = base + HG_TB_MAX;
      uint32_t* rec_slot = cursor + HG_TB_MAX;
      float* rec_v0 = (float*)(rec_slot + nrec);
      float* rec_v1 = rec_v0 + nrec;
      HUGS_REQUIRE(hipMemsetAsync(counts, 0, HG_TB_MAX * sizeof(uint32_t), st) == hipSuccess, -100, "hugs_hashgrid_bwd_ws: hipMemsetAsync");
      const dim3 gs((n + HG_SC_THREADS - 1) / HG_SC_THREADS), bs(HG_SC_THREADS);
      if (d_out_bf16 == 0) k_hg_bin_count<0><<<gs, bs, 0, st>>>(n, n_levels, lv, bn, x01, d_out, row_pitch, l_begin, counts);
      else if (d_out_bf16 == 1) k_hg_bin_count<1><<<gs, bs, 0, st>>>(n, n_levels, lv, bn, x01, d_out, row_pitch, l_begin, counts);
      else k_hg_bin_count<2><<<gs, bs, 0, st>>>(n, n_levels, lv, bn, x01, d_out, row_pitch, l_begin, counts);
      k_hg_bin_scan<<<1, 1024, 0, st>>>((int)tb, counts, base, cursor);
      if (d_out_bf16 == 0) k_hg_bin_scatter<0><<<gs, bs, 0, st>>>(n, n_levels, lv, bn, x01, d_out, row_pitch, l_begin, cursor, rec_slot, rec_v0, rec_v1);
      else if (d_out_bf16 == 1) k_hg_bin_scatter<1><<<gs, bs, 0, st>>>(n, n_levels, lv, bn, x01, d_out, row_pitch, l_begin, cursor, rec_slot, rec_v0, rec_v1);
      else k_hg_bin_scatter<2><<<gs, bs, 0, st>>>(n, n_levels, lv, bn, x01, d_out, row_pitch, l_begin, cursor, rec_slot, rec_v0, rec_v1);
      k_hg_bin_reduce<<<tb, 1024, 0, st>>>(n_levels, lv, bn, l_begin, counts, base, rec_slot, rec_v0, rec_v1, d_table_accum);
      HUGS_CHECK_LAUNCH("hugs_hashgrid_bwd_ws(binned)");
      return 0;
    }
  }
  if (features == 2) {
    HG_BY_DTYPE(d_out_bf16, k_hashgrid_bwd, 2, <<<g, b, 0, st>>>(n, n_levels, lv, x01, d_out, row_pitch, d_table_accum, l_begin));
  } else {
    HG_BY_DTYPE(d_out_bf16, k_hashgrid_bwd, 4, <<<g, b, 0, st>>>(n, n_levels, lv, x01, d_out, row_pitch, d_table_accum, l_begin));
  }
  HUGS_CHECK_LAUNCH("k_hashgrid_bwd");
  return 0;
}

extern "C" int hugs_sh4_fwd(int n, const float* dirs01, int out_bf16, int row_pitch, int col0, void* out, void* stream) {
  HUGS_REQUIRE(row_pitch >= col0 + 16 && col0 >= 0, -2, "hugs_sh4_fwd: row pitch %d, first column %d", row_pitch, col0);
  if (n <= 0) return 0;
  switch (out_bf16) {
    case 0: k_sh4<0><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, dirs01, row_pitch, col0, out); break;
    case 1: k_sh4<1><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, dirs01, row_pitch, col0, out); break;
    default: k_sh4<2><<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, dirs01, row_pitch, col0, out); break;
  }
  HUGS_CHECK_LAUNCH("k_sh4");
  return 0;
}

/* 2-D hash grid of per-RAY image coordinates (nerfacto HA-NeRF ImplicitMask, nerfacto.py:1036-1047,1080-1091): writes the
 * mask MLP's whole input row out[n, :] = [grid(x01[n]) (n_levels*features) | extra[n, :T] | 0 ... (row_pitch)]. */
extern "C" int hugs_hashgrid2d_fwd(int n, int n_levels, int features, const long long* level_offsets,
                                   const int* level_resolutions, const float* level_scales, const float* x01,
                                   const float* table, const float* extra, int T, int out_bf16, int row_pitch, void* out,
                                   void* stream) {
  HgLevels lv;
  int rc = fill_levels(lv, n_levels, level_offsets, level_resolutions, level_scales, "hugs_hashgrid2d_fwd");
  if (rc) return rc;
  HUGS_REQUIRE(features == 2 || features == 4, -2, "hugs_hashgrid2d_fwd: %d features per level (2 or 4)", features);
  HUGS_REQUIRE(T >= 0 && (T == 0 || extra) && row_pitch >= n_levels * features + T, -2, "hugs_hashgrid2d_fwd: row pitch %d < %d + %d",
               row_pitch, n_levels * features, T);
  if (n <= 0) return 0;
  const dim3 g((n + 255) / 256), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (features == 2) {
    HG_BY_DTYPE(out_bf16, k_hashgrid2d_fwd, 2, <<<g, b, 0, st>>>(n, n_levels, lv, x01, table, extra, T, row_pitch, out));
  } else {
    HG_BY_DTYPE(out_bf16, k_hashgrid2d_fwd, 4, <<<g, b, 0, st>>>(n, n_levels, lv, x01, table, extra, T, row_pitch, out));
  }
  HUGS_CHECK_LAUNCH("k_hashgrid2d_fwd");
  return 0;
}

extern "C" int hugs_hashgrid2d_bwd(int n, int n_levels, int features, const long long* level_offsets,
                                   const int* level_resolutions, const float* level_scales, const float* x01,
                                   const void* d_out, int d_out_bf16, int row_pitch, float* d_table_accum, void* stream) {
  HgLevels lv;
  int rc = fill_levels(lv, n_levels, level_offsets, level_resolutions, level_scales, "hugs_hashgrid2d_bwd");
  if (rc) return rc;
  HUGS_REQUIRE(features == 2 || features == 4, -2, "hugs_hashgrid2d_bwd: %d features per level (2 or 4)", features);
  if (n <= 0) return 0;
  const dim3 g((n + 255) / 256), b(256);
  hipStream_t st = (hipStream_t)stream;
  if (features == 2) {
    HG_BY_DTYPE(d_out_bf16, k_hashgrid2d_bwd, 2, <<<g, b, 0, st>>>(n, n_levels, lv, x01, d_out, row_pitch, d_table_accum));
  } else {
    HG_BY_DTYPE(d_out_bf16, k_hashgrid2d_bwd, 4, <<<g, b, 0, st>>>(n, n_levels, lv, x01, d_out, row_pitch, d_table_accum));
  }
  HUGS_CHECK_LAUNCH("k_hashgrid2d_bwd");
  return 0;
}
