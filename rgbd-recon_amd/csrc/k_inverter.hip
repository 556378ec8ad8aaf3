// Inverse calibration volume builder (gfx950), SURVEY.md section 8 f3.
// CalibrationInverter::calculateInverseVolumes (framework/calibration/calibration_inverter.cpp:68-115): for every voxel
// centre of the output grid that lies inside the sensor's frustum, the 8 nearest forward-LUT samples (CGAL kd-tree in the
// reference, nearest_neighbour_search.cpp:13-43) are blended by inverse distance (:57-66) into a normalised LUT coordinate.
//
// MI355X design: the kd-tree is replaced by a counting-sorted uniform grid over the samples and an EXACT expanding-shell
// search, one thread per output voxel:
//   * squared distances in double over the float coordinates and the (distance, enumeration index) total order are those
//     of the oracle, so the selected 8 and their blending order are identical -> results are bit-exact;
//   * the top-8 list lives in registers (unrolled compare/swap chain, no scratch);
//   * a shell ends the search once the 8th distance is strictly below the distance to the unsearched region.
// ~25 M queries x a few dozen candidate samples each: milliseconds, bound by L2 hits on the sorted samples.
#include "tsdf_common.hpp"

namespace rr {

__device__ __forceinline__ int cell_axis(double s, double gmin, double inv, int g) {
  int c = (int)floor((s - gmin) * inv);
  return c < 0 ? 0 : (c >= g ? g - 1 : c);
}

// enumeration index of getXyzSamples (calibration_inverter.cpp:40-55: x outer, z inner) from the volume's x-fastest index
__device__ __forceinline__ uint32_t enum_index(uint32_t v, const InverterGrid& G) {
  const uint32_t x = v % G.rx, y = (v / G.rx) % G.ry, z = v / (G.rx * G.ry);
  return (x * G.ry + y) * G.rz + z;
}

__global__ __launch_bounds__(256) void k_inv_count(InverterGrid G, const float* __restrict__ xyz, uint32_t* __restrict__ count) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= G.n) return;
  const int cx = cell_axis(xyz[3 * v], G.gmin[0], G.inv[0], G.g[0]), cy = cell_axis(xyz[3 * v + 1], G.gmin[1], G.inv[1], G.g[1]),
            cz = cell_axis(xyz[3 * v + 2], G.gmin[2], G.inv[2], G.g[2]);
  atomicAdd(&count[((size_t)cz * G.g[1] + cy) * G.g[0] + cx], 1u);
}

// exclusive scan of `count` in three launches: per-block scan of 2048 items, scan of the block sums, add
__global__ __launch_bounds__(256) void k_inv_scan_blocks(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ sums, uint32_t n) {
  __shared__ uint32_t s[256];
  const uint32_t base = blockIdx.x * 2048 + threadIdx.x * 8;
  uint32_t v[8], t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = base + i < n ? in[base + i] : 0u; t += v[i]; }
  s[threadIdx.x] = t;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t add = threadIdx.x >= (unsigned)off ? s[threadIdx.x - off] : 0u;
    __syncthreads();
    s[threadIdx.x] += add;
    __syncthreads();
  }
  uint32_t run = s[threadIdx.x] - t;
#pragma unroll
  for (int i = 0; i < 8; ++i) { if (base + i < n) out[base + i] = run; run += v[i]; }
  if (threadIdx.x == 255) sums[blockIdx.x] = s[255];
}
__global__ __launch_bounds__(1024) void k_inv_scan_sums(uint32_t* __restrict__ sums, uint32_t nb) {
  __shared__ uint32_t s[1024];
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nb; base += 1024) {
    const uint32_t i = base + threadIdx.x, t = i < nb ? sums[i] : 0u;
    s[threadIdx.x] = t;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const uint32_t add = threadIdx.x >= (unsigned)off ? s[threadIdx.x - off] : 0u;
      __syncthreads();
      s[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nb) sums[i] = carry + s[threadIdx.x] - t;
    carry += s[1023];
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_inv_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ sums, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] += sums[i / 2048];
}

__global__ __launch_bounds__(256) void k_inv_scatter(InverterGrid G, const float* __restrict__ xyz, const uint32_t* __restrict__ start,
                                                     uint32_t* __restrict__ fill, float4* __restrict__ sorted) {
  const uint32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= G.n) return;
  const float x = xyz[3 * v], y = xyz[3 * v + 1], z = xyz[3 * v + 2];
  const int cx = cell_axis(x, G.gmin[0], G.inv[0], G.g[0]), cy = cell_axis(y, G.gmin[1], G.inv[1], G.g[1]), cz = cell_axis(z, G.gmin[2], G.inv[2], G.g[2]);
  const size_t c = ((size_t)cz * G.g[1] + cy) * G.g[0] + cx;
  const uint32_t slot = start[c] + atomicAdd(&fill[c], 1u);
  sorted[slot] = make_float4(x, y, z, __uint_as_float(enum_index(v, G)));
}

struct Best { double d; uint32_t i; };
__device__ __forceinline__ bool before(double d, uint32_t i, const Best& b) { return d < b.d || (d == b.d && i < b.i); }

__global__ __launch_bounds__(256) void k_inv_query(InverterGrid G, InverterQuery Q, const float* __restrict__ xyz, const uint32_t* __restrict__ start,
                                                   const float4* __restrict__ sorted, float4* __restrict__ out) {
  const uint32_t x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), z = blockIdx.z;
  if (x >= Q.res[0] || y >= Q.res[1]) return;
  const size_t o = ((size_t)z * Q.res[1] + y) * Q.res[0] + x;
  // sample_pos = sample_start + fvec3(x, y, z) * sample_step, :84
  const float px = Q.start[0] + (float)x * Q.step[0], py = Q.start[1] + (float)y * Q.step[1], pz = Q.start[2] + (float)z * Q.step[2];
#pragma unroll
  for (int k = 0; k < 6; ++k)                                             // Frustum::inside, frustum.cpp: vec4 dot = (x+y)+(z+w)
    if ((Q.plane[k][0] * px + Q.plane[k][1] * py) + (Q.plane[k][2] * pz + Q.plane[k][3] * 1.0f) < 0.0f) { out[o] = make_float4(-1.0f, -1.0f, -1.0f, -1.0f); return; }

  const double qx = px, qy = py, qz = pz;
  const int cq[3] = {cell_axis(qx, G.gmin[0], G.inv[0], G.g[0]), cell_axis(qy, G.gmin[1], G.inv[1], G.g[1]), cell_axis(qz, G.gmin[2], G.inv[2], G.g[2])};
  Best b0, b1, b2, b3, b4, b5, b6, b7;
  b0.d = b1.d = b2.d = b3.d = b4.d = b5.d = b6.d = b7.d = INFINITY;
  b0.i = b1.i = b2.i = b3.i = b4.i = b5.i = b6.i = b7.i = 0xffffffffu;
  uint32_t found = 0;
  const int rmax = max(G.g[0], max(G.g[1], G.g[2]));
  for (int r = 0; r <= rmax; ++r) {
    for (int dz = -r; dz <= r; ++dz) {
      const int cz = cq[2] + dz;
      if (cz < 0 || cz >= G.g[2]) continue;
      for (int dy = -r; dy <= r; ++dy) {
        const int cy = cq[1] + dy;
        if (cy < 0 || cy >= G.g[1]) continue;
        const bool face = (dz == -r || dz == r || dy == -r || dy == r);   // whole x row belongs to the shell, else only its two ends
        const int xstep = face || r == 0 ? 1 : 2 * r;
        for (int dx = -r; dx <= r; dx += xstep) {
          const int cx = cq[0] + dx;
          if (cx < 0 || cx >= G.g[0]) continue;
          const size_t c = ((size_t)cz * G.g[1] + cy) * G.g[0] + cx;
          const uint32_t s0 = start[c], s1 = start[c + 1];
          for (uint32_t s = s0; s < s1; ++s) {
            const float4 t = sorted[s];
            const double ex = qx - (double)t.x, ey = qy - (double)t.y, ez = qz - (double)t.z;
            double d = 0.0; d += ex * ex; d += ey * ey; d += ez * ez;
            const uint32_t id = __float_as_uint(t.w);
            ++found;
            if (!before(d, id, b7)) continue;
            b7.d = d; b7.i = id;
#define RR_BUBBLE(lo, hi) if (before(hi.d, hi.i, lo)) { const Best tmp = lo; lo = hi; hi = tmp; }
            RR_BUBBLE(b6, b7) RR_BUBBLE(b5, b6) RR_BUBBLE(b4, b5) RR_BUBBLE(b3, b4) RR_BUBBLE(b2, b3) RR_BUBBLE(b1, b2) RR_BUBBLE(b0, b1)
#undef RR_BUBBLE
          }
        }
      }
    }
    // everything within Chebyshev radius r of the query's cell is searched: lower bound on the distance to the rest
    double bound = INFINITY;
    bool more = false;
    const double q[3] = {qx, qy, qz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (cq[a] - r > 0) { more = true; bound = fmin(bound, fmax(0.0, q[a] - (G.gmin[a] + (double)(cq[a] - r) * G.cell[a]))); }
      if (cq[a] + r < G.g[a] - 1) { more = true; bound = fmin(bound, fmax(0.0, (G.gmin[a] + (double)(cq[a] + r + 1) * G.cell[a]) - q[a])); }
    }
    if (!more) break;
    if (found >= 8u && b7.d < bound * bound * (1.0 - 1e-9)) break;
  }

  // inverseDistance(), :57-66, neighbours in ascending (distance, enumeration index)
  float tw = 0.0f, wx = 0.0f, wy = 0.0f, wz = 0.0f;
  const Best* bs[8] = {&b0, &b1, &b2, &b3, &b4, &b5, &b6, &b7};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t id = bs[k]->i;
    if (id == 0xffffffffu) continue;
    const uint32_t iz = id % G.rz, iy = (id / G.rz) % G.ry, ix = id / (G.rz * G.ry);
    const float* t = xyz + 3 * (((size_t)iz * G.ry + iy) * G.rx + ix);
    const float ex = t[0] - px, ey = t[1] - py, ez = t[2] - pz;          // glm::distance(p0, p1) = length(p1 - p0)
    const float w = 1.0f / sqrtf(ex * ex + ey * ey + ez * ez);
    wx += w * (float)ix; wy += w * (float)iy; wz += w * (float)iz;
    tw += w;
  }
  wx /= tw; wy /= tw; wz /= tw;
  out[o] = make_float4((wx + 0.5f) / (float)G.rx, (wy + 0.5f) / (float)G.ry, (wz + 0.5f) / (float)G.rz, 1.0f);   // :95
}

void launch_inverter_build(hipStream_t st, const InverterGrid& G, const float* xyz, uint32_t* count, uint32_t* start, uint32_t* sums, float4* sorted) {
  const uint32_t cells = (uint32_t)G.g[0] * G.g[1] * G.g[2], nb = (cells + 1 + 2047) / 2048;
  (void)hipMemsetAsync(count, 0, (size_t)(cells + 1) * sizeof(uint32_t), st);
  hipLaunchKernelGGL(k_inv_count, dim3((G.n + 255) / 256), dim3(256), 0, st, G, xyz, count);
  hipLaunchKernelGGL(k_inv_scan_blocks, dim3(nb), dim3(256), 0, st, count, start, sums, cells + 1);
  hipLaunchKernelGGL(k_inv_scan_sums, dim3(1), dim3(1024), 0, st, sums, nb);
  hipLaunchKernelGGL(k_inv_scan_add, dim3((cells + 1 + 255) / 256), dim3(256), 0, st, start, sums, cells + 1);
  (void)hipMemsetAsync(count, 0, (size_t)(cells + 1) * sizeof(uint32_t), st);
  hipLaunchKernelGGL(k_inv_scatter, dim3((G.n + 255) / 256), dim3(256), 0, st, G, xyz, start, count, sorted);
}
void launch_inverter_query(hipStream_t st, const InverterGrid& G, const InverterQuery& Q, const float* xyz, const uint32_t* start, const float4* sorted, float4* out) {
  hipLaunchKernelGGL(k_inv_query, dim3((Q.res[0] + 63) / 64, (Q.res[1] + 3) / 4, Q.res[2]), dim3(256), 0, st, G, Q, xyz, start, sorted, out);
}

}  // namespace rr
