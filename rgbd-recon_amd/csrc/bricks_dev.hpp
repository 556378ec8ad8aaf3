// Device-side mark_brick(): glsl/inc_bricks.glsl:40-58, shared by the standalone marking kernel and the normal pass.
#pragma once
#include "sampling.hpp"

namespace rr {

__device__ __forceinline__ float sgnf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

// counters[id] += (number of lanes holding id), one atomic per distinct id in the wave.  Neighbouring pixels of a
// row fall into the same few bricks, so a wave issues a handful of atomics instead of up to 64 contended ones.
// Must be called by every lane of the wave.
__device__ __forceinline__ void wave_count(uint32_t* __restrict__ counters, uint32_t id, bool valid) {
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t lid = __shfl(id, leader);
    const unsigned long long same = __ballot(valid && id == lid) & todo;
    if (lane == leader) atomicAdd(&counters[lid], (uint32_t)__popcll(same));
    todo &= ~same;
  }
}

// Which counters mark_brick(pos) increments: the brick containing pos (+1) and its neighbour across the face nearest
// to pos along the dominant axis of (pos - brick centre) (+1 if |dx| > 0.1 * brick; the shader tests d_abs.x whatever
// the dominant axis).  Positions outside the grid (UB in the shader) mark nothing.
__device__ __forceinline__ void mark_brick_ids(const Bricks& B, float3 pos, bool& own, uint32_t& id_own, bool& nbr, uint32_t& id_nbr) {
  own = nbr = false;
  const float relx = pos.x - B.bbox_min[0], rely = pos.y - B.bbox_min[1], relz = pos.z - B.bbox_min[2];
  const float fx = floorf(relx / B.size[0]), fy = floorf(rely / B.size[1]), fz = floorf(relz / B.size[2]);
  if (!(fx >= 0.0f && fy >= 0.0f && fz >= 0.0f && fx < (float)B.res[0] && fy < (float)B.res[1] && fz < (float)B.res[2])) return;
  const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
  const float cx = (float)ix * B.size[0] + B.bbox_min[0] + 0.5f * B.size[0];
  const float cy = (float)iy * B.size[1] + B.bbox_min[1] + 0.5f * B.size[1];
  const float cz = (float)iz * B.size[2] + B.bbox_min[2] + 0.5f * B.size[2];
  const float dx = pos.x - cx, dy = pos.y - cy, dz = pos.z - cz;
  const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
  const float mv = fmaxf(ax, fmaxf(ay, az));
  const int ox = (int)sgnf(dx * (ax < mv ? 0.0f : 1.0f));
  const int oy = (int)sgnf(dy * (ay < mv ? 0.0f : 1.0f));
  const int oz = (int)sgnf(dz * (az < mv ? 0.0f : 1.0f));
  const int nbx = clampi(ix + ox, 0, B.res[0] - 1), nby = clampi(iy + oy, 0, B.res[1] - 1), nbz = clampi(iz + oz, 0, B.res[2] - 1);
  nbr = ax > B.size[0] * 0.1f;                                          // the neighbour add is 0 otherwise (:52)
  id_nbr = (uint32_t)(((size_t)nbz * B.res[1] + nby) * B.res[0] + nbx);
  own = true;
  id_own = (uint32_t)(((size_t)iz * B.res[1] + iy) * B.res[0] + ix);
}

}  // namespace rr
