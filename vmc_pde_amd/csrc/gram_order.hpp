// Workgroup -> (sample split, lower-triangle tile) mapping shared by the fp64 Gram (gram.hip) and the int8 experiment (gram_i8.hip).
#pragma once
#include <algorithm>
#include <utility>
#include <vector>

#include "mfma_tile.hpp"

namespace vmcpde {

// XCD-aware mapping of a workgroup to its (sample split, tile) unit.  Workgroups are dealt round-robin over the 8 XCDs
// (blockIdx % 8 share an L2) and started in blockIdx order, so slot = blockIdx / 8 is "time".  order 1 (default): all XCDs
// sweep the SAME sample chunk at the same time -- slot -> (split, local tile), XCD x owns a contiguous run of that split's
// tiles -- which keeps the chip-wide working set at one or two chunks of J (Infinity-Cache sized) and neighbouring tiles
// (shared operand panels) on one L2.  order 0: one sample chunk per XCD (eight chunks in flight).
// order 2: the tiles are laid along a Z-order (Morton) curve through the lower triangle (perm: position -> tile index) and every
// XCD owns a contiguous piece of the curve in every sample split, so the 64 workgroups resident on an XCD work on a compact
// 2-D patch of tiles and share ~16 operand panels per K-step in that XCD's L2 instead of ~26 for two or three whole tile rows.
// The pieces are balanced: with ntiles = 8 q + r every XCD gets q tiles per split and the r left-over tiles rotate over the
// XCDs from split to split (XCDs (s r + i) mod 8, i < r), so the per-XCD totals differ by at most one unit (order 1 gives
// seven XCDs ceil(ntiles/8) tiles and the last one the rest: 71 against 64 at C2, i.e. 1.2 % more rounds than necessary).
__device__ __forceinline__ int xcd_extra(int xcd, int split, int r) { return (((xcd - split * r) % 8) + 8) % 8 < r ? 1 : 0; }

__device__ __forceinline__ bool gram_unit(int ntiles, int nsplit, int total_units, int units_per_xcd, int order, const int* __restrict__ perm,
                                          int& split, int& tile) {
  if (order == 2) {
    const int xcd = (int)(blockIdx.x & 7), q = ntiles >> 3, r = ntiles & 7;
    int rem = (int)(blockIdx.x >> 3), s = 0;
    for (; s < nsplit; ++s) {
      const int cnt = q + xcd_extra(xcd, s, r);
      if (rem < cnt) break;
      rem -= cnt;
    }
    if (s == nsplit) return false;
    int start = q * xcd;
    for (int x = 0; x < xcd; ++x) start += xcd_extra(x, s, r);
    split = s;
    tile = perm[start + rem];
    return true;
  }
  if (order == 1) {
    const int slot = (int)(blockIdx.x >> 3), xcd = (int)(blockIdx.x & 7);
    split = slot / units_per_xcd;                      // units_per_xcd = tiles per XCD per split here
    tile = xcd * units_per_xcd + (slot - split * units_per_xcd);
    return tile < ntiles && split < nsplit;
  }
  const int unit = (int)(blockIdx.x & 7) * units_per_xcd + (int)(blockIdx.x >> 3);
  if (unit >= total_units) return false;
  split = unit / ntiles;
  tile = unit - split * ntiles;
  return true;
}


// host side of order 2: the tiles of the lower triangle along a Morton curve (position -> tile index ti (ti + 1) / 2 + tj) ...
inline std::vector<int> morton_tile_order(int nb) {
  std::vector<std::pair<uint64_t, int>> keyed;
  keyed.reserve((size_t)nb * (nb + 1) / 2);
  auto spread = [](uint32_t v) { uint64_t x = v; x = (x | (x << 16)) & 0x0000FFFF0000FFFFull; x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
                                 x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full; x = (x | (x << 2)) & 0x3333333333333333ull;
                                 x = (x | (x << 1)) & 0x5555555555555555ull; return x; };
  for (int ti = 0; ti < nb; ++ti)
    for (int tj = 0; tj <= ti; ++tj) keyed.push_back({(spread((uint32_t)ti) << 1) | spread((uint32_t)tj), ti * (ti + 1) / 2 + tj});
  std::sort(keyed.begin(), keyed.end());
  std::vector<int> host(keyed.size());
  for (size_t i = 0; i < keyed.size(); ++i) host[i] = keyed[i].second;
  return host;
}

// ... and the slots of the busiest XCD under gram_unit's balanced deal (grid = 8 x this)
inline int order2_slots_per_xcd(int ntiles, int nsplit) {
  const int q = ntiles / 8, r = ntiles % 8;
  int most = 0;
  for (int x = 0; x < 8; ++x) {
    int cnt = 0;
    for (int sp = 0; sp < nsplit; ++sp) cnt += q + (((((x - sp * r) % 8) + 8) % 8 < r) ? 1 : 0);
    most = cnt > most ? cnt : most;
  }
  return most;
}

// the device copy of the Morton order for `nb` row blocks, cached per GPU (call under DeviceCtxLock); returns 0 or a negative error code
inline int ensure_gram_perm(DeviceCtx* ctx, int nb, const int** perm) {
  auto it = ctx->gram_perms.find(nb);
  if (it == ctx->gram_perms.end()) {       // first use of this matrix size on this GPU (a blocking upload, once)
    const std::vector<int> host = morton_tile_order(nb);
    int* dev = nullptr;
    VMC_CHECK_HIP(hipMalloc(&dev, host.size() * sizeof(int)));
    VMC_CHECK_HIP(hipMemcpy(dev, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice));
    it = ctx->gram_perms.emplace(nb, dev).first;
  }
  *perm = it->second;
  return 0;
}

}  // namespace vmcpde
