// Wave-level reductions shared by the chains of eig.hip and the Jacobi sweeps of jacobi.hip (gfx950, 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

namespace vmcpde {

// Sum over the 64 lanes, every lane returns it.  Six DPP steps (quad permutes, row rotations, row broadcasts: ~200 cycles) and a
// v_readlane of lane 63 instead of six ds_bpermute exchanges (~1100 cycles: __shfl_xor goes through the LDS crossbar on this chip).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0xb1>(v);       // quad_perm [1,0,3,2]
  v += dpp_move<0x4e>(v);       // quad_perm [2,3,0,1]
  v += dpp_move<0x124>(v);      // row_ror:4
  v += dpp_move<0x128>(v);      // row_ror:8   -- every lane of a row of 16 holds the row's sum
  v += dpp_move<0x142>(v);      // row_bcast:15: lane 15 of each row to the next row (rows without a source add 0)
  v += dpp_move<0x143>(v);      // row_bcast:31: lane 31 to rows 2 and 3 -- lane 63 now holds the sum of all four rows
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

}  // namespace vmcpde
