// camcal_layout_device: the tile layout of include/camcal.h (camcal_layout) for view sets whose CSR offsets
// and points already live on the device.  The tables are what the host form writes -- views sorted by
// descending corner count (stable), 64 slots per tile -- so the packed set and everything downstream is the
// same; only the 20-odd framework launches and two host synchronisations that used to produce them for a
// large set are gone (100 000 views: 0.56 ms of a 3.5 ms solve).  The sort is rocPRIM's radix sort (a plain
// library sort, like a plain GEMM would be hipBLASLt's); the rest are the small kernels below.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/camcal.h"
#include "camcal_math.h"

namespace {
using camcal::LANES;
using camcal::PT_COMP;
constexpr int LTPB = 256;
constexpr int SCAN_TPB = 1024;

#define HIP_TRY(x)                                  \
  do {                                              \
    hipError_t e_ = (x);                            \
    if (e_ != hipSuccess) return (int)e_;           \
  } while (0)

enum : long long { HEAD_NOT_PLANAR = CAMCAL_LAYOUT_NOT_PLANAR, HEAD_BAD_COUNT = CAMCAL_LAYOUT_BAD_COUNT, HEAD_BAD_TOTAL = CAMCAL_LAYOUT_BAD_TOTAL };

// keys = corner counts, vals = view indices; a count outside [0, 2^31) or offsets that do not end at the
// caller's number of points are flagged, not sorted on
__global__ __launch_bounds__(LTPB) void k_layout_keys(int64_t n_views, const int64_t* __restrict__ view_ptr, int64_t n_obs,
                                                      int32_t* __restrict__ keys, int32_t* __restrict__ vals,
                                                      unsigned long long* __restrict__ head) {
  const int64_t stride = (int64_t)gridDim.x * LTPB;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * LTPB + threadIdx.x; i < n_views; i += stride) {
    const int64_t n = view_ptr[i + 1] - view_ptr[i];
    const bool b = n < 0 || n > 2147483647LL;
    bad |= b;
    keys[i] = b ? 0 : (int32_t)n;
    vals[i] = (int32_t)i;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t total = view_ptr[n_views] - view_ptr[0];
    head[0] = (unsigned long long)total;
    if (total != n_obs || view_ptr[0] != 0) atomicOr(&head[2], (unsigned long long)HEAD_BAD_TOTAL);
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&head[2], (unsigned long long)HEAD_BAD_COUNT);
}

// Z != 0 anywhere (NaN counts): the set is not a planar target
__global__ __launch_bounds__(LTPB) void k_layout_planar(int64_t n_obs, const float* __restrict__ obj_xyz,
                                                        unsigned long long* __restrict__ head) {
  const int64_t stride = (int64_t)gridDim.x * LTPB;
  bool off = false;
  for (int64_t i = (int64_t)blockIdx.x * LTPB + threadIdx.x; i < n_obs; i += stride) off |= !(obj_xyz[3 * i + 2] == 0.0f);
  if (__ballot(off) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&head[2], (unsigned long long)HEAD_NOT_PLANAR);
}

// One workgroup: padding slots behind the sorted views, tile_nmax (the first slot of a tile holds its
// largest count: the slots are sorted), tile_off = exclusive prefix sum of the tile sizes, the totals.
__global__ __launch_bounds__(SCAN_TPB) void k_layout_tiles(int64_t n_views, int64_t n_slots, int32_t n_tiles,
                                                           int32_t* __restrict__ slot_view, int32_t* __restrict__ lane_n,
                                                           int32_t* __restrict__ tile_nmax, int64_t* __restrict__ tile_off,
                                                           unsigned long long* __restrict__ head) {
  __shared__ int64_t s_wave[SCAN_TPB / 64];
  __shared__ int64_t s_carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int64_t s = n_views + tid; s < n_slots; s += SCAN_TPB) {
    slot_view[s] = -1;
    lane_n[s] = 0;
  }
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int32_t t0 = 0; t0 < n_tiles; t0 += SCAN_TPB) {
    const int32_t t = t0 + tid;
    int32_t nmax = 0;
    if (t < n_tiles && (int64_t)t * LANES < n_views) nmax = lane_n[(int64_t)t * LANES];
    const int64_t size = (int64_t)nmax * PT_COMP * LANES;
    int64_t incl = size;                               // inclusive scan inside the wavefront
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int64_t up = __shfl_up(incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int64_t before = s_carry;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    if (t < n_tiles) {
      tile_nmax[t] = nmax;
      tile_off[t] = before + incl - size;
    }
    __syncthreads();
    if (tid == SCAN_TPB - 1) s_carry = before + incl;
    __syncthreads();
  }
  if (tid == 0) {
    head[1] = (unsigned long long)s_carry;
    head[3] = n_views > 0 ? (unsigned long long)lane_n[0] : 0ull;
  }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int sort_storage_bytes(int64_t n_views, size_t* bytes) {
  *bytes = 0;
  if (n_views <= 0) return 0;
  const int32_t* k = nullptr;
  int32_t* ko = nullptr;
  HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, *bytes, k, ko, k, ko, (size_t)n_views, 0, 32, (hipStream_t) nullptr));
  return 0;
}
}  // namespace

extern "C" {

int64_t camcal_layout_device_temp_len(int64_t n_views) {
  if (n_views < 0 || n_views > 2147483647LL - LANES) return CAMCAL_E_TOOBIG;
  size_t sort_bytes = 0;
  const int rc = sort_storage_bytes(n_views, &sort_bytes);
  if (rc) return rc > 0 ? -rc - 1000 : rc;
  return (int64_t)(2 * align256((size_t)n_views * sizeof(int32_t)) + align256(sort_bytes) + 256);
}

int camcal_layout_device(int64_t n_views, const int64_t* view_ptr, int64_t n_obs, const float* obj_xyz,
                         int32_t* slot_view, int32_t* lane_n, int32_t* tile_nmax, int64_t* tile_off,
                         int64_t* head, void* temp, int64_t temp_bytes, void* stream) {
  if (n_views < 0 || n_obs < 0 || !view_ptr || !slot_view || !lane_n || !tile_nmax || !tile_off || !head) return CAMCAL_E_BADARG;
  if (n_views > 2147483647LL - LANES) return CAMCAL_E_TOOBIG;
  const int64_t need = camcal_layout_device_temp_len(n_views);
  if (need < 0) return (int)need;
  if (n_views > 0 && (!temp || temp_bytes < need)) return CAMCAL_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  const int64_t T = n_views > 0 ? (n_views + LANES - 1) / LANES : 1;
  const int64_t S = T * LANES;
  unsigned long long* h = (unsigned long long*)head;
  HIP_TRY(hipMemsetAsync(head, 0, 4 * sizeof(int64_t), s));
  char* base = (char*)(((uintptr_t)temp + 255) & ~(uintptr_t)255);
  int32_t* keys = (int32_t*)base;
  int32_t* vals = (int32_t*)(base + align256((size_t)n_views * sizeof(int32_t)));
  void* sort_tmp = base + 2 * align256((size_t)n_views * sizeof(int32_t));
  auto blocks = [](int64_t n) { return (unsigned)((n + LTPB - 1) / LTPB < 4096 ? (n + LTPB - 1) / LTPB : 4096); };
  hipLaunchKernelGGL(k_layout_keys, dim3(n_views > 0 ? blocks(n_views) : 1), dim3(LTPB), 0, s, n_views, view_ptr, n_obs, keys, vals, h);
  HIP_TRY(hipGetLastError());
  if (n_views > 0) {
    size_t sort_bytes = 0;
    const int rc = sort_storage_bytes(n_views, &sort_bytes);
    if (rc) return rc;
    // sorted counts and view indices land in the tables themselves (their first n_views entries)
    HIP_TRY(rocprim::radix_sort_pairs_desc(sort_tmp, sort_bytes, (const int32_t*)keys, lane_n, (const int32_t*)vals, slot_view,
                                           (size_t)n_views, 0, 32, s));
  }
  hipLaunchKernelGGL(k_layout_tiles, dim3(1), dim3(SCAN_TPB), 0, s, n_views, S, (int32_t)T, slot_view, lane_n, tile_nmax, tile_off, h);
  HIP_TRY(hipGetLastError());
  if (obj_xyz && n_obs > 0) {
    hipLaunchKernelGGL(k_layout_planar, dim3(blocks(n_obs)), dim3(LTPB), 0, s, n_obs, obj_xyz, h);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}


int camcal_points_off_plane(int64_t n_obs, const float* obj_xyz, int64_t* flag, void* stream) {
  if (n_obs < 0 || !flag || (n_obs > 0 && !obj_xyz)) return CAMCAL_E_BADARG;
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemsetAsync(flag, 0, 4 * sizeof(int64_t), s));
  if (n_obs == 0) return 0;
  const int64_t nb = (n_obs + LTPB - 1) / LTPB;
  hipLaunchKernelGGL(k_layout_planar, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(LTPB), 0, s, n_obs, obj_xyz,
                     (unsigned long long*)flag);
  HIP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
