// radix_sort.hip -- stable LSD radix sort of (key, int32 value) pairs on a bit range.
// Stands in for cuda_lib.radix_sort_pairs (reference cuda_lib/radix_sort_pairs.cu:8-70, CUB
// DeviceRadixSort) for callers that use that primitive directly (e.g. misc/morton_sort.py) and
// for the reference-shaped mapper path used to cross-check the fused per-tile sort.
//
// wave64 design, 8-bit digits, three kernels per digit pass:
//   histogram : each WAVE owns a contiguous chunk of 1024 pairs (16 rows of 64) and counts its
//               digits in a wave-private LDS table -> hist[digit][wave] (digit-major).
//   scan      : exclusive scan of the digit-major table = global base of every (digit, wave).
//   scatter   : the wave walks its chunk row by row; within a row the rank among equal digits is
//               popcount(peers & lanes-below) with peers from 8 ballots (match-any), the running
//               per-digit base lives in the wave-private LDS table.  Row order + lane order =
//               input order, so the sort is stable.
// No workgroup barrier is needed anywhere: a wave only touches its own LDS table.

#include "gs_common.h"

namespace {

constexpr int ROWS = 16;                 // rows of 64 pairs per wave
constexpr int CHUNK = ROWS * 64;         // pairs per wave
constexpr int WAVES_PER_BLOCK = 4;

template <typename K>
__global__ __launch_bounds__(256) void rs_hist(int64_t n, const K* keys, int shift, int bits, int num_waves, int* hist) {
  __shared__ int s_cnt[WAVES_PER_BLOCK][256];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t wave = int64_t(blockIdx.x) * WAVES_PER_BLOCK + w;
  for (int d = lane; d < 256; d += 64) s_cnt[w][d] = 0;
  __syncthreads();
  const unsigned mask = (1u << bits) - 1u;
  if (wave < num_waves) {
    const int64_t base = wave * CHUNK;
    for (int r = 0; r < ROWS; ++r) {
      const int64_t i = base + r * 64 + lane;
      if (i < n) atomicAdd(&s_cnt[w][unsigned(keys[i] >> shift) & mask], 1);
    }
  }
  __syncthreads();
  if (wave < num_waves)
    for (int d = lane; d < 256; d += 64) hist[int64_t(d) * num_waves + wave] = s_cnt[w][d];
}

template <typename K>
__global__ __launch_bounds__(256) void rs_scatter(int64_t n, const K* keys, const int* vals, K* keys_out, int* vals_out,
                                                  int shift, int bits, int num_waves, const int* offsets) {
  __shared__ int s_base[WAVES_PER_BLOCK][256];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t wave = int64_t(blockIdx.x) * WAVES_PER_BLOCK + w;
  if (wave >= num_waves) return;
  for (int d = lane; d < 256; d += 64) s_base[w][d] = offsets[int64_t(d) * num_waves + wave];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const unsigned mask = (1u << bits) - 1u;
  const uint64_t below = (1ull << lane) - 1ull;
  const int64_t base = wave * CHUNK;
  for (int r = 0; r < ROWS; ++r) {
    const int64_t i = base + r * 64 + lane;
    const bool valid = i < n;
    K key = 0;
    int val = 0;
    if (valid) { key = keys[i]; val = vals[i]; }
    const unsigned d = unsigned(key >> shift) & mask;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    if (valid) {
      const int rank = __popcll(peers & below);
      const int pos = s_base[w][d] + rank;
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
    __builtin_amdgcn_wave_barrier();
    // the highest lane of each peer group advances the group's base
    if (valid && (peers >> lane) == 1ull) s_base[w][d] += __popcll(peers);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename K>
int sort_impl(int64_t k, const K* keys_in, const int* values_in, K* keys_out, int* values_out, int begin_bit,
              int end_bit, char* scratch, hipStream_t s) {
  const int num_waves = int(gs_div_up(k, CHUNK));
  const int blocks = int(gs_div_up(num_waves, WAVES_PER_BLOCK));
  K* tmp_keys = reinterpret_cast<K*>(scratch);
  int* tmp_vals = reinterpret_cast<int*>(scratch + gs_align_up(k * int64_t(sizeof(K)), 256));
  int* hist = reinterpret_cast<int*>(scratch + gs_align_up(k * int64_t(sizeof(K)), 256) + gs_align_up(k * 4, 256));
  const int64_t hist_n = int64_t(256) * num_waves;
  void* scan_scratch = reinterpret_cast<char*>(hist) + gs_align_up((hist_n + 1) * 4, 256);
  const int passes = int(gs_div_up(end_bit - begin_bit, 8));
  if (passes == 0) {
    (void)hipMemcpyAsync(keys_out, keys_in, size_t(k) * sizeof(K), hipMemcpyDeviceToDevice, s);
    (void)hipMemcpyAsync(values_out, values_in, size_t(k) * 4, hipMemcpyDeviceToDevice, s);
    return GS_OK;
  }
  const K* src_k = keys_in;
  const int* src_v = values_in;
  for (int p = 0; p < passes; ++p) {
    const int shift = begin_bit + 8 * p;
    const int bits = min(8, end_bit - shift);
    // last pass must land in the caller's output: passes-1-p even -> out, odd -> tmp
    const bool to_out = ((passes - 1 - p) & 1) == 0;
    K* dst_k = to_out ? keys_out : tmp_keys;
    int* dst_v = to_out ? values_out : tmp_vals;
    hipLaunchKernelGGL((rs_hist<K>), dim3(blocks), dim3(256), 0, s, k, src_k, shift, bits, num_waves, hist);
    // exclusive scan of the digit-major table, in place (multi-block scan of mapper.hip)
    if (int rc = gs_full_cumsum_i32(hist_n, hist, hist, scan_scratch, gs_cumsum_scratch_bytes(hist_n), s)) return rc;
    hipLaunchKernelGGL((rs_scatter<K>), dim3(blocks), dim3(256), 0, s, k, src_k, src_v, dst_k, dst_v, shift, bits,
                       num_waves, hist);
    src_k = dst_k;
    src_v = dst_v;
  }
  GS_CHECK_LAUNCH("gs_radix_sort_pairs");
  return GS_OK;
}

}  // namespace

extern "C" int64_t gs_sort_scratch_bytes(int64_t k, int32_t key_bytes) {
  const int64_t num_waves = gs_div_up(k, CHUNK);
  const int64_t hist_n = 256 * num_waves;
  return gs_align_up(k * key_bytes, 256) + gs_align_up(k * 4, 256) + gs_align_up((hist_n + 1) * 4, 256) +
         gs_cumsum_scratch_bytes(hist_n) + 256;
}

extern "C" int gs_radix_sort_pairs(int64_t k, int32_t key_bytes, const void* keys_in, const int32_t* values_in,
                                   void* keys_out, int32_t* values_out, int32_t begin_bit, int32_t end_bit,
                                   void* scratch, int64_t scratch_bytes, void* stream) {
  GS_REQUIRE(key_bytes == 4 || key_bytes == 8, GS_ERR_UNSUPPORTED, "gs_radix_sort_pairs: key_bytes %d (4 or 8)",
             key_bytes);
  if (end_bit <= 0) end_bit = key_bytes * 8;  // radix_sort_pairs.cu:15
  GS_REQUIRE(begin_bit >= 0 && begin_bit <= end_bit && end_bit <= key_bytes * 8, GS_ERR_INVALID_ARGUMENT,
             "gs_radix_sort_pairs: bit range [%d,%d)", begin_bit, end_bit);
  GS_REQUIRE(k >= 0 && k < (int64_t(1) << 31), GS_ERR_INVALID_ARGUMENT, "gs_radix_sort_pairs: %lld pairs",
             (long long)k);
  if (k == 0) return GS_OK;
  GS_REQUIRE(keys_in && values_in && keys_out && values_out && scratch, GS_ERR_INVALID_ARGUMENT,
             "gs_radix_sort_pairs: NULL buffer");
  GS_REQUIRE(scratch_bytes >= gs_sort_scratch_bytes(k, key_bytes), GS_ERR_SCRATCH_TOO_SMALL,
             "gs_radix_sort_pairs: scratch %lld < %lld", (long long)scratch_bytes,
             (long long)gs_sort_scratch_bytes(k, key_bytes));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (key_bytes == 8)
    return sort_impl<uint64_t>(k, static_cast<const uint64_t*>(keys_in), values_in, static_cast<uint64_t*>(keys_out),
                               values_out, begin_bit, end_bit, static_cast<char*>(scratch), s);
  return sort_impl<uint32_t>(k, static_cast<const uint32_t*>(keys_in), values_in, static_cast<uint32_t*>(keys_out),
                             values_out, begin_bit, end_bit, static_cast<char*>(scratch), s);
}
