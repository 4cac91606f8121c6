// Tile binning for the rasterizer, re-designed for MI355X.
//
// The reference operator duplicates every Gaussian once per touched tile with a 64-bit
// (tile | depth) key and runs a device-wide radix sort over all R instances (6 passes of
// 8-bit digits at 512^2, ~152 B of HBM traffic per instance).  Here the tile part of the key is
// resolved by a counting scatter (count -> scan -> scatter, LDS-privatised), and only the
// depth order inside each tile is sorted — by one workgroup per tile, in registers + wave shuffles with
// only the cross-wave stages in LDS (64 KiB of the CU's 160 KiB covers 8192 instances).  HBM traffic drops to ~28 B per instance
// (8 B key write, 8 B key read, 4 B list write, + counters) and the whole stage is 3 launches.
// Result is identical: per tile, ascending (depth bits, Gaussian index).
//
// Tiles with more than SORT_LDS_CAP instances fall back to the same bitonic network run in place
// on the global key array (correct for any size, slower; exercised in tests).
#include "common.h"

namespace {

constexpr int SCAN_THREADS = 1024;
constexpr size_t SCAN_STAGE_MAX_BYTES = 48 * 1024;   // dynamic LDS of the staged tile scan (next to ~9 KiB static): 12288 tiles
constexpr int SORT_THREADS = 256;
constexpr int SORT_SMALL_CAP = GS_SORT_SMALL_CAP;  // small kernel: <= 8 keys per thread, 16 KiB LDS, ~40 VGPRs -> 8 workgroups per CU
constexpr int SORT_LDS_CAP = 8192;    // large kernel: 16/32 keys per thread, 64 KiB LDS

// ---- K2: exclusive scan of per-tile counts (T <= ~10^5 fits one workgroup comfortably)
// (count and start may alias: every thread reads an element before it overwrites it)
// With `order` set it also writes order[] = tile indices by descending count (counting sort over 1024 count buckets — exact
// order inside a bucket is irrelevant for load balance): the per-tile kernels launch their heaviest tiles first.
// STAGED (frames with more tiles than threads: 1080p has 8160): thread t owns `per` CONSECUTIVE tiles, so its accesses to the
// per-tile arrays are `per` words apart from its neighbour's.  Both directions of that were the kernel (one workgroup = one CU's
// memory pipe and one latency chain): every pass over the counts was a loop of strided, dependent global loads (30.2 us at 8160
// tiles), and the four per-tile results (order, first unit, first short unit, start) left as 4 x 8160 single-word stores, one
// cache line each — the workgroup's body took 13 us and the kernel 24 (tools/ubench/scan_tiles.hip: the store queue of one CU
// drains a line per clock).  So the counts are fetched ONCE, coalesced and all loads in flight together, into LDS, each thread
// takes its own run into registers, and every per-tile result goes back through the same LDS buffer and leaves as whole lines.
#ifdef GS_SCAN_PROBE   // tools/ubench/scan_tiles.hip: phase stamps of the single workgroup (100 MHz wall clock), thread 0
__device__ long long g_scan_probe[16];
#define SCAN_STAMP(k) do { if (threadIdx.x == 0) g_scan_probe[k] = wall_clock64(); } while (0)
#else
#define SCAN_STAMP(k) do { } while (0)
#endif
constexpr int SCAN_PER_MAX = (int)(SCAN_STAGE_MAX_BYTES / (SCAN_THREADS * 4));   // tiles per thread the staged form holds in registers

template <bool STAGED>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_tiles(int T, const uint32_t* count, uint32_t* start, int32_t* num_rendered,
                                                               uint32_t* __restrict__ order, uint32_t* __restrict__ meta,
                                                               uint32_t* __restrict__ seg_first, uint32_t* __restrict__ part_first,
                                                               uint32_t min_units) {
  __shared__ uint32_t wave_tot[SCAN_THREADS / GS_WAVE];
  HIP_DYNAMIC_SHARED(uint32_t, s_stage)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (T + SCAN_THREADS - 1) / SCAN_THREADS;
  const int lo = tid * per, hi = min(T, lo + per);
  SCAN_STAMP(0);
  uint32_t own[STAGED ? SCAN_PER_MAX : 1];
  if constexpr (STAGED) {
    for (int e = tid; e < T; e += SCAN_THREADS) s_stage[e] = count[e];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SCAN_PER_MAX; ++k) own[k] = lo + k < hi ? s_stage[lo + k] : 0u;
  }
  // f(i, c): tile i of this thread's run and its count
  auto for_own = [&](auto&& f) {
    if constexpr (STAGED) {
#pragma unroll
      for (int k = 0; k < SCAN_PER_MAX; ++k)
        if (lo + k < hi) f(lo + k, own[k]);
    } else {
      for (int i = lo; i < hi; ++i) f(i, count[i]);
    }
  };
  // out[i] = value(i, c) for every tile: directly (one tile per thread: already coalesced), or through the LDS buffer
  auto emit = [&](uint32_t* out, auto&& value) {
    if constexpr (STAGED) {
      __syncthreads();   // the buffer's previous readers
      for_own([&](int i, uint32_t c) { s_stage[i] = value(i, c); });
      __syncthreads();
      for (int e = tid; e < T; e += SCAN_THREADS) out[e] = s_stage[e];
    } else {
      for_own([&](int i, uint32_t c) { out[i] = value(i, c); });
    }
  };
  SCAN_STAMP(1);
  uint32_t local = 0;
  for_own([&](int, uint32_t c) { local += c; });
  // wave inclusive scan (DPP: this single-workgroup kernel is a chain of latencies, and a __shfl_up step is an LDS round trip)
  const uint32_t incl = gs_wave_scan_incl_u32(local);
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  uint32_t wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / GS_WAVE; ++w) {
    const uint32_t v = wave_tot[w];
    if (w < wave) wave_off += v;
    total += v;
  }
  SCAN_STAMP(2);
  // ---- the tile order first: it reads count[], which the scan below may overwrite (count and start can alias)
  if (order) {
    __shared__ uint32_t hist[SCAN_THREADS];
    __shared__ uint32_t s_max;
    uint32_t mx = 0;
    for_own([&](int, uint32_t c) { mx = max(mx, c); });
    mx = gs_wave_max_u32(mx);
    hist[tid] = 0u;
    if (tid == 0) s_max = 0u;
    __syncthreads();
    if (lane == 0) atomicMax(&s_max, mx);
    __syncthreads();
    const uint32_t cmax = max(s_max, 1u);
    // (bucket boundaries need not be exact, only the same in both passes and monotone in the count: a float multiply instead of
    // a 64-bit division per tile and pass — a hundred-odd instructions each on this latency chain)
    const float bscale = (float)(SCAN_THREADS - 1) / (float)cmax;
    auto bucket = [&](uint32_t c) { return (uint32_t)(SCAN_THREADS - 1) - min((uint32_t)(SCAN_THREADS - 1), (uint32_t)((float)c * bscale)); };
    SCAN_STAMP(3);
    for_own([&](int, uint32_t c) { atomicAdd(&hist[bucket(c)], 1u); });
    __syncthreads();
    SCAN_STAMP(4);
    // exclusive scan of the 1024 bucket sizes (one per thread)
    const uint32_t h = hist[tid];
    const uint32_t hi_ = gs_wave_scan_incl_u32(h);
    __shared__ uint32_t hist_wave[SCAN_THREADS / GS_WAVE];
    if (lane == 63) hist_wave[wave] = hi_;
    __syncthreads();
    uint32_t hoff = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / GS_WAVE; ++w)
      if (w < wave) hoff += hist_wave[w];
    hist[tid] = hoff + hi_ - h;
    __syncthreads();
    SCAN_STAMP(5);
    if constexpr (STAGED) {   // the permutation is built in LDS (the counts are in registers by now) and leaves as whole lines
      for_own([&](int i, uint32_t c) { s_stage[atomicAdd(&hist[bucket(c)], 1u)] = (uint32_t)i; });
      __syncthreads();
      for (int e = tid; e < T; e += SCAN_THREADS) order[e] = s_stage[e];
    } else {
      for_own([&](int i, uint32_t c) { order[atomicAdd(&hist[bucket(c)], 1u)] = (uint32_t)i; });
    }
    __syncthreads();
    SCAN_STAMP(6);
  }
  if (seg_first) {
    // Backward units (common.h): the unit length of this frame from its true instance count, then the first unit of every tile
    // = exclusive scan of ceil(count / unit length) (same two-level scan as below)
    const uint32_t level = (uint32_t)gs_unit_level_for((long long)total, (long long)min_units);
    const uint32_t chunks = 1u << level;
    const uint32_t seg_len = chunks * GS_SEG;
    // seg_len is a power of two: units and "short last unit" by shift and mask (four 32-bit divisions per tile otherwise, on a
    // kernel that is one workgroup's latency chain)
    static_assert((GS_SEG & (GS_SEG - 1)) == 0, "GS_SEG must be a power of two");
    const uint32_t seg_shift = level + (uint32_t)__builtin_ctz(GS_SEG), seg_mask = seg_len - 1u;
    auto units_of = [&](uint32_t c) { return (c + seg_mask) >> seg_shift; };
    auto short_of = [&](uint32_t c) { return (uint32_t)((c & seg_mask) != 0u); };
    // ... and, in the same pass, the prefix of "this tile's last unit is a short one" (part_first): the backward launches
    // the full-length units first and the short ones last, where they shorten the tail of the kernel (composite.hip)
    uint32_t lseg = 0, lpart = 0;
    for_own([&](int, uint32_t c) { lseg += units_of(c); lpart += short_of(c); });
    const uint32_t iseg = gs_wave_scan_incl_u32(lseg), ipart = gs_wave_scan_incl_u32(lpart);
    __shared__ uint32_t seg_wave[SCAN_THREADS / GS_WAVE];
    __shared__ uint32_t part_wave[SCAN_THREADS / GS_WAVE];
    if (lane == 63) { seg_wave[wave] = iseg; part_wave[wave] = ipart; }
    __syncthreads();
    uint32_t soff = 0, stot = 0, poff = 0, ptot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / GS_WAVE; ++w) {
      if (w < wave) { soff += seg_wave[w]; poff += part_wave[w]; }
      stot += seg_wave[w]; ptot += part_wave[w];
    }
    uint32_t srun = soff + iseg - lseg, prun = poff + ipart - lpart;
    emit(seg_first, [&](int, uint32_t c) { const uint32_t v = srun; srun += units_of(c); return v; });
    emit(part_first, [&](int, uint32_t c) { const uint32_t v = prun; prun += short_of(c); return v; });
    if (tid == 0) { part_first[T] = ptot; meta[3] = ptot; }
    if (tid == 0) { seg_first[T] = stot; meta[1] = stot; meta[2] = chunks; }
  }
  SCAN_STAMP(7);
  uint32_t run = wave_off + incl - local;
  // (count and start can alias: the staged form holds the counts in registers, the other reads count[i] right before it writes start[i])
  emit(start, [&](int, uint32_t c) { const uint32_t v = run; run += c; return v; });
  if (tid == 0) {
    start[T] = total;
    *num_rendered = (int32_t)total;
  }
  SCAN_STAMP(8);
}

// ---- large exclusive scan (used for the kNN cell table, up to 2^24 entries): per-block sums -> scan of the sums by
// k_scan_tiles -> per-block rescan with offset.  SCAN_CHUNK consecutive elements per workgroup, coalesced.
constexpr int SCAN_CHUNK = 4096;

// RECT: the input is not an array of counters but the Gaussians' packed tile rectangles, and element i counts the tiles of
// rectangle i (element n - 1 = 0, so that out[n - 1] is the total): the deterministic backward's row offsets (composite.hip),
// without a kernel and an array for the areas.
template <bool RECT>
__device__ __forceinline__ uint32_t scan_input(const uint32_t* __restrict__ in, int i, int n) {
  if (i >= n) return 0u;
  if constexpr (!RECT) return in[i];
  else {
    if (i == n - 1) return 0u;
    const uint2 r = reinterpret_cast<const uint2*>(in)[i];
    const int x0 = r.x & 0xffff, y0 = r.x >> 16, x1 = r.y & 0xffff, y1 = r.y >> 16;
    return (x1 > x0 && y1 > y0) ? (uint32_t)((x1 - x0) * (y1 - y0)) : 0u;
  }
}

template <bool RECT>
__global__ __launch_bounds__(256) void k_scan_block_sums(int n, const uint32_t* __restrict__ in, uint32_t* __restrict__ sums) {
  __shared__ uint32_t s_red[4];
  const int base = blockIdx.x * SCAN_CHUNK;
  uint32_t acc = 0;
  for (int i = threadIdx.x; i < SCAN_CHUNK; i += 256) acc += scan_input<RECT>(in, base + i, n);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

template <bool RECT>
__global__ __launch_bounds__(256) void k_scan_apply(int n, const uint32_t* __restrict__ in, const uint32_t* __restrict__ block_off,
                                                    uint32_t* __restrict__ out) {
  __shared__ uint32_t s_wave[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t run = block_off[blockIdx.x];
  for (int it = 0; it < SCAN_CHUNK / 256; ++it) {
    const int i = blockIdx.x * SCAN_CHUNK + it * 256 + tid;
    const uint32_t v = scan_input<RECT>(in, i, n);
    const uint32_t incl = gs_wave_scan_incl_u32(v);
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) woff += w < wave ? s_wave[w] : 0u;
    if (i < n) out[i] = run + woff + incl - v;
    run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  }
}

// ---- K1b / K3: per-tile counting and scatter with LDS-privatised counters.
// A workgroup owns BIN_CHUNK consecutive Gaussians and a private histogram over all T tiles in LDS
// (dynamic LDS, 4 B per tile: 4 KiB at 512^2, 32 KiB at 1080p).  Instances hit the histogram with
// LDS atomics; global memory sees one atomic per (workgroup, touched tile) instead of one per
// instance — measured 395 us -> (see profiles/) for 1M instances, where same-address global atomics
// serialise at the L2.  The scatter kernel turns each private count into a reserved range of its
// tile's segment (one returning global atomic), then hands out slots with returning LDS atomics.
// Order inside a tile is arbitrary here; K4's sort makes it deterministic.
constexpr int BIN_CHUNK = GS_BIN_CHUNK;   // Gaussians per workgroup: 384 workgroups at 196k Gaussians (2048 left 160 of 256 CUs idle)
static_assert(BIN_CHUNK == 512, "two Gaussians per thread (own_depth selection in k_scatter_lds)");
constexpr int BIN_MAX_LDS_TILES = 16384;  // 64 KiB of LDS; larger grids use the direct-atomic kernels
static_assert(BIN_MAX_LDS_TILES <= 64 * 256, "k_count_tiles_lds keeps one bit per bin of a thread in a 64-bit mask");

__device__ __forceinline__ bool unpack_rect(uint2 r, int& x0, int& y0, int& x1, int& y1) {
  x0 = r.x & 0xffff; y0 = r.x >> 16; x1 = r.y & 0xffff; y1 = r.y >> 16;
  return x1 > x0 && y1 > y0;
}

// A thread walks its own Gaussian's tile rectangle only when that is short: one Gaussian that has grown over hundreds of
// tiles would otherwise keep a single lane busy (with a dependent LDS-atomic -> global-store chain per tile in the scatter)
// long after the rest of the grid has finished — measured on the C3 run as k_scatter_lds drifting from 8 to 60 us while
// the instance count grew by 13 %.  Rectangles above BIN_WIDE tiles are queued in LDS and walked by a 16-lane row each.
constexpr int BIN_WIDE = 8;

constexpr int BIN_PER_THREAD = BIN_CHUNK / 256;

// f(i, j, x, y): instance (Gaussian i, tile (x, y)); j = the calling thread's own slot of Gaussian i (index into what it
// preloaded), or -1 on the wide path where another thread's Gaussian is walked.  own[j] = rectangle of Gaussian
// lo + tid + 256 j, preloaded by the caller ((0,0,0,0) past the end): the kernels that use this run in one resident round,
// every workgroup in the same phase, and a load issued inside the loop is a round trip to L2/HBM nobody covers.
// The queue holds the rectangle next to the index: at 1080p with a million Gaussians a third of them are "wide" (3 x 3 tiles
// and up), a workgroup's 16 rows then take ~10 queue rounds each, and a round that began with a dependent global load of the
// rectangle (plus two integer divisions per instance) made count and scatter latency chains (25 / 77 us at C4).
template <class F>
__device__ __forceinline__ void for_each_instance(int lo, int hi, const uint2 (&own)[BIN_PER_THREAD], const uint2* __restrict__ rects,
                                                  uint32_t* s_wide, uint2* s_wide_rect, uint32_t* s_nwide, F&& f) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int j = 0; j < BIN_PER_THREAD; ++j) {
    const int i = lo + tid + 256 * j;
    int x0, y0, x1, y1;
    if (i >= hi || !unpack_rect(own[j], x0, y0, x1, y1)) continue;
    if ((x1 - x0) * (y1 - y0) > BIN_WIDE) {
      const uint32_t slot = atomicAdd(s_nwide, 1u);
      s_wide[slot] = (uint32_t)i; s_wide_rect[slot] = own[j];
      continue;
    }
    for (int y = y0; y < y1; ++y)
      for (int x = x0; x < x1; ++x) f(i, j, x, y);
  }
  __syncthreads();
  const int nwide = (int)*s_nwide;
  for (int q = tid >> 4; q < nwide; q += 16) {  // one 16-lane row per queued Gaussian
    const int i = (int)s_wide[q];
    int x0, y0, x1, y1;
    unpack_rect(s_wide_rect[q], x0, y0, x1, y1);
    const int w = x1 - x0, n = w * (y1 - y0);
    // row-major walk without a division per instance: (column, row) of position j advance by (16 mod w, 16 div w) per step
    const int dq = 16 / w, dr = 16 - dq * w;
    int j = tid & 15, row = j / w, col = j - row * w;
    for (; j < n; j += 16) {
      f(i, -1, x0 + col, y0 + row);
      col += dr; row += dq;
      if (col >= w) { col -= w; row += 1; }
    }
  }
}

// The count kernel also leaves what it learned for the scatter kernel of the same frame: the workgroup's touched tiles with
// their private counts, packed (tile | count << 16; tile < 2^14, count <= BIN_CHUNK), at a fixed place per workgroup
// (entries[GS_BIN_ENTRIES * workgroup ...], entry_n[workgroup]).  The scatter then neither zeroes a histogram over all tiles nor
// walks every rectangle a second time to recount, nor sweeps all T bins for the non-empty ones — at 1080p that was 40 of its
// 60 us with the key stores taken out (profiles/r04_ab_binning_c4_count_scatter.txt).  A workgroup that touches more than
// GS_BIN_ENTRIES tiles says so (entry_n = ~0) and its scatter workgroup recounts as before.
__global__ __launch_bounds__(256) void k_count_tiles_lds(int P, int T, int gx, const uint2* __restrict__ rects,
                                                          uint32_t* __restrict__ tile_count, uint32_t* __restrict__ entries,
                                                          uint32_t* __restrict__ entry_n) {
  HIP_DYNAMIC_SHARED(uint32_t, s_bins)
  __shared__ uint32_t s_wide[BIN_CHUNK];
  __shared__ uint2 s_wide_rect[BIN_CHUNK];
  __shared__ uint32_t s_nwide;
  __shared__ uint32_t s_wave_nz[4];
  const int tid = threadIdx.x;
  const int lo = blockIdx.x * BIN_CHUNK, hi = min(P, lo + BIN_CHUNK);
  uint2 own[BIN_PER_THREAD];
#pragma unroll
  for (int j = 0; j < BIN_PER_THREAD; ++j) own[j] = rects[min(lo + tid + 256 * j, P - 1)];
  for (int t = tid; t < T; t += 256) s_bins[t] = 0;
  if (tid == 0) s_nwide = 0;
  __syncthreads();
  for_each_instance(lo, hi, own, rects, s_wide, s_wide_rect, &s_nwide, [&](int, int, int x, int y) { atomicAdd(&s_bins[y * gx + x], 1u); });
  __syncthreads();
  // (which of this thread's bins were non-empty is kept as a bit mask — T <= 16384 is 64 bins per thread — so that writing
  // the list below revisits those bins only)
  unsigned long long mask = 0ull;
  {
    int k = 0;
    for (int t = tid; t < T; t += 256, ++k) {
      const uint32_t c = s_bins[t];
      if (c) { atomicAdd(&tile_count[t], c); mask |= 1ull << k; }
    }
  }
  const uint32_t nz = (uint32_t)__popcll(mask);
  // this thread's run of the entry list: exclusive scan of the per-thread non-empty counts over the workgroup
  const uint32_t incl = gs_wave_scan_incl_u32(nz);
  if ((tid & 63) == 63) s_wave_nz[tid >> 6] = incl;
  __syncthreads();
  uint32_t off = incl - nz, total = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    if (w < (tid >> 6)) off += s_wave_nz[w];
    total += s_wave_nz[w];
  }
  if (tid == 0) entry_n[blockIdx.x] = total <= (uint32_t)GS_BIN_ENTRIES ? total : 0xffffffffu;
  if (total > (uint32_t)GS_BIN_ENTRIES) return;
  uint32_t* mine = entries + (size_t)blockIdx.x * GS_BIN_ENTRIES + off;
  while (mask) {
    const int t = tid + 256 * (__ffsll((long long)mask) - 1);
    mask &= mask - 1ull;
    *mine++ = (uint32_t)t | (s_bins[t] << 16);
  }
}

__global__ __launch_bounds__(256) void k_scatter_lds(int P, int T, int gx, const float* __restrict__ depths,
                                                      const uint2* __restrict__ rects, const uint32_t* __restrict__ start,
                                                      uint32_t* __restrict__ cursor, uint64_t* __restrict__ keys, uint32_t capacity,
                                                      const uint32_t* __restrict__ entries, const uint32_t* __restrict__ entry_n) {
  HIP_DYNAMIC_SHARED(uint32_t, s_bins)
  __shared__ uint32_t s_wide[BIN_CHUNK];
  __shared__ uint2 s_wide_rect[BIN_CHUNK];
  __shared__ uint32_t s_nwide;
  const int tid = threadIdx.x;
  const int lo = blockIdx.x * BIN_CHUNK, hi = min(P, lo + BIN_CHUNK);
  // this thread's Gaussians: rectangle and depth (the high half of the sort key), requested before anything waits
  uint2 own[BIN_PER_THREAD];
  uint32_t own_depth[BIN_PER_THREAD];
#pragma unroll
  for (int j = 0; j < BIN_PER_THREAD; ++j) {
    const int ic = min(lo + tid + 256 * j, P - 1);
    own[j] = rects[ic];
    own_depth[j] = __float_as_uint(depths[ic]);
  }
  // Private counts -> first slot of this workgroup's range in each touched tile: from the count kernel's entry list (only
  // the touched tiles' bins are written, and only those are read by the walk below) ...
  const uint32_t n_entries = entry_n[blockIdx.x];
  if (n_entries != 0xffffffffu) {
    const uint32_t* mine = entries + (size_t)blockIdx.x * GS_BIN_ENTRIES;
    // (four returning atomics per thread in flight before the first result is consumed)
    for (uint32_t e0 = 0; e0 < n_entries; e0 += 1024) {
      uint32_t en[4], r[4], st[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t e = e0 + tid + 256 * j;
        en[j] = e < n_entries ? mine[e] : 0u;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r[j] = 0u; st[j] = 0u;
        if (en[j]) { r[j] = atomicAdd(&cursor[en[j] & 0xffffu], en[j] >> 16); st[j] = start[en[j] & 0xffffu]; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (en[j]) s_bins[en[j] & 0xffffu] = st[j] + r[j];
    }
  } else {
    // ... or, for a workgroup that touched more tiles than the list holds, by counting again
    for (int t = tid; t < T; t += 256) s_bins[t] = 0;
    if (tid == 0) s_nwide = 0;
    __syncthreads();
    for_each_instance(lo, hi, own, rects, s_wide, s_wide_rect, &s_nwide, [&](int, int, int x, int y) { atomicAdd(&s_bins[y * gx + x], 1u); });
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += 1024) {
      uint32_t c[4], r[4], st[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = t0 + tid + 256 * j;
        c[j] = t < T ? s_bins[t] : 0u;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = t0 + tid + 256 * j;
        r[j] = 0u; st[j] = 0u;
        if (c[j]) { r[j] = atomicAdd(&cursor[t], c[j]); st[j] = start[t]; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c[j]) s_bins[t0 + tid + 256 * j] = st[j] + r[j];
    }
  }
  if (tid == 0) s_nwide = 0;
  __syncthreads();
  for_each_instance(lo, hi, own, rects, s_wide, s_wide_rect, &s_nwide, [&](int i, int j, int x, int y) {
    const uint32_t depth = j == 0 ? own_depth[0] : (j == 1 ? own_depth[BIN_PER_THREAD - 1] : __float_as_uint(depths[i]));
    const uint64_t key = ((uint64_t)depth << 32) | (uint32_t)i;
    const uint32_t pos = atomicAdd(&s_bins[y * gx + x], 1u);
    if (pos < capacity) keys[pos] = key;
  });
}

// direct global-atomic variants for tile grids too large for an LDS histogram
__global__ __launch_bounds__(256) void k_count_tiles_direct(int P, int gx, const uint2* __restrict__ rects,
                                                             uint32_t* __restrict__ tile_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int x0, y0, x1, y1;
  if (!unpack_rect(rects[i], x0, y0, x1, y1)) return;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) atomicAdd(&tile_count[y * gx + x], 1u);
}

__global__ __launch_bounds__(256) void k_scatter_direct(int P, int gx, const float* __restrict__ depths, const uint2* __restrict__ rects,
                                                         const uint32_t* __restrict__ start, uint32_t* __restrict__ cursor,
                                                         uint64_t* __restrict__ keys, uint32_t capacity) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int x0, y0, x1, y1;
  if (!unpack_rect(rects[i], x0, y0, x1, y1)) return;
  const uint64_t key = ((uint64_t)__float_as_uint(depths[i]) << 32) | (uint32_t)i;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      const int t = y * gx + x;
      const uint32_t pos = start[t] + atomicAdd(&cursor[t], 1u);
      if (pos < capacity) keys[pos] = key;
    }
}

// ---- K4: per-tile sort.  Bitonic network in the "flip + disperse" form: every compare-exchange puts
// the smaller key at the lower index, so slots >= n behave as +inf without being materialised and
// any n (not only powers of two) sorts correctly.
template <class KeyPtr>
__device__ __forceinline__ void bitonic_sort_any(KeyPtr a, int n, int tid, int nthreads) {
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  const int half = np2 >> 1;
  for (int k = 2; k <= np2; k <<= 1) {
    const int hk = k >> 1;
    for (int t = tid; t < half; t += nthreads) {
      const int blk = t / hk, off = t - blk * hk;
      const int i = blk * k + off, j = blk * k + (k - 1 - off);
      if (j < n) {
        const uint64_t x = a[i], y = a[j];
        if (x > y) { a[i] = y; a[j] = x; }
      }
    }
    __syncthreads();
    for (int d = hk >> 1; d >= 1; d >>= 1) {
      for (int t = tid; t < half; t += nthreads) {
        const int i = (t / d) * (2 * d) + (t % d), j = i + d;
        if (j < n) {
          const uint64_t x = a[i], y = a[j];
          if (x > y) { a[i] = y; a[j] = x; }
        }
      }
      __syncthreads();
    }
  }
}

// Lane exchange for the sort network without the LDS crossbar (ds_bpermute, two per 64-bit key): every partner pattern
// of a bitonic network inside a wave is a DPP move or a gfx950 row swap.
//   xor 1, 2: quad_perm          xor 4: row_shl:4 into banks 0,2 + row_shr:4 into banks 1,3      xor 8: row_ror:8
//   xor 16 / 32: v_permlane16_swap / v_permlane32_swap of the value with itself, then pick by row / half
//   flips (xor 2^k - 1): quad_perm [3,2,1,0], row_half_mirror, row_mirror (+ the xor-16 / xor-32 steps above)
template <int CTRL>
__device__ __forceinline__ uint32_t sort_dpp(uint32_t v) {
  // every lane has an in-range source for these patterns, so bound_ctrl only spares the compiler a zero-initialised `old`
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t sort_xor16(uint32_t u) {
  const auto p = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return (threadIdx.x & 16) ? p[0] : p[1];
}
__device__ __forceinline__ uint32_t sort_xor32(uint32_t u) {
  const auto p = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return (threadIdx.x & 32) ? p[0] : p[1];
}
#ifndef GS_SORT_BPERMUTE
#define GS_SORT_BPERMUTE 2   // shipped: 14.35 -> 13.55 us at C3 (profiles/r03_ab_sort_bpermute_kernel_avg.txt)
#endif
// A/B build switch (tools/build_variant.sh): lane exchanges of the network through the LDS crossbar (ds_bpermute_b32: no
// VALU issue cycles, its own pipe) instead of DPP moves / row swaps
// (0: none — rounds 1-2; 1: all exchanges; 2: only the ones that would need a v_permlane*_swap + select — lane masks 16, 31,
// 32, 63: 12 VALU issue cycles per 32-bit word against one ds_bpermute.  Measured at C3: 0: 14.3 us, 1: 18.7 us (the network is
// a chain of dependent stages and the crossbar's latency is not covered when every stage takes it), 2: 13.5 us)
#ifndef GS_SORT_TWO_RUNS
#define GS_SORT_TWO_RUNS 2   // A/B build switch: 0 = one network per tile; 1 = tiles of 1025 ... 1536 keys as two sorted runs merged by rank (sort_tile_two_runs); 2 = also 513 ... 768; 3 = also 1537 ... 2048 (no gain)
#endif
template <int LM>
__device__ __forceinline__ uint32_t lane_xchg32(uint32_t v) {
  constexpr bool SORT_BPERMUTE = GS_SORT_BPERMUTE == 1 || (GS_SORT_BPERMUTE == 2 && LM >= 16);
  if constexpr (SORT_BPERMUTE) return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63) ^ LM) << 2), (int)v);
  else if constexpr (LM == 1) return sort_dpp<0xB1>(v);
  else if constexpr (LM == 2) return sort_dpp<0x4E>(v);
  else if constexpr (LM == 3) return sort_dpp<0x1B>(v);
  else if constexpr (LM == 4) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);         // banks 0,2 <- lane + 4
    return (uint32_t)__builtin_amdgcn_update_dpp(lo, (int)v, 0x114, 0xF, 0xA, false);     // banks 1,3 <- lane - 4
  }
  else if constexpr (LM == 7) return sort_dpp<0x141>(v);
  else if constexpr (LM == 8) return sort_dpp<0x128>(v);
  else if constexpr (LM == 15) return sort_dpp<0x140>(v);
  else if constexpr (LM == 16) return sort_xor16(v);
  else if constexpr (LM == 31) return sort_xor16(sort_dpp<0x140>(v));
  else if constexpr (LM == 32) return sort_xor32(v);
  else {
    static_assert(LM == 63, "lane mask of a bitonic network inside a wave64");
    return sort_xor32(sort_xor16(sort_dpp<0x140>(v)));
  }
}
template <int LM>
__device__ __forceinline__ uint64_t lane_xchg64(uint64_t v) {
  return ((uint64_t)lane_xchg32<LM>((uint32_t)(v >> 32)) << 32) | lane_xchg32<LM>((uint32_t)v);
}

// Register/wave/LDS bitonic sort of one tile's keys: thread t owns E keys; key index
//   i = wave * (64*E) + lane * E + e          (e = low bits, lane = middle 6 bits, wave = top 2 bits)
// so compare-exchange partners at distance < E sit in the same thread, at distance < 64*E in the same wave
// (VALU lane exchanges, no barrier) and only the last two distance bits (across the 4 waves) go through
// LDS: 3 barrier stages in total instead of one per network stage (55 for 1024 keys).  The network is unrolled at
// compile time (template recursion), so every partner pattern is a constant.
// Slots >= n hold 0xFFFF... (sorts last, never written back).  Keys are distinct (they end in the Gaussian index), so
// "the lower position keeps the minimum" is one 64-bit compare: take the partner's key iff (partner < mine) == lower.
template <int E, int KK>
__device__ __forceinline__ void sort_flip_stage(uint64_t (&k)[E], uint64_t* __restrict__ s_keys, int base) {
  // partner = i ^ (KK-1); the lower index keeps the minimum
  constexpr int m = KK - 1;
  if constexpr (KK <= E) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int pe = e ^ (m & (E - 1));
      if (e < pe) { const uint64_t a = k[e], b = k[pe]; if (a > b) { k[e] = b; k[pe] = a; } }
    }
  } else if constexpr (KK <= 64 * E) {
    constexpr int lm = (m / E) & 63;  // lane bits of the mask
    uint64_t other[E];
#pragma unroll
    for (int e = 0; e < E; ++e) other[e] = lane_xchg64<lm>(k[E - 1 - e]);
    const bool lower = (base & (KK >> 1)) == 0;
#pragma unroll
    for (int e = 0; e < E; ++e) { const uint64_t a = k[e], b = other[e]; k[e] = ((b < a) == lower) ? b : a; }
  } else {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) s_keys[base + e] = k[e];
    __syncthreads();
    const bool lower = (base & (KK >> 1)) == 0;
#pragma unroll
    for (int e = 0; e < E; ++e) { const uint64_t a = k[e], b = s_keys[(base + e) ^ m]; k[e] = ((b < a) == lower) ? b : a; }
  }
}

template <int E, int D>
__device__ __forceinline__ void sort_disperse_from(uint64_t (&k)[E], uint64_t* __restrict__ s_keys, int base) {
  // partner = i ^ D, then D/2, ... 1
  if constexpr (D >= 1) {
    if constexpr (D < E) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int pe = e ^ D;
        if (e < pe) { const uint64_t a = k[e], b = k[pe]; if (a > b) { k[e] = b; k[pe] = a; } }
      }
    } else if constexpr (D < 64 * E) {
      const bool lower = (base & D) == 0;
#pragma unroll
      for (int e = 0; e < E; ++e) { const uint64_t a = k[e], b = lane_xchg64<D / E>(a); k[e] = ((b < a) == lower) ? b : a; }
    } else {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) s_keys[base + e] = k[e];
      __syncthreads();
      const bool lower = (base & D) == 0;
#pragma unroll
      for (int e = 0; e < E; ++e) { const uint64_t a = k[e], b = s_keys[(base + e) ^ D]; k[e] = ((b < a) == lower) ? b : a; }
    }
    sort_disperse_from<E, D / 2>(k, s_keys, base);
  }
}

template <int E, int KK, int NP2>
__device__ __forceinline__ void sort_merge_from(uint64_t (&k)[E], uint64_t* __restrict__ s_keys, int base) {
  if constexpr (KK <= NP2) {
    sort_flip_stage<E, KK>(k, s_keys, base);
    sort_disperse_from<E, KK / 4>(k, s_keys, base);
    sort_merge_from<E, KK * 2, NP2>(k, s_keys, base);
  }
}

// keys_out == null: the sorted Gaussian indices go to list[s ...]; otherwise the sorted KEYS go to keys_out[s ...]
// (one sorted run of a tile too long for the workgroup's registers, see sort_long_tile)
template <int E>
__device__ __forceinline__ void sort_tile_regs(uint64_t* __restrict__ s_keys, const uint64_t* keys, uint32_t* __restrict__ list,
                                               uint32_t s, int n, uint64_t* keys_out = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base = wave * (64 * E) + lane * E;
  uint64_t k[E];
#pragma unroll
  for (int e = 0; e < E; ++e) k[e] = (base + e < n) ? keys[s + base + e] : ~0ull;
  sort_merge_from<E, 2, SORT_THREADS * E>(k, s_keys, base);
  if (keys_out) {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (base + e < n) keys_out[s + base + e] = k[e];
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (base + e < n) list[s + base + e] = (uint32_t)k[e];
  }
}

// Tiles a little longer than a power of two — at 1080p with a million Gaussians a fifth of the tiles hold 1025 ... 1400 keys
// (mean 890) and the 2048-key network costs 2.4 x the 1024-key one (66 stages of 8 keys per thread against 55 of 4): they took
// 40 % of the kernel.  Such a tile is sorted as TWO runs by the register network — the first SORT_THREADS * EA keys and the
// rest (up to SORT_THREADS * EB) — which are then merged by rank: both sorted runs are parked in LDS, every key counts the
// keys of the OTHER run below it (a branch-free binary search; keys are distinct) and that count plus its position in its own
// run is its place in the tile.  55 + 45 (or 36) register stages on 4 + 2 (1) keys and ~11 dependent LDS reads per key.
template <int NMAX>
__device__ __forceinline__ uint32_t sort_count_below(const uint64_t* __restrict__ run, int n, uint64_t key) {
  int lo = 0;   // number of run[0 .. n) below key, n <= NMAX (a power of two)
#pragma unroll
  for (int step = NMAX; step >= 1; step >>= 1)
    if (lo + step <= n && run[lo + step - 1] < key) lo += step;
  return (uint32_t)lo;
}
template <int EA, int EB>
__device__ __forceinline__ void sort_tile_two_runs(uint64_t* __restrict__ s_keys, const uint64_t* keys, uint32_t* __restrict__ list,
                                                   uint32_t s, int n) {
  constexpr int NA = SORT_THREADS * EA, NBMAX = SORT_THREADS * EB;
  static_assert(NA + NBMAX <= SORT_SMALL_CAP, "both sorted runs are parked in the kernel's LDS array");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int base_a = wave * (64 * EA) + lane * EA, base_b = wave * (64 * EB) + lane * EB, nb = n - NA;   // NA < n <= NA + NBMAX
  uint64_t ka[EA], kb[EB];
#pragma unroll
  for (int e = 0; e < EA; ++e) ka[e] = keys[s + base_a + e];
#pragma unroll
  for (int e = 0; e < EB; ++e) kb[e] = (base_b + e < nb) ? keys[s + NA + base_b + e] : ~0ull;
  sort_merge_from<EA, 2, NA>(ka, s_keys, base_a);
  sort_merge_from<EB, 2, NBMAX>(kb, s_keys, base_b);
  __syncthreads();   // the networks' last LDS stages have been read
#pragma unroll
  for (int e = 0; e < EA; ++e) s_keys[base_a + e] = ka[e];
#pragma unroll
  for (int e = 0; e < EB; ++e) s_keys[NA + base_b + e] = kb[e];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < EA; ++e) list[s + base_a + e + sort_count_below<NBMAX>(s_keys + NA, nb, ka[e])] = (uint32_t)ka[e];
#pragma unroll
  for (int e = 0; e < EB; ++e)
    if (base_b + e < nb) list[s + base_b + e + sort_count_below<NA>(s_keys, NA, kb[e])] = (uint32_t)kb[e];
}

// Tiles longer than SORT_SMALL_CAP (rare: the C3 maximum is under 1100): sort runs of SORT_SMALL_CAP keys with the register
// network (keys rewritten in place), then place every key by RANK — its index in its own run plus, for every other run staged
// in LDS, the number of keys below it (keys are distinct, so the ranks are a permutation).  Same registers and LDS as the
// short-tile path, so the kernel keeps its occupancy and no second kernel is launched for the long tiles.
__device__ __forceinline__ void sort_long_tile(uint64_t* __restrict__ s_keys, uint64_t* keys, uint32_t* __restrict__ list, uint32_t s, int n) {
  constexpr int RUN = SORT_SMALL_CAP, E = RUN / SORT_THREADS;
  const int nrun = (n + RUN - 1) / RUN, tid = threadIdx.x;
  for (int r = 0; r < nrun; ++r) {
    __syncthreads();  // the previous run's LDS stages are done
    sort_tile_regs<E>(s_keys, keys, list, s + (uint32_t)r * RUN, min(RUN, n - r * RUN), keys);
  }
  for (int r = 0; r < nrun; ++r) {
    const int rn = min(RUN, n - r * RUN);
    uint64_t k[E];
    uint32_t rank[E];
    __syncthreads();  // this workgroup's sorted runs are in global memory (visible to the workgroup after the barrier)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int i = tid * E + e;
      k[e] = i < rn ? keys[s + (uint32_t)r * RUN + i] : ~0ull;
      rank[e] = (uint32_t)i;  // position inside the own (sorted) run; the other runs' contributions are added below
    }
    for (int r2 = 0; r2 < nrun; ++r2) {
      if (r2 == r) continue;
      const int n2 = min(RUN, n - r2 * RUN);
      __syncthreads();
      for (int i = tid; i < n2; i += SORT_THREADS) s_keys[i] = keys[s + (uint32_t)r2 * RUN + i];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) {
        int lo = 0, hi = n2;  // number of keys of run r2 below k[e]
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (s_keys[mid] < k[e]) lo = mid + 1; else hi = mid;
        }
        rank[e] += (uint32_t)lo;
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (tid * E + e < rn) list[s + rank[e]] = (uint32_t)k[e];
  }
}

// One kernel for every tile length: the register network up to SORT_SMALL_CAP keys, sorted runs + rank placement up to
// SORT_LDS_CAP (sort_long_tile), the in-place global-memory network beyond.  (A second kernel with 16/32 keys per thread
// for the long tiles cost a 4.6 us launch per frame that almost never had anything to do.)
__device__ __forceinline__ void sort_one_tile(int tile, const uint32_t* __restrict__ start, uint64_t* keys,
                                              uint32_t* __restrict__ list, uint32_t capacity) {
  __shared__ uint64_t s_keys[SORT_SMALL_CAP];
  const uint32_t s = min(start[tile], capacity), e = min(start[tile + 1], capacity);
  const int n = (int)(e - s);
  if (n <= 0) return;
  if (n <= SORT_THREADS) sort_tile_regs<1>(s_keys, keys, list, s, n);
  else if (n <= SORT_THREADS * 2) sort_tile_regs<2>(s_keys, keys, list, s, n);
#if GS_SORT_TWO_RUNS >= 2
  else if (n <= SORT_THREADS * 3) sort_tile_two_runs<2, 1>(s_keys, keys, list, s, n);
#endif
#if GS_SORT_TWO_RUNS >= 4
  else if (n <= SORT_THREADS * 4) sort_tile_two_runs<2, 2>(s_keys, keys, list, s, n);
#endif
  else if (n <= SORT_THREADS * 4) sort_tile_regs<4>(s_keys, keys, list, s, n);
#if GS_SORT_TWO_RUNS >= 1
  else if (n <= SORT_THREADS * 5) sort_tile_two_runs<4, 1>(s_keys, keys, list, s, n);
  else if (n <= SORT_THREADS * 6) sort_tile_two_runs<4, 2>(s_keys, keys, list, s, n);
#endif
#if GS_SORT_TWO_RUNS >= 3
  else if (n <= SORT_SMALL_CAP) sort_tile_two_runs<4, 4>(s_keys, keys, list, s, n);
#endif
  else if (n <= SORT_SMALL_CAP) sort_tile_regs<8>(s_keys, keys, list, s, n);
  else if (n <= SORT_LDS_CAP) sort_long_tile(s_keys, keys, list, s, n);
  else {
    uint64_t* seg = keys + s;
    bitonic_sort_any(seg, n, (int)threadIdx.x, SORT_THREADS);
    for (int i = threadIdx.x; i < n; i += SORT_THREADS) list[s + i] = (uint32_t)seg[i];
  }
}

// One workgroup per tile, heaviest tiles first (order[] from k_scan_tiles).
__global__ __launch_bounds__(SORT_THREADS) void k_sort_tiles(int T, const uint32_t* __restrict__ start, uint64_t* keys,
                                                              uint32_t* __restrict__ list, uint32_t capacity,
                                                              const uint32_t* __restrict__ order) {
  sort_one_tile((int)order[blockIdx.x], start, keys, list, capacity);
}

}  // namespace

static int g_min_units = GS_MIN_UNITS;
thread_local int g_min_units_pinned = 0;   // a trainer handle's snapshot of the knob, in force for the duration of its calls
extern "C" int mi355gs_tune_min_units(int min_units) {
  const int old = g_min_units;
  if (min_units > 0) g_min_units = min_units;
  return old;
}
int gs_min_units() { return g_min_units_pinned > 0 ? g_min_units_pinned : g_min_units; }
void gs_pin_min_units(int v) { g_min_units_pinned = v; }

int gs_launch_scan_tiles(hipStream_t stream, int T, const uint32_t* count, uint32_t* start, int32_t* num_rendered, uint32_t* order,
                         uint32_t* meta, uint32_t* seg_first, uint32_t* part_first) {
  // more tiles than threads and the counts fit the workgroup's dynamic LDS: the staged form (see the kernel)
  const size_t stage_bytes = (size_t)((T + SCAN_THREADS - 1) / SCAN_THREADS) * SCAN_THREADS * 4;   // T words, rounded up to whole rounds of the workgroup
  if (T > SCAN_THREADS && stage_bytes <= SCAN_STAGE_MAX_BYTES)
    hipLaunchKernelGGL(k_scan_tiles<true>, dim3(1), dim3(SCAN_THREADS), stage_bytes, stream, T, count, start, num_rendered, order, meta,
                       seg_first, part_first, (uint32_t)gs_min_units());
  else
    hipLaunchKernelGGL(k_scan_tiles<false>, dim3(1), dim3(SCAN_THREADS), 0, stream, T, count, start, num_rendered, order, meta, seg_first,
                       part_first, (uint32_t)gs_min_units());
  return 0;
}

// exclusive scan of n counters into out[0..n] (out[n] = total); block_sums: ceil(n / 4096) + 2 uint32 of scratch
template <bool RECT>
static int scan_large(hipStream_t stream, int n, const uint32_t* in, uint32_t* out, uint32_t* block_sums, int32_t* total) {
  const int nb = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  hipLaunchKernelGGL(k_scan_block_sums<RECT>, dim3(nb), dim3(256), 0, stream, n, in, block_sums);
  hipLaunchKernelGGL(k_scan_tiles<false>, dim3(1), dim3(SCAN_THREADS), 0, stream, nb, (const uint32_t*)block_sums, block_sums, total,
                     (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u);  // in place
  hipLaunchKernelGGL(k_scan_apply<RECT>, dim3(nb), dim3(256), 0, stream, n, in, (const uint32_t*)block_sums, out);
  return 0;
}
int gs_launch_scan_large(hipStream_t stream, int n, const uint32_t* in, uint32_t* out, uint32_t* block_sums, int32_t* total) {
  return scan_large<false>(stream, n, in, out, block_sums, total);
}
// off[g] = tiles of the rectangles of Gaussians 0 .. g - 1, g = 0 .. P (off[P] = all instances); *total receives it as well
int gs_launch_scan_rect_areas(hipStream_t stream, int P, const uint2* rects, uint32_t* off, uint32_t* block_sums, int32_t* total) {
  return scan_large<true>(stream, P + 1, reinterpret_cast<const uint32_t*>(rects), off, block_sums, total);
}

int gs_launch_count_tiles(hipStream_t stream, int P, int T, int gx, const uint2* rects, uint32_t* tile_count, uint32_t* entries,
                          uint32_t* entry_n) {
  if (P <= 0) return 0;
  if (T <= BIN_MAX_LDS_TILES) {
    GsProfScope prof(5, stream);
    hipLaunchKernelGGL(k_count_tiles_lds, dim3((P + BIN_CHUNK - 1) / BIN_CHUNK), dim3(256), (size_t)T * 4, stream, P, T, gx, rects, tile_count,
                       entries, entry_n);
  }
  else
    hipLaunchKernelGGL(k_count_tiles_direct, dim3((P + 255) / 256), dim3(256), 0, stream, P, gx, rects, tile_count);
  return 0;
}

int gs_launch_binning(hipStream_t stream, int P, int T, int gx, const float* depths, const uint2* rects, const uint32_t* start,
                      uint32_t* cursor, uint64_t* keys, uint32_t* list, uint32_t capacity, const uint32_t* order, const uint32_t* entries,
                      const uint32_t* entry_n) {
  if (P <= 0 || capacity == 0) return 0;
  if (T <= BIN_MAX_LDS_TILES)
    hipLaunchKernelGGL(k_scatter_lds, dim3((P + BIN_CHUNK - 1) / BIN_CHUNK), dim3(256), (size_t)T * 4, stream, P, T, gx, depths, rects,
                       start, cursor, keys, capacity, entries, entry_n);
  else
    hipLaunchKernelGGL(k_scatter_direct, dim3((P + 255) / 256), dim3(256), 0, stream, P, gx, depths, rects, start, cursor, keys, capacity);
  {
    GsProfScope prof(4, stream);
    hipLaunchKernelGGL(k_sort_tiles, dim3(T), dim3(SORT_THREADS), 0, stream, T, start, keys, list, capacity, order);
  }
  return 0;
}
