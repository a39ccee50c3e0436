// gfx950 (MI355X, CDNA4) kernels of the OH predictor.
//
// What they replace: libxgboost's CPU predictor behind XGBoosterPredict, called
// once per OH alarm tick from predict_OH_with_XGB
// (/root/reference OH_GridComp/OH_GridCompMod.F90:356), plus the passes around
// it: the SoA->AoS gather (:308-345), the DMatrix copy (:347) and the
// 10**pred scatter (:364-374).
//
// Prediction semantics (xgboost 1.6.0 cpu_predictor / predict_fn):
//   pred = base_score; for t in order: pred += leaf_t(row)      (float32, sequential)
//   walk: missing (NaN, or == `missing`) -> default child; else x < cond -> left, else right
//
// Execution shape: memory/latency bound pointer chasing, no MFMA.  One lane owns
// one gridcell; the wave's 64 feature vectors sit in LDS feature-major
// ([feature][lane], bank = lane % 32, conflict-free for any per-lane feature
// index); CHAINS trees are walked at once per lane so that several dependent
// node loads are in flight per lane, and their leaves are added in tree order.
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdint>

#include "kernels.hpp"

namespace ohx {

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;                  // 4 waves
constexpr int kWavesPerBlock = kBlock / kWave;

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// a pointer that is KNOWN to point into LDS: loads through it are ds_read, never flat_load (a generic pointer
// into LDS whose origin the compiler loses track of goes through the texture addresser like a global load)
typedef const __attribute__((address_space(3))) char* lds_cptr;
typedef const __attribute__((address_space(3))) u32x4* lds_u32x4_ptr;
typedef __attribute__((address_space(3))) uint32_t* lds_u32_ptr;
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// The packed 8-byte nodes are read through a buffer descriptor: one 64-bit load per node that the
// compiler cannot split into dword loads (it does split a plain uint2 load when only one half
// is needed early, which doubles the gathers), a 32-bit offset instead of a 64-bit address,
// and a hardware range check.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint2 load_node8(__amdgpu_buffer_rsrc_t r, uint32_t slot) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(slot << 3), 0, 0);
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ bool is_inf(float v) { return __builtin_isinf(v); }

// ------------------------------------------------------------------ lanes -> rows

// Row of this lane in tile `tile_id` (TileShape, kernels.hpp).  Tiles are numbered brick-i
// fastest, so the waves of a block and the blocks of a launch walk neighbouring bricks.
__device__ __forceinline__ uint64_t tile_row(const TileShape& sh, uint64_t tile_id, int lane, uint64_t nrow,
                                             bool* valid) {
  if (sh.im == 0) {
    const uint64_t row = tile_id * kWave + lane;
    *valid = row < nrow;
    return row;
  }
  uint32_t t = (uint32_t)tile_id;
  const uint32_t bi = t % sh.nbi;
  t /= sh.nbi;
  const uint32_t bj = t % sh.nbj;
  const uint32_t bk = t / sh.nbj;
  const uint32_t l = (uint32_t)lane;
  uint32_t di, dj, dk;
  if (sh.k_fastest) {
    dk = l & ((1u << sh.lk) - 1u);
    di = (l >> sh.lk) & ((1u << sh.li) - 1u);
    dj = l >> (sh.lk + sh.li);
  } else {
    di = l & ((1u << sh.li) - 1u);
    dj = (l >> sh.li) & ((1u << sh.lj) - 1u);
    dk = l >> (sh.li + sh.lj);
  }
  const uint32_t i = (bi << sh.li) + di;
  const uint32_t j = (bj << sh.lj) + dj;
  const uint32_t k = sh.k_first + (bk << sh.lk) + dk;
  const uint64_t m = (uint64_t)i + (uint64_t)sh.im * ((uint64_t)j + (uint64_t)sh.jm * (uint64_t)k);
  *valid = i < sh.im && j < sh.jm && m >= sh.row0 && m - sh.row0 < sh.nrow;
  return m - sh.row0;
}

// the same for a launch: rows grouped by the clustering pass come through the permutation
__device__ __forceinline__ uint64_t launch_row(const PredictArgs& a, uint64_t tile_id, int lane, bool* valid,
                                               uint64_t perm_slots = ~0ull) {
  if (a.perm != nullptr) {
    const uint64_t slot = tile_id * kWave + lane;
    *valid = a.perm_count != nullptr ? slot < perm_slots : slot < a.nrow;
    const uint32_t row = *valid ? a.perm[slot] : 0u;
    if (row == 0xFFFFFFFFu && a.perm_count != nullptr) *valid = false;      // a slot of the deferred list nobody wrote
    return *valid ? (uint64_t)row : 0;
  }
  return tile_row(a.shape, tile_id, lane, a.nrow, valid);
}

// ------------------------------------------------------------------ tile fill

// Row-major rows -> LDS tile[f * 64 + lane].  `missing` values become NaN so the
// walk has one notion of missing.  Returns true if this lane's row has any NaN.
__device__ __forceinline__ bool fill_tile_rows(float* __restrict__ tile, const float* __restrict__ rows,
                                               uint64_t row, bool valid, uint32_t ncol, uint32_t nfeat,
                                               float missing, bool missing_is_nan, uint32_t* flags) {
  bool any_nan = false;
  bool any_inf = false;
  const float qnan = __builtin_nanf("");
  const float* p = rows + row * (uint64_t)ncol;
  uint32_t f = 0;
  if (valid) {
    for (; f + 4 <= ncol; f += 4) {
      f4u v = __builtin_nontemporal_load(reinterpret_cast<const f4u*>(p + f));
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float x = v[c];
        any_inf |= is_inf(x);
        if (!missing_is_nan && x == missing) x = qnan;
        any_nan |= (x != x);
        tile[(f + c) * kWave] = x;
      }
    }
    for (; f < ncol; ++f) {
      float x = __builtin_nontemporal_load(p + f);
      any_inf |= is_inf(x);
      if (!missing_is_nan && x == missing) x = qnan;
      any_nan |= (x != x);
      tile[f * kWave] = x;
    }
    // a booster with more features than the matrix has columns sees them as missing
    for (; f < nfeat; ++f) {
      tile[f * kWave] = qnan;
      any_nan = true;
    }
    if (any_inf && !is_inf(missing) && flags) atomicOr(flags, kFlagInfInput);
  } else {
    for (; f < nfeat; ++f) tile[f * kWave] = 0.0f;
  }
  return any_nan;
}

// One lane's row of the fused path into its column of the wave's tile: field after field (the reference's gather,
// OH_GridCompMod.F90:308-345, PL / 100 at :314), `missing` -> NaN.  The fields come one after the other, each load
// waited for before the next is issued: issuing the 27 loads together is SLOWER (35.2 against 33.6 ms per C360 step) - a
// burst of 27 four-byte gathers from one wave stands in front of the other waves' node gathers at the texture
// addresser (profiles/r03_sweeps.txt).  Returns whether the lane's row holds a missing value.
__device__ __forceinline__ bool fill_tile_fields(const FieldsArgs& a, float* __restrict__ tile, uint32_t nfeat, uint64_t m,
                                                 uint64_t col, uint64_t slab, bool valid, bool missing_is_nan) {
  const float qnan = __builtin_nanf("");
  bool lane_nan = false, any_inf = false;
  for (uint32_t f = 0; f < nfeat; ++f) {
    float x = qnan;
    if (valid && f < a.nfield) {
      const float* src = a.field[f];
      x = ((a.is2d_mask >> f) & 1u) ? src[col] : __builtin_nontemporal_load(src + slab + m);
      if (f == a.pl_feature) x = x / 100.0f;
      any_inf |= is_inf(x);
      if (!missing_is_nan && x == a.missing) x = qnan;
    }
    if (!valid) x = 0.0f;
    lane_nan |= (x != x);
    tile[f * kWave] = x;
  }
  if (any_inf && !is_inf(a.missing) && a.flags) atomicOr(a.flags, kFlagInfInput);
  return lane_nan;
}

// The same for a SMALL slab (a GEOS rank's block; PredictArgs::leaf_buf: a tile is filled by every wave that walks a run
// of its trees): nine loads in flight at a time.  There the chip is far from full, nobody's gathers stand behind a burst
// of loads, and 27 round trips in a row were a tenth of the kernel (48 x 24 x 72 block: 74 -> 67 us; profiles/r04_sweeps.txt).
__device__ __forceinline__ bool fill_tile_fields_burst(const FieldsArgs& a, float* __restrict__ tile, uint32_t nfeat, uint64_t m,
                                                       uint64_t col, uint64_t slab, bool valid, bool missing_is_nan) {
  constexpr uint32_t GROUP = 9;
  const float qnan = __builtin_nanf("");
  bool lane_nan = false, any_inf = false;
  const uint64_t at3 = valid ? slab + m : slab, at2 = valid ? col : 0;      // a lane without a gridcell: the slab's first
  for (uint32_t f0 = 0; f0 < nfeat; f0 += GROUP) {
    float x[GROUP];
#pragma unroll
    for (uint32_t g = 0; g < GROUP; ++g) {
      const uint32_t f = f0 + g;
      x[g] = qnan;
      if (f < nfeat && f < a.nfield) {
        const float* src = a.field[f];
        x[g] = ((a.is2d_mask >> f) & 1u) ? src[at2] : __builtin_nontemporal_load(src + at3);
      }
    }
#pragma unroll
    for (uint32_t g = 0; g < GROUP; ++g) {
      const uint32_t f = f0 + g;
      float v = x[g];
      if (f == a.pl_feature) {
        asm volatile("");      // a branch, taken for one field, not a division computed for all and selected
        v = v / 100.0f;
      }
      any_inf |= f < nfeat && is_inf(v);
      if (!missing_is_nan && v == a.missing) v = qnan;
      if (!valid) v = 0.0f;
      lane_nan |= f < nfeat && (v != v);
      if (f < nfeat) tile[f * kWave] = v;
    }
  }
  if (any_inf && !is_inf(a.missing) && a.flags) atomicOr(a.flags, kFlagInfInput);
  return lane_nan;
}

// Where slab row m = i + im * (j + jm * k') sits in the 3-D arrays, counted from the slab's first level, and in the 2-D
// ones: m itself and m % (im * jm) - unless the (im, jm) at hand is a range of j of a wider grid (FieldsArgs::level_stride)
struct CellAt {
  uint64_t at3, at2;
};
__device__ __forceinline__ CellAt cell_at(const FieldsArgs& a, uint64_t plane, uint64_t m) {
  const uint64_t k = m / plane, col = m - k * plane;
  return CellAt{a.level_stride ? k * a.level_stride + col : m, col};
}
__device__ __forceinline__ uint64_t level_floats(const FieldsArgs& a, uint64_t plane) {
  return a.level_stride ? a.level_stride : plane;
}

// 10.0**x for a float32 x (OH_ML = 10.0 ** xx_pred, OH_GridCompMod.F90:369), rounded ONCE from a double that is good to
// 2e-14: x log2(10) in double, split into an integer and |f| <= 1/2, 2**f as its Taylor polynomial of degree 11 (last
// term 6e-15), scaled by v_ldexp_f64.  It agrees with a correctly rounded powf wherever the double does not sit within
// 2e-14 of a float32 rounding boundary (3 in 10 million values; 0 of 20 000 against 60-digit arithmetic on the host).
// Until round 4 this was (float)pow(10.0, (double)x): the library's general double pow held the fused ring kernel at 121
// vector registers - every register of the CU, four waves per SIMD at 128 - where the walk itself needs 88 (r5): with it
// nothing could run beside a fused walk (OH Run1's streaming kernels, a collective's kernels).
__device__ __forceinline__ float pow10_f32(float x) {
  double y = (double)x * 3.321928094887362;           // log2(10)
  y = y < -1100.0 ? -1100.0 : (y > 1100.0 ? 1100.0 : y);     // +-inf -> 0 / inf below; NaN passes through
  const double n = __builtin_rint(y);
  const double f = y - n;                             // exact
  double p = 4.44553827187081e-10;                    // ln(2)**k / k!, k = 11 .. 0
  p = __builtin_fma(p, f, 7.054911620801121e-09);
  p = __builtin_fma(p, f, 1.0178086009239696e-07);
  p = __builtin_fma(p, f, 1.3215486790144305e-06);
  p = __builtin_fma(p, f, 1.5252733804059838e-05);
  p = __builtin_fma(p, f, 0.00015403530393381606);
  p = __builtin_fma(p, f, 0.0013333558146428441);
  p = __builtin_fma(p, f, 0.009618129107628477);
  p = __builtin_fma(p, f, 0.055504108664821576);
  p = __builtin_fma(p, f, 0.2402265069591007);
  p = __builtin_fma(p, f, 0.6931471805599453);
  p = __builtin_fma(p, f, 1.0);
  return (float)__builtin_ldexp(p, (int)n);
}

// what the fused path stores for a gridcell: the margin (optional) and 10**margin * OHscale (:369, :1569)
__device__ __forceinline__ void store_oh(const FieldsArgs& a, float* __restrict__ out, float* __restrict__ margin_out,
                                         uint64_t slab_out, uint64_t m, uint64_t at3, float acc) {
  if (margin_out) margin_out[m] = acc;
  float oh = acc;
  if (a.apply_pow10) oh = pow10_f32(acc);
  oh = oh * a.scale;
  out[slab_out + at3] = oh;
}

// The OH shape (27 columns, 27 features): a row held in registers, so the NEXT tile's rows can
// be in flight from HBM while the current tile is walked (the walk needs no registers of it).
struct Row27 {
  f4u v[7];   // a lane's own row in v[0] .. v[6][2]; or seven 16-byte pieces of the wave's tile (load_pieces)
};

__device__ __forceinline__ void load_row27(Row27& r, const float* __restrict__ rows, uint64_t row) {
  const float* p = rows + row * 27u;
#pragma unroll
  for (int q = 0; q < 6; ++q) r.v[q] = __builtin_nontemporal_load(reinterpret_cast<const f4u*>(p + 4 * q));
#pragma unroll
  for (int q = 0; q < 3; ++q) r.v[6][q] = __builtin_nontemporal_load(p + 24 + q);
}

__device__ __forceinline__ bool store_row27(float* __restrict__ tile, const Row27& r, bool valid, float missing,
                                            bool missing_is_nan, uint32_t* flags) {
  bool any_nan = false, any_inf = false;
  const float qnan = __builtin_nanf("");
#pragma unroll
  for (int f = 0; f < 27; ++f) {
    float x = r.v[f / 4][f % 4];
    if (!valid) x = 0.0f;
    any_inf |= is_inf(x);
    if (!missing_is_nan && x == missing) x = qnan;
    any_nan |= (x != x);
    tile[f * kWave] = x;
  }
  if (any_inf && !is_inf(missing) && flags) atomicOr(flags, kFlagInfInput);
  return any_nan;
}

// The same rows fetched by the wave TOGETHER.  A tile of a grid is made of runs of 2^run_log rows that follow each
// other in memory (a brick's cells along i; all 64 rows where no grid is known): 27 * 2^run_log floats, a multiple of
// 16 bytes from run_log = 2 on.  Lane l fetches the 16-byte pieces l, l + 64, ... of the tile's 432, whatever
// rows they belong to: consecutive lanes read consecutive memory, every 64-byte line is asked for once, and a quad
// of lanes asks the L1 for one or two blocks instead of four to eight - per-row loads made the row stream the
// texture addresser's second-largest customer.  The pieces go to the tile by element.
//   run g of a tile: its first row is the row of lane  lane_of(g) = g's low run_lo_bits | rest << (run_lo_bits + run_log)
//   row r of run g belongs to lane  lane_of(g) | r << run_lo_bits
// (level-fastest bricks: run_lo_bits = lk; i-fastest and gridless tiles: 0.)
constexpr uint32_t kPiecesPerTile = kWave * 27u / 4u;   // 432

__device__ __forceinline__ uint32_t div27(uint32_t x) { return (x * 1214u) >> 15; }   // exact below 1728

__device__ __forceinline__ uint32_t run_first_lane(uint32_t g, uint32_t run_log, uint32_t lo_bits) {
  return (g & ((1u << lo_bits) - 1u)) | ((g >> lo_bits) << (lo_bits + run_log));
}

// `row`: this lane's own row (launch_row), meaningful or not; the runs' first rows are taken from their lanes
__device__ __forceinline__ void load_pieces(Row27& r, const PredictArgs& a, uint64_t row, int lane) {
  // what follows depends on the lane and the launch only: kept from being hoisted out of the tile loop, where it
  // would sit in some forty registers across the walk
  asm volatile("" : "+v"(lane));
  const uint32_t ppr = 27u << (a.run_log - 2u);            // pieces per run
  const int64_t total = (int64_t)a.nrow * 27;
  const uint32_t row_lo = (uint32_t)row, row_hi = (uint32_t)(row >> 32);
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const uint32_t p = (uint32_t)lane + 64u * (uint32_t)q;
    const uint32_t pc = p < kPiecesPerTile ? p : kPiecesPerTile - 1u;
    const uint32_t g = div27(pc) >> (a.run_log - 2u);
    const uint32_t w = pc - g * ppr;
    const int src = (int)(run_first_lane(g, a.run_log, a.run_lo_bits) << 2);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)row_lo);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)row_hi);
    const int64_t first = (int64_t)(((uint64_t)hi << 32) | lo) * 27 + (int64_t)(4u * w);   // float index of the piece
    // pieces of rows outside the matrix are read from inside it (their lanes carry no row: store_pieces writes
    // zeros); the tiles that hold the matrix's first or last row are not fetched this way (see the kernel)
    const int64_t inside = first < 0 ? 0 : (first + 4 > total ? total - 4 : first);
    r.v[q] = __builtin_nontemporal_load(reinterpret_cast<const f4u*>(a.rows + inside));
  }
}

// tile_base: the wave's tile (feature f of lane l at tile_base[f * 64 + l]); vmask: the lanes that carry a row
__device__ __forceinline__ bool store_pieces(float* __restrict__ tile_base, const Row27& r, const PredictArgs& a,
                                             uint64_t vmask, int lane, bool missing_is_nan) {
  bool any_nan = false, any_inf = false;
  const float qnan = __builtin_nanf("");
  asm volatile("" : "+v"(lane));   // as in load_pieces
  const uint32_t ppr = 27u << (a.run_log - 2u);
  const bool all_valid = vmask == ~0ull;
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const uint32_t p = (uint32_t)lane + 64u * (uint32_t)q;
    if (p < kPiecesPerTile) {
      const uint32_t g = div27(p) >> (a.run_log - 2u);
      const uint32_t w4 = 4u * (p - g * ppr);                 // element of the run this piece starts at
      const uint32_t col0 = run_first_lane(g, a.run_log, a.run_lo_bits);
      uint32_t rr = div27(w4);
      uint32_t f = w4 - 27u * rr;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t col = col0 | (rr << a.run_lo_bits);
        float x = r.v[q][c];
        if (!all_valid && !((vmask >> col) & 1ull)) x = 0.0f;
        any_inf |= is_inf(x);
        if (!missing_is_nan && x == a.missing) x = qnan;
        any_nan |= (x != x);
        tile_base[f * kWave + col] = x;
        ++f;
        if (f == 27u) {
          f = 0u;
          ++rr;
        }
      }
    }
  }
  if (any_inf && !is_inf(a.missing) && a.flags) atomicOr(a.flags, kFlagInfInput);
  return any_nan;
}

// ------------------------------------------------------------------ walks

// Packed walk, CHAINS trees at once per lane.  The loop body is straight-line code: all
// feature reads (LDS), one wait, all compares, all node gathers - so the chains' LDS and
// memory latencies overlap.  A chain that has reached its leaf re-reads that leaf (its
// index no longer moves) until every lane of the wave is done with all chains; with
// depth-capped trees that costs a few percent and keeps control flow out of the loop.
template <int CHAINS, bool HAS_MISSING>
__device__ __forceinline__ float walk_packed(__amdgpu_buffer_rsrc_t nodes, const uint32_t* __restrict__ roots,
                                             uint32_t t0, uint32_t t1, float acc, const float* __restrict__ tile) {
  for (uint32_t t = t0; t < t1; t += CHAINS) {
    uint2 nd[CHAINS];
    uint32_t cur[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      const uint32_t tt = (t + c < t1) ? t + c : t1 - 1;  // clamp: the duplicate walk is discarded
      cur[c] = roots[tt];
      nd[c] = load_node8(nodes, cur[c]);
    }
    bool more;
    do {
      float x[CHAINS];
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) x[c] = tile[(nd[c].y & 31u) * kWave];
      more = false;
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        const uint32_t m = nd[c].y;
        bool go_left = x[c] < __uint_as_float(nd[c].x);
        if (HAS_MISSING) go_left = go_left || ((x[c] != x[c]) && (m & 32u));
        const uint32_t n = (m >> 6) - (go_left ? 1u : 0u);
        cur[c] = (m != 0u) ? n : cur[c];
        more = more || (m != 0u);
      }
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) nd[c] = load_node8(nodes, cur[c]);
    } while (__any(more));
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
      if (t + c < t1) acc += __uint_as_float(nd[c].x);
  }
  return acc;
}

template <bool HAS_MISSING>
__device__ __forceinline__ float walk_wide_tile(const uint4* __restrict__ nodes, const uint32_t* __restrict__ roots,
                                                uint32_t t0, uint32_t t1, float acc, const float* __restrict__ tile) {
  for (uint32_t t = t0; t < t1; ++t) {
    uint4 nd = nodes[roots[t]];
    while (nd.y != 0u) {
      const float x = tile[(nd.z & 0x7FFFFFFFu) * kWave];
      bool go_left = x < __uint_as_float(nd.x);
      if (HAS_MISSING) go_left = go_left || ((x != x) && (nd.z >> 31));
      nd = nodes[nd.y + (go_left ? 0u : 1u)];
    }
    acc += __uint_as_float(nd.x);
  }
  return acc;
}

// Super-node walk: one 16-byte gather brings a node and both its children, so a lane
// descends TWO levels per gather (flatten.hpp).  Leaves are recognised by feature code 31.
// Same phase structure as walk_packed: the chains' first-level feature reads, then their
// second-level reads, then their gathers, with nothing but selects in between.  A chain that
// has found its leaf keeps the value and re-reads the super-node it stopped at until the whole
// wave is done.
//
// "pin_super": hipcc splits a plain uint4 load into dword + dwordx3 loads when the components are
// consumed at different points (twice the gathers).  Passing each loaded vector through an
// empty asm statement as ONE 128-bit register tuple, at its first use (never right after the
// load: the wait would land there), keeps every gather a single global_load_dwordx4 and leaves
// the other chains' gathers in flight behind a counted vmcnt.

__device__ __forceinline__ bool go_left_or_default(float x, float thr, bool default_left) {
  // three compares whose results stay lane masks in scalar registers and meet in two scalar operations (as
  // integers in vector registers, the way this was first written, a decision cost eight vector instructions and
  // a batch with missing values walked 46 % slower than one without)
  const bool lt = x < thr;
  const bool missing = x != x;     // a split condition is never NaN
  return lt | (missing & default_left);
}

// One step (two tree levels) of all chains.  There is no "finished" state: a lane that has taken its
// leaf simply walks on, and emit_super guarantees that it lands on filler super-nodes from then on
// (internal-looking, no code 31, leading to group 0 = four fillers), so "the child's value if its code
// is 31" is taken exactly once per walk, at the leaf.  The fillers of one tree share a cache line, so
// lanes that are done cost the L1 next to nothing.
// LAST: the tree's final step - nothing is fetched after it, only the leaf is taken.
// The missing-aware step keeps its decisions as LANE MASKS in scalar registers from the compare to the last use: the
// compares are the mask-valued intrinsics, "right = not-less and (ordered or default-right)" is two scalar operations,
// and the selects and the add-with-carry take the mask as it is (inline assembly: from a bool that is the AND / OR of
// compares hipcc rebuilds every mask it needs with v_cndmask 0/1 + v_cmp_ne, 4 vector instructions per chain and
// step).  21 vector instructions per chain and two-level step, against 25 as plain C and 14 for the walk without
// missing values; C360 with 1e-3 of the entries missing (every tile walks this form): 34.6 -> 30.9 ms (profiles/r04_sweeps.txt;
// behind a wave-uniform branch on "any lane read a NaN at this level": 30.8 ms at 1e-3, 33.3 against 31.3 at 1e-2, not kept).
using lane_mask = uint64_t;
constexpr int kFcmpOrdered = 7, kFcmpNotLess = 11 /* unordered or >= */, kIcmpEq = 32;

// mask ? if_set : if_clear, per lane
__device__ __forceinline__ uint32_t pick(lane_mask m, uint32_t if_clear, uint32_t if_set) {
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
  return r;
}

// lanes that go RIGHT at a node: x is not less than the threshold and is a number, or x is missing and the node's
// default is right (`default_left_bit` of `w` clear)
__device__ __forceinline__ lane_mask right_or_default(float x, float thr, uint32_t w, uint32_t default_left_bit) {
  const lane_mask not_less = __builtin_amdgcn_fcmpf(x, thr, kFcmpNotLess);
  const lane_mask number = __builtin_amdgcn_fcmpf(x, x, kFcmpOrdered);
  const lane_mask default_right = __builtin_amdgcn_uicmp(w & default_left_bit, 0u, kIcmpEq);
  return not_less & (number | default_right);
}

// 2 v + carry-in, per lane
__device__ __forceinline__ uint32_t double_and_carry(uint32_t v, lane_mask carry) {
  uint32_t r;
  asm("v_addc_co_u32_e64 %0, vcc, %1, %1, %2" : "=v"(r) : "v"(v), "s"(carry) : "vcc");
  return r;
}

template <int CHAINS, bool HAS_MISSING, bool LAST>
__device__ __forceinline__ void super_step(u32x4 (&s)[CHAINS], uint32_t (&rel)[CHAINS], uint32_t (&leafb)[CHAINS],
                                           const char* __restrict__ tile_b) {
  float x0[CHAINS], x1[CHAINS];
  uint32_t thr1[CHAINS], f1[CHAINS];
  bool l0[CHAINS];
  lane_mask r0[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) {
    asm volatile("" : "+v"(s[c]));   // one 128-bit tuple: see "pin_super" above
    // meta & 0x1F00 is the byte offset of the node's feature row; a leaf (feature 31) reads row 31, which
    // is the next wave's tile or the first-step table (tile_lds_bytes keeps it inside the block's
    // allocation) and the value is not used
    x0[c] = *reinterpret_cast<const float*>(tile_b + (s[c].w & 0x1F00u));
  }
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) {
    const uint32_t w = s[c].w;
    if constexpr (HAS_MISSING) {
      // missing (NaN) -> default child; bit 5 of w: the node's default is left
      r0[c] = right_or_default(x0[c], __uint_as_float(s[c].x), w, 32u);
      thr1[c] = pick(r0[c], s[c].y, s[c].z);
      f1[c] = (w >> pick(r0[c], 0u, 13u)) & 31u;
    } else {
      const bool l = x0[c] < __uint_as_float(s[c].x);
      l0[c] = l;
      thr1[c] = l ? s[c].y : s[c].z;
      f1[c] = (w >> (l ? 0u : 13u)) & 31u;
    }
    if (!LAST) x1[c] = *reinterpret_cast<const float*>(tile_b + (f1[c] << 8));
  }
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) {
    const uint32_t w = s[c].w;
    leafb[c] = (f1[c] == 31u) ? thr1[c] : leafb[c];
    if (!LAST) {
      // record number of the next step = 4 g + 2 (go right at the node) + (go right at the child), as two doublings with
      // the decisions carried in: two v_addc_co_u32 whose carry-in is the compare's lane mask - 14 VALU instructions per
      // chain and step instead of 16 (hipcc turns the C form, in whatever spelling, back into shifts, selects and ors);
      // depth 18 / 10 / 6: -0.4 / -1.7 / -1.8 %, the fused kernel -1.2 % (profiles/r04_sweeps.txt)
      const uint32_t g = w >> 18;
      if constexpr (HAS_MISSING) {
        // bits 6 / 7 of w: the left / right child's default is left
        const lane_mask r1 = right_or_default(x1[c], __uint_as_float(thr1[c]), w, pick(r0[c], 64u, 128u));
        rel[c] = double_and_carry(double_and_carry(g, r0[c]), r1);
      } else {
        const bool l1 = x1[c] < __uint_as_float(thr1[c]);
        const lane_mask right0 = __builtin_amdgcn_ballot_w64(!l0[c]), right1 = __builtin_amdgcn_ballot_w64(!l1);
        rel[c] = double_and_carry(double_and_carry(g, right0), right1);
      }
    }
  }
}

// The same step (never the last one) CHAIN BY CHAIN, each chain's next record asked for the moment its number is known:
// chain c's two feature reads, its decisions, its gather - and only then chain c + 1, whose first feature has been asked
// for meanwhile (the skew).  super_step above goes stage by stage through all chains and leaves the four gathers for the
// end of the step: the LDS reads of four chains overlap nicely, but every record then has to arrive between the end of
// one step and the start of the next, with nothing of this wave to do.  Chain by chain a record has the other chains'
// work - three quarters of a step - to arrive in.  (r6) The ring kernels' deep steps, where the texture addresser is the
// bound and the vector ALU has slack: C360 step 24.19 -> 23.35 ms (-3.5 %), fused fields 26.64 -> 25.81, OH Run1 slab
// 20.02 -> 19.35, 1e-3 missing 30.9 -> 29.2, depth 14 17.47 -> 17.03; the resident (LDS) steps the same way are 6 %
// SLOWER (their records come in tens of cycles: there the stage-by-stage overlap of the feature reads is what counts).
// Without the sched_barrier hipcc re-merges the chains (23.56); chains in pairs 23.38; without the skew 23.52; with both
// children's features read speculatively beside the node's own (one LDS latency less, three vector instructions more) 25.0
// (profiles/r06_sweeps.txt).  `fetch(c, record)` starts the load of chain c's next record and returns its registers.
template <int CHAINS, bool HAS_MISSING, class Fetch>
__device__ __forceinline__ void super_step_by_chain(u32x4 (&s)[CHAINS], uint32_t (&rel)[CHAINS], uint32_t (&leafb)[CHAINS],
                                                    const char* __restrict__ tile_b, Fetch fetch) {
  asm volatile("" : "+v"(s[0]));   // one 128-bit tuple: see "pin_super" above
  float x0_next = *reinterpret_cast<const float*>(tile_b + (s[0].w & 0x1F00u));
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) {
    const uint32_t w = s[c].w;
    const float x0 = x0_next;
    if (c + 1 < CHAINS) {
      asm volatile("" : "+v"(s[c + 1]));
      x0_next = *reinterpret_cast<const float*>(tile_b + (s[c + 1].w & 0x1F00u));
    }
    uint32_t thr1, f1;
    bool l0 = false;
    lane_mask r0 = 0;
    if constexpr (HAS_MISSING) {
      r0 = right_or_default(x0, __uint_as_float(s[c].x), w, 32u);
      thr1 = pick(r0, s[c].y, s[c].z);
      f1 = (w >> pick(r0, 0u, 13u)) & 31u;
    } else {
      l0 = x0 < __uint_as_float(s[c].x);
      thr1 = l0 ? s[c].y : s[c].z;
      f1 = (w >> (l0 ? 0u : 13u)) & 31u;
    }
    const float x1 = *reinterpret_cast<const float*>(tile_b + (f1 << 8));
    const uint32_t g = w >> 18;
    if constexpr (HAS_MISSING) {
      const lane_mask r1 = right_or_default(x1, __uint_as_float(thr1), w, pick(r0, 64u, 128u));
      rel[c] = double_and_carry(double_and_carry(g, r0), r1);
    } else {
      const bool l1 = x1 < __uint_as_float(thr1);
      const lane_mask right0 = __builtin_amdgcn_ballot_w64(!l0), right1 = __builtin_amdgcn_ballot_w64(!l1);
      rel[c] = double_and_carry(double_and_carry(g, right0), right1);
    }
    s[c] = fetch(c, rel[c]);
    leafb[c] = (f1 == 31u) ? thr1 : leafb[c];      // off the path to the gather: behind it (23.32 -> 23.12 ms)
    __builtin_amdgcn_sched_barrier(0);      // the next chain's work stays behind this chain's gather
  }
}

// `first`: the block's LDS copy of every tree's first-step super-nodes (kFirstTrees trees, two
// entries each: the root's children, or the root's own super-node twice), filled by the kernel.
// The first gather of a walk is the cheapest for the L1 (one block per tree) but costs the
// texture addresser its fixed ~11 cycles like any other; from LDS it costs it nothing.
constexpr uint32_t kFirstTrees = kFirstStepTrees;
constexpr uint32_t kFirstBytes = kFirstTrees * 2 * 16;

// Hands every lane the record that lane `rel` of the wave holds (rel < 64): four ds_bpermute_b32, which use the
// LDS crossbar but neither LDS memory nor the texture addresser.
template <int CHAINS>
__device__ __forceinline__ void take_from_lanes(u32x4 (&s)[CHAINS], const u32x4 (&top)[CHAINS],
                                                const uint32_t (&rel)[CHAINS]) {
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) {
    const int a = (int)(rel[c] << 2);
    s[c].x = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)top[c].x);
    s[c].y = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)top[c].y);
    s[c].z = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)top[c].z);
    s[c].w = (uint32_t)__builtin_amdgcn_ds_bpermute(a, (int)top[c].w);
  }
}

// TOPS: fetch the first kSuperTopSlots records of each tree with ONE coalesced load per wave (a record per lane)
// and take the super-nodes of steps 2 and 3 from the lane that holds them, instead of two gathers per lane.
// A gather costs the texture addresser its fixed 16 cycles even when the whole wave wants the same two or
// three records, as it does at the top of a tree.  It pays where the addresser is the bound -
// deep trees, whose lower steps keep it busy (C360 step, depth 18: 34.5 -> 33.4 ms) - and costs where it is not:
// the load brings twelve 64-byte lines where the two gathers touch two to four, and the eight ds_bpermute per
// chain are work the shallow walk has no slack for (depth 6: 11.5 -> 14.6 ms, depth 10: 16.5 -> 18.3 ms).  The
// host picks the kernel by the forest's mean step count (capi.cpp device_forest).
template <int CHAINS, bool HAS_MISSING, bool TOPS>
__device__ __forceinline__ float walk_super(const uint4* __restrict__ nodes, const SuperTreeHead* __restrict__ heads,
                                            uint32_t t0, uint32_t t1, float acc, const float* __restrict__ tile,
                                            const char* __restrict__ first, uint32_t nfirst, uint32_t nodes_bytes,
                                            float* __restrict__ leaf_out = nullptr) {
  if (t0 >= t1) return acc;
  // the gathers below the tree tops go through a buffer descriptor over the whole forest: 0.7 % faster than the
  // same loads as global loads (31.03 against 31.26 ms, three runs each), and an index that strays reads zeros
  const __amdgpu_buffer_rsrc_t forest = make_rsrc(nodes, nodes_bytes);
  // lane l holds record min(l, kSuperTopSlots - 1) of the tree: the walk never asks for one beyond
  const uint32_t lane_id = threadIdx.x & (kWave - 1);
  const uint32_t top_off = (lane_id < kSuperTopSlots ? lane_id : kSuperTopSlots - 1u) << 4;
  const lds_cptr first_lds = (lds_cptr)first;
  const u32x4* __restrict__ nodes_v = reinterpret_cast<const u32x4*>(nodes);
  const char* tile_b = reinterpret_cast<const char*>(tile);
  // tree heads are wave-uniform (scalar loads); the next group's are fetched a whole walk ahead
  SuperTreeHead hn[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) hn[c] = heads[(t0 + c < t1) ? t0 + c : t1 - 1];
  for (uint32_t t = t0; t < t1; t += CHAINS) {
    SuperTreeHead h[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      h[c] = hn[c];
      const uint32_t tn = t + CHAINS + c;
      hn[c] = heads[tn < t1 ? tn : t1 - 1];               // clamp: a duplicate walk is discarded below
    }
    u32x4 s[CHAINS], top[CHAINS];
    // a lane's place in its tree: super-node index relative to the tree's first; the tree's own
    // address is wave-uniform, so a gather is base (SGPR pair) + 32-bit byte offset
    const char* tb[CHAINS];
    uint32_t rel[CHAINS], leafb[CHAINS];
    float xr[CHAINS];
    // the trees' tops first: they are in flight during the root's compare and the first step
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      tb[c] = reinterpret_cast<const char*>(nodes_v + h[c].base);
      if (TOPS) top[c] = *reinterpret_cast<const u32x4*>(tb[c] + (uint64_t)top_off);
    }
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) xr[c] = tile[(h[c].root_meta & 31u) * kWave];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      bool l = xr[c] < h[c].root_thr;
      if (HAS_MISSING) l = go_left_or_default(xr[c], h[c].root_thr, (h[c].root_meta & 32u) != 0u);
      // group 1 holds the root's super-node (phase 0) or those of its two children (phase 1)
      rel[c] = 4u + (((h[c].root_meta & 0x100u) && !l) ? 1u : 0u);
      leafb[c] = 0u;
    }
    // The trip count is the tree's (a scalar from its head), not a vote of the lanes: no branch
    // waits for the chains' compares, and the last step fetches nothing.  The shallower tree of a group
    // walks on through fillers (flatten.hpp).
    uint32_t nsteps = h[0].steps;
#pragma unroll
    for (int c = 1; c < CHAINS; ++c) nsteps = h[c].steps > nsteps ? h[c].steps : nsteps;
    if (t + CHAINS <= nfirst) {
      // step 1 from the block's LDS table: it does not wait for the tops
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        const uint32_t tt = (t + c < t1) ? t + c : t1 - 1;
        s[c] = *(lds_u32x4_ptr)(first_lds + ((2u * tt + (rel[c] - 4u)) << 4));
      }
    } else if (TOPS) {
      take_from_lanes<CHAINS>(s, top, rel);
    } else {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) s[c] = *reinterpret_cast<const u32x4*>(tb[c] + (uint64_t)(rel[c] << 4));
    }
    if (TOPS) {
      // a shallow tree in a deep forest takes four steps as well: past its leaf a walk only meets fillers
      nsteps = nsteps < 4u ? 4u : nsteps;
      super_step<CHAINS, HAS_MISSING, false>(s, rel, leafb, tile_b);
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) asm volatile("" : "+v"(top[c]));   // one global_load_dwordx4 each
      take_from_lanes<CHAINS>(s, top, rel);
      super_step<CHAINS, HAS_MISSING, false>(s, rel, leafb, tile_b);
      take_from_lanes<CHAINS>(s, top, rel);
      for (uint32_t step = 3; step < nsteps; ++step) {
        super_step<CHAINS, HAS_MISSING, false>(s, rel, leafb, tile_b);
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
          s[c] = __builtin_amdgcn_raw_buffer_load_b128(forest, (int)((h[c].base + rel[c]) << 4), 0, 0);
      }
    } else {
      for (uint32_t step = 1; step < nsteps; ++step) {
        super_step<CHAINS, HAS_MISSING, false>(s, rel, leafb, tile_b);
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
          s[c] = __builtin_amdgcn_raw_buffer_load_b128(forest, (int)((h[c].base + rel[c]) << 4), 0, 0);
      }
    }
    super_step<CHAINS, HAS_MISSING, true>(s, rel, leafb, tile_b);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
      if (t + c < t1) acc += __uint_as_float(leafb[c]);
    // trees split over waves (PredictArgs::leaf_buf): the leaves themselves, for the launch that sums them in order
    if (leaf_out != nullptr) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c)
        if (t + c < t1) leaf_out[(size_t)(t + c) * kWave] = __uint_as_float(leafb[c]);
    }
  }
  return acc;
}

// FMT: 0 = wide 16-byte nodes, 1 = packed 8-byte nodes, 2 = 16-byte super-nodes
template <int FMT, int CHAINS, bool TOPS>
__device__ __forceinline__ float walk_tile_any(const DeviceForest& fr, const SuperTreeHead* __restrict__ heads, uint32_t t0,
                                               uint32_t t1, const float* tile, bool wave_has_missing, const char* first,
                                               uint32_t nfirst, float* __restrict__ leaf_out);

// a wave that walks is issued before a wave that fills its tile or stores (s_setprio): depth 6 / 10 / 18: -3.2 / -2.2 /
// -0.7 % (profiles/r04_sweeps.txt; the ring kernels, where it is worth 8 %, have their own two levels)
template <int FMT, int CHAINS, bool TOPS>
__device__ __forceinline__ float walk_tile(const DeviceForest& fr, const SuperTreeHead* __restrict__ heads, uint32_t t0,
                                           uint32_t t1, const float* tile, bool wave_has_missing, const char* first,
                                           uint32_t nfirst, float* __restrict__ leaf_out = nullptr) {
  __builtin_amdgcn_s_setprio(1);
  const float acc = walk_tile_any<FMT, CHAINS, TOPS>(fr, heads, t0, t1, tile, wave_has_missing, first, nfirst, leaf_out);
  __builtin_amdgcn_s_setprio(0);
  return acc;
}

template <int FMT, int CHAINS, bool TOPS>
__device__ __forceinline__ float walk_tile_any(const DeviceForest& fr, const SuperTreeHead* __restrict__ heads, uint32_t t0,
                                               uint32_t t1, const float* tile, bool wave_has_missing, const char* first,
                                               uint32_t nfirst, float* __restrict__ leaf_out) {
  float acc = fr.base_score;
  if constexpr (FMT == 1) {
    const __amdgpu_buffer_rsrc_t nodes = make_rsrc(fr.packed, fr.packed_bytes);
    return wave_has_missing ? walk_packed<CHAINS, true>(nodes, fr.roots, t0, t1, acc, tile)
                            : walk_packed<CHAINS, false>(nodes, fr.roots, t0, t1, acc, tile);
  } else if constexpr (FMT == 2) {
    // the deep gathers go through a buffer descriptor over the forest (walk_super); the tree tops are plain loads
    const uint4* nodes = reinterpret_cast<const uint4*>(fr.super);
    return wave_has_missing ? walk_super<CHAINS, true, TOPS>(nodes, heads, t0, t1, acc, tile, first, nfirst, fr.super_bytes, leaf_out)
                            : walk_super<CHAINS, false, TOPS>(nodes, heads, t0, t1, acc, tile, first, nfirst, fr.super_bytes, leaf_out);
  } else {
    const uint4* nodes = reinterpret_cast<const uint4*>(fr.wide);
    return wave_has_missing ? walk_wide_tile<true>(nodes, fr.roots, t0, t1, acc, tile)
                            : walk_wide_tile<false>(nodes, fr.roots, t0, t1, acc, tile);
  }
}

// Fills the block's first-step table (walk_super) - one entry per thread - and returns how many
// trees it covers.  Every thread of the block must call it (it ends in a barrier).
template <int FMT>
__device__ __forceinline__ uint32_t fill_first_steps(const DeviceForest& fr, const SuperTreeHead* __restrict__ heads,
                                                     char* first) {
  if constexpr (FMT != 2) return 0u;
  const uint32_t nfirst = fr.num_trees < kFirstTrees ? fr.num_trees : kFirstTrees;
  for (uint32_t e = threadIdx.x; e < 2u * nfirst; e += blockDim.x) {
    const SuperTreeHead h = heads[e >> 1];
    const uint32_t slot = h.base + 4u + ((h.root_meta & 0x100u) ? (e & 1u) : 0u);
    reinterpret_cast<u32x4*>(first)[e] = reinterpret_cast<const u32x4*>(fr.super)[slot];
  }
  __syncthreads();
  return nfirst;
}

// ------------------------------------------------------------------ kernels

// AoS rows in, margins out.  PREFETCH27: 27-column rows, next tile's rows prefetched into
// registers during the walk (used when a launch gives every wave more than one tile).
template <int FMT, int CHAINS, bool PREFETCH27, bool TOPS>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(CHAINS > 3 ? 4 : 5))) void predict_rows_tile_kernel(
    DeviceForest fr, PredictArgs a, const SuperTreeHead* __restrict__ heads, float* __restrict__ out) {
  // `heads` (= fr.super_heads) and `out` (= a.out) are kernel arguments of their own so that they carry
  // noalias: with the margins' stores provably elsewhere, the wave-uniform head records stay scalar
  // loads (as members of the by-value structs they turn into one more vector load per walk)
  extern __shared__ float lds[];
  // the launch behind a ring train (launch_rows_ring): nothing to do unless a block of that train gave up
  if (a.only_if_train != 0u) {
    if (__hip_atomic_load(a.flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.only_if_train) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.flags + 1, 1u);
  }
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  // the waves' feature tiles first, the first-step table behind them (see tile_lds_bytes)
  float* tile = lds + (size_t)wave * fr.num_feature * kWave + lane;
  char* first = reinterpret_cast<char*>(lds + (size_t)kWavesPerBlock * fr.num_feature * kWave);
  const uint32_t nfirst = fill_first_steps<FMT>(fr, heads, first);
  const bool missing_is_nan = a.missing != a.missing;
  // Blocks b and b + 8 share an XCD (round-robin dispatch, observed, speed only): give
  // every XCD one contiguous run of tiles so that its CUs walk neighbouring gridcells and
  // share node lines in that XCD's L2.
  uint32_t block = blockIdx.x;
  if (a.xcd_remap && (gridDim.x & 7u) == 0u) block = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint64_t wave_id = (uint64_t)block * kWavesPerBlock + wave;
  const uint64_t nwaves = (uint64_t)gridDim.x * kWavesPerBlock;
  if constexpr (PREFETCH27) {
    uint64_t tile_id = a.tile_begin + wave_id;
    // The wave fetches its tile's rows together where they come in runs (load_pieces), else every lane its own;
    // so does the tile that holds the matrix's first or last row: a 16-byte piece of those could hang over the
    // end of the caller's buffer.
    float* tile_base = tile - lane;
    Row27 regs;
    bool valid = false, together = false;
    uint64_t row = 0;
    if (tile_id < a.tile_end) {
      row = launch_row(a, tile_id, lane, &valid);
      together = a.run_log >= 2u && !__any(valid && (row == 0 || row + 1 == a.nrow));
      if (together) load_pieces(regs, a, row, lane);
      else load_row27(regs, a.rows, valid ? row : 0);
    }
    while (tile_id < a.tile_end) {
      // a tile none of whose lanes carries a row (bricks are laid over whole levels: a matrix that ends inside a
      // level, or is smaller than one, leaves such tiles behind it) is not filled and not walked
      const bool live = __any(valid);
      const bool lane_nan = !live ? false
                            : together ? store_pieces(tile_base, regs, a, __ballot(valid), lane, missing_is_nan)
                                       : store_row27(tile, regs, valid, a.missing, missing_is_nan, a.flags);
      const uint64_t next = tile_id + nwaves;
      const uint64_t this_row = row;
      const bool this_valid = valid;
      if (next < a.tile_end) {
        row = launch_row(a, next, lane, &valid);
        together = a.run_log >= 2u && !__any(valid && (row == 0 || row + 1 == a.nrow));
        if (together) load_pieces(regs, a, row, lane);   // in flight during the walk
        else load_row27(regs, a.rows, valid ? row : 0);
      }
      bool wave_nan = __any(lane_nan);
      bool keep = this_valid;
      // fetched together, the tile was written by all lanes for all lanes: LDS serves a wave's instructions in
      // order; the fences and the wave barriers keep the compiler from moving reads and writes across
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // rows with missing values leave for the second launch: this wave then walks without missing-value logic
      // (a NaN compares false everywhere: those lanes walk some path of every tree to its end and are not stored)
      if (wave_nan && a.defer_count != nullptr && a.defer_cap == 0u) {
        // only counting (the host saw too many such rows in the last batch for the second launch to pay): an
        // estimate from every eighth tile, so that nearly every wave of a batch full of missing values does not
        // queue at one counter; nobody waits for the sum
        if ((tile_id & 7u) == 0u) {
          const uint32_t n = 8u * (uint32_t)__popcll(__ballot(lane_nan));
          if (lane == 0) atomicAdd(a.defer_count, n);
        }
      } else if (wave_nan && a.defer_count != nullptr) {
        // whose row it is that holds the NaN: a lane that filled pieces of the tile saw other lanes' elements
        bool mine = false;
#pragma unroll
        for (int f = 0; f < 27; ++f) {
          const float x = tile[f * kWave];
          mine |= (x != x);
        }
        const bool leaves = mine && this_valid;
        const uint64_t who = __ballot(leaves);
        const uint32_t n = (uint32_t)__popcll(who);
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(a.defer_count, n);
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (at + n <= a.defer_cap) {               // room for all of them (a full list stays full: later waves walk as before)
          if (leaves) a.defer_list[at + (uint32_t)__popcll(who & ((1ull << lane) - 1ull))] = (uint32_t)this_row;
          keep = this_valid && !mine;
          wave_nan = false;
        }
      }
      if (live) {
        const float acc = walk_tile<FMT, CHAINS, TOPS>(fr, heads, a.tree_begin, a.tree_end, tile, wave_nan, first, nfirst);
        if (keep) __builtin_nontemporal_store(acc, out + this_row);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      tile_id = next;
    }
    return;
  }
  if constexpr (FMT == 2) {
    if (a.leaf_buf != nullptr) {
      // a small batch: work item = (run of trees, tile), tiles fastest, so that the waves of a block still walk the same
      // trees on neighbouring tiles; every item writes its trees' leaves, combine_leaves_kernel sums them
      const uint32_t ntree = a.tree_end - a.tree_begin;
      const uint32_t per = ((ntree + a.tree_split - 1) / a.tree_split + (CHAINS - 1)) / CHAINS * CHAINS;
      const uint64_t ntiles = a.tile_end - a.tile_begin, items = ntiles * a.tree_split;
      for (uint64_t item = wave_id; item < items; item += nwaves) {
        const uint64_t tile_id = a.tile_begin + item % ntiles;
        const uint32_t t0 = a.tree_begin + (uint32_t)(item / ntiles) * per;
        const uint32_t t1 = t0 + per < a.tree_end ? t0 + per : a.tree_end;
        bool valid;
        const uint64_t row = launch_row(a, tile_id, lane, &valid);
        if (!__any(valid) || t0 >= t1) continue;
        const bool lane_nan = fill_tile_rows(tile, a.rows, row, valid, a.ncol, fr.num_feature, a.missing,
                                             missing_is_nan, a.flags);
        float* leaves = a.leaf_buf + ((size_t)(tile_id - a.tile_begin) * ntree - a.tree_begin) * kWave + lane;
        (void)walk_tile<FMT, CHAINS, TOPS>(fr, heads, t0, t1, tile, __any(lane_nan), first, nfirst, leaves);
      }
      return;
    }
  }
  // the second launch of a deferred-rows predict: how many slots the first one filled is only known on the device
  uint64_t slots = ~0ull, tile_end = a.tile_end;
  if (a.perm_count != nullptr) {
    const uint32_t filled = *a.perm_count;
    slots = filled < a.perm_slots ? filled : a.perm_slots;
    const uint64_t need = (slots + kWave - 1) / kWave;
    tile_end = need < tile_end ? need : tile_end;
  }
  for (uint64_t tile_id = a.tile_begin + wave_id; tile_id < tile_end; tile_id += nwaves) {
    bool valid;
    const uint64_t row = launch_row(a, tile_id, lane, &valid, slots);
    if (!__any(valid)) continue;                      // nothing of the matrix in this tile
    const bool lane_nan = fill_tile_rows(tile, a.rows, row, valid, a.ncol, fr.num_feature, a.missing,
                                         missing_is_nan, a.flags);
    const bool wave_nan = __any(lane_nan);
    // the tile is private to this wave: its own LDS writes are ordered before its reads
    const float acc = walk_tile<FMT, CHAINS, TOPS>(fr, heads, a.tree_begin, a.tree_end, tile, wave_nan, first, nfirst);
    if (valid) __builtin_nontemporal_store(acc, out + row);
  }
}

// ------------------------------------------------------------------ tree tops resident in LDS: the ring kernels
//
// The tile kernels above pay the texture addresser ~25 cycles for every vector-memory instruction and a
// wave issues seven per tree.  A `ds_read_b128` whose lanes ask for a handful of distinct records costs the LDS 4-8
// cycles (MI355X_MICROARCH.md §LDS: four groups of 16 lanes, a cycle each when the records differ in banks or are
// the same).  The ring kernels keep the records of a walk's first FOUR steps - breadth first they are a tree's first
// 176 (4 fillers + 4 + 8 + 32 + 128; emit_super) - in LDS for the trees being walked, and only steps 5 .. 9 go through
// the addresser: 5 gathers + 0.17 staging loads per wave and tree, against 6 gathers + 1 tree top.
//
// LDS cannot hold that for every wave's own trees, so the 16 waves of a block (1 024 threads, 16 tiles = 110.6 KB, one
// block per CU) walk the SAME trees at about the same time, in groups of four (four chains per lane: with 16 waves
// instead of 20 the walk needs them to keep the addresser busy): a ring of four buffers of 4 x 176 records (44 KB).
// With a barrier per group every wave would wait for the slowest of the block, group after group: measured, that
// costs all the gain and more (32.8 ms against 26.7 with the barriers taken out, 31.1 for the tile kernel;
// profiles/r04_sweeps.txt).  So the waves only meet through three kinds of words in LDS:
//   progress[w]  groups wave w is done with (written by w)
//   claim        next group nobody has started to stage (compare-and-swap by the first wave to reach the group before)
//   filled       groups that are complete in the ring (written by their stager)
// The wave that is first at group g stages group g + 1: `buffer_load ... lds` straight into the ring (no registers, no
// ds_write), once every wave is done with the group whose buffer that is (g + 1 - 4); it then walks group g like
// everybody else and publishes g + 1 from inside that walk, behind the LDS steps, when its DMA has had their time to
// land.  Nobody waits for anybody unless it is two groups ahead of the slowest wave or its next group has not landed.
// Every wait is on work with a lower group number, so the waves cannot wait in a circle; the spins are bounded all the
// same.  A wave that gives up says so in a fourth word, `abort`, which every spinning wave of the block reads: from then
// on nobody in the block waits, stages, walks or stores - the block runs through its rounds in no time, raises
// the id of its launch train in flags[2], and the launch the launcher put behind the train (the tile kernel, predicated
// on that id) walks the train's rows instead.  (Until round 4 a time-out was an error: rc -1 into the caller's _ASSERT,
// OH_GridCompMod.F90:358; and the waves that had not given up themselves spun on, limit after limit.)
//
// C360 step: 27.4 ms against the tile kernel's 31.1, and 24.3 with issue priorities (ring_walk_group); with them the ring
// wins from 5 steps per tree on (depth 10: 13.8 against 14.2 ms; depth 8: 12.0 against 11.8), which is where the host
// starts to pick these kernels (capi.cpp pick_kernel, kRingMinMeanSteps).
constexpr int kRingChains = 4;
constexpr int kRingWaves = 16;
constexpr int kRingBlock = kRingWaves * kWave;
constexpr uint32_t kRingSteps = 4;
constexpr uint32_t kRingSlots = 176;                                   // records of a tree's first four steps
constexpr uint32_t kRingBuffers = 4;
constexpr uint32_t kRingTreeBytes = kRingSlots * 16u;
constexpr uint32_t kRingBufBytes = kRingChains * kRingTreeBytes;        // 11 264
constexpr uint32_t kRingDmaLoads = kRingChains * kRingSlots / kWave;    // 11 wave instructions per group
static_assert(kRingChains * kRingSlots % kWave == 0, "a group is whole wave loads");
#ifndef OHX_EXP_RING_SPIN
#define OHX_EXP_RING_SPIN (1u << 21)      // scratch builds set it to 1 to see what a time-out does (tests/test_gpu_parity.py)
#endif
constexpr uint32_t kRingSpinLimit = OHX_EXP_RING_SPIN;
constexpr size_t kRingLdsBytes = (size_t)kRingWaves * 27 * kWave * sizeof(float) + (size_t)kRingBuffers * kRingBufBytes +
                                 (size_t)(kRingWaves + 4) * sizeof(uint32_t);
static_assert(kRingLdsBytes <= 160 * 1024, "LDS of a block");

__device__ __forceinline__ uint32_t ring_load(lds_u32_ptr p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ring_store(lds_u32_ptr p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS serves a wave's requests in order; this keeps the compiler from moving LDS accesses across and drains the
// wave's outstanding LDS operations (not its vector-memory ones: a fence would wait for the rows in flight too)
__device__ __forceinline__ void ring_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// One wave instruction of LDS-DMA: 64 x 16 bytes from the buffer `desc` describes, at byte `off` per lane
// (range-checked), to LDS at lds_addr + 16 * lane.  In assembly because hipcc puts `s_waitcnt vmcnt(0)` in front of
// every LDS read that follows a DMA it knows of; the caller waits (vmcnt) before anybody reads what it wrote.
__device__ __forceinline__ void dma_to_lds_b128(u32x4 desc, uint32_t off, uint32_t lds_addr) {
#if defined(__gfx950__)
  uint32_t keep;
  // (the LDS address is the same in every lane; said so explicitly: where hipcc's divergence analysis loses track of
  // that - the staging now sits behind "has the block aborted?" - it would hand M0 a vector register)
  const uint32_t m0_value = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(off), "s"(desc), "s"(m0_value)
      : "memory");
#else
  // 16-byte LDS-DMA is a gfx950 instruction; this library is built for nothing else (a sanitizer build of the HOST
  // side drags a default device target along: it must compile, it never runs)
  (void)desc; (void)off; (void)lds_addr;
  __builtin_trap();
#endif
}

// What a wave knows of its block's ring.
struct TopRing {
  lds_cptr ring;              // [kRingBuffers][kRingChains][kRingSlots] records
  uint32_t ring_addr;         // the same as an LDS byte address (for the DMA's M0)
  lds_u32_ptr progress, filled, claim, abort;
  u32x4 desc;                 // buffer descriptor over the forest's super-nodes
  __amdgpu_buffer_rsrc_t forest;
  const SuperTreeHead* heads;
  uint32_t t0, t1, ngroups;   // trees of this launch, in groups of kRingChains
  uint32_t total;             // groups every wave of the block goes through: rounds x ngroups
  uint32_t g;                 // groups this wave is done with
  bool gave_up;
};

// group `first_tree` .. + kRingChains - 1 (clamped to the last tree: a duplicate walk is discarded) -> buffer `buf`
__device__ __forceinline__ void ring_stage(const TopRing& rg, uint32_t first_tree, uint32_t buf, int lane) {
  uint32_t base[kRingChains];
#pragma unroll
  for (int c = 0; c < kRingChains; ++c) base[c] = rg.heads[(first_tree + c < rg.t1) ? first_tree + c : rg.t1 - 1].base;
#pragma unroll
  for (uint32_t q = 0; q < kRingDmaLoads; ++q) {
    const uint32_t L = q * kWave + (uint32_t)lane;     // record L of the group: tree L / 176, record L % 176
    uint32_t b = base[0], rec = L;
#pragma unroll
    for (int k = 1; k < kRingChains; ++k)
      if (L >= (uint32_t)k * kRingSlots) b = base[k], rec = L - (uint32_t)k * kRingSlots;
    dma_to_lds_b128(rg.desc, (b + rec) << 4, rg.ring_addr + buf * kRingBufBytes + q * 1024u);   // past the forest: zeros
  }
}

// Every thread of the block: sets the ring up behind the block's tiles, stages group 0, one barrier.
__device__ __forceinline__ void ring_begin(TopRing& rg, char* ring, const DeviceForest& fr, const SuperTreeHead* heads,
                                           uint32_t t0, uint32_t t1, uint32_t rounds) {
  const int lane = threadIdx.x & (kWave - 1);
  rg.ring = (lds_cptr)ring;
  rg.ring_addr = (uint32_t)(uintptr_t)(lds_void_ptr)ring;
  rg.progress = (lds_u32_ptr)(ring + (size_t)kRingBuffers * kRingBufBytes);
  rg.filled = rg.progress + kRingWaves;
  rg.claim = rg.progress + kRingWaves + 1;
  rg.abort = rg.progress + kRingWaves + 2;
  const uint64_t addr = reinterpret_cast<uint64_t>(fr.super);
  rg.desc.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)addr);
  rg.desc.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(addr >> 32) & 0xFFFFu));
  rg.desc.z = (uint32_t)__builtin_amdgcn_readfirstlane((int)fr.super_bytes);
  rg.desc.w = 0x00020000u;
  rg.forest = make_rsrc(fr.super, fr.super_bytes);
  rg.heads = heads;
  rg.t0 = t0;
  rg.t1 = t1;
  rg.ngroups = (t1 - t0 + kRingChains - 1) / kRingChains;
  rg.total = rounds * rg.ngroups;
  rg.g = 0;
  rg.gave_up = false;
  if (threadIdx.x < kWave) {            // wave 0
    ring_stage(rg, t0, 0u, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (threadIdx.x < (uint32_t)kRingWaves) ring_store(rg.progress + threadIdx.x, 0u);
  if (threadIdx.x == 0) {
    ring_store(rg.filled, 1u);
    ring_store(rg.claim, 1u);
    ring_store(rg.abort, 0u);
  }
  __syncthreads();
}

// one group of kRingChains trees for the wave's tile: steps 1 .. 4 from `tops` (LDS), the rest gathered.
// publish_to: this wave staged the next group before this walk; it says so from in here (see below).
template <bool HAS_MISSING>
__device__ __forceinline__ float ring_walk_group(const SuperTreeHead (&h)[kRingChains], uint32_t ntrees_here,
                                                 __amdgpu_buffer_rsrc_t forest, const float* __restrict__ tile,
                                                 lds_cptr tops, float acc, lds_u32_ptr publish_to,
                                                 uint32_t publish_value, bool ahead) {
  constexpr int CHAINS = kRingChains;
  const char* tile_b = reinterpret_cast<const char*>(tile);
  // Issue priority (s_setprio): a wave that walks goes before a wave that fills its tile, stores, stages or spins
  // (priority 0), and the further down its trees a wave is, the sooner it is issued - first step 1, the other LDS
  // steps 2, the deep steps' gathers (the walk's scarce resource) 3: the C360 step 27.4 -> 24.8 ms, the fused fields
  // kernel 30.4 -> 27.3 ms (profiles/r04_sweeps.txt; walk 1 / deep 2: 25.3, deep steps from the sixth on at 3: 25.0).
  // `ahead`: this wave was first at the group (it staged the next one) - it walks one level lower (1 / 1 / 2), so that
  // the waves behind it, which every wave of the block ends up waiting for, catch up: 24.7 -> 24.4 ms
  __builtin_amdgcn_s_setprio(1);
  u32x4 s[CHAINS];
  uint32_t rel[CHAINS], leafb[CHAINS];
  float xr[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) xr[c] = tile[(h[c].root_meta & 31u) * kWave];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) {
    bool l = xr[c] < h[c].root_thr;
    if (HAS_MISSING) l = go_left_or_default(xr[c], h[c].root_thr, (h[c].root_meta & 32u) != 0u);
    rel[c] = 4u + (((h[c].root_meta & 0x100u) && !l) ? 1u : 0u);
    leafb[c] = 0u;
  }
  uint32_t nsteps = h[0].steps;
#pragma unroll
  for (int c = 1; c < CHAINS; ++c) nsteps = h[c].steps > nsteps ? h[c].steps : nsteps;
  nsteps = nsteps < kRingSteps ? kRingSteps : nsteps;      // a shallow tree walks on through its fillers
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s[c] = *(lds_u32x4_ptr)(tops + (uint32_t)c * kRingTreeBytes + (rel[c] << 4));
#pragma unroll
  for (uint32_t step = 1; step < kRingSteps; ++step) {
    if (step == 2 && !ahead) __builtin_amdgcn_s_setprio(2);
    super_step<CHAINS, HAS_MISSING, false>(s, rel, leafb, tile_b);
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s[c] = *(lds_u32x4_ptr)(tops + (uint32_t)c * kRingTreeBytes + (rel[c] << 4));
  }
  // the stager: its DMA of the next group was issued before this walk and has had the LDS steps' time to land;
  // nothing else of this wave is in flight here but its next rows
  if (publish_to != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & (kWave - 1)) == 0) ring_store(publish_to, publish_value);
  }
  if (ahead) __builtin_amdgcn_s_setprio(2);      // the gathers of the deep steps first (see above)
  else __builtin_amdgcn_s_setprio(3);
  uint32_t base[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) base[c] = h[c].base;
  for (uint32_t step = kRingSteps; step < nsteps; ++step)
    super_step_by_chain<CHAINS, HAS_MISSING>(s, rel, leafb, tile_b, [&](int c, uint32_t record) -> u32x4 {
      return __builtin_amdgcn_raw_buffer_load_b128(forest, (int)((base[c] + record) << 4), 0, 0);
    });
  super_step<CHAINS, HAS_MISSING, true>(s, rel, leafb, tile_b);
  __builtin_amdgcn_s_setprio(1);      // the ring's bookkeeping between two groups is part of the walk (ring_walk_tile)
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
    if ((uint32_t)c < ntrees_here) acc += __uint_as_float(leafb[c]);
  return acc;
}

// has a wave of the block given up?  (wave-uniform: what the answer guards - staging through M0 - must stay scalar)
__device__ __forceinline__ bool ring_aborted(const TopRing& rg) {
  return __builtin_amdgcn_readfirstlane((int)ring_load(rg.abort)) != 0;
}

// This wave stops waiting, and tells the block.
__device__ __forceinline__ void ring_give_up(TopRing& rg, int lane) {
  rg.gave_up = true;
  if (lane == 0) ring_store(rg.abort, 1u);
}

// All trees of the launch for the wave's tile (live: the wave has one; a wave without still goes round: the others
// count on its progress).  `last_round`: nothing is staged behind the last group of the block's last round.
__device__ __forceinline__ float ring_walk_tile(TopRing& rg, float acc, const float* __restrict__ tile, bool live,
                                                bool wave_nan, int lane, int wave) {
  if (rg.gave_up) return acc;           // (the block's other waves see `abort` at their next wait)
  for (uint32_t p = 0; p < rg.ngroups; ++p, ++rg.g) {
    const uint32_t g = rg.g;
    const uint32_t t = rg.t0 + p * kRingChains;
    // ---- first at group g?  then group g + 1 is this wave's to stage.  The claim and the look at `filled` are one LDS
    //      round trip, made at the walk's priority (a wave that comes out of a walk at priority 0 waits for every
    //      walker on its SIMD before it may as much as ask): per-wave clocks showed 6 % of a wave's time in these two
    //      questions (profiles/r04_sweeps.txt); 24.7 -> 24.4 ms, with `ahead` 24.3
    uint32_t won = 0;
    uint32_t filled_now = 0;
    if (lane == 0) {
      // the claim and the look at `filled` in one LDS round trip
      uint32_t expected = g + 1u;
      const bool may = g + 1u < rg.total && !rg.gave_up;
      filled_now = ring_load(rg.filled);
      if (may) won = __hip_atomic_compare_exchange_strong(rg.claim, &expected, g + 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
    }
    won = (uint32_t)__builtin_amdgcn_readfirstlane((int)won);
    filled_now = (uint32_t)__builtin_amdgcn_readfirstlane((int)filled_now);
    if (won || filled_now < g + 1u) __builtin_amdgcn_s_setprio(0);      // about to wait or to stage: out of the walkers' way
    if (won) {
      // its buffer held group g + 1 - kRingBuffers: every wave must be done with that one
      const uint32_t need = g + 2u > kRingBuffers ? g + 2u - kRingBuffers : 0u;
      uint32_t spin = 0;
      while (!__all(lane >= kRingWaves || ring_load(rg.progress + (lane < kRingWaves ? lane : 0)) >= need)) {
        __builtin_amdgcn_s_sleep(2);
        if (++spin > kRingSpinLimit || ring_aborted(rg)) { ring_give_up(rg, lane); break; }
      }
      ring_order();
      // a wave that gave up must not write over a buffer the others may still be reading
      if (!rg.gave_up) ring_stage(rg, (p + 1 == rg.ngroups) ? rg.t0 : t + kRingChains, (g + 1u) % kRingBuffers, lane);
    }
    // ---- group g complete in the ring?
    if (filled_now < g + 1u && !rg.gave_up) {
      uint32_t spin = 0;
      while ((uint32_t)__builtin_amdgcn_readfirstlane((int)ring_load(rg.filled)) < g + 1u) {      // (uniform: see ring_aborted)
        __builtin_amdgcn_s_sleep(1);
        if (++spin > kRingSpinLimit || ring_aborted(rg)) { ring_give_up(rg, lane); break; }
      }
    }
    ring_order();       // the buffer is read after `filled` said so, not before
    // group g was complete before this wave walks it, so nobody else publishes meanwhile: g + 1 is next
    const lds_u32_ptr pub = won ? rg.filled : (lds_u32_ptr) nullptr;
    if (live && !rg.gave_up) {
      SuperTreeHead h[kRingChains];
#pragma unroll
      for (int c = 0; c < kRingChains; ++c) h[c] = rg.heads[(t + c < rg.t1) ? t + c : rg.t1 - 1];
      const uint32_t here = rg.t1 - t < (uint32_t)kRingChains ? rg.t1 - t : (uint32_t)kRingChains;
      const lds_cptr buf = rg.ring + (g % kRingBuffers) * kRingBufBytes;
      acc = wave_nan ? ring_walk_group<true>(h, here, rg.forest, tile, buf, acc, pub, g + 2u, won != 0u)
                     : ring_walk_group<false>(h, here, rg.forest, tile, buf, acc, pub, g + 2u, won != 0u);
    } else if (won) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // a stager without a tile
      if (lane == 0 && !rg.gave_up) ring_store(rg.filled, g + 2u);
    }
    ring_order();                                             // the walk's reads of the buffer are done
    if (lane == 0) ring_store(rg.progress + wave, g + 1u);
  }
  __builtin_amdgcn_s_setprio(0);
  return acc;
}

// AoS rows (27 columns, 27 features) in, margins out: predict_rows_tile_kernel<2, .., true, ..>'s row handling (rows
// fetched by the wave together, the next tile's in flight during the walk, rows with missing values left to the
// second launch) around the ring walk.  Every wave of a block goes round as often as the block's first wave.
__global__ __launch_bounds__(kRingBlock) __attribute__((amdgpu_waves_per_eu(4))) void predict_rows_ring_kernel(
    DeviceForest fr, PredictArgs a, const SuperTreeHead* __restrict__ heads, float* __restrict__ out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  float* tile = lds + (size_t)wave * 27 * kWave + lane;
  float* tile_base = tile - lane;
  const bool missing_is_nan = a.missing != a.missing;
  uint32_t block = blockIdx.x;
  if (a.xcd_remap && (gridDim.x & 7u) == 0u) block = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint64_t wave_id = (uint64_t)block * kRingWaves + wave;
  const uint64_t nwaves = (uint64_t)gridDim.x * kRingWaves;
  const uint64_t first_tile = a.tile_begin + (uint64_t)block * kRingWaves;
  if (first_tile >= a.tile_end) return;
  const uint32_t rounds = (uint32_t)((a.tile_end - first_tile + nwaves - 1) / nwaves);
  TopRing rg;
  ring_begin(rg, reinterpret_cast<char*>(lds + (size_t)kRingWaves * 27 * kWave), fr, heads, a.tree_begin, a.tree_end, rounds);
  uint64_t tile_id = a.tile_begin + wave_id;
  Row27 regs;
  bool valid = false, together = false;
  uint64_t row = 0;
  if (tile_id < a.tile_end) {
    row = launch_row(a, tile_id, lane, &valid);
    together = a.run_log >= 2u && !__any(valid && (row == 0 || row + 1 == a.nrow));
    if (together) load_pieces(regs, a, row, lane);
    else load_row27(regs, a.rows, valid ? row : 0);
  }
  for (uint32_t r = 0; r < rounds; ++r) {
    const bool live = tile_id < a.tile_end && __any(valid);
    const bool lane_nan = !live ? false
                          : together ? store_pieces(tile_base, regs, a, __ballot(valid), lane, missing_is_nan)
                                     : store_row27(tile, regs, valid, a.missing, missing_is_nan, a.flags);
    const uint64_t next = tile_id + nwaves;
    const uint64_t this_row = row;
    const bool this_valid = valid;
    if (next < a.tile_end) {
      row = launch_row(a, next, lane, &valid);
      together = a.run_log >= 2u && !__any(valid && (row == 0 || row + 1 == a.nrow));
      if (together) load_pieces(regs, a, row, lane);   // in flight during the walk
      else load_row27(regs, a.rows, valid ? row : 0);
    } else {
      valid = false;
    }
    bool wave_nan = __any(lane_nan);
    bool keep = this_valid;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // rows with missing values leave for the second launch, as in predict_rows_tile_kernel
    if (wave_nan && a.defer_count != nullptr && a.defer_cap == 0u) {
      if ((tile_id & 7u) == 0u) {
        const uint32_t n = 8u * (uint32_t)__popcll(__ballot(lane_nan));
        if (lane == 0) atomicAdd(a.defer_count, n);
      }
    } else if (wave_nan && a.defer_count != nullptr) {
      bool mine = false;
#pragma unroll
      for (int f = 0; f < 27; ++f) {
        const float x = tile[f * kWave];
        mine |= (x != x);
      }
      const bool leaves = mine && this_valid;
      const uint64_t who = __ballot(leaves);
      const uint32_t n = (uint32_t)__popcll(who);
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(a.defer_count, n);
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
      if (at + n <= a.defer_cap) {
        if (leaves) a.defer_list[at + (uint32_t)__popcll(who & ((1ull << lane) - 1ull))] = (uint32_t)this_row;
        keep = this_valid && !mine;
        wave_nan = false;
      }
    }
    const float acc = ring_walk_tile(rg, fr.base_score, tile, live, wave_nan, lane, wave);
    if (live && keep && !rg.gave_up) __builtin_nontemporal_store(acc, out + this_row);
    tile_id = next;
  }
  if (rg.gave_up && a.flags && lane == 0) atomicExch(a.flags + 2, a.train_id);
}

// The fused path (predict_fields_kernel's fill and store) around the ring walk.
__global__ __launch_bounds__(kRingBlock) __attribute__((amdgpu_waves_per_eu(4))) void predict_fields_ring_kernel(
    DeviceForest fr, FieldsArgs a, const SuperTreeHead* __restrict__ heads, float* __restrict__ out,
    float* __restrict__ margin_out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  float* tile = lds + (size_t)wave * 27 * kWave + lane;
  const bool missing_is_nan = a.missing != a.missing;
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  const uint64_t nrow = plane * (uint64_t)(a.k2 - a.k1 + 1);
  const uint64_t slab = level_floats(a, plane) * (uint64_t)(a.k1 - a.src_k0);
  const uint64_t slab_out = level_floats(a, plane) * (uint64_t)(a.k1 - a.out_k0);
  uint32_t block = blockIdx.x;
  if (a.xcd_remap && (gridDim.x & 7u) == 0u) block = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint64_t wave_id = (uint64_t)block * kRingWaves + wave;
  const uint64_t nwaves = (uint64_t)gridDim.x * kRingWaves;
  const uint64_t first_tile = a.tile_begin + (uint64_t)block * kRingWaves;
  if (first_tile >= a.tile_end) return;
  const uint32_t rounds = (uint32_t)((a.tile_end - first_tile + nwaves - 1) / nwaves);
  TopRing rg;
  ring_begin(rg, reinterpret_cast<char*>(lds + (size_t)kRingWaves * 27 * kWave), fr, heads, a.tree_begin, a.tree_end, rounds);
  uint64_t tile_id = a.tile_begin + wave_id;
  for (uint32_t r = 0; r < rounds; ++r, tile_id += nwaves) {
    bool valid = false;
    uint64_t m = 0;
    if (tile_id < a.tile_end) m = tile_row(a.shape, tile_id, lane, nrow, &valid);
    const bool live = __any(valid);
    const CellAt at = cell_at(a, plane, valid ? m : 0);
    const bool lane_nan = live && fill_tile_fields(a, tile, 27u, at.at3, at.at2, slab, valid, missing_is_nan);
    bool wave_nan = __any(lane_nan);
    bool keep = valid;
    if (wave_nan && a.defer_count != nullptr && a.defer_cap == 0u) {
      if ((tile_id & 7u) == 0u) {
        const uint32_t n = 8u * (uint32_t)__popcll(__ballot(lane_nan && valid));
        if (lane == 0) atomicAdd(a.defer_count, n);
      }
    } else if (wave_nan && a.defer_count != nullptr) {
      const bool leaves = lane_nan && valid;
      const uint64_t who = __ballot(leaves);
      const uint32_t n = (uint32_t)__popcll(who);
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(a.defer_count, n);
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
      if (at + n <= a.defer_cap) {
        if (leaves) a.defer_list[at + (uint32_t)__popcll(who & ((1ull << lane) - 1ull))] = (uint32_t)m;
        keep = valid && !lane_nan;
        wave_nan = false;
      }
    }
    const float acc = ring_walk_tile(rg, fr.base_score, tile, live, wave_nan, lane, wave);
    if (keep && !rg.gave_up) store_oh(a, out, margin_out, slab_out, m, at.at3, acc);
  }
  if (rg.gave_up && a.flags && lane == 0) atomicExch(a.flags + 2, a.train_id);
}

// The second launch of a small batch (PredictArgs::leaf_buf): one wave per tile, margin = ((base + leaf_0) + leaf_1) + ...
__global__ __launch_bounds__(kBlock) void combine_leaves_kernel(PredictArgs a, float base_score, float* __restrict__ out) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint64_t ntiles = a.tile_end - a.tile_begin, nwaves = (uint64_t)gridDim.x * kWavesPerBlock;
  const uint32_t ntree = a.tree_end - a.tree_begin;
  for (uint64_t w = (uint64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave; w < ntiles; w += nwaves) {
    bool valid;
    const uint64_t row = launch_row(a, a.tile_begin + w, lane, &valid);
    if (!__any(valid)) continue;
    const float* leaves = a.leaf_buf + (size_t)w * ntree * kWave + lane;
    float acc = base_score;
#pragma unroll 4
    for (uint32_t t = 0; t < ntree; ++t) acc += leaves[(size_t)t * kWave];
    if (valid) out[row] = acc;
  }
}

// Any feature count, no LDS: every lane reads its own row from global memory.
template <bool PRED_LEAF>
__global__ __launch_bounds__(kBlock) void predict_rows_direct_kernel(DeviceForest fr, PredictArgs a) {
  const uint4* nodes = reinterpret_cast<const uint4*>(fr.wide);
  const bool missing_is_nan = a.missing != a.missing;
  const bool missing_is_inf = is_inf(a.missing);
  const uint32_t ntree = a.tree_end - a.tree_begin;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; row < a.nrow; row += stride) {
    const float* x = a.rows + row * (uint64_t)a.ncol;
    float acc = fr.base_score;
    bool any_inf = false;
    for (uint32_t t = a.tree_begin; t < a.tree_end; ++t) {
      uint4 nd = nodes[fr.roots[t]];
      while (nd.y != 0u) {
        const uint32_t f = nd.z & 0x7FFFFFFFu;
        bool miss = true;
        bool lt = false;
        if (f < a.ncol) {
          const float v = x[f];
          any_inf |= is_inf(v);
          miss = (v != v) || (!missing_is_nan && v == a.missing);
          lt = v < __uint_as_float(nd.x);
        }
        const bool go_left = miss ? (nd.z >> 31) != 0u : lt;
        nd = nodes[nd.y + (go_left ? 0u : 1u)];
      }
      if (PRED_LEAF) a.out[row * (uint64_t)ntree + (t - a.tree_begin)] = (float)(int32_t)nd.w;
      else acc += __uint_as_float(nd.x);
    }
    if (!PRED_LEAF) a.out[row] = acc;
    if (any_inf && !missing_is_inf && a.flags) atomicOr(a.flags, kFlagInfInput);
  }
}

// The fused path: reads the MAPL fields in place (lane = consecutive i, coalesced),
// applies PL/100 (OH_GridCompMod.F90:314), walks, writes 10**pred * OHscale
// (OH_GridCompMod.F90:369,1569) into OH_ML(i,j,k1..k2).
template <int FMT, int CHAINS, bool TOPS>
__global__ __launch_bounds__(kBlock) void predict_fields_kernel(DeviceForest fr, FieldsArgs a, const SuperTreeHead* __restrict__ heads,
                                                                float* __restrict__ out,
                                                                float* __restrict__ margin_out) {
  extern __shared__ float lds[];
  if (a.only_if_train != 0u) {      // the launch behind a ring train: nothing to do unless a block of that train gave up
    if (__hip_atomic_load(a.flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.only_if_train) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.flags + 1, 1u);
  }
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  float* tile = lds + (size_t)wave * fr.num_feature * kWave + lane;
  char* first = reinterpret_cast<char*>(lds + (size_t)kWavesPerBlock * fr.num_feature * kWave);
  const uint32_t nfirst = fill_first_steps<FMT>(fr, heads, first);
  const bool missing_is_nan = a.missing != a.missing;
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  const uint64_t nrow = plane * (uint64_t)(a.k2 - a.k1 + 1);
  const uint64_t slab = level_floats(a, plane) * (uint64_t)(a.k1 - a.src_k0);
  const uint64_t slab_out = level_floats(a, plane) * (uint64_t)(a.k1 - a.out_k0);
  // as in the rows kernel: blocks b and b + 8 share an XCD, give every XCD one contiguous run of tiles
  uint32_t block = blockIdx.x;
  if (a.xcd_remap && (gridDim.x & 7u) == 0u) block = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint64_t wave_id = (uint64_t)block * kWavesPerBlock + wave;
  const uint64_t nwaves = (uint64_t)gridDim.x * kWavesPerBlock;
  if constexpr (FMT == 2) {
    if (a.leaf_buf != nullptr) {
      // a small slab (a rank's block): work item = (run of trees, tile) as in predict_rows_tile_kernel; every item
      // fills its tile itself and writes its trees' leaves, combine_leaves_fields_kernel sums and stores them
      const uint32_t ntree = a.tree_end - a.tree_begin;
      const uint32_t per = ((ntree + a.tree_split - 1) / a.tree_split + (CHAINS - 1)) / CHAINS * CHAINS;
      const uint64_t ntiles = a.tile_end - a.tile_begin, items = ntiles * a.tree_split;
      for (uint64_t item = wave_id; item < items; item += nwaves) {
        const uint64_t tile_id = a.tile_begin + item % ntiles;
        const uint32_t t0 = a.tree_begin + (uint32_t)(item / ntiles) * per;
        const uint32_t t1 = t0 + per < a.tree_end ? t0 + per : a.tree_end;
        bool valid;
        const uint64_t m = tile_row(a.shape, tile_id, lane, nrow, &valid);
        if (!__any(valid) || t0 >= t1) continue;
        const CellAt at = cell_at(a, plane, valid ? m : 0);
        const bool lane_nan = fill_tile_fields_burst(a, tile, fr.num_feature, at.at3, at.at2, slab, valid, missing_is_nan);
        float* leaves = a.leaf_buf + ((size_t)(tile_id - a.tile_begin) * ntree - a.tree_begin) * kWave + lane;
        (void)walk_tile<FMT, CHAINS, TOPS>(fr, heads, t0, t1, tile, __any(lane_nan), first, nfirst, leaves);
      }
      return;
    }
  }
  // the second launch of a deferred-rows call (PredictArgs::defer_list): the rows of the list, as many as the first filled
  uint64_t slots = 0, tile_end = a.tile_end;
  if (a.perm != nullptr) {
    const uint32_t filled = *a.perm_count;
    slots = filled < a.perm_slots ? filled : a.perm_slots;
    const uint64_t need = (slots + kWave - 1) / kWave;
    tile_end = need < tile_end ? need : tile_end;
  }
  for (uint64_t tile_id = a.tile_begin + wave_id; tile_id < tile_end; tile_id += nwaves) {
    bool valid;
    uint64_t m;
    if (a.perm != nullptr) {
      const uint64_t slot = tile_id * kWave + lane;
      valid = slot < slots;
      const uint32_t listed = valid ? a.perm[slot] : 0u;
      valid = valid && listed != 0xFFFFFFFFu;
      m = valid ? listed : 0u;
    } else {
      m = tile_row(a.shape, tile_id, lane, nrow, &valid);
    }
    if (!__any(valid)) continue;                       // a brick without a row of the slab, a tile of empty slots
    const CellAt at = cell_at(a, plane, valid ? m : 0);
    const bool lane_nan = fill_tile_fields(a, tile, fr.num_feature, at.at3, at.at2, slab, valid, missing_is_nan);
    bool wave_nan = __any(lane_nan);
    bool keep = valid;
    // rows with missing values leave for the second launch, as in the rows kernel (every lane filled its own row here)
    if (wave_nan && a.defer_count != nullptr && a.defer_cap == 0u) {
      if ((tile_id & 7u) == 0u) {
        const uint32_t n = 8u * (uint32_t)__popcll(__ballot(lane_nan && valid));
        if (lane == 0) atomicAdd(a.defer_count, n);
      }
    } else if (wave_nan && a.defer_count != nullptr) {
      const bool leaves = lane_nan && valid;
      const uint64_t who = __ballot(leaves);
      const uint32_t n = (uint32_t)__popcll(who);
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(a.defer_count, n);
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
      if (at + n <= a.defer_cap) {
        if (leaves) a.defer_list[at + (uint32_t)__popcll(who & ((1ull << lane) - 1ull))] = (uint32_t)m;
        keep = valid && !lane_nan;
        wave_nan = false;
      }
    }
    const float acc = walk_tile<FMT, CHAINS, TOPS>(fr, heads, a.tree_begin, a.tree_end, tile, wave_nan, first, nfirst);
    if (keep) store_oh(a, out, margin_out, slab_out, m, at.at3, acc);
  }
}

// The second launch of a small slab (FieldsArgs::leaf_buf): margin = ((base + leaf_0) + leaf_1) + ..., then 10** and OHscale
__global__ __launch_bounds__(kBlock) void combine_leaves_fields_kernel(FieldsArgs a, float base_score, float* __restrict__ out,
                                                                        float* __restrict__ margin_out) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  const uint64_t nrow = plane * (uint64_t)(a.k2 - a.k1 + 1);
  const uint64_t slab_out = level_floats(a, plane) * (uint64_t)(a.k1 - a.out_k0);
  const uint64_t ntiles = a.tile_end - a.tile_begin, nwaves = (uint64_t)gridDim.x * kWavesPerBlock;
  const uint32_t ntree = a.tree_end - a.tree_begin;
  for (uint64_t w = (uint64_t)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave; w < ntiles; w += nwaves) {
    bool valid;
    const uint64_t m = tile_row(a.shape, a.tile_begin + w, lane, nrow, &valid);
    if (!__any(valid)) continue;
    const float* leaves = a.leaf_buf + (size_t)w * ntree * kWave + lane;
    float acc = base_score;
#pragma unroll 4
    for (uint32_t t = 0; t < ntree; ++t) acc += leaves[(size_t)t * kWave];
    if (valid) store_oh(a, out, margin_out, slab_out, m, cell_at(a, plane, m).at3, acc);
  }
}

__global__ __launch_bounds__(kBlock) void scan_dense_kernel(const float* __restrict__ data, uint64_t count,
                                                            float missing, uint32_t* flags) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  bool any_inf = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride)
    any_inf |= is_inf(data[i]);
  if (any_inf && !is_inf(missing)) atomicOr(flags, kFlagInfInput);
}

// Is a column of the row-major matrix periodic in the rows with period nrow / k?  blockIdx.x picks k
// (kmax - blockIdx.x, so ascending periods; a k that does not divide nrow is marked 2 = "no candidate"),
// blockIdx.y the column; 4 x blockDim sampled row pairs each, bit for bit.  This is how a matrix gathered
// level by level from a grid shows its level size without being told: a 2-D field (LAT, feature 0 of
// the OH gather, OH_GridCompMod.F90:313) repeats exactly from one level to the next.
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

__global__ __launch_bounds__(kBlock) void detect_period_kernel(const float* __restrict__ data, uint64_t nrow,
                                                               uint32_t ncol, PeriodColumns cols, uint32_t kmax,
                                                               uint32_t* __restrict__ verdict) {
  const uint32_t k = kmax - blockIdx.x;
  uint32_t* slot = verdict + (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const uint32_t col = cols.col[blockIdx.y];
  if (nrow % k != 0 || col >= ncol) {
    if (threadIdx.x == 0) *slot = 2u;
    return;
  }
  const uint64_t period = nrow / k;
  const uint64_t span = nrow - period;
  const uint32_t* bits = reinterpret_cast<const uint32_t*>(data);
  bool bad = false;
  for (uint32_t s = threadIdx.x; s < 4u * blockDim.x; s += blockDim.x) {
    const uint64_t m = ((uint64_t)mix32(mix32(0x5eedu ^ s) ^ blockIdx.x) * 2654435761ull + s) % span;
    bad |= bits[m * ncol + col] != bits[(m + period) * ncol + col];
  }
  // the first and the last pair as well
  if (threadIdx.x == 0) bad |= bits[col] != bits[period * ncol + col] ||
                               bits[(span - 1) * ncol + col] != bits[(nrow - 1) * ncol + col];
  if (bad) atomicOr(slot, 1u);
}

// ------------------------------------------------------------------ rows nobody can describe: clustering pass
//
// What a tree walk costs on gfx950 is set by how many different node lines the lanes of a wave - and above
// all the four lanes of a quad - ask for.  Rows that come in grid order are tiled into bricks of
// neighbouring gridcells; rows in no particular order (a caller that shuffled, filtered or concatenated its
// gather: the DMatrix contract allows any order, OH_GridCompMod.F90:275-345 is just one caller) have no
// neighbours to offer, and 64 arbitrary rows per wave run at a third of the speed.  This pass finds them
// neighbours: every row gets a key made of the decisions it takes at the top of the first few trees of the
// booster itself, (key, row number) pairs are sorted by key (a device radix sort, sort_pairs.hip), and the walk
// then takes 64 rows that are neighbours in key order per wave through the sorted row numbers.  Predictions cannot change - rows are independent and
// every row still walks every tree in order; only which rows share a wave does.
//
// All of it is plain integer work next to the walk: one more read of the rows and a sort of eight bytes per row.
// One lane = one row.  The wave's rows go through the same LDS tile as in the walk (coalesced row loads, missing ->
// NaN), the key walk reads its features from there.  Key = for each of `ntrees` trees the root decision (trees
// whose super-nodes start below the root) and two decisions per super-node step, most significant first; a row
// that reaches a leaf early keeps walking on fixed decisions.  zorder: the trees' decisions interleaved step by
// step instead of tree after tree.  Also counts how many rows agree with the row before them on the first three
// decisions of the first tree - rows in grid order mostly do (74 % on the C360 batch), shuffled rows mostly do
// not (21 %) - and how often each of the eight outcomes occurs, so that the host can tell "ordered" from "a
// tree top that sends nearly every row the same way".
constexpr int kKeyTileStride = kWave + 1;

// NT = number of trees in the key (compile time: their walks are independent chains and must be in flight together)
template <int NT>
__global__ __launch_bounds__(kBlock) void cluster_keys_kernel(DeviceForest fr, ClusterArgs a) {
  extern __shared__ float lds[];
  const u32x4* __restrict__ nodes = reinterpret_cast<const u32x4*>(fr.super);
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  float* wave_tile = lds + (size_t)wave * fr.num_feature * kKeyTileStride;
  float* tile = wave_tile + lane;
  const bool missing_is_nan = a.missing != a.missing;
  const uint64_t ntiles = (a.nrow + kWave - 1) / kWave;
  const uint64_t nwaves = (uint64_t)gridDim.x * kWavesPerBlock;
  SuperTreeHead h[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) h[t] = fr.super_heads[t];
  uint32_t agree = 0, seen = 0;
  for (uint64_t tile_id = (uint64_t)blockIdx.x * kWavesPerBlock + wave; tile_id < ntiles; tile_id += nwaves) {
    const uint64_t row = tile_id * kWave + lane;
    const bool valid = row < a.nrow;
    // The wave's 64 rows are one contiguous piece of the matrix: stream it in coalesced (lane l takes elements l,
    // l + 64, ...) and drop every element at [feature][row] of the tile.  Row-per-lane loads (as the walk kernels
    // do, hidden behind the walk there) touch every line seven times and run this pass at 0.8 TB/s.  The tile rows
    // are 65 floats apart so that consecutive elements - consecutive features of one row - fall in different banks.
    {
      const uint64_t first_row = tile_id * kWave;
      const uint32_t rows_here = (uint32_t)((a.nrow - first_row) < (uint64_t)kWave ? (a.nrow - first_row) : (uint64_t)kWave);
      const uint32_t count = rows_here * a.ncol;
      const float* base = a.rows + first_row * (uint64_t)a.ncol;
      const uint32_t dr = (uint32_t)kWave / a.ncol, df = (uint32_t)kWave % a.ncol;
      uint32_t r = (uint32_t)lane / a.ncol, f = (uint32_t)lane % a.ncol;
      for (uint32_t e = (uint32_t)lane; e < count; e += kWave) {
        float x = __builtin_nontemporal_load(base + e);
        if (!missing_is_nan && x == a.missing) x = __builtin_nanf("");
        wave_tile[f * kKeyTileStride + r] = x;
        r += dr;
        f += df;
        if (f >= a.ncol) {
          f -= a.ncol;
          r += 1;
        }
      }
      // features the matrix does not have are missing; rows past the end are never looked at
      for (uint32_t ff = a.ncol; ff < fr.num_feature; ++ff) tile[ff * kKeyTileStride] = __builtin_nanf("");
    }
    // lanes read what other lanes of the same wave wrote: LDS serves a wave's instructions in order, the fence
    // and the wave barrier keep the compiler from moving the reads above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t key = 0, lead = 0;
    uint32_t rel[NT], part[NT];
    u32x4 nd[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      rel[t] = 4u;
      part[t] = 0;
      if (h[t].root_meta & 0x100u) {        // phase 1: the root is evaluated from the head
        const float x = tile[(h[t].root_meta & 31u) * kKeyTileStride];
        const bool l = (x != x) ? (h[t].root_meta & 32u) != 0u : x < h[t].root_thr;
        part[t] = l ? 0u : 1u;
        rel[t] = 4u + (l ? 0u : 1u);
      }
      nd[t] = nodes[h[t].base + rel[t]];
    }
    for (uint32_t s = 0; s < a.nsteps; ++s) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        uint32_t bits = 0;
        if (s < h[t].steps) {
          const uint32_t w = nd[t].w;
          const uint32_t f0 = (w >> 8) & 31u;
          if (f0 != kSuperLeaf) {
            const float x0 = tile[f0 * kKeyTileStride];
            const bool l0 = (x0 != x0) ? (w & 32u) != 0u : x0 < __uint_as_float(nd[t].x);
            const uint32_t f1 = (w >> (l0 ? 0u : 13u)) & 31u;
            bool l1 = true;
            if (f1 != kSuperLeaf) {
              const float x1 = tile[f1 * kKeyTileStride];
              l1 = (x1 != x1) ? (w & (l0 ? 64u : 128u)) != 0u : x1 < __uint_as_float(l0 ? nd[t].y : nd[t].z);
            }
            bits = (l0 ? 0u : 2u) | (l1 ? 0u : 1u);
            rel[t] = ((w >> 18) << 2) + bits;
          } else {
            rel[t] = (w >> 18) << 2;        // fillers lead to fillers
          }
        }
        part[t] = (part[t] << 2) | bits;
        if (a.zorder) key = (key << 2) | bits;
      }
      if (s + 1 < a.nsteps) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (s + 1 < h[t].steps) nd[t] = nodes[h[t].base + rel[t]];
      }
      if (s == 0) lead = part[0];            // root and first step of the first tree: 3 decisions
    }
    if (a.zorder) {
      uint32_t roots = 0;                    // the root decisions in front of the interleaved steps
#pragma unroll
      for (int t = 0; t < NT; ++t) roots = (roots << 1) | (part[t] >> (2u * a.nsteps));
      key |= roots << (2u * a.nsteps * NT);
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) key = (key << (1u + 2u * a.nsteps)) | part[t];
    }
    if (valid) {
      a.keys[row] = key;
      a.vals[row] = (uint32_t)row;
    }
    // neighbours in row order: lane l against lane l - 1 (the first lane of a wave sits out)
    const uint32_t prev = __shfl_up(lead, 1);
    const bool same = valid && lane != 0 && prev == lead;
    agree += __popcll(__ballot(same));
    // how often each of the eight outcomes occurs at all: what two rows picked at random would agree on
#pragma unroll
    for (uint32_t v = 0; v < 8; ++v) {
      const uint32_t n = (uint32_t)__popcll(__ballot(valid && (lead & 7u) == v));
      if (lane == (int)v) seen += n;
    }
    __builtin_amdgcn_wave_barrier();         // the next tile overwrites what this one's lanes were reading
  }
  if (lane == 0 && agree) atomicAdd(a.agree, agree);
  if (lane < 8 && seen) atomicAdd(a.agree + 1 + lane, seen);
}

// ------------------------------------------------------------------ OH Run1: before and after the predict

// Feature engineering in two kernels.  (1) pointwise, one thread per gridcell: PL_BST, the
// layer AOD, stratO3 - plain coalesced streams.  (2) column sums, one wave per 64 consecutive
// columns and per layer field: the column goes through LDS [k][lane] once, SUM(x(1:k)) is the
// running sum of that pass, and every SUM(x(k:km)) is then accumulated FROM ZERO in ascending
// level order (O(km^2) LDS reads), which is how the reference's SUM intrinsic rounds
// (OH_GridCompMod.F90:1468-1478); a suffix scan from the bottom would round differently.
__global__ __launch_bounds__(kBlock) void feature_pointwise_kernel(PrepArgs a, float* __restrict__ aod) {
#pragma clang fp contract(off)
  // blockIdx.y = level; the block's threads stride over the columns col0 .. col0 + ncols - 1
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  const uint64_t ncols = a.ncols ? a.ncols : plane;
  const uint64_t level = plane * (uint64_t)blockIdx.y;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += stride) {
    const uint64_t col = a.col0 + c, m = level + col;
    a.pl_bst[m] = (a.ple_bst[m] + a.ple_bst[m + plane]) * 0.5f;               // :1488
    const float thick = a.zle_bst[m] - a.zle_bst[m + plane];                  // ZLE(k-1) - ZLE(k), :1451
    float sc = a.sca[0][m] + a.sca[1][m];                                      // BC+OC+BR+DU+SU+SS+NI, :1456-1457
    sc = sc + a.sca[2][m];
    sc = sc + a.sca[3][m];
    sc = sc + a.sca[4][m];
    sc = sc + a.sca[5][m];
    sc = sc + a.sca[6][m];
    aod[m] = thick * sc;
    if (blockIdx.y == 0) a.strato3[col] = a.gmito3[col] - a.gmitto3[col];      // :1446
  }
}

// blockIdx.y selects the field: 0 TAUCLW, 1 TAUCLI, 2 aod
__global__ __launch_bounds__(kBlock) void feature_column_sums_kernel(PrepArgs a, const float* __restrict__ aod) {
#pragma clang fp contract(off)
  extern __shared__ float lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  const uint64_t ncols = a.ncols ? a.ncols : plane;
  const uint64_t col = ((uint64_t)blockIdx.x * kWavesPerBlock + wave) * kWave + lane;
  const bool valid = col < ncols;
  const uint64_t c = a.col0 + (valid ? col : ncols - 1);
  const float* src = blockIdx.y == 0 ? a.tauclw : (blockIdx.y == 1 ? a.taucli : aod);
  float* up = blockIdx.y == 0 ? a.tauclwup : (blockIdx.y == 1 ? a.taucliup : a.aodup);
  float* dn = blockIdx.y == 0 ? a.tauclwdn : (blockIdx.y == 1 ? a.tauclidn : a.aoddn);
  float* col_lds = lds + (size_t)wave * a.km * kWave + lane;
  // the column comes in with independent loads (eight in flight per lane), the sums run from LDS
#pragma unroll 8
  for (int k = 0; k < a.km; ++k) col_lds[(size_t)k * kWave] = __builtin_nontemporal_load(src + c + plane * (uint64_t)k);
  float run = 0.0f;
  for (int k = 0; k < a.km; ++k) {
    run = run + col_lds[(size_t)k * kWave];                                    // SUM(x(1:k))
    if (valid) __builtin_nontemporal_store(run, up + c + plane * (uint64_t)k);
  }
  for (int k = 0; k < a.km; ++k) {                                             // SUM(x(k:km)), from zero
    float s = 0.0f;
    for (int kk = k; kk < a.km; ++kk) s = s + col_lds[(size_t)kk * kWave];
    if (valid) __builtin_nontemporal_store(s, dn + c + plane * (uint64_t)k);
  }
}

// The same with the column in registers (KM known at compile time): the O(km^2) additions of the SUM(x(k:km)) sums
// are then VALU work on registers (2 628 adds per column at 72 levels) instead of as many LDS reads, and the column's
// loads are all in flight together - 0.79 ms -> see profiles/r04_sweeps.txt.  Same order of additions, same bits.
template <int KM>
__global__ __launch_bounds__(kBlock) void feature_column_sums_reg_kernel(PrepArgs a, const float* __restrict__ aod) {
#pragma clang fp contract(off)
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  const uint64_t ncols = a.ncols ? a.ncols : plane;
  uint64_t col = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (col >= ncols) return;
  col += a.col0;
  const float* src = (blockIdx.y == 0 ? a.tauclw : (blockIdx.y == 1 ? a.taucli : aod)) + col;
  float* up = (blockIdx.y == 0 ? a.tauclwup : (blockIdx.y == 1 ? a.taucliup : a.aodup)) + col;
  float* dn = (blockIdx.y == 0 ? a.tauclwdn : (blockIdx.y == 1 ? a.tauclidn : a.aoddn)) + col;
  float x[KM];
#pragma unroll
  for (int k = 0; k < KM; ++k) x[k] = __builtin_nontemporal_load(src + plane * (uint64_t)k);
  float run = 0.0f;
#pragma unroll
  for (int k = 0; k < KM; ++k) {
    run = run + x[k];                                                          // SUM(x(1:k))
    __builtin_nontemporal_store(run, up + plane * (uint64_t)k);
  }
#pragma unroll
  for (int k = 0; k < KM; ++k) {                                               // SUM(x(k:km)), from zero
    float s = 0.0f;
#pragma unroll
    for (int kk = k; kk < KM; ++kk) s = s + x[kk];
    __builtin_nontemporal_store(s, dn + plane * (uint64_t)k);
  }
}

// The same for a SMALL plane (a GEOS rank's block: 48 x 24 = 1 152 columns): a WAVE per column and field, a lane per
// level (two with more than 64 levels).  With a thread per column a rank's block is 18 waves on the whole chip, each
// with 72 strided loads and 2 628 dependent additions in front of it: 87 us of a 370 us tick (profiles/r05_sweeps.txt).
// Here every lane runs its own two chains over the column in LDS - SUM(x(1:k)) and SUM(x(k:km)), each from zero in
// ascending level order, the additions the reference's SUM makes (:1468-1478) - 72 steps for all levels at once.
constexpr int kColWaveMaxKm = 128;
__global__ __launch_bounds__(kBlock) void feature_column_sums_wave_kernel(PrepArgs a, const float* __restrict__ aod) {
#pragma clang fp contract(off)
  __shared__ float xs[kWavesPerBlock][kColWaveMaxKm];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  const uint64_t ncols = a.ncols ? a.ncols : plane;
  const uint64_t item = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
  if (item >= ncols) return;
  const uint64_t col = a.col0 + item;
  const float* src = (blockIdx.y == 0 ? a.tauclw : (blockIdx.y == 1 ? a.taucli : aod)) + col;
  float* up = (blockIdx.y == 0 ? a.tauclwup : (blockIdx.y == 1 ? a.taucliup : a.aodup)) + col;
  float* dn = (blockIdx.y == 0 ? a.tauclwdn : (blockIdx.y == 1 ? a.tauclidn : a.aoddn)) + col;
  const int km = a.km;
  const int ka = lane, kb = lane + kWave;                 // this lane's one or two levels
  if (ka < km) xs[wave][ka] = src[plane * (uint64_t)ka];
  if (kb < km) xs[wave][kb] = src[plane * (uint64_t)kb];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float up_a = 0.0f, dn_a = 0.0f, up_b = 0.0f, dn_b = 0.0f;
#pragma unroll 8
  for (int kk = 0; kk < km; ++kk) {
    const float v = xs[wave][kk];                         // every lane the same word: a broadcast
    if (kk <= ka) up_a = up_a + v;
    if (kk >= ka) dn_a = dn_a + v;
    if (kk <= kb) up_b = up_b + v;
    if (kk >= kb) dn_b = dn_b + v;
  }
  if (ka < km) {
    up[plane * (uint64_t)ka] = up_a;
    dn[plane * (uint64_t)ka] = dn_a;
  }
  if (kb < km) {
    up[plane * (uint64_t)kb] = up_b;
    dn[plane * (uint64_t)kb] = dn_b;
  }
}

// ksubcount = max over columns of COUNT(pl > tropp | tropp_min) (:275-298)
__global__ __launch_bounds__(kBlock) void k_slab_kernel(SlabArgs a) {
#pragma clang fp contract(off)
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  int best = 0, bad = 0;
  for (uint64_t col = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; col < plane; col += stride) {
    const float tp = a.tropp[col];
    const float lim = a.dynamic_k_range ? tp : a.tropp_min;
    bad += (tp <= a.tropp_min) ? 1 : 0;
    // a column's km + 1 edges, each read once, eight in flight (one after the other a rank-sized block - a handful of
    // waves on the whole chip - spent 16 us here on 146 dependent-looking loads per column)
    int cnt = 0;
    float upper = a.ple_mod[col];
    for (int k0 = 0; k0 < a.km; k0 += 8) {
      float e[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) {      // (past the column's end: its last edge again, loaded and not counted - no branches)
        const int edge = k0 + d + 1 < a.km ? k0 + d + 1 : a.km;
        e[d] = a.ple_mod[col + plane * (uint64_t)edge];
      }
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const float pl = (upper + e[d]) * 0.5f;
        cnt += (k0 + d < a.km && pl > lim) ? 1 : 0;
        upper = e[d];
      }
    }
    best = cnt > best ? cnt : best;
  }
  atomicMax(&a.result[0], best);
  if (bad) atomicAdd(&a.result[1], bad);
}

// tropopause mask and mol/mol -> molec/cm3 (:1247-1257, 1579-1595)
__global__ __launch_bounds__(kBlock) void post_process_kernel(PostArgs a) {
#pragma clang fp contract(off)
  // blockIdx.y = level; the block's threads stride over the columns col0 .. col0 + ncols - 1
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  const uint64_t ncols = a.ncols ? a.ncols : plane;
  const uint64_t level = plane * (uint64_t)blockIdx.y;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += stride) {
    const uint64_t col = a.col0 + c, m = level + col;
    const float pl = (a.ple_mod[m] + a.ple_mod[m + plane]) * 0.5f;
    const float q = a.q_mod[m];
    const float tv = a.t_mod[m] * (1.0f + q / a.epsilon) / (1.0f + q);
    const float ndwet = (a.avogad * pl) / (a.runiv * tv);
    const float ohv = (pl > a.tropp[col]) ? a.oh_ml[m] : a.default_oh[m];
    a.oh[m] = (ohv * ndwet) * 1.0e-6f;
    if (a.ndwet) a.ndwet[m] = ndwet;
  }
}

// latarr = LATS*MAPL_RADIANS_TO_DEGREES (:1444) and computeSolarZenithAngle_LocalNoon (:401-466), the
// reference's float32 expressions in its order of evaluation.  sin/asin/cos/acos are the device
// library's: results agree with a host libm to a few ulp, and acos turns one ulp of cosz near +-1
// into hundredths of a degree - which is why the library never computes SZA behind the caller's back.
__global__ __launch_bounds__(kBlock) void solar_geometry_kernel(SolarArgs a) {
#pragma clang fp contract(off)
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  const float sindec = 0.3978f * sinf(0.9863f * ((float)a.jday - 80.0f) * a.deg2rad);
  const float soldek = asinf(sindec);
  const float cosdec = cosf(soldek);
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; m < plane; m += stride) {
    const float lat = a.lats[m];
    if (a.lat_deg) a.lat_deg[m] = lat * a.rad2deg;
    if (!a.sza_noon) continue;
    const float lon = a.lons[m];
    const float sinlat = sinf(lat);
    const float sollat = asinf(sinlat);
    const float coslat = cosf(sollat);
    float mylon = lon * a.rad2deg;
    if (mylon > 180.0f) mylon = mylon - 360.0f;
    if (mylon < -180.0f) mylon = mylon + 360.0f;
    const float tau = 12.0f + (mylon / -180.0f) * 12.0f;
    const float loct = ((tau * 15.0f) - 180.0f) * a.deg2rad + lon;
    float cosz = cosdec * coslat * cosf(loct) + sindec * sinlat;
    cosz = fminf(1.0f, cosz);
    cosz = fmaxf(-1.0f, cosz);
    a.sza_noon[m] = acosf(cosz) * a.rad2deg;
  }
}

int grid_for(uint64_t work_items, int num_cus, int blocks_per_cu) {
  uint64_t blocks = (work_items + kBlock - 1) / kBlock;
  uint64_t cap = (uint64_t)num_cus * (uint64_t)blocks_per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// LDS of a block: the feature tiles of its waves, then (super-nodes) the first-step table.  The super-node
// walk reads "feature row 31" for a leaf and ignores the value; that row must be memory the block owns.
// Row 31 of a wave's tile lies in the next wave's tile, and for the last wave in the first-step table, as
// long as the allocation reaches (3 * num_feature + 32) rows - which tiles + table already do for the OH
// booster (27 features: 31 744 B against 28 928 B, so five blocks still share a CU); boosters with few
// features get the padding.
size_t tile_lds_bytes(uint32_t num_feature, bool super_format) {
  const size_t tiles = (size_t)kWavesPerBlock * num_feature * kWave * sizeof(float);
  if (!super_format) return tiles;
  const size_t reach = ((size_t)(kWavesPerBlock - 1) * num_feature + 32) * kWave * sizeof(float);
  return tiles + kFirstBytes > reach ? tiles + kFirstBytes : reach;
}

template <class K>
int tile_grid(K kernel, size_t lds_bytes, uint64_t ntiles, int num_cus) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds_bytes) != hipSuccess || per_cu < 1)
    per_cu = 1;
  uint64_t blocks = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
  const uint64_t cap = (uint64_t)num_cus * (uint64_t)per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

template <class K>
hipError_t ensure_lds(K kernel, size_t lds_bytes) {
  if (lds_bytes > 64 * 1024)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)lds_bytes);
  return hipSuccess;
}

// Which stream the k-th launch of a train goes to (TrainStreams, kernels.hpp).  Inside a group the launches
// alternate between the caller's stream and the side stream; at a group's end both streams meet.  Order between
// launches never matters for the results (disjoint rows); it only shapes how the chip is filled.
struct TrainCursor {
  hipStream_t main;
  const TrainStreams* tr;
  int group;
  int k = 0;
  bool side_open = false;   // the side stream carries launches nobody has waited for yet
  TrainCursor(hipStream_t m, const LaunchTuning& tune)
      : main(m), tr((tune.overlap_group >= 2 && tune.train && tune.train->side) ? tune.train : nullptr),
        group(tune.overlap_group) {}
  hipError_t meet() {
    if (!side_open) return hipSuccess;
    side_open = false;
    hipError_t e = hipEventRecord(tr->join, tr->side);
    if (e != hipSuccess) return e;
    return hipStreamWaitEvent(main, tr->join, 0);
  }
  hipError_t next(hipStream_t* out) {
    *out = main;
    if (tr == nullptr) return hipSuccess;
    const int pos = k++ % group;
    if (pos == 0) {
      hipError_t e = meet();
      if (e != hipSuccess) return e;
    }
    if (pos & 1) {
      if (!side_open) {   // the side stream starts behind everything the caller's stream holds so far
        hipError_t e = hipEventRecord(tr->fork, main);
        if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(tr->side, tr->fork, 0);
        if (e != hipSuccess) return e;
        side_open = true;
      }
      *out = tr->side;
    }
    return hipSuccess;
  }
};

// which rows a wave takes: bricks when the caller named the grid the rows come from
void shape_rows(PredictArgs& a, const LaunchTuning& tune) {
  if (a.perm == nullptr && tune.grid_im > 0 && tune.grid_jm > 0 && a.nrow > 0 &&
      (tune.brick_li < 0 || tune.brick_li + tune.brick_lj + tune.brick_lk == 6)) {
    if (tune.brick_li < 0) a.shape.set_grid_auto((uint32_t)tune.grid_im, (uint32_t)tune.grid_jm, tune.grid_row0, a.nrow);
    else a.shape.set_grid((uint32_t)tune.grid_im, (uint32_t)tune.grid_jm, tune.grid_row0, a.nrow, (uint32_t)tune.brick_li,
                          (uint32_t)tune.brick_lj, (uint32_t)tune.brick_lk);
    a.shape.k_fastest = (uint32_t)tune.brick_k_fastest;
  }
  if (a.shape.ntiles(a.nrow) >= 0xFFFFFFFFull) a.shape = TileShape();   // tile_row numbers bricks in 32 bits
}

// A small batch: the trees in `split` runs, a wave per (run, tile), then the launch that sums the leaves in tree order.
template <class K>
hipError_t launch_rows_split(K kernel, size_t lds, const DeviceForest& fr, PredictArgs a, int num_cus, hipStream_t stream,
                             const LaunchTuning& tune, uint32_t split) {
  hipError_t e = ensure_lds(kernel, lds);
  if (e != hipSuccess) return e;
  a.xcd_remap = tune.xcd_remap;
  a.run_log = 0;
  a.run_lo_bits = 0;
  a.tile_begin = 0;
  a.tile_end = a.shape.ntiles(a.nrow);
  a.leaf_buf = tune.leaf_buf;
  a.tree_split = split;
  const int grid = tile_grid(kernel, lds, a.tile_end * split, num_cus);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, stream, fr, a, fr.super_heads, a.out);
  const uint64_t blocks = (a.tile_end + kWavesPerBlock - 1) / kWavesPerBlock;
  hipLaunchKernelGGL(combine_leaves_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(kBlock), 0, stream, a,
                     fr.base_score, a.out);
  return hipGetLastError();
}

// One launch, or a train of launches of one residency (grid x 4 tiles) each.
template <class K>
hipError_t launch_rows_tiled(K kernel, size_t lds, const DeviceForest& fr, PredictArgs a, int num_cus,
                             hipStream_t stream, const LaunchTuning& tune) {
  lds += (size_t)tune.lds_pad;
  hipError_t e = ensure_lds(kernel, lds);
  if (e != hipSuccess) return e;
  shape_rows(a, tune);
  const uint64_t ntiles = a.shape.ntiles(a.nrow);
  const int grid = tile_grid(kernel, lds, ntiles, num_cus);
  a.xcd_remap = tune.xcd_remap;
  // runs of consecutive rows inside a tile (load_pieces): a brick's cells along i, or all 64 rows without a grid
  a.run_log = 0;
  a.run_lo_bits = 0;
  if (tune.coop_rows && a.perm == nullptr && a.ncol == 27) {
    if (a.shape.im == 0) {
      a.run_log = 6;
    } else if (a.shape.li >= 2) {
      a.run_log = a.shape.li;
      a.run_lo_bits = a.shape.k_fastest ? a.shape.lk : 0u;
    }
  }
  if (tune.launches_per_residency <= 0) {
    a.tile_begin = 0;
    a.tile_end = ntiles;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, stream, fr, a, fr.super_heads, a.out);
    return hipGetLastError();
  }
  const uint64_t per_launch = (uint64_t)grid * kWavesPerBlock * (uint64_t)tune.launches_per_residency;
  TrainCursor train(stream, tune);
  for (uint64_t t0 = 0; t0 < ntiles; t0 += per_launch) {
    a.tile_begin = t0;
    a.tile_end = t0 + per_launch < ntiles ? t0 + per_launch : ntiles;
    const uint64_t blocks = (a.tile_end - a.tile_begin + kWavesPerBlock - 1) / kWavesPerBlock;
    hipStream_t s;
    e = train.next(&s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3((unsigned)(blocks < (uint64_t)grid ? blocks : (uint64_t)grid)), dim3(kBlock), lds,
                       s, fr, a, fr.super_heads, a.out);
  }
  e = train.meet();
  if (e != hipSuccess) return e;
  return hipGetLastError();
}

// ids of ring launch trains, process-wide, never 0 (PredictArgs::train_id)
uint32_t next_train_id() {
  static std::atomic<uint32_t> counter{0};
  uint32_t id = counter.fetch_add(1, std::memory_order_relaxed) + 1;
  if (id == 0) id = counter.fetch_add(1, std::memory_order_relaxed) + 1;
  return id;
}

// The train of launches of predict_rows_ring_kernel: one 1 024-thread block per CU, `ring_rounds` tiles per wave
// and launch (the waves of a block walk the same trees by construction; what a launch boundary still buys is that the
// blocks of an XCD start on tree 0 together).
hipError_t launch_rows_ring(const DeviceForest& fr, PredictArgs a, int num_cus, hipStream_t stream, const LaunchTuning& tune) {
  hipError_t e = ensure_lds(predict_rows_ring_kernel, kRingLdsBytes);
  if (e != hipSuccess) return e;
  shape_rows(a, tune);
  const uint64_t ntiles = a.shape.ntiles(a.nrow);
  uint64_t grid = (ntiles + kRingWaves - 1) / kRingWaves;
  const uint64_t cus = tune.reserve_cus > 0 && tune.reserve_cus < num_cus ? (uint64_t)(num_cus - tune.reserve_cus) : (uint64_t)num_cus;
  if (grid > cus) grid = cus;
  a.xcd_remap = tune.xcd_remap;
  a.run_log = 0;
  a.run_lo_bits = 0;
  if (tune.coop_rows && a.perm == nullptr && a.ncol == 27) {
    if (a.shape.im == 0) {
      a.run_log = 6;
    } else if (a.shape.li >= 2) {
      a.run_log = a.shape.li;
      a.run_lo_bits = a.shape.k_fastest ? a.shape.lk : 0u;
    }
  }
  // rows that are not known to be neighbours on a grid (no hint and no level size found, or rows that come through the
  // clustering pass's permutation) keep the short launches: what their lanes ask for has little in common, the XCD's L2
  // is all they share, and it only holds while the blocks walk the same trees - shuffled C360 rows 47.2 ms at 16 rounds,
  // 50.3 at 64 (without the clustering pass 84 against 125); rows on a grid 24.28 / 24.13 (profiles/r04_sweeps.txt)
  // (through the permutation: 47.4 ms at 16, 45.9 at 4; 64 consecutive rows per wave: 34.05 at 16, 34.5 at 4)
  int rounds = tune.ring_rounds;
  const int no_grid = a.perm != nullptr ? kRingRoundsPermuted : kRingRoundsNoGrid;
  if (a.shape.im == 0 && (rounds <= 0 || rounds > no_grid)) rounds = no_grid;      // (0 = one launch: for rows on a grid only)
  const uint64_t per_launch = rounds <= 0 ? ntiles : grid * kRingWaves * (uint64_t)rounds;
  a.train_id = next_train_id();
  TrainCursor train(stream, tune);
  for (uint64_t t0 = 0; t0 < ntiles; t0 += per_launch) {
    a.tile_begin = t0;
    a.tile_end = t0 + per_launch < ntiles ? t0 + per_launch : ntiles;
    const uint64_t blocks = (a.tile_end - a.tile_begin + kRingWaves - 1) / kRingWaves;
    hipStream_t s;
    e = train.next(&s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(predict_rows_ring_kernel, dim3((unsigned)(blocks < grid ? blocks : grid)), dim3(kRingBlock),
                       kRingLdsBytes, s, fr, a, fr.super_heads, a.out);
  }
  e = train.meet();
  if (e != hipSuccess) return e;
  // A ring block that gave up waiting has not written its rows, and says so by writing this train's id to flags[2].
  // Behind the train: the tile kernel over ALL of the train's rows, every block of which leaves at once unless it finds
  // that id there (5 us per step when it does not), missing-aware and without deferring; its first block counts the
  // event.  Stream-ordered, so the device forms need no host in the loop (include/ohxgb.h: OHXBoosterRingReruns).
  if (a.flags != nullptr) {
    auto again = predict_rows_tile_kernel<2, 2, true, true>;
    const size_t lds2 = tile_lds_bytes(fr.num_feature, true);
    e = ensure_lds(again, lds2);
    if (e != hipSuccess) return e;
    PredictArgs b = a;
    b.only_if_train = a.train_id;
    b.defer_list = nullptr;
    b.defer_count = nullptr;
    b.defer_cap = 0;
    b.tile_begin = 0;
    b.tile_end = ntiles;
    hipLaunchKernelGGL(again, dim3(tile_grid(again, lds2, ntiles, num_cus)), dim3(kBlock), lds2, stream, fr, b, fr.super_heads,
                       b.out);
  }
  return hipGetLastError();
}

// The second launch of a deferred-rows predict: the rows of the list, 64 per wave, each lane its own row, missing-aware.
template <class K>
hipError_t launch_deferred(K kernel, size_t lds, const DeviceForest& fr, PredictArgs a, int num_cus, hipStream_t stream) {
  hipError_t e = ensure_lds(kernel, lds);
  if (e != hipSuccess) return e;
  a.perm = a.defer_list;
  a.perm_count = a.defer_count;
  a.perm_slots = a.defer_cap;
  a.defer_list = nullptr;
  a.defer_count = nullptr;
  a.shape = TileShape();
  a.run_log = 0;
  a.run_lo_bits = 0;
  a.tile_begin = 0;
  a.tile_end = ((uint64_t)a.defer_cap + kWave - 1) / kWave;
  const int grid = tile_grid(kernel, lds, a.tile_end, num_cus);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, stream, fr, a, fr.super_heads, a.out);
  return hipGetLastError();
}


}  // namespace

const char* kernel_kind_name(KernelKind k) {
  switch (k) {
    case KernelKind::Wide: return "wide";
    case KernelKind::Packed1: return "packed1";
    case KernelKind::Packed2: return "packed2";
    case KernelKind::Packed4: return "packed4";
    case KernelKind::Super1: return "super1";
    case KernelKind::Super2: return "super2";
    case KernelKind::Super3: return "super3";
    case KernelKind::Super4: return "super4";
    case KernelKind::Ring: return "ring";
  }
  return "?";
}

// What launch_predict does with a batch, decided in one place so that OHXBoosterKernelSymbolRows can say it too.
struct RowsPlan {
  bool ok = true;          // false: the forest has no array this kind of kernel could read
  bool is_super = false;
  bool direct = false;     // predict_rows_direct_kernel (wide nodes, no LDS)
  bool pf = false;         // 27-column rows, the next tile's in flight during a walk
  bool defer = false, listing = false;      // rows with missing values counted / left to a second launch
  bool ring = false;       // predict_rows_ring_kernel
  uint32_t split = 0;      // > 0: a small batch, its trees in this many runs over waves
  TileShape split_shape;
  size_t lds = 0;
};

RowsPlan plan_rows(KernelKind kind, const DeviceForest& fr, const PredictArgs& a, int num_cus, const LaunchTuning& tune) {
  RowsPlan p;
  p.is_super = kind == KernelKind::Super1 || kind == KernelKind::Super2 || kind == KernelKind::Super3 ||
               kind == KernelKind::Super4 || kind == KernelKind::Ring;
  p.lds = tile_lds_bytes(fr.num_feature, p.is_super);
  const bool tile_ok = fr.num_feature >= 1 && p.lds <= 160 * 1024 && a.ncol <= fr.num_feature;
  if (p.is_super && fr.super == nullptr) p.ok = false;
  if (a.pred_leaf || kind == KernelKind::Wide || !tile_ok || (!p.is_super && fr.packed == nullptr)) {
    p.direct = true;
    return p;
  }
  // every wave gets more than one tile per launch and the rows are the OH shape: prefetch
  p.pf = a.ncol == 27 && fr.num_feature == 27 && tune.launches_per_residency != 1 && tune.prefetch;
  // rows with missing values are left to a second launch (PredictArgs::defer_list)
  constexpr uint64_t kDeferMinRows = 1u << 18;
  p.defer = p.pf && a.perm == nullptr && tune.defer_buf != nullptr && tune.defer_words >= 2 && a.nrow < 0xFFFFFFF0ull &&
            (tune.defer_missing > 0 || (tune.defer_missing < 0 && a.nrow >= kDeferMinRows));
  p.listing = p.defer && !tune.defer_count_only;
  // A small batch leaves most of the chip's wave slots empty and takes as long as one tile's walk of ALL trees
  // (165 us for the OH booster whatever N, profiles/r03_latency_rows.json): cut the trees into runs walked by
  // different waves.  Decided on the live tiles of the batch against the chip's 20 waves per CU.
  if (p.is_super && a.perm == nullptr && tune.tree_split != 0 && tune.leaf_buf != nullptr && a.tree_end - a.tree_begin >= 8) {
    PredictArgs probe = a;
    shape_rows(probe, tune);
    // bricks laid over a level of which the batch holds a small part are mostly empty: 64 consecutive rows per wave then
    if (probe.shape.im != 0 && probe.shape.live_tiles() * 2 < probe.shape.ntiles(a.nrow)) probe.shape = TileShape();
    const uint64_t ntiles = probe.shape.ntiles(a.nrow);
    const uint64_t live = probe.shape.im != 0 && probe.shape.live_tiles() < ntiles ? probe.shape.live_tiles() : ntiles;
    const uint64_t slots = (uint64_t)num_cus * 20u;
    const uint32_t ntree = a.tree_end - a.tree_begin;
    uint64_t want = tune.tree_split > 1 ? (uint64_t)tune.tree_split : (live * 2 <= slots ? slots / (live ? live : 1) : 0);
    if (want > 10) want = 10;
    if (want * 4 > ntree) want = ntree / 4;
    if (want >= 2 && ntiles * ntree * kWave <= tune.leaf_words) {
      p.split = (uint32_t)want;
      p.split_shape = probe.shape;
    }
  }
  // the ring kernel takes the OH shape with the next rows in flight (a big batch); everything else of a booster
  // that asked for it - small batches with their trees split over waves, other column counts, the second launch of
  // the deferred rows - goes the super2 way
  p.ring = kind == KernelKind::Ring && p.pf && !p.split && a.tree_end > a.tree_begin;
  return p;
}

hipError_t launch_predict(KernelKind kind, const DeviceForest& fr, const PredictArgs& a_in, int num_cus,
                          hipStream_t stream, const LaunchTuning& tune) {
  if (a_in.nrow == 0) return hipSuccess;
  PredictArgs a = a_in;
  const RowsPlan plan = plan_rows(kind, fr, a, num_cus, tune);
  if (!plan.ok) return hipErrorInvalidValue;
  const size_t lds = plan.lds;
  if (plan.direct) {
    if (fr.wide == nullptr) return hipErrorInvalidValue;
    const int grid = grid_for(a.nrow, num_cus, 8);
    if (a.pred_leaf) hipLaunchKernelGGL(predict_rows_direct_kernel<true>, dim3(grid), dim3(kBlock), 0, stream, fr, a);
    else hipLaunchKernelGGL(predict_rows_direct_kernel<false>, dim3(grid), dim3(kBlock), 0, stream, fr, a);
    return hipGetLastError();
  }
  const bool pf = plan.pf, listing = plan.listing;
  if (plan.defer) {
    const uint64_t want = a.nrow / 32 + 1024;                                   // room for ~3 % of the rows
    a.defer_cap = listing ? (uint32_t)(want < tune.defer_words - 1 ? want : tune.defer_words - 1) : 0u;
    a.defer_count = tune.defer_buf;
    a.defer_list = tune.defer_buf + 1;
    hipError_t e = hipMemsetAsync(a.defer_count, 0, sizeof(uint32_t), stream);
    if (e == hipSuccess && listing) e = hipMemsetAsync(a.defer_list, 0xFF, (size_t)a.defer_cap * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
  }
  const uint32_t split = plan.split;
  if (split) a.shape = plan.split_shape;
#define OHX_ROWS_T(FMT, CH, TOPS)                                                                                  \
  {                                                                                                                \
    if (split) return launch_rows_split(predict_rows_tile_kernel<FMT, CH, false, TOPS>, lds, fr, a, num_cus, stream, tune, split); \
    hipError_t e_ = pf ? launch_rows_tiled(predict_rows_tile_kernel<FMT, CH, true, TOPS>, lds, fr, a, num_cus, stream, tune)   \
                       : launch_rows_tiled(predict_rows_tile_kernel<FMT, CH, false, TOPS>, lds, fr, a, num_cus, stream, tune); \
    if (e_ == hipSuccess && listing)                                                                               \
      e_ = launch_deferred(predict_rows_tile_kernel<FMT, CH, false, TOPS>, lds, fr, a, num_cus, stream);           \
    return e_;                                                                                                     \
  }
#define OHX_ROWS(FMT, CH)                    \
  if (fr.tree_tops) OHX_ROWS_T(FMT, CH, true); \
  OHX_ROWS_T(FMT, CH, false)
  if (plan.ring) {
    hipError_t e_ = launch_rows_ring(fr, a, num_cus, stream, tune);
    if (e_ == hipSuccess && listing)
      e_ = launch_deferred(predict_rows_tile_kernel<2, 2, false, true>, lds, fr, a, num_cus, stream);
    return e_;
  }
  switch (kind) {
    case KernelKind::Packed1: OHX_ROWS_T(1, 1, false);
    case KernelKind::Packed2: OHX_ROWS_T(1, 2, false);
    case KernelKind::Packed4: OHX_ROWS_T(1, 4, false);
    case KernelKind::Super1: OHX_ROWS(2, 1);
    case KernelKind::Super3: OHX_ROWS(2, 3);
    case KernelKind::Super4: OHX_ROWS(2, 4);
    default: OHX_ROWS(2, 2);
  }
#undef OHX_ROWS
#undef OHX_ROWS_T
}

std::string predict_kernel_symbol(KernelKind kind, const DeviceForest& fr, uint32_t ncol, const LaunchTuning& tune) {
  const bool is_super = kind == KernelKind::Super1 || kind == KernelKind::Super2 || kind == KernelKind::Super3 ||
                        kind == KernelKind::Super4 || kind == KernelKind::Ring;
  const size_t lds = tile_lds_bytes(fr.num_feature, is_super);
  const bool tile_ok = fr.num_feature >= 1 && lds <= 160 * 1024 && ncol <= fr.num_feature;
  if (kind == KernelKind::Wide || !tile_ok || (!is_super && fr.packed == nullptr)) return "predict_rows_direct_kernel<false>";
  const bool pf = ncol == 27 && fr.num_feature == 27 && tune.launches_per_residency != 1 && tune.prefetch;
  if (kind == KernelKind::Ring && pf) return "predict_rows_ring_kernel";
  int chains = 2;
  if (kind == KernelKind::Packed1 || kind == KernelKind::Super1) chains = 1;
  if (kind == KernelKind::Super3) chains = 3;
  if (kind == KernelKind::Packed4 || kind == KernelKind::Super4) chains = 4;
  return std::string("predict_rows_tile_kernel<") + (is_super ? "2," : "1,") + std::to_string(chains) +
         (pf ? ",true," : ",false,") + (is_super && fr.tree_tops ? "true>" : "false>");
}

// Every __global__ a predict on this batch launches, in order, joined by " + " (OHXBoosterKernelSymbolRows).
std::string predict_kernel_symbols_rows(KernelKind kind, const DeviceForest& fr, const PredictArgs& a, int num_cus,
                                        const LaunchTuning& tune) {
  const RowsPlan p = plan_rows(kind, fr, a, num_cus, tune);
  if (a.nrow == 0 || !p.ok) return "";
  if (p.direct) return a.pred_leaf ? "predict_rows_direct_kernel<true>" : "predict_rows_direct_kernel<false>";
  int chains = 2;
  if (kind == KernelKind::Packed1 || kind == KernelKind::Super1) chains = 1;
  if (kind == KernelKind::Super3) chains = 3;
  if (kind == KernelKind::Packed4 || kind == KernelKind::Super4) chains = 4;
  const bool tops = p.is_super && fr.tree_tops;
  auto tile = [&](bool prefetch) {
    return std::string("predict_rows_tile_kernel<") + (p.is_super ? "2," : "1,") + std::to_string(chains) +
           (prefetch ? ",true," : ",false,") + (tops ? "true>" : "false>");
  };
  if (p.split) return tile(false) + " + combine_leaves_kernel";
  std::string out;
  if (p.ring) {
    out = "predict_rows_ring_kernel + predict_rows_tile_kernel<2,2,true,true> (only after a ring time-out)";
    if (p.listing) out += " + predict_rows_tile_kernel<2,2,false,true> (rows with missing values)";
    return out;
  }
  out = tile(p.pf);
  if (p.listing) out += " + " + tile(false) + " (rows with missing values)";
  return out;
}

// Same train of launches as the row kernels: a launch per `launches_per_residency` residencies
// keeps the waves of an XCD on the same few trees.
template <class K>
hipError_t launch_fields_tiled(K kernel, size_t lds, const DeviceForest& fr, FieldsArgs a, uint64_t nrow, int num_cus,
                               hipStream_t stream, const LaunchTuning& tune, int waves_per_block = kWavesPerBlock,
                               int rounds = -1) {
  if (rounds < 0) rounds = tune.launches_per_residency;
  const bool ring = waves_per_block != kWavesPerBlock;
  const unsigned threads = (unsigned)waves_per_block * kWave;
  hipError_t e = ensure_lds(kernel, lds);
  if (e != hipSuccess) return e;
  if (tune.brick_li < 0) a.shape.set_grid_auto((uint32_t)a.im, (uint32_t)a.jm, 0, nrow);
  else if (tune.brick_li + tune.brick_lj + tune.brick_lk == 6)
    a.shape.set_grid((uint32_t)a.im, (uint32_t)a.jm, 0, nrow, (uint32_t)tune.brick_li, (uint32_t)tune.brick_lj,
                     (uint32_t)tune.brick_lk);
  a.shape.k_fastest = (uint32_t)tune.brick_k_fastest;
  a.xcd_remap = tune.xcd_remap;
  if (a.shape.ntiles(nrow) >= 0xFFFFFFFFull) a.shape = TileShape();
  const uint64_t ntiles = a.shape.ntiles(nrow);
  int grid = tile_grid(kernel, lds, ntiles, num_cus);
  if (ring) {        // one block per CU, less the CUs left free for a collective's kernels (LaunchTuning::reserve_cus)
    const uint64_t blocks = (ntiles + waves_per_block - 1) / waves_per_block;
    const uint64_t cus = tune.reserve_cus > 0 && tune.reserve_cus < num_cus ? (uint64_t)(num_cus - tune.reserve_cus) : (uint64_t)num_cus;
    grid = (int)(blocks < cus ? blocks : cus);
  }
  // rows with missing values are left to a second launch (PredictArgs::defer_list)
  constexpr uint64_t kDeferMinRows = 1u << 18;
  if (tune.defer_buf != nullptr && tune.defer_words >= 2 && nrow < 0xFFFFFFF0ull && fr.num_feature == 27 &&
      (tune.defer_missing > 0 || (tune.defer_missing < 0 && nrow >= kDeferMinRows))) {
    const uint64_t want = nrow / 32 + 1024;
    a.defer_cap = tune.defer_count_only ? 0u : (uint32_t)(want < tune.defer_words - 1 ? want : tune.defer_words - 1);
    a.defer_count = tune.defer_buf;
    a.defer_list = tune.defer_buf + 1;
    e = hipMemsetAsync(a.defer_count, 0, sizeof(uint32_t), stream);
    if (e == hipSuccess && a.defer_cap) e = hipMemsetAsync(a.defer_list, 0xFF, (size_t)a.defer_cap * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
  }
  const uint64_t per_launch = rounds <= 0 ? ntiles : (uint64_t)grid * waves_per_block * (uint64_t)rounds;
  if (ring) a.train_id = next_train_id();
  TrainCursor train(stream, tune);
  for (uint64_t t0 = 0; t0 < ntiles; t0 += per_launch) {
    a.tile_begin = t0;
    a.tile_end = t0 + per_launch < ntiles ? t0 + per_launch : ntiles;
    const uint64_t blocks = (a.tile_end - a.tile_begin + waves_per_block - 1) / waves_per_block;
    hipStream_t s;
    e = train.next(&s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3((unsigned)(blocks < (uint64_t)grid ? blocks : (uint64_t)grid)), dim3(threads), lds,
                       s, fr, a, fr.super_heads, a.out, a.margin_out);
  }
  e = train.meet();
  if (e != hipSuccess) return e;
  if (ring && a.flags != nullptr) {
    // as launch_rows_ring: the tile kernel over all of the slab, predicated on this train's id in flags[2]
    auto again = predict_fields_kernel<2, 2, true>;
    const size_t lds2 = tile_lds_bytes(fr.num_feature, true);
    e = ensure_lds(again, lds2);
    if (e != hipSuccess) return e;
    FieldsArgs b = a;
    b.only_if_train = a.train_id;
    b.defer_list = nullptr;
    b.defer_count = nullptr;
    b.defer_cap = 0;
    b.tile_begin = 0;
    b.tile_end = ntiles;
    hipLaunchKernelGGL(again, dim3(tile_grid(again, lds2, ntiles, num_cus)), dim3(kBlock), lds2, stream, fr, b, fr.super_heads,
                       b.out, b.margin_out);
  }
  if (a.defer_list != nullptr && a.defer_cap != 0u) {
    // the second launch: the listed rows, 64 per wave, every lane gathering its own row from the fields
    a.perm = a.defer_list;
    a.perm_count = a.defer_count;
    a.perm_slots = a.defer_cap;
    a.defer_list = nullptr;
    a.defer_count = nullptr;
    a.tile_begin = 0;
    a.tile_end = ((uint64_t)a.defer_cap + kWave - 1) / kWave;
    if (ring) {      // the listed rows go through the tile kernel (a lane per row, missing-aware)
      auto second = predict_fields_kernel<2, 2, true>;
      const size_t lds2 = tile_lds_bytes(fr.num_feature, true);
      e = ensure_lds(second, lds2);
      if (e != hipSuccess) return e;
      const int grid2 = tile_grid(second, lds2, a.tile_end, num_cus);
      hipLaunchKernelGGL(second, dim3(grid2), dim3(kBlock), lds2, stream, fr, a, fr.super_heads, a.out, a.margin_out);
    } else {
      const int grid2 = tile_grid(kernel, lds, a.tile_end, num_cus);
      hipLaunchKernelGGL(kernel, dim3(grid2), dim3(kBlock), lds, stream, fr, a, fr.super_heads, a.out, a.margin_out);
    }
  }
  return hipGetLastError();
}

// A small slab: the trees in `split` runs, a wave per (run, tile), then the launch that sums the leaves in tree order
template <class K>
hipError_t launch_fields_split(K kernel, size_t lds, const DeviceForest& fr, FieldsArgs a, int num_cus, hipStream_t stream,
                               const LaunchTuning& tune, uint32_t split) {
  hipError_t e = ensure_lds(kernel, lds);
  if (e != hipSuccess) return e;
  a.xcd_remap = tune.xcd_remap;
  a.tile_begin = 0;
  a.tile_end = a.shape.ntiles((uint64_t)a.im * a.jm * (uint64_t)(a.k2 - a.k1 + 1));
  a.leaf_buf = tune.leaf_buf;
  a.tree_split = split;
  const int grid = tile_grid(kernel, lds, a.tile_end * split, num_cus);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), lds, stream, fr, a, fr.super_heads, a.out, a.margin_out);
  const uint64_t blocks = (a.tile_end + kWavesPerBlock - 1) / kWavesPerBlock;
  hipLaunchKernelGGL(combine_leaves_fields_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(kBlock), 0, stream, a,
                     fr.base_score, a.out, a.margin_out);
  return hipGetLastError();
}

hipError_t launch_predict_fields(KernelKind kind, const DeviceForest& fr, const FieldsArgs& a, int num_cus,
                                 hipStream_t stream, const LaunchTuning& tune) {
  if (a.k2 < a.k1 || a.im <= 0 || a.jm <= 0) return hipSuccess;
  const uint64_t nrow = (uint64_t)a.im * (uint64_t)a.jm * (uint64_t)(a.k2 - a.k1 + 1);
  const bool is_super = kind == KernelKind::Super1 || kind == KernelKind::Super2 || kind == KernelKind::Super3 ||
                        kind == KernelKind::Super4 || kind == KernelKind::Ring;
  const size_t lds = tile_lds_bytes(fr.num_feature, is_super);
  if (fr.num_feature < 1 || lds > 160 * 1024) return hipErrorInvalidValue;
  if (is_super && fr.super == nullptr) return hipErrorInvalidValue;
  const bool use_wide = kind == KernelKind::Wide || (!is_super && fr.packed == nullptr);
  if (use_wide && fr.wide == nullptr) return hipErrorInvalidValue;
#define OHX_LAUNCH_FIELDS_T(FMT, CH, TOPS) \
  return launch_fields_tiled(predict_fields_kernel<FMT, CH, TOPS>, lds, fr, a, nrow, num_cus, stream, tune)
#define OHX_LAUNCH_FIELDS(FMT, CH)                    \
  if (fr.tree_tops) OHX_LAUNCH_FIELDS_T(FMT, CH, true); \
  OHX_LAUNCH_FIELDS_T(FMT, CH, false)
  if (use_wide) OHX_LAUNCH_FIELDS_T(0, 1, false);
  // A small slab - a GEOS rank's block - leaves most of the chip's wave slots empty and takes as long as one tile's walk
  // of ALL trees: its trees are cut into runs walked by different waves, as launch_predict does for small row batches
  if (is_super && tune.tree_split != 0 && tune.leaf_buf != nullptr && a.tree_end - a.tree_begin >= 8) {
    FieldsArgs probe = a;
    if (tune.brick_li < 0) probe.shape.set_grid_auto((uint32_t)a.im, (uint32_t)a.jm, 0, nrow);
    else if (tune.brick_li + tune.brick_lj + tune.brick_lk == 6)
      probe.shape.set_grid((uint32_t)a.im, (uint32_t)a.jm, 0, nrow, (uint32_t)tune.brick_li, (uint32_t)tune.brick_lj,
                           (uint32_t)tune.brick_lk);
    probe.shape.k_fastest = (uint32_t)tune.brick_k_fastest;
    const uint64_t ntiles = probe.shape.ntiles(nrow);
    const uint64_t live = probe.shape.im != 0 && probe.shape.live_tiles() < ntiles ? probe.shape.live_tiles() : ntiles;
    const uint64_t slots = (uint64_t)num_cus * 20u;
    const uint32_t ntree = a.tree_end - a.tree_begin;
    uint64_t want = tune.tree_split > 1 ? (uint64_t)tune.tree_split : (live * 2 <= slots ? slots / (live ? live : 1) : 0);
    if (want > 10) want = 10;
    if (want * 4 > ntree) want = ntree / 4;
    if (want >= 2 && ntiles < 0xFFFFFFFFull && ntiles * ntree * kWave <= tune.leaf_words) {
      if (kind == KernelKind::Super1) return launch_fields_split(predict_fields_kernel<2, 1, false>, lds, fr, probe, num_cus, stream, tune, (uint32_t)want);
      if (kind == KernelKind::Super4) return launch_fields_split(predict_fields_kernel<2, 4, false>, lds, fr, probe, num_cus, stream, tune, (uint32_t)want);
      if (fr.tree_tops) return launch_fields_split(predict_fields_kernel<2, 2, true>, lds, fr, probe, num_cus, stream, tune, (uint32_t)want);
      return launch_fields_split(predict_fields_kernel<2, 2, false>, lds, fr, probe, num_cus, stream, tune, (uint32_t)want);
    }
  }
  // the ring kernel for slabs that fill the chip at least twice; smaller ones (a rank-sized block) the super2 way:
  // a block of the ring kernel is 16 tiles that wait for each other's trees
  if (kind == KernelKind::Ring && fr.num_feature == 27 && a.tree_end > a.tree_begin &&
      nrow >= (uint64_t)num_cus * kRingWaves * kWave * 2u)
    return launch_fields_tiled(predict_fields_ring_kernel, kRingLdsBytes, fr, a, nrow, num_cus, stream, tune, kRingWaves,
                               tune.ring_rounds);
  switch (kind) {
    case KernelKind::Packed1: OHX_LAUNCH_FIELDS_T(1, 1, false);
    case KernelKind::Packed2: OHX_LAUNCH_FIELDS_T(1, 2, false);
    case KernelKind::Packed4: OHX_LAUNCH_FIELDS_T(1, 4, false);
    case KernelKind::Super1: OHX_LAUNCH_FIELDS(2, 1);
    case KernelKind::Super3: OHX_LAUNCH_FIELDS(2, 3);
    case KernelKind::Super4: OHX_LAUNCH_FIELDS(2, 4);
    default: OHX_LAUNCH_FIELDS(2, 2);
  }
#undef OHX_LAUNCH_FIELDS
#undef OHX_LAUNCH_FIELDS_T
}

uint32_t cluster_key_bits(const ClusterArgs& a) { return a.ntrees * (1u + 2u * a.nsteps); }

template <int NT>
hipError_t launch_cluster_keys_nt(const DeviceForest& fr, const ClusterArgs& a, size_t lds, int num_cus, hipStream_t stream) {
  hipError_t e = ensure_lds(cluster_keys_kernel<NT>, lds);
  if (e != hipSuccess) return e;
  const int grid = tile_grid(cluster_keys_kernel<NT>, lds, (a.nrow + kWave - 1) / kWave, num_cus);
  hipLaunchKernelGGL(cluster_keys_kernel<NT>, dim3(grid), dim3(kBlock), lds, stream, fr, a);
  return hipGetLastError();
}

hipError_t launch_cluster_keys(const DeviceForest& fr, const ClusterArgs& a, int num_cus, hipStream_t stream) {
  if (a.nrow == 0) return hipSuccess;
  if (fr.super == nullptr || cluster_key_bits(a) > 32u || a.ntrees == 0 || a.ntrees > 4 || a.ntrees > fr.num_trees ||
      a.ncol > fr.num_feature)
    return hipErrorInvalidValue;
  const size_t lds = (size_t)kWavesPerBlock * fr.num_feature * kKeyTileStride * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  switch (a.ntrees) {
    case 1: return launch_cluster_keys_nt<1>(fr, a, lds, num_cus, stream);
    case 2: return launch_cluster_keys_nt<2>(fr, a, lds, num_cus, stream);
    case 3: return launch_cluster_keys_nt<3>(fr, a, lds, num_cus, stream);
    default: return launch_cluster_keys_nt<4>(fr, a, lds, num_cus, stream);
  }
}

// grid of the kernels that take a level per blockIdx.y and stride over a range of columns
static dim3 level_grid(uint64_t ncols, int km) {
  uint64_t bx = (ncols + (uint64_t)kBlock * 4 - 1) / ((uint64_t)kBlock * 4);     // four columns per thread
  if (bx < 1) bx = 1;
  if (bx > 1024) bx = 1024;
  return dim3((unsigned)bx, (unsigned)km);
}

hipError_t launch_feature_prep(const PrepArgs& a, float* aod_scratch, hipStream_t stream) {
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  if (plane == 0 || a.km <= 0) return hipSuccess;
  if (a.col0 + a.ncols > plane || a.km > 65535) return hipErrorInvalidValue;
  const uint64_t ncols = a.ncols ? a.ncols : plane;
  hipLaunchKernelGGL(feature_pointwise_kernel, level_grid(ncols, a.km), dim3(kBlock), 0, stream, a, aod_scratch);
  // a rank's block: a wave per column.  Also for a piece whose features are computed beside another piece's walk: the
  // kernel with a column in registers needs 143 of them, this one 26 and 2 KB of LDS
  if ((ncols <= 8192 || a.beside_a_walk) && a.km <= kColWaveMaxKm) {
    hipLaunchKernelGGL(feature_column_sums_wave_kernel, dim3((unsigned)((ncols + kWavesPerBlock - 1) / kWavesPerBlock), 3),
                       dim3(kBlock), 0, stream, a, (const float*)aod_scratch);
    return hipGetLastError();
  }
  if (a.km == 72) {        // GEOS's 72 levels: the column in registers
    hipLaunchKernelGGL(feature_column_sums_reg_kernel<72>, dim3((unsigned)((ncols + kBlock - 1) / kBlock), 3), dim3(kBlock), 0,
                       stream, a, (const float*)aod_scratch);
    return hipGetLastError();
  }
  const size_t lds = (size_t)kWavesPerBlock * a.km * kWave * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipError_t e = ensure_lds(feature_column_sums_kernel, lds);
  if (e != hipSuccess) return e;
  const uint64_t cols_per_block = (uint64_t)kWavesPerBlock * kWave;
  hipLaunchKernelGGL(feature_column_sums_kernel, dim3((unsigned)((ncols + cols_per_block - 1) / cols_per_block), 3),
                     dim3(kBlock), lds, stream, a, (const float*)aod_scratch);
  return hipGetLastError();
}

hipError_t launch_solar_geometry(const SolarArgs& a, hipStream_t stream) {
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  if (plane == 0) return hipSuccess;
  hipLaunchKernelGGL(solar_geometry_kernel, dim3(grid_for(plane, 256, 8)), dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_k_slab(const SlabArgs& a, hipStream_t stream) {
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  if (plane == 0) return hipSuccess;
  hipLaunchKernelGGL(k_slab_kernel, dim3(grid_for(plane, 256, 8)), dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_post_process(const PostArgs& a, hipStream_t stream) {
  const uint64_t plane = (uint64_t)a.im * (uint64_t)a.jm;
  if (plane == 0 || a.km <= 0) return hipSuccess;
  if (a.col0 + a.ncols > plane || a.km > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(post_process_kernel, level_grid(a.ncols ? a.ncols : plane, a.km), dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}

// Many arrays moved by ONE launch: array a of the list is copied by the blocks (*, a).  Either side may be the
// caller's registered host memory, which the GPU reads and writes over PCIe directly (capi.cpp HostRegistry): forty
// copies of a rank-sized block's arrays cost forty times a copy's fixed price, one launch costs one.
namespace {
__global__ __launch_bounds__(kBlock) void copy_arrays_kernel(CopyList l) {
  const uint32_t a = blockIdx.y;
  const float* __restrict__ src = l.src[a];
  float* __restrict__ dst = l.dst[a];
  const uint64_t n = l.n[a];
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
    const uint64_t n4 = n / 4;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (uint64_t i = t; i < n4; i += stride) d4[i] = s4[i];
    for (uint64_t i = n4 * 4 + t; i < n; i += stride) dst[i] = src[i];
  } else {
    for (uint64_t i = t; i < n; i += stride) dst[i] = src[i];
  }
}
}  // namespace

hipError_t launch_copy_arrays(const CopyList& l, hipStream_t stream, uint32_t max_blocks) {
  if (l.count == 0) return hipSuccess;
  uint64_t longest = 0;
  for (uint32_t a = 0; a < l.count; ++a) longest = l.n[a] > longest ? l.n[a] : longest;
  uint64_t blocks = (longest / 4 + kBlock - 1) / kBlock;
  // `max_blocks` counts the launch's blocks over all arrays (0 = up to 2 048 per array)
  const uint64_t per_array = max_blocks == 0 ? 2048 : (max_blocks / l.count > 0 ? max_blocks / l.count : 1);
  if (blocks > per_array) blocks = per_array;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(copy_arrays_kernel, dim3((unsigned)blocks, l.count), dim3(kBlock), 0, stream, l);
  return hipGetLastError();
}

hipError_t launch_scan_dense(const float* data, uint64_t count, float missing, uint32_t* flags, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  hipLaunchKernelGGL(scan_dense_kernel, dim3(grid_for(count / 4 + 1, 256, 8)), dim3(kBlock), 0, stream, data, count,
                     missing, flags);
  return hipGetLastError();
}

hipError_t launch_detect_period(const float* data, uint64_t nrow, uint32_t ncol, const PeriodColumns& cols,
                                uint32_t ncols, uint32_t kmax, uint32_t* d_verdict, hipStream_t stream) {
  if (kmax < 2 || ncols == 0) return hipSuccess;
  hipLaunchKernelGGL(detect_period_kernel, dim3(kmax - 1, ncols), dim3(kBlock), 0, stream, data, nrow, ncol, cols, kmax,
                     d_verdict);
  return hipGetLastError();
}

}  // namespace ohx
