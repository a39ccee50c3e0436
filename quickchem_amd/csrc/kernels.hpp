// Launchers for the gfx950 kernels (definitions in kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>

#include "flatten.hpp"

namespace ohx {

// Flag bits the kernels raise in a device word
enum : uint32_t {
  kFlagInfInput = 1u,  // an input value was +-inf while `missing` is finite
  // (2u was the ring kernels' time-out until round 4, an error.  A time-out is now recorded as the id of its launch
  // train in the third word of the flags buffer - flags[2] - and settled on the stream: PredictArgs::train_id.)
};

// trees whose first-step super-nodes the tile kernels keep in LDS (kernels.hip)
constexpr uint32_t kFirstStepTrees = 128;

struct DeviceForest {
  const PackedNode* packed = nullptr;   // may be null (booster does not qualify)
  const WideNode* wide = nullptr;       // may be null until first needed
  const uint32_t* roots = nullptr;
  const int32_t* packed_orig_id = nullptr;
  const SuperNode* super = nullptr;     // may be null (booster does not qualify)
  const SuperTreeHead* super_heads = nullptr; // per tree
  uint32_t num_trees = 0;
  uint32_t num_feature = 0;
  float base_score = 0.0f;
  // sizes in bytes of the packed / super arrays (< 4 GiB: they are read through buffer
  // descriptors, whose range check also turns any stray index into a harmless zero read)
  uint32_t packed_bytes = 0;
  uint32_t super_bytes = 0;
  // super-nodes: walk with one coalesced "tree top" load per tree (kernels.hip walk_super); chosen by the host
  uint32_t tree_tops = 0;
};

// Which 64 rows a wave takes.  Without a grid (im == 0) tile t is rows 64t .. 64t+63.  With the
// grid the rows were gathered from (m = i + im*(j + jm*k), OH_GridCompMod.F90:309-345) a tile is a
// brick of 2^li x 2^lj x 2^lk neighbouring gridcells: neighbours in all three directions have
// similar features, so the four lanes of a quad - the unit the L1 looks tags up for - and the
// lanes of a wave share more tree nodes than 64 cells of one latitude line do.
// The rows may be any contiguous range [row0, row0 + nrow) of the grid's rows (a rank's shard);
// bricks overhang the range where an extent is not a multiple of the brick's; those lanes idle.
struct TileShape {
  uint32_t im = 0, jm = 0;
  uint32_t k_first = 0, nk = 0;         // levels the row range touches
  uint64_t row0 = 0, nrow = 0;
  uint32_t li = 2, lj = 2, lk = 2;      // li + lj + lk == 6
  uint32_t nbi = 0, nbj = 0, nbk = 0;   // bricks along i, j, k
  uint32_t k_fastest = 0;               // lane order inside a brick: 0 = i, j, k; 1 = k, i, j
  uint64_t ntiles(uint64_t nrow_) const { return im ? (uint64_t)nbi * nbj * nbk : (nrow_ + 63) / 64; }
  // Bricks that hold at least one row.  Bricks are laid over whole levels; the row range may begin and end inside a
  // level, and the bricks of a slab of levels none of which is complete only count along the j they touch (the
  // kernels skip a brick without a row).
  uint64_t live_tiles() const {
    if (!im) return (nrow + 63) / 64;
    const uint64_t plane = (uint64_t)im * jm, last = row0 + nrow - 1;
    uint64_t live = 0;
    for (uint32_t b = 0; b < nbk; ++b) {
      uint32_t jlo = jm, jhi = 0;                                  // j range the slab's levels touch, inclusive
      for (uint32_t d = 0; d < (1u << lk); ++d) {
        const uint64_t k = (uint64_t)k_first + ((uint64_t)b << lk) + d;
        const uint64_t lev0 = k * plane, lev1 = lev0 + plane - 1;
        if (lev1 < row0 || lev0 > last) continue;
        const uint64_t a = row0 > lev0 ? row0 - lev0 : 0, z = (last < lev1 ? last : lev1) - lev0;
        const uint32_t j0 = (uint32_t)(a / im), j1 = (uint32_t)(z / im);
        jlo = j0 < jlo ? j0 : jlo;
        jhi = j1 > jhi ? j1 : jhi;
      }
      if (jhi >= jlo && jlo < jm) live += (uint64_t)nbi * ((jhi >> lj) - (jlo >> lj) + 1);
    }
    return live ? live : 1;
  }
  // fraction of the lanes of the bricks that are walked that carry a row
  double fill() const { return im ? (double)nrow / (64.0 * (double)live_tiles()) : 1.0; }
  void set_grid(uint32_t im_, uint32_t jm_, uint64_t row0_, uint64_t nrow_, uint32_t li_, uint32_t lj_, uint32_t lk_) {
    im = im_; jm = jm_; row0 = row0_; nrow = nrow_; li = li_; lj = lj_; lk = lk_;
    const uint64_t plane = (uint64_t)im * jm;
    k_first = (uint32_t)(row0 / plane);
    nk = (uint32_t)((row0 + nrow - 1) / plane) - k_first + 1;
    nbi = (im + (1u << li) - 1) >> li;
    nbj = (jm + (1u << lj) - 1) >> lj;
    nbk = (nk + (1u << lk) - 1) >> lk;
  }
  // The brick that wastes the fewest lanes, preferring the more compact one on a tie within a few
  // percent (measured on the C360 step: 4x4x4 36.4 ms, 8x4x2 36.6 ms, 8x8x1 37.1 ms).
  void set_grid_auto(uint32_t im_, uint32_t jm_, uint64_t row0_, uint64_t nrow_) {
    // jm == 1: only the level size is known (im = cells per level): runs of cells x levels, measured on the
    // C360 step at 34.8 / 35.2 / 37.0 ms for 8x1x8 / 16x1x4 / 32x1x2 (4x4x4 with the full grid: 34.6 ms)
    static const uint32_t cand3[3][3] = {{2, 2, 2}, {3, 2, 1}, {3, 3, 0}};
    static const uint32_t cand1[3][3] = {{3, 0, 3}, {4, 0, 2}, {5, 0, 1}};
    static const double speed3[3] = {1.0, 0.994, 0.98}, speed1[3] = {1.0, 0.99, 0.94};
    const uint32_t (*cand)[3] = jm_ == 1 ? cand1 : cand3;
    const double* speed = jm_ == 1 ? speed1 : speed3;
    double best = -1.0;
    int pick = 0;
    for (int q = 0; q < 3; ++q) {
      set_grid(im_, jm_, row0_, nrow_, cand[q][0], cand[q][1], cand[q][2]);
      const double score = fill() * speed[q];
      if (score > best) best = score, pick = q;
    }
    set_grid(im_, jm_, row0_, nrow_, cand[pick][0], cand[pick][1], cand[pick][2]);
  }
};

struct PredictArgs {
  const float* rows = nullptr;  // [nrow][ncol] row-major (the reference's xx_carr(27,N))
  uint64_t nrow = 0;
  uint32_t ncol = 0;
  float missing = 0.0f;
  uint32_t tree_begin = 0, tree_end = 0;
  float* out = nullptr;         // [nrow] margins, or [nrow][ntree] leaf ids when pred_leaf
  bool pred_leaf = false;
  // three device words: [0] OR-ed with kFlag*; [1] ring re-runs (counted by the launch that re-runs); [2] the id of the
  // last ring launch train in which a block gave up waiting for another and left its rows unwritten
  uint32_t* flags = nullptr;
  // ring kernels: the id of this launch train (never 0), written to flags[2] by a block that gives up.  Tile kernels: not
  // 0 = this is the launch the launchers put behind train `only_if_train`: every block leaves at once unless flags[2]
  // holds that id - then the launch predicts the train's rows again (and block 0 counts it in flags[1])
  uint32_t train_id = 0, only_if_train = 0;
  uint64_t tile_begin = 0;      // first 64-row tile of this launch (tile kernels)
  uint64_t tile_end = 0;        // one past the last tile of this launch
  int xcd_remap = 1;            // give each XCD a contiguous range of tiles
  const uint32_t* perm = nullptr;  // rows grouped by the clustering pass: lane l of tile t takes row perm[64 t + l]
  // ... or the rows a first pass left for a second one (below): then the number of slots is read from *perm_count on
  // the device (capped by nrow_slots) and a slot holding 0xFFFFFFFF is empty
  const uint32_t* perm_count = nullptr;
  uint32_t perm_slots = 0;
  // Deferred rows (27-column rows with prefetch, big batches).  A wave some of whose rows hold missing values
  // appends those rows to defer_list (room reserved with one atomicAdd on *defer_count; if the list is full the wave
  // walks missing-aware as before) and walks the tile WITHOUT missing-value logic; the listed rows are predicted by a
  // second, small launch through the list.  Zero *defer_count and fill the list with 0xFF before the first launch.
  uint32_t* defer_list = nullptr;
  uint32_t* defer_count = nullptr;
  uint32_t defer_cap = 0;
  // Small batches (fewer tiles than the chip has wave slots): the trees are cut into tree_split runs and a tile is
  // walked by tree_split waves, one per run, each writing its trees' LEAF VALUES to leaf_buf[(tile * ntree + t) * 64 +
  // lane]; a second launch adds them up in tree order (float32 addition is not associative: partial sums would not
  // give the margin of the sequential sum).  Latency of a predict: that of 100 / tree_split trees.
  float* leaf_buf = nullptr;
  uint32_t tree_split = 1;
  // 27-column rows of a tile fetched by the wave together, run of consecutive rows by run (kernels.hip RowPieces):
  // log2 of the rows per run (0 = every lane fetches its own row), and where a run's rows sit among the lanes
  uint32_t run_log = 0, run_lo_bits = 0;
  TileShape shape;              // lanes -> rows
};

// A second stream for the launch train (owned by the booster): launches of a group alternate between the caller's
// stream and this one, so that the waves of a launch that finish early are replaced by the next launch's at once
// instead of idling until the last wave of their launch is done (LaunchTuning::overlap_group).
struct TrainStreams {
  hipStream_t side = nullptr;      // non-blocking
  hipEvent_t fork = nullptr, join = nullptr;
};

// ring kernels: tiles per wave and launch at most, for rows not known to lie on a grid, and for rows that come through
// the clustering pass's permutation (launch_rows_ring)
constexpr int kRingRoundsNoGrid = 16;
constexpr int kRingRoundsPermuted = 4;

struct LaunchTuning {
  // 0: one launch for the whole batch (waves stride over tiles).  > 0: one launch per this many
  // "waves of tiles": every launch starts all resident waves on tree 0 together, so the waves of
  // an XCD walk the same few trees at the same time and share their node lines in that XCD's L2.
  int launches_per_residency = 2;
  int xcd_remap = 1;
  // launches of the train in groups of this many, alternating between two streams inside a group and meeting at
  // its end (0 or 1 = every launch waits for the one before, as a single stream does)
  int overlap_group = 0;
  const TrainStreams* train = nullptr;
  // tree tops (walk_super): -1 = by the forest's mean step count (deep forests), 0 = never, 1 = always
  int tree_tops = -1;
  int prefetch = 1;   // 27-column rows: prefetch the next tile's rows into registers during a walk
  int coop_rows = 1;  // ... fetched by the wave together where a tile is made of runs of >= 4 consecutive rows
  int lds_pad = 0;    // experiment knob: extra LDS bytes per block, to lower occupancy
  // ring kernels: tiles per wave and launch (0 = one launch for the whole batch).  C360 step 24.28 ms at 16, 24.13 at 64,
  // 24.07 in one launch; the fused fields kernel 26.85 / 26.65 / 27.05 (profiles/r04_sweeps.txt)
  int ring_rounds = 64;
  // OH Run1 in ranges of j walked one after the other, the next one's feature engineering and the last one's
  // post-processing beside the walk (capi.cpp run1_device): an experiment knob - 0 and 1 = one piece (the default; pieces
  // measured slower), n > 1 = n pieces
  int run1_pieces = 0;
  // ring kernels: CUs left free (a ring block owns its CU - all of its vector registers and LDS - for the length of a
  // launch, so a collective's kernels enqueued beside it only get on the chip at a launch boundary; 0 = take them all)
  int reserve_cus = 0;
  // small batches: trees split over several waves per tile (PredictArgs::leaf_buf): -1 = when the batch leaves half of
  // the chip's wave slots empty, 0 = never, n > 1 = always in n runs; needs the booster's leaf buffer
  int tree_split = -1;
  float* leaf_buf = nullptr;
  size_t leaf_words = 0;
  // rows with missing values go to a list and a second small launch (PredictArgs::defer_list) instead of slowing
  // their whole wave down for every tree: -1 = for batches of 262 144 rows and more, 0 = never, 1 = always;
  // needs the booster's list buffer (defer_words words: the count, then the list)
  int defer_missing = -1;
  // the rows with missing values are only counted (*defer_buf), none is listed: the host saw too many of them in the
  // last batch for the second launch to pay (capi.cpp)
  int defer_count_only = 0;
  uint32_t* defer_buf = nullptr;
  size_t defer_words = 0;
  // rows in no known order: -1 = decide per matrix (cluster unless the rows look ordered), 0 = never, 1 = always
  int cluster = -1;
  int cluster_trees = 2, cluster_steps = 7, cluster_zorder = 0;
  // rows API: the grid the rows were gathered from and the grid row the matrix starts at, filled per call from
  // the DMatrix (OHXDMatrixSetGrid, or the level size the library found).  0 = unknown: 64 consecutive rows.
  int grid_im = 0, grid_jm = 0;
  uint64_t grid_row0 = 0;
  // log2 extents of a wave's brick; -1 = chosen per call (TileShape::set_grid_auto); all 0 = no bricks
  int brick_li = -1, brick_lj = -1, brick_lk = -1;
  // lanes of a brick ordered level-fastest: the four lanes of a quad then share a column and with it the 2-D
  // features, and ask the L1 for fewer distinct blocks (33.2 against 33.5 ms on the C360 step)
  int brick_k_fastest = 1;
};

// 27 SoA fields of the MAPL state (OH_GridCompMod.F90:313-339), device pointers.
struct FieldsArgs {
  const float* field[32];
  uint32_t is2d_mask = 0;      // bit f set => field f is (im,jm), else (im,jm,km)
  uint32_t pl_feature = 1;     // feature index that is divided by 100 (Pa -> hPa), 0xFFFFFFFF for none
  uint32_t nfield = 0;
  int im = 0, jm = 0, km = 0, k1 = 0, k2 = 0;  // k1..k2 inclusive, 0-based
  // floats between a gridcell and the one above it in the 3-D arrays; 0 = im * jm.  Not 0: the (im, jm) given here is
  // a range of j of a wider grid - OH Run1 walks a big slab in such pieces so that the feature engineering of the next
  // and the post-processing of the last run beside the walk (capi.cpp run1_device) - and every field pointer (2-D ones
  // too) and `out` point at the piece's first column
  uint64_t level_stride = 0;
  int src_k0 = 0;              // level the 3-D source arrays start at (0 = whole arrays, k1 = slab-only copies)
  int out_k0 = 0;              // same for `out`
  float missing = 0.0f;
  uint32_t tree_begin = 0, tree_end = 0;
  int apply_pow10 = 1;         // OH_ML = 10**pred (OH_GridCompMod.F90:369)
  float scale = 1.0f;          // then * OHscale (OH_GridCompMod.F90:1569)
  float* out = nullptr;        // (im,jm,km) array; only levels k1..k2 are written
  float* margin_out = nullptr; // optional [N] raw margins in slab row order
  uint32_t* flags = nullptr;   // as PredictArgs::flags
  uint32_t train_id = 0, only_if_train = 0;      // as PredictArgs
  uint64_t tile_begin = 0, tile_end = 0;   // 64-row tiles of this launch
  int xcd_remap = 1;           // give each XCD a contiguous range of tiles
  TileShape shape;             // lanes -> gridcells of the slab
  // rows with missing values left to a second launch, as in PredictArgs: the list holds slab rows m
  uint32_t* defer_list = nullptr;
  uint32_t* defer_count = nullptr;
  uint32_t defer_cap = 0;
  const uint32_t* perm = nullptr;        // second launch: lane l of tile t takes slab row perm[64 t + l]
  const uint32_t* perm_count = nullptr;
  uint32_t perm_slots = 0;
  // small slabs: the trees split over waves, as PredictArgs::leaf_buf / tree_split
  float* leaf_buf = nullptr;
  uint32_t tree_split = 1;
};

// OH Run1's feature engineering and post-processing (include/ohxgb.h part 3), device pointers.
struct PrepArgs {
  int im = 0, jm = 0, km = 0;
  uint64_t col0 = 0, ncols = 0;      // columns [col0, col0 + ncols) of the im * jm only (ncols 0 = all): Run1's pieces
  int beside_a_walk = 0;             // the kernels must fit in what a ring block leaves of a CU (64 VGPRs a wave, 4 KB of LDS)
  const float *ple_bst = nullptr, *zle_bst = nullptr, *tauclw = nullptr, *taucli = nullptr;
  const float* sca[7] = {};
  const float *gmito3 = nullptr, *gmitto3 = nullptr;
  float *pl_bst = nullptr, *tauclwdn = nullptr, *tauclidn = nullptr, *taucliup = nullptr, *tauclwup = nullptr;
  float *aodup = nullptr, *aoddn = nullptr, *strato3 = nullptr;
};

struct SlabArgs {
  int im = 0, jm = 0, km = 0;
  int dynamic_k_range = 1;
  float tropp_min = 4000.0f;
  const float *ple_mod = nullptr, *tropp = nullptr;
  int32_t* result = nullptr;   // [0] = ksubcount (max over columns), [1] = COUNT(tropp <= tropp_min)
};

struct PostArgs {
  int im = 0, jm = 0, km = 0;
  uint64_t col0 = 0, ncols = 0;      // as PrepArgs
  float avogad = 0, runiv = 0, epsilon = 0;
  const float *ple_mod = nullptr, *t_mod = nullptr, *q_mod = nullptr, *tropp = nullptr;
  const float *default_oh = nullptr, *oh_ml = nullptr;
  float *oh = nullptr, *ndwet = nullptr;
};

// OH Run1's solar geometry (OH_GridCompMod.F90:401-466, 1444): 2-D, device pointers.
struct SolarArgs {
  int im = 0, jm = 0, jday = 0;
  float deg2rad = 0, rad2deg = 0;          // MAPL_DEGREES_TO_RADIANS, MAPL_RADIANS_TO_DEGREES
  const float *lats = nullptr, *lons = nullptr;   // radians
  float *lat_deg = nullptr, *sza_noon = nullptr;  // either may be null
};
hipError_t launch_solar_geometry(const SolarArgs& a, hipStream_t stream);

hipError_t launch_feature_prep(const PrepArgs& a, float* aod_scratch, hipStream_t stream);  // aod_scratch: (im,jm,km)
hipError_t launch_k_slab(const SlabArgs& a, hipStream_t stream);
hipError_t launch_post_process(const PostArgs& a, hipStream_t stream);

// The clustering pass in front of the walk for rows in no known order (kernels.hip).  Device pointers.
struct ClusterArgs {
  const float* rows = nullptr;
  uint64_t nrow = 0;
  uint32_t ncol = 0;
  float missing = 0.0f;
  uint32_t ntrees = 2, nsteps = 7;   // key = top of the first `ntrees` (<= 4) trees, `nsteps` super-node steps each (<= 32 bits)
  uint32_t zorder = 0;               // interleave the trees' decisions step by step instead of tree after tree
  uint32_t* keys = nullptr;          // [nrow]
  uint32_t* vals = nullptr;          // [nrow] the row numbers 0 .. nrow-1, to be sorted along with the keys
  uint32_t* agree = nullptr;         // nine words, zero before: [0] rows whose first three decisions (tree 0) equal the
                                     // previous row's, [1..8] rows per outcome of those three decisions
};
uint32_t cluster_key_bits(const ClusterArgs& a);
hipError_t launch_cluster_keys(const DeviceForest& forest, const ClusterArgs& a, int num_cus, hipStream_t stream);
// (key, value) pairs sorted by the low `key_bits` bits of the key.  Called with temp == nullptr it only reports
// the scratch bytes needed in *temp_bytes.  *sorted_vals = whichever of vals_a / vals_b holds the result.
hipError_t sort_pairs_u32(void* temp, size_t* temp_bytes, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a,
                          uint32_t* vals_b, uint64_t n, unsigned key_bits, hipStream_t stream, uint32_t** sorted_vals);

// Ring: super-nodes, the records of a walk's first four steps resident in LDS (predict_rows_ring_kernel /
// predict_fields_ring_kernel); falls back to Super2 for what those kernels do not take (small batches, other shapes)
enum class KernelKind { Wide, Packed1, Packed2, Packed4, Super1, Super2, Super3, Super4, Ring };

const char* kernel_kind_name(KernelKind k);
// the __global__ launch_predict would launch for rows of `ncol` columns (for profiles and bench.py)
std::string predict_kernel_symbol(KernelKind kind, const DeviceForest& forest, uint32_t ncol, const LaunchTuning& tune);
// ... and every kernel a predict on the batch `a` launches, in order, joined by " + "
std::string predict_kernel_symbols_rows(KernelKind kind, const DeviceForest& fr, const PredictArgs& a, int num_cus,
                                        const LaunchTuning& tune);

hipError_t launch_predict(KernelKind kind, const DeviceForest& forest, const PredictArgs& a, int num_cus,
                          hipStream_t stream, const LaunchTuning& tune = LaunchTuning());
hipError_t launch_predict_fields(KernelKind kind, const DeviceForest& forest, const FieldsArgs& a, int num_cus,
                                 hipStream_t stream, const LaunchTuning& tune = LaunchTuning());
// up to kCopyListMax arrays moved by one launch (kernels.hip copy_arrays_kernel); pointers the device can reach
constexpr uint32_t kCopyListMax = 48;
struct CopyList {
  const float* src[kCopyListMax];
  float* dst[kCopyListMax];
  uint64_t n[kCopyListMax];     // floats
  uint32_t count = 0;
};
hipError_t launch_copy_arrays(const CopyList& l, hipStream_t stream, uint32_t max_blocks = 0);

hipError_t launch_scan_dense(const float* data, uint64_t count, float missing, uint32_t* flags, hipStream_t stream);
// Level-size search (capi.cpp infer_level_size): block (x, y) tests whether column cols.col[y] repeats with
// period nrow / (kmax - x); d_verdict[y * (kmax - 1) + x] must be zero before and is 0 (periodic), 1 (not) or
// 2 (kmax - x does not divide nrow) afterwards.  All device pointers.
struct PeriodColumns {
  uint32_t col[4];
};
hipError_t launch_detect_period(const float* data, uint64_t nrow, uint32_t ncol, const PeriodColumns& cols,
                                uint32_t ncols, uint32_t kmax, uint32_t* d_verdict, hipStream_t stream);

}  // namespace ohx
