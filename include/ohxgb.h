/*
 * ohxgb.h — C ABI of libohxgb.so, the MI355X (gfx950) OH-chemistry predictor.
 *
 * Drop-in boundary: GEOS-ESM/QuickChem reaches XGBoost through the Fortran
 * module xgb_fortran_api, which binds eleven XGBoost C-API symbols with
 * ISO_C_BINDING.  libohxgb.so exports those same symbols, with the C signatures
 * the bindings imply, so QuickChem links against it instead of libxgboost 1.6.0
 * (reference Shared/CMakeLists.txt:8-12) without touching a line of Fortran.
 * Each declaration below cites the reference binding it replaces
 * (paths relative to the QuickChem tree).
 *
 * Everything computes on the GPU.  There is no CPU fallback: a compute call
 * made where no HIP device is usable returns -1 and XGBGetLastError() says so.
 *
 * Conventions (xgboost 1.6.0 c_api.h): every function returns 0 on success and
 * -1 on failure; the message is then available, per thread, from
 * XGBGetLastError().  Handles are opaque pointers created by the library and
 * released by the matching *Free call.
 */
#ifndef OHXGB_H_
#define OHXGB_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* DMatrixHandle;
typedef void* BoosterHandle;
typedef uint64_t bst_ulong;

/* ------------------------------------------------------------------------
 * Part 1 — the XGBoost C-API subset QuickChem binds
 * ------------------------------------------------------------------------ */

/* Last error message of the calling thread.  Not bound by the reference (it
 * only asserts rc == 0, OH_GridComp/OH_GridCompMod.F90:252-265,353-378). */
const char* XGBGetLastError(void);

/* Shared/xgb_fortran_api.F90:86-94, called at OH_GridCompMod.F90:251 (1x27
 * dummy) and :347 (the N x 27 batch).  `data` is row-major [nrow][ncol]
 * (Fortran xx_carr(27,N)); it is copied to HBM before the call returns, the
 * caller may free it at once.  Entries equal to `missing`, or NaN, are missing
 * values.  As in xgboost 1.6.0, +-inf in the data is an error unless `missing`
 * is itself infinite. */
int XGDMatrixCreateFromMat(const float* data, bst_ulong nrow, bst_ulong ncol, float missing, DMatrixHandle* out);

/* Shared/xgb_fortran_api.F90:45-48, called at OH_GridCompMod.F90:264,377. */
int XGDMatrixFree(DMatrixHandle handle);

/* Shared/xgb_fortran_api.F90:98-103 and :107-112 (bound, unused by OH). */
int XGDMatrixNumRow(DMatrixHandle handle, bst_ulong* out);
int XGDMatrixNumCol(DMatrixHandle handle, bst_ulong* out);

/* Shared/xgb_fortran_api.F90:35-40 and :53-58 (bound, unused by OH).
 * SaveBinary writes this library's own dense container ("OHXDMAT1");
 * CreateFromFile reads that container, or CSV text when the name ends in
 * ".csv" or carries "?format=csv" (all columns are features). */
int XGDMatrixSaveBinary(DMatrixHandle handle, const char* fname, int silent);
int XGDMatrixCreateFromFile(const char* fname, int silent, DMatrixHandle* out);

/* Shared/xgb_fortran_api.F90:76-82, called at OH_GridCompMod.F90:256.  The
 * reference passes ONE handle by value with len == 0 ("setting this to 27
 * results in a Seg Fault", :255): `dmats` is never dereferenced when len == 0,
 * and cached matrices are not needed for prediction in any case. */
int XGBoosterCreate(const DMatrixHandle dmats[], bst_ulong len, BoosterHandle* out);

/* Shared/xgb_fortran_api.F90:116-119 (the reference leaves it commented out,
 * OH_GridCompMod.F90:389-392). */
int XGBoosterFree(BoosterHandle handle);

/* Shared/xgb_fortran_api.F90:19-23, called at OH_GridCompMod.F90:261.  Format
 * by extension as in xgboost 1.6.0: ".json" JSON, ".ubj" UBJSON (draft 12),
 * anything else the legacy binary format of the production ".model"/".bin"
 * files (OH_GridComp/OH_instance_OH.rc:17-20). */
int XGBoosterLoadModel(BoosterHandle handle, const char* fname);

/* Shared/xgb_fortran_api.F90:27-31 (bound, unused by OH). */
int XGBoosterSaveModel(BoosterHandle handle, const char* fname);

/* Same loader from memory (xgboost c_api.h; not bound by the reference). */
int XGBoosterLoadModelFromBuffer(BoosterHandle handle, const void* buf, bst_ulong len);

/* Shared/xgb_fortran_api.F90:62-72, called at OH_GridCompMod.F90:356 with
 * option_mask = 0, ntree_limit = 0, training = 0 (:231-235).
 *   option_mask: 0 normal, 1 output margin, 16 leaf indices; others refused.
 *   ntree_limit: 0 = all trees.
 * *out_result points at a host buffer owned by the booster, valid until the
 * next predict on that booster or XGBoosterFree (the reference never frees
 * it, :362,381).  *out_len = nrow (asserted at :359). */
int XGBoosterPredict(BoosterHandle handle, DMatrixHandle dmat, int option_mask, unsigned ntree_limit, int training,
                     bst_ulong* out_len, const float** out_result);

/* xgboost c_api.h; not bound by the reference.  Understood names:
 *   "ohx_kernel"      auto | ring | super1 | super2 | super3 | super4 | packed1 | packed2 |
 *                     packed4 | wide : node format and trees in flight per lane.  ring: super-nodes, the records
 *                     of a walk's first four steps resident in LDS, 16 wavefronts per block walking the same four
 *                     trees at a time (the big batches of a 27-feature booster; everything else of such a booster
 *                     goes the super2 way).  auto = ring for 27-feature boosters of 5 or more steps per tree
 *                     (where it is faster: 20 % at the OH booster's 9 steps), else super2
 *   "ohx_ring_rounds" ring kernels: tiles per wavefront and launch (default 64; 0 = one launch);
 *                     at most 16 for rows not known to lie on a grid, 4 for rows in no order (clustering pass)
 *   "ohx_reserve_cus" ring kernels (rows): compute units left free, 0..128 (default 0).  A ring block owns its CU for the
 *                     length of a launch; a collective enqueued beside the predict (OHXAllGatherOH, torch.distributed)
 *                     otherwise only gets on the chip at a launch boundary
 *   "ohx_run1_pieces" experiment knob of OHXBoosterRun1[Device]: n > 1 walks n ranges of j one after the other, with the
 *                     feature engineering of the next and the post-processing of the last on a second stream beside
 *                     the walk; 0 and 1 = one piece (default: measured, pieces are slower - profiles/r05_sweeps.txt)
 *   "ohx_copy_engine" kernel (default) | dma | auto, process-wide (the handle may be NULL): how REGISTERED host arrays
 *                     cross PCIe.  kernel = a list of arrays per launch of a copy kernel (the GPU reads / writes the
 *                     caller's memory itself; the one list that crosses under the walk by DMA); dma = every array by
 *                     hipMemcpyAsync (the DMA engines: a fixed price per array, no wave on any CU); auto = the host form
 *                     of OHXBoosterRun1 times its own first ticks both ways (two of warm-up, then eight of each) and keeps
 *                     the faster, trying again every 512 ticks.  Measured at 1, 2, 3 and 6 ranks on a card
 *                     (profiles/r06_ranks_per_gpu_block_48x24_engines.json): the kernels win at every count (a rank's tick
 *                     0.29 against 0.61 ms alone, 1.35 against 3.13 ms at six), auto says so every time, and its trial
 *                     costs the tail - hence not the default.  The same bits either way.
 *   "ohx_register_host"  0 | 1, process-wide (the handle may be NULL): the host arrays handed to OHXBoosterRun1,
 *                     OHXOHPostProcess and OHXBoosterPredictFields are registered with the GPU driver the first time
 *                     they are seen and moved by DMA - a rank-sized block's forty arrays by ONE copy launch - from
 *                     then on (a 48 x 24 x 72 block's Run1 tick: 0.98 -> 0.52 ms; six ranks sharing the GPU: 3.3 ->
 *                     0.7 ms).  A CONTRACT: every array passed while this is on must stay allocated until
 *                     OHXUnregisterHost(array), OHXReleaseScratch() or the end of the process (MAPL's state arrays
 *                     do); an array that is freed while registered is reached through a registration the library
 *                     cannot check - ROCm's follows the process's page tables, so the next copy faults rather than
 *                     reads stale pages, but it fails.  Nothing is unregistered under a copy in flight (the device
 *                     is synchronised first).  Default 0 (the OH shell turns it on: register_host_arrays, default T)
 *   "ohx_copy_blocks" process-wide: blocks per launch of the kernel that moves registered host arrays (default 64; 0 = a
 *                     block per KiB).  Small on purpose: a chip full of wavefronts waiting on PCIe moves a rank's arrays
 *                     slower and keeps every other stream's kernels from starting meanwhile.
 *                     Launches that WRITE the caller's arrays (a tick's results, at its end) use 512 blocks
 *                     (OHX_COPY_BACK_BLOCKS in the environment, read once; 0 = as "ohx_copy_blocks")
 *   "ohx_tree_tops"   auto | on | off : super-nodes: fetch a tree's first records with one coalesced load per
 *                     wavefront (auto = forests of 7 or more steps per tree, where it is faster)
 *   "ohx_cluster"     auto | on | off : group rows of no known order by the decisions they take at the top of
 *                     the first trees before walking them ("ohx_cluster_trees", "ohx_cluster_steps",
 *                     "ohx_cluster_zorder" shape the key)
 *   "ohx_tree_split"  auto | off | 2..10 : a batch that leaves half of the chip's wave slots empty has its trees cut
 *                     into runs walked by different wavefronts, the leaves summed in tree order by a second launch
 *                     (a predict on 10 000 .. 300 000 rows takes a third of the time; margins unchanged, bit for bit)
 *   "ohx_defer_missing"  auto | on | off : rows that hold missing values are predicted by a second, small launch
 *                     instead of putting their whole wavefront on the missing-aware walk (auto = batches of 262 144
 *                     rows and more)
 *   "ohx_launches_per_residency"  tiles per wave per launch (default 2; 0 = one launch)
 *   "ohx_brick" = "a,b,c", "ohx_brick_k_fastest", "ohx_prefetch", "ohx_coop_rows", "ohx_xcd_remap", "ohx_lds_pad", "ohx_overlap_group"
 *                     launch-shape knobs behind profiles/ *_sweeps.txt; the defaults are the measured best
 *   "ohx_top_levels", "ohx_line_slots", "ohx_min_chunk"  placement of the packed format
 *   "ohx_device"      HIP device ordinal for this booster
 * None of them changes a prediction.
 * xgboost's own parameter names ("nthread", "predictor", ...) are accepted and
 * ignored. */
int XGBoosterSetParam(BoosterHandle handle, const char* name, const char* value);

/* ------------------------------------------------------------------------
 * Part 2 — device-resident and fused entry points (additive)
 * ------------------------------------------------------------------------ */

/* HIP device count, or -1 with an error message when HIP is unusable. */
int OHXDeviceCount(int* out);

/* As XGDMatrixCreateFromMat, but `d_data` already lives in HBM.  The matrix
 * BORROWS the pointer (no copy); it must stay valid until XGDMatrixFree.
 * The inf check of the host path is folded into the predict kernels instead:
 * a predict on data holding +-inf fails (OHXBoosterCheck reports it).
 * Ordering: the library reads the rows on streams of its own.  The host-form
 * calls on such a matrix (XGBoosterPredict, XGDMatrixSaveBinary) first wait for
 * everything enqueued so far on the legacy default stream, hence on every
 * blocking stream; rows filled on a NON-blocking stream of the caller's must
 * be synchronised by the caller, or the *Device forms used, which take the
 * producing stream as an argument. */
int OHXDMatrixCreateFromDevice(const float* d_data, bst_ulong nrow, bst_ulong ncol, float missing, DMatrixHandle* out);

/* Optional hint, any DMatrix: its rows are rows row0, row0+1, ... of the gather
 * m = (i-1) + im*((j-1) + jm*(k-k1)) that predict_OH_with_XGB builds
 * (OH_GridCompMod.F90:309-345); row0 > 0 for a rank's contiguous shard.  Predictions do not
 * change.  The kernels then give each wavefront a brick of 4x4x4 (or 8x4x2, 8x8x1) neighbouring
 * gridcells instead of 64 consecutive rows, which measures 1.19x faster on C360 L72 because
 * neighbours in all three directions walk the same tree nodes.  im = jm = 0 says "no grid": 64 consecutive
 * rows per wavefront, and the library does not look for a level size either. */
int OHXDMatrixSetGrid(DMatrixHandle handle, int im, int jm, bst_ulong row0);

/* What the library knows about the rows of a DMatrix (any pointer may be NULL).  Without a hint the
 * library looks for the level size by itself, ONCE per matrix, at the first predict on it: the reference's
 * gather stacks levels and its first column, LAT, is a 2-D field (OH_GridCompMod.F90:313), so that column
 * repeats bit for bit with period im*jm.  A period found that way is reported as im = level size, jm = 1,
 * inferred = 1 and is worth as much as the full hint to within 1 % (runs of 8 cells x 8 levels per
 * wavefront).  For a matrix the library copied itself (XGDMatrixCreateFromMat) this call looks if nobody
 * has yet; for a matrix over device memory it reports what is known so far. */
int OHXDMatrixGetGrid(DMatrixHandle handle, int* im, int* jm, bst_ulong* row0, int* inferred);

/* The same search on demand.  Waits for `stream` (the rows must be there), then looks; *found (may be
 * NULL) says whether a level size was found.  Replaces any earlier hint.  A device-resident caller that
 * wants OHXBoosterPredictDevice never to wait calls this (or OHXDMatrixSetGrid) beforehand: the first
 * predict on a matrix nobody has described waits for its stream once to look.  For matrices the library copied
 * itself (XGDMatrixCreateFromMat) that is once per SHAPE: what the first predict finds is remembered by (rows,
 * columns) for the life of the process, so a host that creates and frees its matrix every tick, as the reference
 * does (OH_GridCompMod.F90:347,377), pays the search and the wait at the first tick only.  This call and
 * OHXDMatrixGetGrid always look. */
int OHXDMatrixInferGrid(DMatrixHandle handle, void* stream, int* found);

/* Predict straight into device memory: d_out[nrow] margins (or [nrow][ntree]
 * leaf ids with option_mask 16).  `stream` is a hipStream_t (NULL = default
 * stream); the call only enqueues work (but see OHXDMatrixInferGrid for the first
 * predict on an undescribed matrix).  OHXBoosterCheck surfaces errors the
 * kernels raised (inf in the input; without it a device form's caller never
 * learns of them).  A booster keeps single scratch buffers
 * (error flags, Run1 intermediates, staging): calls on ONE booster must not run
 * concurrently on two streams or threads; different boosters are independent.
 * hipGraphs: this call and OHXBoosterPredictFieldsDevice may be made on a stream
 * that is being captured - what they enqueue is launches, memsets and copies on
 * `stream` - once ONE plain call of the same shape has been made on the booster
 * and the matrix (it allocates the booster's buffers, uploads the model and looks
 * at the matrix).  A capture that would have to allocate or to wait returns -1
 * with a message that says so and enqueues nothing.  While capturing the library
 * leaves out what its host side does beside the launches to adapt the NEXT call
 * (the read-back of how many rows went to the second launch): a replay walks the
 * way the last plain call decided, and is bit-identical to a plain call on the
 * same contents (tests/test_gpu_graph.py).  OHXBoosterRun1Device is not
 * capturable: it waits for its slab count. */
int OHXBoosterPredictDevice(BoosterHandle handle, DMatrixHandle dmat, int option_mask, unsigned ntree_limit,
                            float* d_out, void* stream);
int OHXBoosterCheck(BoosterHandle handle, void* stream);

/* The whole of predict_OH_with_XGB's RUN section in one kernel
 * (OH_GridCompMod.F90:303-383): gathers the 27 MAPL fields in place (field f is
 * (im,jm,km) Fortran order, or (im,jm) when is2d[f] != 0; feature order of
 * :313-339), divides field `pl_feature` by 100 (Pa -> hPa, :314; pass -1 for
 * none), predicts rows m = i + im*(j + jm*(k-k1)) for k = k1..k2 (1-based,
 * inclusive, the slab of :300-301), and stores
 *   oh_ml(i,j,k) = 10**pred * ohscale      (:369 and :1569)
 * leaving the other levels of oh_ml untouched.  apply_pow10 = 0 stores the raw
 * margin times ohscale instead.  margin (optional, may be NULL) receives the raw
 * xx_pred(m).  Host-pointer form: stages through HBM and returns when oh_ml is
 * complete.  Device form: all pointers are device pointers, work is enqueued. */
int OHXBoosterPredictFields(BoosterHandle handle, const float* const fields[], const int32_t is2d[], int nfield,
                            int pl_feature, int im, int jm, int km, int k1, int k2, float missing, int apply_pow10,
                            float ohscale, float* oh_ml, float* margin);
int OHXBoosterPredictFieldsDevice(BoosterHandle handle, const float* const d_fields[], const int32_t is2d[],
                                  int nfield, int pl_feature, int im, int jm, int km, int k1, int k2, float missing,
                                  int apply_pow10, float ohscale, float* d_oh_ml, float* d_margin, void* stream);

/* ------------------------------------------------------------------------
 * Part 3 — the steps either side of the predict call (additive)
 * ------------------------------------------------------------------------
 * OHXBoosterRun1 does the arithmetic of OH Run1 from the imports to the INTERNAL
 * field OH (OH_GridComp/OH_GridCompMod.F90:1240-1257, 1444-1478, 1488, 1557-1595)
 * in HBM, so the engineered features never round-trip to the host:
 *   PL_MOD = (PLE_MOD(k-1)+PLE_MOD(k))*0.5, TV_MOD, NDWET_MOD            (:1247-1257)
 *   stratO3 = GMITO3 - GMITTO3                                           (:1446)
 *   gridBoxThickness = ZLE(k-1)-ZLE(k); aod = thickness * (BC+OC+BR+DU+SU+SS+NI)   (:1451-1458)
 *   tauclwDN/taucliDN/aodDN(k) = SUM(x(k:km)), taucliUP/tauclwUP/aodUP(k) = SUM(x(1:k)),
 *       each sum accumulated from zero in ascending level order          (:1468-1478)
 *   PL_BST = (PLE_BST(k-1)+PLE_BST(k))*0.5                               (:1488)
 *   the k-slab, predict_OH_with_XGB, OH_ML *= OHscale                    (:1559-1569)
 *   OH = PL_MOD > TROPP ? OH_ML : default_OH ; OH = (OH*NDWET_MOD)*1.0e-6   (:1579-1595)
 * What stays with the caller: the choice of import per OH_data_source.  LAT in degrees and the
 * local-noon SZA are 2-D inputs of the struct; OHXSolarGeometry below computes them the reference's
 * way for a caller who wants that on the GPU too.
 * All arrays are Fortran order: 3-D (im,jm,km), edge fields (im,jm,0:km), 2-D
 * (im,jm).  MAPL's constants are passed in, not restated.  Host form stages
 * through HBM and returns when the outputs are complete; device form takes device
 * pointers and enqueues (it waits once, for the slab count, which runs on the
 * library's second stream beside the feature kernels).
 * Host form: the arrays cross PCIe in the order the tick needs them - PLE and TROPP of
 * the model (the slab count), the feature engineering's inputs, then what only the walk
 * reads (of the sixteen 3-D fields among those, the slab's levels only), last and under
 * the walk what the mask and the unit conversion read.  The same array may be passed
 * for several members (ONLINE_INST: T is t_mod and t_bst): it crosses once.  With
 * "ohx_register_host" every list is one launch of a small copy kernel.
 * Streams: the library owns two per device for the life of the process - one for kernels, one
 * for copies - and nothing of it runs on the null stream; every stream a process has used
 * is a hardware queue, and a GPU that several ranks share time-slices queues once they
 * outnumber its slots.  Experiment knobs in the environment, read
 * once: OHX_RUN1_STREAMS=1 (copies on the kernels' stream), OHX_RUN1_GATE=1 (a kernel
 * is launched when this thread has seen its inputs arrive, not enqueued behind a wait for
 * them), OHX_RUN1_SLAB_IN_PLACE=1 (the slab count reads the caller's registered PLE and
 * TROPP over PCIe); both measured worth nothing, both off. */
typedef struct OHXRun1Args {
  int32_t im, jm, km;
  int32_t dynamic_k_range;           /* .NOT. compute_once_per_day (:1561) */
  float tropp_min;                   /* Pa, 4000.0 (:1563) */
  float ohscale;                     /* :1569 */
  float missing;                     /* -999.0 (:213) */
  float avogad, runiv, epsilon;      /* MAPL_AVOGAD, MAPL_RUNIV, MAPL_EPSILON */
  /* the model's own state: slab, tropopause mask, number density */
  const float *ple_mod, *t_mod, *q_mod, *tropp_mod;
  /* inputs to the engineered features */
  const float *ple_bst, *zle_bst, *tauclw, *taucli;
  const float *scacoef[7];           /* BC OC BR DU SU SS NI at the chosen wavelength */
  const float *gmito3, *gmitto3;
  /* features used as they are (order of :313-339 where not engineered) */
  const float *lat_deg, *t_bst, *no2, *o3, *ch4, *co, *isop, *acet, *c2h6, *c3h8, *prpe, *alk4, *mp, *h2o2;
  const float *cloud, *qv, *albuv, *ch2o, *sza;
  const float *default_oh;           /* import oh_OH, mol/mol (:1548) */
  /* outputs */
  float *oh;                         /* INTERNAL OH, molec/cm3 */
  float *oh_boost;                   /* export OH_boost = OH_ML*OHscale (may be NULL) */
  float *ndwet;                      /* DIAG_NDWET (may be NULL) */
  int32_t *k1, *k2;                  /* HOST pointers, 1-based slab (may be NULL) */
  /* optional dumps of the engineered features - the reference's DIAG_* exports, its only in-model
   * debugging hook (:1607-1640, OH_StateSpecs.rc:41-73); each may be NULL.  (im,jm,km), strato3 (im,jm):
   * PL_BST (Pa), tauclwDN, taucliDN, taucliUP, tauclwUP, aodUP, aodDN, the layer aod, stratO3 */
  float *diag_pl_bst, *diag_tauclwdn, *diag_tauclidn, *diag_taucliup, *diag_tauclwup;
  float *diag_aodup, *diag_aoddn, *diag_aod, *diag_strato3;
} OHXRun1Args;

int OHXBoosterRun1(BoosterHandle handle, const OHXRun1Args* args);
int OHXBoosterRun1Device(BoosterHandle handle, const OHXRun1Args* args, void* stream);

/* The tail of OH Run1 alone, for a tick that does not call Boost (compute_once_per_day and nhms > 0,
 * OH_GridCompMod.F90:1189-1193): with oh_ml = the OH_ML*OHscale kept from the last Boost,
 *   PL_MOD, TV_MOD, NDWET_MOD as above (:1247-1257);
 *   OH = PL_MOD > TROPP ? oh_ml : default_OH ; OH = (OH*NDWET_MOD)*1.0e-6      (:1579-1595)
 * ndwet may be NULL.  Host form stages through HBM; device form takes device pointers and enqueues. */
int OHXOHPostProcess(int im, int jm, int km, float avogad, float runiv, float epsilon, const float* ple_mod,
                     const float* t_mod, const float* q_mod, const float* tropp_mod, const float* default_oh,
                     const float* oh_ml, float* oh, float* ndwet);
int OHXOHPostProcessDevice(int im, int jm, int km, float avogad, float runiv, float epsilon, const float* d_ple_mod,
                           const float* d_t_mod, const float* d_q_mod, const float* d_tropp_mod,
                           const float* d_default_oh, const float* d_oh_ml, float* d_oh, float* d_ndwet, void* stream);

/* The solar geometry of OH Run1.  OHXJulianDay: JulianDay(nymd) with the reference's leap_year
 * (OH_GridCompMod.F90:1905-1970; host integer arithmetic).  OHXSolarGeometry: latarr =
 * LATS*MAPL_RADIANS_TO_DEGREES (:1444) and sza_noon = computeSolarZenithAngle_LocalNoon(jday, LATS,
 * LONS) (:401-466, 1481-1482) for (im,jm) arrays in radians; either output may be NULL.  float32 in
 * the reference's order of evaluation; sin/asin/cos/acos are the device library's, so the result
 * agrees with a host run to a few ulp of cos(zenith) - which acos turns into up to ~0.05 degree
 * where the sun is overhead.  SZA feeds tree splits: a caller that needs OH bit-identical to a CPU
 * run passes its own sza to OHXBoosterRun1 instead. */
int OHXJulianDay(int nymd, int* jday);
int OHXSolarGeometry(int jday, const float* lats, const float* lons, int im, int jm, float degrees_to_radians,
                     float radians_to_degrees, float* lat_deg, float* sza_noon);
int OHXSolarGeometryDevice(int jday, const float* d_lats, const float* d_lons, int im, int jm, float degrees_to_radians,
                           float radians_to_degrees, float* d_lat_deg, float* d_sza_noon, void* stream);

/* Model facts for roofline accounting: info[0] trees, [1] nodes in the model,
 * [2] node slots in HBM, [3] bytes of the node array the selected kernel reads,
 * [4] max depth, [5] features, [6] node format in use (0 wide, 1 packed, 2 super-nodes),
 * [7] vector-memory instructions one wavefront issues to walk the whole forest once (super-nodes). */
int OHXBoosterGetInfo(BoosterHandle handle, bst_ulong info[8]);
/* Name of the GPU kernel XGBoosterPredict / OHXBoosterPredictDevice launch for rows of `ncol` columns with the
 * booster's current parameters, as a profiler prints it (without namespaces and arguments), e.g.
 * "predict_rows_tile_kernel<2,2,true,true>".  *out stays valid until the next call on this handle. */
int OHXBoosterKernelSymbol(BoosterHandle handle, bst_ulong ncol, const char** out);
/* Every GPU kernel a margin predict on THIS matrix launches with the booster's current parameters, in launch order,
 * joined by " + " - the size of the batch decides (a small one has its trees split over waves and a second launch that
 * sums the leaves in tree order; a big one of the OH booster goes through the ring kernel and the launches behind it).
 * Decided by the code that launches (kernels.hip plan_rows).  Not listed: the clustering pass in front of rows that
 * are in no known order, the level-size search of a matrix nobody described.  *out as above. */
int OHXBoosterKernelSymbolRows(BoosterHandle handle, DMatrixHandle dmat, const char** out);
/* How often a block of the ring kernels (the default for the OH booster's big batches) gave up waiting for another
 * since the booster's forest went to the GPU.  Never seen in 6 700 whole-batch calls (tools/ring_soak.py), and not an
 * error: every ring launch train is followed, on the same stream, by a launch of the tile kernel that only runs when a
 * block gave up and then predicts the train's rows again - same bits, the host is not involved, so the device forms
 * (OHXBoosterPredictDevice, ...FieldsDevice, Run1Device) are covered without OHXBoosterCheck.  The first sighting is
 * said on stderr by the next call that reads the flags back.  Waits for `stream`.  (The reference asserts rc == 0 on
 * XGBoosterPredict, OH_GridComp/OH_GridCompMod.F90:356-358: a time-out must not end a model run.) */
int OHXBoosterRingReruns(BoosterHandle handle, void* stream, bst_ulong* out);
/* What "ohx_copy_engine" = auto (when asked for) has decided for OHXBoosterRun1's host form on this booster: *choice = -1 while it is
 * still trying (or when auto is not in charge: the engine was set, or the arrays are not registered), 0 = copy kernels,
 * 1 = DMA; *trials = how many trials have ended, *picked_dma = how many of them picked DMA.  Any of the three may be NULL. */
int OHXBoosterCopyEngineChoice(BoosterHandle handle, int* choice, unsigned* trials, unsigned* picked_dma);

/* ------------------------------------------------------------------------
 * Part 4 — reassembling the OH field across the GPUs of a node (additive)
 * ------------------------------------------------------------------------
 * Inside GEOS nothing is exchanged: every rank keeps the block it predicted
 * (OH_GridCompMod.F90:1199-1202, 1565).  For a caller that wants the whole field on every GPU
 * (BASELINE.json configs #4/#5) the gridcell rows are cut into contiguous shards, rank r of N holding
 * OHXShardRows' rows, and ONE collective - an RCCL all-gather over xGMI - puts every shard at its rows of
 * d_full on every rank.  Set-up as RCCL's own: rank 0 calls OHXCommGetUniqueId, the host distributes the
 * OHX_UNIQUE_ID_BYTES bytes by whatever it has (MPI_Bcast in a GEOS-like host), every rank - its HIP device
 * already current - calls OHXCommInitRank.  OHXAllGatherOH only enqueues on `stream`; d_shard may be
 * d_full + row0 (in place).  Equal shards are one ncclAllGather; ragged ones - and equal ones when the
 * environment said OHX_ALLGATHER=pairs at OHXCommInitRank (read once, there; every rank must be started with the
 * same setting) - the direct exchange of SURVEY.md §8e: one group of ncclSend / ncclRecv
 * between all pairs of ranks, every shard travelling its own xGMI link.  The communicator belongs to the HIP
 * device that was current at OHXCommInitRank: a call with another device current is refused.
 * librccl.so is loaded at the first of these calls, not linked (nor is its header needed to build). */
typedef void* OHXCommHandle;
#define OHX_UNIQUE_ID_BYTES 128
int OHXCommGetUniqueId(void* id);
int OHXCommInitRank(const void* id, int nranks, int rank, OHXCommHandle* out);
int OHXCommFree(OHXCommHandle comm);
/* RCCL's version code (ncclGetVersion), for a benchmark's record */
int OHXCommInfo(int* rccl_version);
/* rows [*row0, *row0 + *nrows) of rank `rank`: contiguous, sizes differing by at most one row */
int OHXShardRows(bst_ulong nrows_total, int nranks, int rank, bst_ulong* row0, bst_ulong* nrows);
int OHXAllGatherOH(OHXCommHandle comm, const float* d_shard, bst_ulong nrows_local, bst_ulong nrows_total,
                   float* d_full, void* stream);

/* An array that was handed over while "ohx_register_host" was on is about to be freed or reallocated: its registration
 * goes (after the device has been synchronised).  An array the library never registered is not an error. */
int OHXUnregisterHost(const void* array);

/* Returns the device buffers the library keeps between calls to the driver: freed DMatrix storage parked
 * for the next XGDMatrixCreateFromMat (the reference creates and frees its matrix on every OH tick,
 * OH_GridCompMod.F90:347,377; at most two buffers are kept; OHX_DMATRIX_POOL=0 in the environment keeps
 * none), and drops every host registration ("ohx_register_host").  Live handles are not touched. */
int OHXReleaseScratch(void);

#ifdef __cplusplus
}
#endif
#endif /* OHXGB_H_ */
