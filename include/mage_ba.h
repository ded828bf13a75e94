/*
 * mage_ba.h -- C ABI of the MI355X bundle-adjustment back-end (libmageslam_hip.so).
 *
 * Drop-in boundary for the reference's BundlerLib facade:
 *     Dependencies/BundlerLib/Include/BundlerLib.h:20-66   (class mage::BundlerLib)
 *     Dependencies/BundlerLib/Source/BundlerLib.cpp        (behaviour; cited per entry point)
 * Every entry point below replaces exactly one BundlerLib method; include/BundlerLib.h is the C++
 * shim with the reference's class name and method names that forwards to these symbols, and
 * INTEGRATION.md shows the binding a MAGE-SLAM maintainer would add.
 *
 * Conventions (same as the reference): poses are world->camera; `R_colmajor` is a 3x3 rotation in
 * column-major order (Eigen::Map<const Matrix3f>); intrinsics are (cx, cy, fx, fy) and -- like the
 * reference, BundlerLib.cpp:266 -- only fx, cx, cy are used.  All inputs are float32 and are widened
 * to float64 on entry; all arithmetic on the device is float64.
 *
 * Error behaviour: the reference has no error channel (asserts, gsl::narrow throws).  Here every
 * function returns a mage_status; nothing throws, nothing aborts.  A handle is confined to one
 * host thread at a time; distinct handles may be used concurrently (each owns a HIP stream).
 */
#ifndef MAGE_BA_H
#define MAGE_BA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mage_status {
    MAGE_OK = 0,
    MAGE_ERR_INVALID_ARGUMENT = 1,   /* null handle/pointer, index out of range, allocate-twice */
    MAGE_ERR_OUT_OF_MEMORY = 2,
    MAGE_ERR_DEVICE = 3,             /* HIP runtime error; see mage_last_error() */
    MAGE_ERR_UNSUPPORTED = 4,        /* part of the surface that is not built (none of the BundlerLib surface at present; ORB/match variants) */
    MAGE_ERR_NO_DEVICE = 5           /* no gfx950 device visible: the HIP path never falls back to a CPU */
} mage_status;

/* Text of the last error raised on the calling thread ("" if none). */
const char* mage_last_error(void);

/* DIAGNOSTIC (tests only): the dense solver of the reduced camera system on its own.  Solves A x = b for a symmetric positive
 * definite A (column-major, lower triangle read, host memory) with the same tiled Cholesky + substitutions the bundle adjustment
 * runs; *ok = 0 when a pivot is not positive.  Lets the factorisation be checked against LAPACK / rocSOLVER. */
mage_status mage_debug_dense_solve(int device, int n, const double* A_colmajor, const double* b, double* x, int* ok);
mage_status mage_debug_dense_solve_skyline(int device, int n, const double* A_colmajor, const double* b, double* x, int* ok, const int* tile_env);

/* Device buffers of destroyed handles (BA, ORB, matcher) are parked per device and reused by the next handle, because the
 * reference creates and destroys a bundler per optimisation (BundleAdjust.cpp:348-351) and a fresh 0.8 GB allocation costs
 * ~25 ms.  The parked total is bounded by MAGE_DEVICE_CACHE_MB (default 4096, 0 = no caching); this call returns all of it
 * to the HIP runtime now. */
void mage_release_cached_memory(void);

typedef struct mage_ba mage_ba;

/* mage::BundlerParameters (BundlerLib.h:15-18) + device placement. */
typedef struct mage_ba_params {
    int are_points_fixed;   /* BundlerParameters::ArePointsFixed */
    int device;             /* HIP device ordinal; -1 = the calling thread's current device */
} mage_ba_params;

/* BundlerLib::BundlerLib / ~BundlerLib  (BundlerLib.cpp:184-196, :352) */
mage_status mage_ba_create(const mage_ba_params* params, mage_ba** out);
void        mage_ba_destroy(mage_ba* h);

/* AllocateCameras / SetCameraPose / FixCameraPose  (BundlerLib.cpp:198-207, 261-281) */
mage_status mage_ba_alloc_cameras(mage_ba* h, size_t count);
mage_status mage_ba_set_camera(mage_ba* h, size_t idx, const float position[3], const float R_colmajor[9],
                               const float cx_cy_fx_fy[4], int is_fixed);
mage_status mage_ba_fix_camera(mage_ba* h, size_t idx, int is_fixed);

/* AllocateMapPoints / SetMapPoint  (BundlerLib.cpp:209-217, 283-292) */
mage_status mage_ba_alloc_points(mage_ba* h, size_t count);
mage_status mage_ba_set_point(mage_ba* h, size_t idx, const float xyz[3]);

/* AllocateObservations / SetObservation  (BundlerLib.cpp:219-229, 294-309) */
mage_status mage_ba_alloc_observations(mage_ba* h, size_t count);
mage_status mage_ba_set_observation(mage_ba* h, size_t idx, const float uv[2], uint64_t camera_index,
                                    uint64_t point_index, float information_scalar);

/* Bulk forms of the three setters (same semantics as calling the scalar setter for idx = 0..count-1);
 * they exist because BuildDataForG2O (BundleAdjust.cpp:25-193) makes one call per element, which
 * dominates once the solve is fast. */
mage_status mage_ba_set_cameras_bulk(mage_ba* h, size_t count, const float* positions3, const float* R_colmajor9,
                                     const float* cx_cy_fx_fy4, const uint8_t* is_fixed);
mage_status mage_ba_set_points_bulk(mage_ba* h, size_t count, const float* xyz3);
/* EXTENSION (no counterpart in BundlerLib.h): new poses for cameras that are already set, without touching the graph.
 * The reference re-creates the bundler to re-seed poses (BundleAdjust.cpp:281-354); a map sharded by keyframe window
 * (mageslam_amd/windowed.py) refreshes the fixed halo cameras of every window once per outer iteration, and a
 * rebuild of the structure each time would cost more than the LM iteration it precedes.  The optimiser restarts
 * (iteration 0, lambda re-seeded) exactly as after SetCurrentLambda. */
mage_status mage_ba_update_camera_poses(mage_ba* h, size_t count, const uint32_t* indices, const float* positions3,
                                        const float* R_colmajor9);

/* EXTENSION -- device-resident pose exchange for a map sharded by keyframe window (north star: "RCCL all-reduce of the
 * pose block over xGMI").  A POSE BLOCK is device memory of rows x 8 float64, one row per keyframe of the whole map in the
 * solver's own state format: qx qy qz qw tx ty tz 0 (world -> camera).  Nothing is staged through the host and nothing
 * synchronises: export / import run on the handle's stream but are ordered AS IF ENQUEUED ON `stream` (a hipStream_t of the
 * caller; NULL = the device's null stream) -- they wait for what `stream` holds and `stream` waits for them.
 *   bind    the handle's cameras that are published (export_cameras[k] -> block row export_rows[k]) and the ones that are
 *           re-seeded from the block (import_cameras[k] <- row import_rows[k]); host arrays, copied once.
 *   export  block[row] = current estimate (+0.0, so a row is bit-identical whether or not it went through a SUM with other
 *           ranks' zero rows).
 *   import  both device state buffers of the bound cameras <- block rows.  Same effect as mage_ba_update_camera_poses: the graph is kept, the optimiser
 *           restarts (iteration 0; re-seed lambda with mage_ba_set_lambda to carry the damping over).
 * The caller zero-fills the rows nobody exports, runs its collective (ncclAllReduce SUM of rows x 8 doubles: rows are disjoint
 * between ranks, so the sum is exact) on its own stream, and imports.  tools/windowed_rccl.cpp / mage_window.h do exactly this. */
mage_status mage_ba_bind_pose_exchange(mage_ba* h, size_t n_export, const uint32_t* export_cameras, const uint32_t* export_rows,
                                       size_t n_import, const uint32_t* import_cameras, const uint32_t* import_rows);
mage_status mage_ba_export_poses_device(mage_ba* h, double* block_device, void* stream);
mage_status mage_ba_import_poses_device(mage_ba* h, const double* block_device, void* stream);
/* Blocks until everything enqueued on the handle's stream has finished (the handle's stream is private). */
mage_status mage_ba_synchronize(mage_ba* h);

/* ---- Landmark-sharded solve of ONE map on several GPUs (SURVEY 8e "exact algorithm"; no counterpart in the reference, whose
 * optimiser is single-threaded g2o behind StepOptimizer::Step, BundlerLib.cpp:132-149).
 * Every rank creates a handle, gives it ALL cameras (same order, same fixed flags, same poses) and only ITS OWN share of the map
 * points with every observation of those points (and any share of the tether edges: each edge on exactly one rank), then calls
 * mage_ba_set_landmark_shard before the first step.  mage_ba_step must then be called by all ranks together with the same arguments.
 * Per Levenberg-Marquardt trial a rank builds the camera system of its own landmarks -- U_r, b_r and -sum W V^-1 W^T over its
 * landmarks; the damping of the camera blocks is added by rank 0 alone -- and the ranks ADD them: one call of `allreduce` on the
 * lower tiles of S packed behind each other with the right-hand side at the end (6n(6n+128)/2 + 6n doubles for n free cameras; 148 MB at
 * 1 000 poses).  Every rank then factorises the same matrix (bit-identical results on identical devices), moves the cameras, and
 * back-substitutes its own landmarks.  Besides that call a trial exchanges two scalars (chi^2 of the trial, the gain denominator), an
 * iteration one (chi^2) plus, when lambda is seeded, the diagonal of U (6n doubles) and its maximum, and a step the three sums of
 * the outlier pass and one flag: a step's mean error is the map's, the outlier list a rank gets holds its own observations, and
 * the optimiser re-initialises on all ranks when any rank removed one (BundlerLib.cpp:135-138).  Every free camera is part of
 * the system on every rank whether or not the rank (or anyone) observes it; a map without a free camera is refused
 * (MAGE_ERR_UNSUPPORTED: its landmarks are independent, run them unsharded).  The small-problem and pose-only fast paths are off.
 *
 * allreduce(user, buffer, count, op, stream): in-place all-reduce over the ranks of `count` doubles at DEVICE address `buffer`
 * (op 0: sum, 1: max), ordered after the work already enqueued on the HIP stream `stream` and before any enqueued later --
 * ncclAllReduce(buffer, buffer, count, ncclDouble, op, comm, stream) is exactly that (tools/sharded_rccl.cpp).  Every rank
 * must receive the same bytes (RCCL's ring and tree algorithms do).  Non-zero return = failure (the step returns MAGE_ERR_DEVICE).
 * Failures are collective where the library can make them so: a rank that cannot build its part of the problem (out of memory,
 * an unsupported shape) or whose dense solve reports a stalled hand-off tells the others through the step's flag / the trial's
 * scalars, and mage_ba_step returns an error on EVERY rank at the same point.  A failing `allreduce` callback cannot be made
 * collective by the library (the communicator itself is broken): the caller must abort the communicator (ncclCommAbort) so that
 * the other ranks leave their pending collective.
 * n_ranks = 0 switches sharding off. */
typedef int (*mage_ba_allreduce_fn)(void* user, double* buffer_device, size_t count, int op, void* stream);
mage_status mage_ba_set_landmark_shard(mage_ba* h, int rank, int n_ranks, mage_ba_allreduce_fn allreduce, void* user);
/* Who owns which map point: owner[p] in [0, n_ranks), the heaviest point (k (k + 1) / 2 blocks of S for k observations: SURVEY 8e)
 * first onto the lightest rank, ties to the lower point index / lower rank -- a pure function of its arguments, so every rank
 * computes the same table on its own.  Host only (no device needed). */
mage_status mage_ba_partition_landmarks(size_t n_points, size_t n_observations, const uint32_t* point_index, int n_ranks, int32_t* owner);
/* All-reduce among buffers ONE process can address -- several handles on one device, or devices with peer access enabled:
 * every bufs[r][i] becomes the op over r (sum in rank order) of bufs[r][i]; enqueued on `stream`, whose device runs the kernel.
 * The building block of an `allreduce` callback for a process that drives all shards itself (mageslam_amd/sharded.py ThreadGroup). */
mage_status mage_device_allreduce_local(double* const* bufs_device, int n_bufs, size_t count, int op, void* stream);

mage_status mage_ba_set_observations_bulk(mage_ba* h, size_t count, const float* uv2, const uint32_t* camera_index,
                                          const uint32_t* point_index, const float* information_scalar);

/* Tether edges: pose-pose constraints between two cameras (stereo rigs, inertial fusion; gathered by
 * BundleAdjust.cpp:57-112, handed over at :155-192).  Allocate* once each (BundlerLib.cpp:231-259; a count of zero is
 * what monocular maps pass), then one Set* per index.  Quaternions are Eigen::Quaternionf coefficient order x, y, z, w.
 *   fixed distance      BundlerLib.cpp:24-54, 311-322   e = (distance - |t_2 - t_1|) * weight, t = pose translation
 *   relative rotation   BundlerLib.cpp:56-90, 324-336   e = angle between (T_1^-1 T_2).rotation and delta_rotation, * weight
 *   relative transform  BundlerLib.cpp:338-350          g2o EdgeSE3Expmap: e = log(T_2^-1 * SE3(delta_rotation, delta_position) * T_1),
 *                                                       information weight * I6
 * The first two are differentiated numerically exactly as g2o's BaseMultiEdge does (central differences, step 1e-9).
 * A tether is active unless both cameras are fixed; it takes no part in the outlier pass or the returned mean error
 * (the reference walks tether edges there through a mis-typed vertex cast, BundlerLib.cpp:402-403 -- undefined behaviour
 * that is not reproduced).  camera_index_1 == camera_index_2 is rejected with MAGE_ERR_INVALID_ARGUMENT. */
mage_status mage_ba_alloc_fixed_distance_constraints(mage_ba* h, size_t count);
mage_status mage_ba_alloc_relative_rotation_constraints(mage_ba* h, size_t count);
mage_status mage_ba_alloc_relative_transform_constraints(mage_ba* h, size_t count);
mage_status mage_ba_set_fixed_distance_constraint(mage_ba* h, size_t idx, size_t camera_index_1, size_t camera_index_2,
                                                  float distance, float weight);
mage_status mage_ba_set_relative_rotation_constraint(mage_ba* h, size_t idx, size_t camera_index_1, size_t camera_index_2,
                                                     const float delta_rotation_xyzw[4], float weight);
mage_status mage_ba_set_relative_transform_constraint(mage_ba* h, size_t idx, size_t camera_index_1, size_t camera_index_2,
                                                      const float delta_position[3], const float delta_rotation_xyzw[4], float weight);

/* SetCurrentLambda / GetCurrentLambda  (BundlerLib.cpp:123-130, 354-362) */
mage_status mage_ba_set_lambda(mage_ba* h, float user_lambda);
mage_status mage_ba_get_lambda(const mage_ba* h, float* lambda_out);

/* StepBundleAdjustment  (BundlerLib.cpp:364-447): one Levenberg-Marquardt iteration per Huber width,
 * then classification of every active observation (behind the camera, or squared reprojection error
 * above max_error_square -> removed and reported).  Outlier observation indices are written in
 * ascending order to outliers[0..min(*n_outliers, capacity)); *n_outliers is the full count
 * (the reference appends to a std::vector); the full list stays readable through mage_ba_get_outliers
 * until the next step, so a short buffer never loses removed observations.  *mean_square_error is the reference's return value
 * (NaN when no inlier remains). */
mage_status mage_ba_step(mage_ba* h, const float* huber_width_per_iteration, size_t n_iterations,
                         float max_error_square, uint32_t* outliers, size_t capacity, size_t* n_outliers,
                         float* mean_square_error);

/* The complete outlier list of the most recent mage_ba_step, ascending (the same entries the step wrote, without its capacity
 * limit): *count is the full length, outliers[0..min(*count, capacity)) is filled (outliers may be NULL to query the length).
 * A caller that cannot bound the list up front -- the reference's callers pass a growing std::vector, BundleAdjust.cpp:316-320 --
 * steps with capacity 0 and reads the list here; nothing is ever dropped. */
mage_status mage_ba_get_outliers(const mage_ba* h, uint32_t* outliers, size_t capacity, size_t* count);

/* GetPose / GetPoint  (BundlerLib.cpp:457-471) */
mage_status mage_ba_get_pose(const mage_ba* h, size_t idx, float position[3], float R_colmajor[9]);
mage_status mage_ba_get_point(const mage_ba* h, size_t idx, float xyz[3]);
mage_status mage_ba_get_poses_bulk(const mage_ba* h, size_t count, float* positions3, float* R_colmajor9);
mage_status mage_ba_get_points_bulk(const mage_ba* h, size_t count, float* xyz3);

/* ---- Diagnostics (no counterpart in the reference; used by tests and bench.py) ---- */

/* float64 state as the solver holds it: poses as (qx,qy,qz,qw,tx,ty,tz), points as (x,y,z). */
mage_status mage_ba_get_state_f64(const mage_ba* h, double* poses7, double* points3);

typedef struct mage_ba_iter_stats {
    int    code;          /* 0 OK, 1 Terminate, 2 Fail (g2o SolverResult) */
    int    trials;        /* damped trials taken in this iteration */
    double chi2_before;   /* robustified chi2 at linearisation */
    double chi2_after;    /* robustified chi2 of the kept state */
    double lambda;        /* lambda after the iteration */
} mage_ba_iter_stats;

/* Statistics of the iterations run by the most recent mage_ba_step (up to 64 kept). */
mage_status mage_ba_get_iter_stats(const mage_ba* h, mage_ba_iter_stats* out, size_t capacity, size_t* count);

/* Timing of the dense reduced-camera factorisation, measured with HIP events on the handle's own
 * stream when enabled (adds one event pair per factorisation). */
typedef struct mage_ba_profile {
    uint64_t n_factorizations;
    double   factor_ms_total;      /* sum of event-timed factorisation spans */
    double   factor_flops_each;    /* n^3/3 for the system order n = 6 * free cameras (padding not counted) */
    uint64_t schur_launches;
    double   schur_ms_total;
    int      system_order;         /* 6 * free cameras */
    int      padded_order;
    /* the HBM-bound stages of an LM iteration (everything but the dense factorisation), HIP-event spans on the handle's stream,
     * and their ALGORITHMIC bytes per launch for this problem and this data layout (every array a stage must
     * read or write counted once, W blocks materialised) */
    uint64_t linearize_launches;   /* k_error + k_linearize_lm + k_linearize_cam (+ tethers): once per LM iteration */
    double   linearize_ms_total;
    double   linearize_bytes_each;
    double   schur_bytes_each;     /* zero-fill of S + k_lm_invert + k_schur_block + k_schur_rhs: once per LM trial */
    uint64_t update_launches;      /* k_backsub + k_pose_update + k_error of the trial state: once per LM trial */
    double   update_ms_total;
    double   update_bytes_each;
    /* Health of the dense solve's in-launch hand-offs (always counted, profiling on or off).  Both stay 0 in a process that has the
     * GPU to itself; they move when several PROCESSES oversubscribe one GPU (DESIGN.md "forward progress"): a deployment that sees
     * them grow is paying for re-run trials and should give each process its own GPU or set MAGE_CHOL_NO_MERGED_TRSM=1. */
    uint64_t trials_rerun_after_stall;        /* LM trials of THIS handle run again because a bounded wait between workgroups timed out */
    uint64_t fallback_to_separate_launches;   /* 1 once this PROCESS has switched the panel solve back to its own launches after such a stall */
} mage_ba_profile;
/* enable: 0 off; 1 every stage bracketed by event records (seven per LM trial: each costs a few microseconds of idle stream);
 * 2 only the dense factorisation + solves (two per trial) -- what a timed run keeps on.  Enabling resets the sums. */
mage_status mage_ba_enable_profiling(mage_ba* h, int enable);
mage_status mage_ba_get_profile(const mage_ba* h, mage_ba_profile* out);
/* The reduced camera matrix of a trajectory map is block-banded: cameras far apart share no landmark.  The reference's LinearSolverDense
 * (BundlerLib.cpp:184-196) factors it as a dense matrix all the same, and so does this library by default.  enable = 1:
 * the dense solve's task-graph schedule skips every 128 x 128 tile left of the matrix's skyline (by tile rows: Schur blocks, tether pairs;
 * fill-in never leaves a row's envelope) -- tiles that are zero and stay zero in the factor, so the numbers are the same to the bit and
 * the factorisation is bound by its chain of diagonal tiles alone (1k-pose map 2.05 -> 1.86 ms, 2k-pose map 11.7 -> 3.7 ms).  Takes
 * effect with the next structure build; MAGE_BA_SKYLINE=1 makes it the default of every new handle.  Not for landmark-sharded handles. */
mage_status mage_ba_use_skyline(mage_ba* h, int enable);

/* A/B switch of the Schur build's launch on maps of more than 2 048 blocks: 0 (default) resident wavefronts work through per-compute-unit
 * lists of blocks (k_schur_stream); 1 one wavefront per 6 x 6 block (k_schur_block_compact, rounds 2-5).  The same sums in the same order:
 * the reduced system is the same to the bit (tests/test_ba_gpu.py).  MAGE_BA_SCHUR_BLOCKS=1 makes 1 the default of every new handle. */
mage_status mage_ba_debug_schur_per_block(mage_ba* h, int enable);

/* One list of the graph structure as it sits in HBM after the first step (names as in mageslam_amd/csrc/ba_kernels.h:
 * "cam2hc", "hc2cam", "L_edge", "L_uv", "L_info", "L_cam", "L_pt", "L_slot", "lm_ptr", "lm_pt", "lm_wptr", "w_hc", "w_lm", "camE_ptr",
 * "camE", "camS_ptr", "camS", "blk_ptr", "blk_ij", "con", "blk_order", "stream_ptr", "stream_blks" (k_schur_stream's lists; empty without them); "sizes" = 12 ints: n_L, n_lm, n_fc, n_w, n_blk, n_blk_slots,
 * n_con, dup_slots, n_pad, built_on_device, 0, 0).  *bytes = the list's size; it is copied when capacity_bytes suffices.
 * The structure is what g2o's initializeOptimization + buildStructure produce (BundlerLib.cpp:156-166); it is built on the
 * device by default and on the host with MAGE_BA_BUILD=host -- the tests compare the two element for element. */
mage_status mage_ba_debug_structure(mage_ba* h, const char* name, void* out, size_t capacity_bytes, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* MAGE_BA_H */
