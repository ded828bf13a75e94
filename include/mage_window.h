/*
 * mage_window.h -- C ABI of the window-sharded map driver (libmageslam_hip.so): ONE large map cut into windows of
 * consecutive keyframes, every window a bundle-adjustment problem of its own (mage_ba.h), the windows' poses reconciled
 * once per outer iteration through a device-resident pose block and one all-reduce (BASELINE.json configs[4]: "8k-pose
 * map sharded by keyframe window, RCCL pose-block all-reduce over xGMI").
 *
 * Reference semantics a window follows: the local bundle adjustment's problem set,
 *     Core/MAGESLAM/Source/Map/ThreadSafeMap.cpp:868-971   (GetMapPointsAndDistantKeyframes: the window's keyframes are
 *                                                            free, every map point they see is in, and every OTHER
 *                                                            keyframe seeing one of those points enters FIXED -- :939)
 *     Core/MAGESLAM/Source/BundleAdjustment/BundleAdjust.cpp:281-354   (one BundlerLib per problem, stepped, written back)
 * The dense reduced camera system of the whole map (48 000^2 f64 = 18 GB at 8k poses) is never formed.  One OUTER ITERATION:
 *   1. every window owned by this rank takes `inner` LM iterations against its frozen halo (windows are independent here:
 *      they are stepped from `threads` concurrent host threads, each handle on its own HIP stream);
 *   2. the exchange, entirely on the device: the pose block (n_cams x 8 f64: qx qy qz qw tx ty tz 0) is zero-filled, every
 *      window writes the rows of the keyframes it OWNS, the caller-supplied all-reduce sums the block over the ranks (rows
 *      are disjoint, the sum is exact), every window re-seeds the cameras it does not own (overlap + halo) from the block.
 *      No host staging, no host synchronisation; the damping of every window carries over (as MappingWorker carries lambda
 *      from one local BA to the next).
 * This is block-Jacobi (restricted additive Schwarz with `overlap`): the same fixed point as the monolithic solve, not the
 * same iterates; since a window's step depends only on its own state and the block, results are bit-identical for every
 * assignment of windows to ranks and for every `threads`.
 *
 * The library itself links no collective library: the all-reduce is a callback (tools/windowed_rccl.cpp binds it to
 * ncclAllReduce on the rank's RCCL communicator; a single-rank run needs none).
 */
#ifndef MAGE_WINDOW_H
#define MAGE_WINDOW_H

#include "mage_ba.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mage_wmap mage_wmap;

typedef struct mage_wmap_params {
    int n_windows;   /* windows of consecutive keyframes: window w owns keyframes [w n / W, (w+1) n / W) */
    int overlap;     /* keyframes either side of a window that are free in it too; only the owner's value is ever published */
    int rank, world; /* this process owns the windows w with floor(w * world / n_windows) == rank (contiguous runs) */
    int device;      /* HIP device ordinal of this rank; -1 = the calling thread's current device */
    int threads;     /* windows of this rank stepped concurrently (>= 1) */
} mage_wmap_params;

/* In-place SUM of `count` float64 at `block_device` over all ranks, enqueued on `hip_stream` (a hipStream_t); returns 0 on
 * success.  With RCCL: ncclAllReduce(block, block, count, ncclDouble, ncclSum, comm, stream). */
typedef int (*mage_allreduce_fn)(void* ctx, double* block_device, size_t count, void* hip_stream);

/* The whole map in the float32 form of the BundlerLib surface (same arrays as the mage_ba_set_*_bulk calls).  Cuts the
 * windows, builds the sub-problems of the windows this rank owns and binds their exchange lists. */
mage_status mage_wmap_create(const mage_wmap_params* params, size_t n_cams, const float* positions3, const float* R_colmajor9,
                             const float* cx_cy_fx_fy4, const uint8_t* is_fixed, size_t n_pts, const float* xyz3, size_t n_obs,
                             const float* uv2, const uint32_t* camera_index, const uint32_t* point_index,
                             const float* information_scalar, mage_wmap** out);
void        mage_wmap_destroy(mage_wmap* h);

/* fn == NULL: single rank (the block is exchanged between this rank's windows only). */
mage_status mage_wmap_set_allreduce(mage_wmap* h, mage_allreduce_fn fn, void* ctx);

/* One outer iteration (see above).  *mean_square_error = observation-weighted mean of the windows' StepBundleAdjustment
 * returns (before the exchange), NaN when no inlier remains. */
mage_status mage_wmap_outer_iteration(mage_wmap* h, float huber_width, float max_error_square, int inner_iterations,
                                      double* mean_square_error);

/* The pose block as of the last exchange, copied to host memory (n_cams x 8 float64).  All zero before the first exchange. */
mage_status mage_wmap_get_pose_block(mage_wmap* h, double* poses8);
/* Device address of the pose block and the stream the exchange runs on (for callers that keep consuming on the device). */
mage_status mage_wmap_pose_block_device(mage_wmap* h, double** block_device, void** hip_stream);

/* Window w as cut from the map (any w, owned or not): own keyframes, cameras (own + overlap + halo), points, observations. */
mage_status mage_wmap_window_info(const mage_wmap* h, int w, size_t* n_own, size_t* n_cameras, size_t* n_points, size_t* n_observations,
                                  int* owned_by_this_rank);

/* DIAGNOSTIC: the bundle-adjustment handle of an owned window (owned by the map: never destroy it, do not step it while an
 * outer iteration runs).  Lets a test read a window's float64 state and iteration statistics. */
mage_status mage_wmap_window_handle(const mage_wmap* h, int w, mage_ba** out);

#ifdef __cplusplus
}
#endif
#endif /* MAGE_WINDOW_H */
