// ba_kernels.h -- device-side data layout and kernel launchers of the bundle-adjustment path.
//
// Everything the kernels touch lives in HBM as struct-of-arrays, float64 for state and
// accumulators, float32/u32 for the observation records exactly as they cross the BundlerLib
// surface (BundlerLib.h:28-39).  Observations are kept in LANDMARK ORDER (all observations of a
// map point contiguous, free cameras first and ascending) -- the CSR-by-landmark graph of
// DESIGN.md section 4; per-camera index lists give the CSR-by-camera view.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mage {

constexpr int SCHUR_SPLIT_BLOCKS_BELOW = 2048;       // Schur blocks: at most this many -> four wavefronts per block; more: k_schur_stream
constexpr int SCHUR_WAVES = 1;     // wavefronts (= blocks of S) per workgroup of k_schur_block; the slot table is built for it

struct ConPos { int a, b, lm; };
struct BaDeviceView {
    // ---- sizes
    int n_cams, n_pts;        // allocated cameras / points
    int n_L;                  // active observations (landmark order)
    int n_lm;                 // landmarks with >= 1 active observation
    int n_fc;                 // free cameras in the system (hessian-indexed)
    int n_w;                  // W slots: distinct (free camera, free landmark) pairs
    int n_blk;                // non-empty upper 6x6 blocks of the reduced camera matrix
    int points_free;          // 0 when BundlerParameters::ArePointsFixed
    int n_pad;                // padded order of the reduced camera system (multiple of the tile)
    int dup_slots;            // 1 when some landmark is observed twice by one free camera (several observations share a W slot)

    // ---- state (current = accepted estimate, trial = LM candidate; swapped on accept)
    double* pose_cur;   double* pose_trial;   // n_cams x 8 : qx qy qz qw tx ty tz pad
    double* pt_cur;     double* pt_trial;     // n_pts  x 4 : x y z pad
    const double* camK;                        // n_cams x 4 : f cx cy pad
    const int* cam2hc;                         // n_cams : hessian index or -1
    const int* hc2cam;                         // n_fc

    // ---- observations, landmark order
    const float2* L_uv; const float* L_info; const uint32_t* L_cam; const uint32_t* L_pt;
    const int* L_slot;                         // W slot of the observation or -1
    uint8_t* L_active;                         // 0 once the observation was removed as an outlier (soft removal, no rebuild)
    const uint32_t* L_edge;                    // original observation index
    const int* lm_ptr;                         // n_lm + 1 offsets into L_*
    const int* lm_pt;                          // n_lm : point index of the landmark
    const int* lm_wptr;                        // n_lm + 1 offsets into the slot arrays
    const int* w_hc; const int* w_lm;          // n_w : camera hessian index / landmark of a slot
    // ---- camera views
    const int* camE_ptr; const int* camE;      // per free camera: positions (in L order) of its observations
    const int* camS_ptr; const int* camS;      // per free camera: its W slots
    // ---- reduced-camera-matrix structure
    const int* blk_ptr; const int2* blk_ij; const int2* con;   // contributions (slot_a, slot_b) per block
    const int* blk_order; int n_blk_slots;                     // wavefront slot -> block (-1: none): rows of S are pinned to XCDs (ba_host.hip)
    // ---- where a slot's COMPACT record lives (round 4).  Slots are numbered landmark-major (lm_wptr); a block (i, j) of the Schur
    // complement reads the records of the landmarks cameras i and j share -- ten slots apart in that numbering, one 128-byte line per
    // 32-byte record.  The compact records are therefore STORED camera-major (position p of camS <-> slot camS[p]: a camera's records
    // in ascending landmark), where the records a block reads are neighbours.  Placement only: the same values meet in the same
    // order.  MAGE_BA_W_LANDMARK_MAJOR=1 keeps position == slot (A/B).  Valid while `compact` (ba_launch_build_positions).
    const int* w_pos;                          // n_w : slot -> position
    const int* pos_lm;                         // n_w : landmark of a position
    const int* slot_order;                     // blk_order with every XCD's run sorted longest block first (k_schur_block_compact: the diagonal
                                               // blocks are four times the average and sat at regular intervals up to the END of the grid)
    const int* con_soa; int con_soa_pitch;     // the same list as three arrays (a | b | lm, each con_soa_pitch ints) for k_schur_stream; null without it
    long long* stream_stamps;                  // debug (MAGE_BA_SCHUR_TRACE): 4 words per wavefront of k_schur_stream (8 per workgroup), else null
    const int* stream_ptr; const void* stream_blks; int n_stream_groups;   // k_schur_stream: per workgroup (= compute unit) its list of blocks (n_stream_groups + 1 offsets; 16-byte descriptors c_begin, c_end, i, j); null: one wavefront per block
    const struct ConPos* con_pos;              // the contributions of `con` as (position a, position b, landmark): 12 bytes, ONE trip to memory
                                               // between a block's list and its records (the landmark used to be a look-up of its own)

    // ---- tether edges (pose-pose constraints; active ones only, all three kinds in one list)
    int n_T, n_tc, n_tp;                       // tethers / cameras carrying tethers / free-camera pairs joined by tethers
    const int* T_kind; const int2* T_cam; const int2* T_fixed;   // kind, (camera a, camera b), (a fixed, b fixed)
    const double* T_meas;                      // n_T x 8 : measurement q(x y z w) t(x y z) distance
    const double* T_w;                         // n_T : weight
    double* T_out;                             // n_T x TETHER_OUT_STRIDE : H_aa(36) H_bb(36) H_ab(36) b_a(6) b_b(6)
    const int* tc_hc; const int* tc_ptr; const int* tc_item;     // per camera: (tether << 1 | side) in tether order
    const int2* tp_ij; const int* tp_ptr; const int* tp_item;    // per pair i < j: (tether << 1 | transposed)

    // ---- linear system
    double* errL;          // n_L x 2   residual of the last error evaluation
    double* U; double* bc; // n_fc x 36, n_fc x 6
    double* V; double* bp; // n_lm x 6 (sym: 00 01 02 11 12 22), n_lm x 4
    double* W;             // n_w x 18  (6x3 row-major); in COMPACT form the same memory holds n_w x 4: x/z, y/z, 1/z of the point in the
                           // slot's camera and the observation's robust weight -- W = Jc^T w Jp is a function of those and of the camera
    int compact;           // 1: large problems without shared slots keep W in that form (set per LM iteration by the host)
    double* camR;          // n_fc x 12 : f and the rotation R (row-major) of every free camera at the linearisation point (COMPACT form)
    double* Dinv; double* db;  // n_lm x 6, n_lm x 4
    const int* tile_env;   // per 128-row tile row R of S: the first tile column that can ever hold a non-zero (structure + fill-in stay inside this
                           // skyline); non-null only while every tile left of it is KNOWN to hold zeros: the zero-fill then clears the skyline alone
    double* S;             // n_pad x n_pad column-major, lower triangle valid
    double* y;             // n_pad : reduced rhs b_s (forward-substituted in place by the factorisation)
    double* xc;            // n_pad : camera increments x_c
    double* xl;            // n_lm x 4 landmark increments
    // ---- scalars / scratch
    double* partial;       // scratch for two-level reductions (>= 4096 doubles)
    double* scal;          // device scalars, see enum Scal
};

enum TetherKind { TETHER_DISTANCE = 0, TETHER_ROTATION = 1, TETHER_TRANSFORM = 2 };
constexpr int TETHER_OUT_STRIDE = 120;

// SC_CHI: robust chi2 of the current estimate; SC_CHI_TRIAL: of the LM candidate (separate slots: one host read fetches both)
enum Scal { SC_CHI = 0, SC_SCALE = 1, SC_MAXDIAG = 2, SC_CHOL_OK = 3, SC_ERRSUM = 4, SC_ERRCNT = 5, SC_NOUT = 6, SC_CHI_TRIAL = 7, SC_CHOL_STALL = 8,
            SC_SHARD_FLAG = 9, SC_NOUT_OWN = 10,      // landmark-sharded maps: "some rank re-initialises", this rank's outlier count (SC_NOUT then holds the map's)
            SC_SPEC_DONE = 11,                        // 1.0: the post-pass queued behind the trial (ba_launch_classify_after_trial) found the call finished and ran
            SC_COUNT = 12 };

// All launchers enqueue on `st` and return immediately.
void ba_launch_error(const BaDeviceView& v, bool trial, double huber_delta, hipStream_t st);         // -> scal[SC_CHI] / scal[SC_CHI_TRIAL]
void ba_launch_linearize(const BaDeviceView& v, double huber_delta, hipStream_t st);                 // U,bc,V,bp,W
void ba_launch_maxdiag(const BaDeviceView& v, hipStream_t st, const double* udiag_sum = nullptr);                                       // -> scal[SC_MAXDIAG]
void ba_launch_schur(const BaDeviceView& v, double lambda, hipStream_t st);                          // Dinv,db,S,y
void ba_launch_build_positions(const BaDeviceView& v, int* w_pos, int* pos_lm, ConPos* con_pos, int* slot_order, int* con_soa, int con_soa_pitch, hipStream_t st); // w_pos / pos_lm / con_pos from camS, w_lm, con
int ba_schur_stream_groups(int n_cu);                                                                 // workgroups of k_schur_stream on a device of n_cu compute units
int ba_schur_stream_rounds(int n_blk_slots, int n_groups);                                            // blocks per workgroup at most
void ba_launch_build_stream_lists(const BaDeviceView& v, int n_groups, int* group_blocks, int* group_ptr, void* blks, hipStream_t st);   // from slot_order / blk_ptr / blk_ij; group_blocks: n_groups x rounds ints; blks: n_blk descriptors of 16 bytes
void ba_launch_tile_envelope(const BaDeviceView& v, int* tile_env, hipStream_t st);                  // the skyline of S by tile rows, from blk_ij and the tether pairs
void ba_launch_update(const BaDeviceView& v, double lambda, hipStream_t st);                         // xl, trial state, scal[SC_SCALE]
// landmark-sharded maps (include/mage_ba.h: mage_ba_set_landmark_shard)
void ba_launch_schur(const BaDeviceView& v, double lambda, double lambda_cam, double pad_diag, hipStream_t st);
void ba_launch_schur(const BaDeviceView& v, double lambda, double lambda_cam, double pad_diag, int fold_chi_partials, hipStream_t st);   // + the deferred chi2 fold of ba_fused_linearize
void ba_launch_update(const BaDeviceView& v, double lambda, double lambda_cam, hipStream_t st);
bool ba_update_and_trial_error_fuses(const BaDeviceView& v);                                             // free points: update + trial chi2 in three launches
void ba_launch_update_and_trial_error(const BaDeviceView& v, double lambda, double lambda_cam, double huber_delta, hipStream_t st);
size_t ba_packed_doubles(int n_pad);                                                                   // lower tiles of S + y
void ba_launch_pack_lower(const BaDeviceView& v, double* packed, bool to_packed, hipStream_t st);
void ba_launch_gather_udiag(const BaDeviceView& v, double* out6_per_camera, hipStream_t st);
bool ba_launch_allreduce_local(double* const* bufs, int n, size_t count, int op, hipStream_t st);
void ba_launch_classify(const BaDeviceView& v, double max_err_sq, uint32_t* out_ids, int* out_count, int out_base, hipStream_t st); // scal[SC_ERRSUM..SC_NOUT]
// The same post-pass queued BEHIND an LM trial, before the host has seen the trial's scalars: every workgroup repeats the host's
// decision (OptimizationAlgorithmLevenberg::solve) from scal[] -- accepted / rejected, trial loop over or not -- and
// classifies only when this StepBundleAdjustment call has taken its last trial (it then reads the KEPT estimate: the trial's
// buffers when accepted, the current ones when not; residuals are the last trial's either way, BundlerLib.cpp:386-425).  Otherwise
// it leaves everything untouched and scal[SC_SPEC_DONE] = 0.  One host round trip per call instead of two.
struct ClassifyAfterTrial {
    double chi_ref; int chi_on_device;    // chi2 of the current estimate: scal[SC_CHI] (first trial of an iteration) or the host's value
    int trials_done;                      // trials of this iteration including the one just queued
    int last_iteration;                   // no LM iteration follows in this call
};
void ba_launch_classify_after_trial(const BaDeviceView& v, const ClassifyAfterTrial& c, double max_err_sq, uint32_t* out_ids, int* out_count, int out_base, hipStream_t st);

// Small problems (reduced camera system of order <= 128, no tethers): one LM trial in five launches instead of ~22
// (ba_kernels.hip, "SMALL PROBLEMS").  `counter` is one zero-initialised device int owned by the handle (the kernels leave it 0).
bool ba_small_applies(const BaDeviceView& v);
bool ba_small_shape_applies(int n_fc, int n_tethers, long long n_L);                                   // the same predicate before a view exists (structure build)
void ba_small_init_device();                                                                          // once per device: LDS opt-in
bool ba_compact_w_enabled();                                                                          // false with MAGE_BA_MATERIAL_W=1 (A/B, tests)
bool ba_fused_linearize_applies(const BaDeviceView& v);                                              // large, one observation per W slot
int ba_fused_linearize(const BaDeviceView& v, double huber_delta, int* counter, hipStream_t st, bool defer_chi_fold = false);    // = ba_launch_error(current) + ba_launch_linearize in one launch
void ba_small_linearize(const BaDeviceView& v, double huber_delta, bool want_maxdiag, int* counter, hipStream_t st);   // U,bc,V,bp,W, S/y zeroed, scal[SC_CHI] (+ SC_MAXDIAG)
void ba_small_solve_trial(const BaDeviceView& v, double lambda, double huber_delta, double* linv_ws, int* counter, hipStream_t st);    // S, y, xc, trial state, SC_SCALE, SC_CHI_TRIAL, SC_CHOL_OK/STALL
void ba_small_classify(const BaDeviceView& v, double max_err_sq, uint32_t* out_ids, int* out_count, int out_base, int* counter, double* result, hipStream_t st);   // result: the kept estimate (poses x 8, points x 4) or nullptr
// mirror (may be null): the device address of the host's pinned mirror of v.scal -- the launch writes the first mirror_doubles doubles of the
// run that starts at v.scal there itself (mirror_scalars when the call turns out not to be over): no read-back copy is queued
void ba_small_classify_after_trial(const BaDeviceView& v, const ClassifyAfterTrial& c, double max_err_sq, uint32_t* out_ids, int* out_count, int out_base, int* counter, double* result,
                                   double* mirror, int mirror_scalars, int mirror_doubles, int ids_prefix, hipStream_t st);      // (+ up to ids_prefix outlier ids behind the mirror_doubles)

// Pose-only problems (points fixed): the whole StepBundleAdjustment call in ONE launch (ba_kernels.hip, "POSE-ONLY problems").
constexpr int POSE_LM_MAX_ITERS = 16;
struct PoseLmArgs {
    int n_huber; float huber[POSE_LM_MAX_ITERS];      // one LM iteration per Huber width
    double max_err_sq;                                // outlier threshold of the post-pass
    double lambda, user_lambda, ni; int iteration;    // LM state on entry
};
struct PoseLmIter { int code, trials; double chi2_before, chi2_after, lambda; };
struct PoseLmResult {
    double lambda, ni; int iteration, n_stats, flips; // LM state on exit; flips = accepted trials (each swaps the two pose buffers)
    double err_sum, err_cnt, n_out;                   // post-pass
    PoseLmIter stats[POSE_LM_MAX_ITERS];
};
bool ba_pose_lm_applies(const BaDeviceView& v, size_t n_huber);
// out_pose (may be null: the two pose buffers of `v` are then the result, in place): 2 x n_cams x 8 doubles, buffer 0 then buffer 1
void ba_launch_pose_lm(const BaDeviceView& v, const PoseLmArgs& a, PoseLmResult* out_device, uint8_t* flag_by_edge, double* out_pose, hipStream_t st);
// the same solve with every array staged in LDS (only the two pose buffers are written back): false = the problem does not fit
constexpr int POSE_LM_STAGED_MAX_BYTES = 140 * 1024;
bool ba_pose_lm_staged_fits(const BaDeviceView& v);     // the staged form's LDS image fits: the launch below will be taken
// (its inputs must be ONE packed run, ba_host.hip's frame image, read once and never written: v.pose_cur = start of
//  [pose0 | pose1 | camK | pt | L_uv | L_info | L_cam | L_pt | camE | camE_ptr | hc2cam | L_active] + 16 bytes of slack)
bool ba_launch_pose_lm_staged(const BaDeviceView& v, const PoseLmArgs& a, PoseLmResult* out_device, uint8_t* flag_by_edge, double* out_pose, hipStream_t st);

// Pose exchange of a window-sharded map (mage_ba_export_poses_device / mage_ba_import_poses_device).  A block row is 8 doubles
// (qx qy qz qw tx ty tz 0).  export: block[row[k]] = pose[cam[k]] (+0.0, so that -0.0 leaves as +0.0 -- what a SUM with the
// zero rows of the other ranks would make of it anyway); import: pose0[cam[k]] = pose1[cam[k]] = block[row ? row[k] : k].
void ba_launch_export_poses(const double* pose, const uint32_t* cam, const uint32_t* row, size_t n, double* block, hipStream_t st);
void ba_launch_import_poses(double* pose0, double* pose1, const uint32_t* cam, const uint32_t* row, size_t n, const double* block, hipStream_t st);

// tether_kernels.hip (called by the launchers above when the problem carries tethers)
void tether_launch_error(const BaDeviceView& v, bool trial, hipStream_t st);      // same slot += tether chi2
void tether_launch_linearize(const BaDeviceView& v, hipStream_t st);              // U, bc += tether blocks
void tether_launch_schur(const BaDeviceView& v, hipStream_t st);                  // S += pose-pose blocks

}  // namespace mage
