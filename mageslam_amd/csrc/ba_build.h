// ba_build.h -- the graph structure of a bundle-adjustment problem built ON THE DEVICE (ba_build.hip).
//
// What is built is what SparseOptimizer::initializeOptimization + BlockSolver::buildStructure produce in g2o (SURVEY.md
// appendix A.5; driven by StepOptimizer::InitializeOptimization, BundlerLib.cpp:156-166), re-expressed as the flat lists of
// ba_kernels.h: the hessian index map of the cameras, the observations in landmark order, the W slots, the per-camera views and
// the block lists of the reduced camera matrix.  The reference rebuilds this for EVERY optimisation -- a bundler lives for one
// BundleAdjust call (BundleAdjust.cpp:293, 348-351) and its default local BA is one LM iteration (MageSettings.h:42-44) -- so
// the build is on the caller's critical path, not set-up.  The host build of ba_host.hip stays as the A/B twin
// (MAGE_BA_BUILD=host); both produce the same lists, element for element (tests/test_ba_build_gpu.py).
//
// Every list is a deterministic function of the Set* records: counts and offsets come from integer atomics and scans, orders
// from keys that are total orders (observation index, slot index), never from arrival order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mage {

// One SetObservation record as the host keeps it (pinned) and as it is uploaded: BundlerLib.h:36-39 plus the two flags.
struct ObsRecord {
    float u = 0, v = 0, info = 0;
    uint32_t cam = 0, pt = 0;
    uint8_t set = 0, removed = 0, pad0 = 0, pad1 = 0;
};
static_assert(sizeof(ObsRecord) == 24, "ObsRecord layout");

// Sizes the device build leaves for the host: one read-back when the rows of S are counted, a second one (large problems: the XCD runs) at the end.
struct BuildCounts {
    int n_L, n_fc, n_lm, n_w;          // active observations, free cameras in the system, landmarks, W slots
    int slot_obs;                      // observations that own or share a slot (!= n_w: some slot is shared)
    int n_blk;                         // non-empty upper blocks of the reduced camera matrix
    int xcd_longest;                   // longest run of blocks handed to one XCD
    int pad;
    unsigned long long n_con;          // Schur contributions
    int xcd_first[9];                  // first block of every XCD's run (+ n_blk)
    int pad2;
};

// Device arrays the build reads and writes.  Inputs first; outputs are the lists of BaDeviceView (same names); the rest is scratch.
struct BuildArgs {
    // ---- inputs
    const ObsRecord* obs; int n_obs;
    const uint8_t* cam_fixed; int n_cams;
    const int* cam_extra_deg;          // per camera: active tether edges touching it (null: none)
    int n_pts;
    int points_fixed;                  // BundlerParameters::ArePointsFixed
    int keep_all_free_cameras;         // landmark-sharded maps: every free camera is in the system on every rank
    // ---- outputs
    int* cam2hc; int* hc2cam;
    float2* L_uv; float* L_info; uint32_t* L_cam; uint32_t* L_pt; int* L_slot; uint32_t* L_edge;
    int* lm_ptr; int* lm_pt; int* lm_wptr; int* w_hc; int* w_lm;
    int* camE_ptr; int* camE; int* camS_ptr; int* camS;
    int* blk_ptr; int2* blk_ij; int2* con; int* blk_order;
    // ---- scratch
    int* cam_deg; int* pt_deg; int* pt2lm;         // n_cams, n_pts (pt_deg doubles as the fill cursor of the bucket pass), n_pts
    unsigned long long* bucket;                    // n_obs keys (camera key << 32 | observation)
    int* L_hc; int* L_lm; int* where;              // per position: hessian camera or -1, landmark; per observation: position or -1
    int* w_end;                                    // per slot: one past the last slot of its landmark
    unsigned long long* scan_tmp;                  // block sums of the scans
    int* hist;                                     // stable splits: chunk histograms (both splits), then the group totals
    unsigned long long* row;                       // per row of S: (blocks << 40 | contributions), then their exclusive scan
    BuildCounts* counts;
};

constexpr int BUILD_SCAN_BLOCK = 2048;             // elements per workgroup of the scans
constexpr int BUILD_ROW_KC = 12;                   // the row kernels read this many entries of w_hc past a slot unconditionally: pad w_hc by it
inline size_t build_scan_tmp_elems(size_t n) { return (n + BUILD_SCAN_BLOCK - 1) / BUILD_SCAN_BLOCK + 2; }
// counts, cam_deg and pt_deg sit behind each other in ONE allocation so that one fill clears them:
//   [BuildCounts | cam_deg: n_cams + 1 ints | pt_deg: n_pts + 1 ints]
inline size_t build_zeroed_bytes(int n_cams, int n_pts) { return sizeof(BuildCounts) + ((size_t)n_cams + 1 + (size_t)n_pts + 1) * sizeof(int); }
size_t build_hist_ints(int n_obs, int n_fc_max);   // `hist` of the stable splits by camera (both splits, with their group totals)
int build_row_waves(int n_fc);                     // wavefronts per row of S in the block-list kernels

void build_init_device();                          // once per device: LDS opt-in of the row kernels

// Phase 1, everything whose size the host can bound from what was allocated through the surface (n_fc_max = cameras that are
// not fixed): index maps, landmark order, slots, per-camera views, and the rows of S COUNTED.  Leaves n_L, n_fc, n_lm, n_w,
// slot_obs, n_blk and n_con in *counts -- the one read-back a small problem needs.
void build_launch_phase1(const BuildArgs& a, int n_fc_max, hipStream_t st);
// Phase 2 (`con`, `blk_ptr`, `blk_ij` allocated): the block lists; want_xcd_runs: also the XCD runs of k_schur_block in *counts
// (the second read-back, large problems only).
void build_launch_row_fill(const BuildArgs& a, int n_fc, bool want_xcd_runs, hipStream_t st);
// Phase 3 (xcd_first known): the slot -> block table of k_schur_block.
void build_launch_blk_order(const BuildArgs& a, int n_blk_slots, hipStream_t st);

}  // namespace mage
