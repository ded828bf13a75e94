// ba_build.h -- the graph structure of a bundle-adjustment problem built ON THE DEVICE (ba_build.hip).
//
// What is built is what SparseOptimizer::initializeOptimization + BlockSolver::buildStructure produce in g2o (SURVEY.md
// appendix A.5; driven by StepOptimizer::InitializeOptimization, BundlerLib.cpp:156-166), re-expressed as the flat lists of
// ba_kernels.h: the hessian index map of the cameras, the observations in landmark order, the W slots, the per-camera views and
// the block lists of the reduced camera matrix.  The reference rebuilds this for EVERY optimisation -- a bundler lives for one
// BundleAdjust call (BundleAdjust.cpp:293, 348-351) and its default local BA is one LM iteration (MageSettings.h:42-44) -- so
// the build is on the caller's critical path, not set-up.  The host build of ba_host.hip stays as the A/B twin
// (MAGE_BA_BUILD=host); both produce the same lists, element for element (tests/test_ba_gpu.py).
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

// Sizes the device build leaves for the host (one small read-back in the middle, one at the end).
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
    unsigned long long* scan_tmp;                  // block sums of the scans
    int* hist;                                     // multi-split histograms: blocks x n_fc
    unsigned long long* row;                       // per row of S: (blocks << 40 | contributions), then their exclusive scan
    BuildCounts* counts;
};

constexpr int BUILD_SCAN_BLOCK = 2048;             // elements per workgroup of the scans
inline size_t build_scan_tmp_elems(size_t n) { return (n + BUILD_SCAN_BLOCK - 1) / BUILD_SCAN_BLOCK + 2; }
int build_split_blocks(int n_items, int n_fc);     // workgroups (one wavefront each) of a stable split by camera
int build_row_waves(int n_fc);                     // wavefronts per row of S in the block-list kernels

void build_init_device();                          // once per device: LDS opt-in of the row kernels

// Phase 1 (sizes known: allocated cameras / points / observations): index maps, landmark order, slots.  Leaves n_L, n_fc, n_lm,
// n_w, slot_obs and n_con in *counts.
void build_launch_phase1(const BuildArgs& a, hipStream_t st);
// Phase 2 (n_fc, n_w known on the host): per-camera views.
void build_launch_camera_views(const BuildArgs& a, int n_fc, int n_w, hipStream_t st);
// Phase 3 (n_con known: `con` allocated): rows of S -> row[], n_blk; then (blk arrays allocated for the bound n_blk_max) the
// block lists, and the XCD runs in *counts.
void build_launch_row_count(const BuildArgs& a, int n_fc, hipStream_t st);
void build_launch_row_fill(const BuildArgs& a, int n_fc, hipStream_t st);
// Phase 4 (n_blk, xcd_first known): the slot -> block table of k_schur_block.
void build_launch_blk_order(const BuildArgs& a, int n_blk_slots, hipStream_t st);

}  // namespace mage
