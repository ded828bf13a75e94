// orb_kernels.h -- launchers of the ORB front-end kernels (orb_kernels.hip) and of the Hamming matcher (match_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mage_match.h"
#include "../../include/mage_orb.h"

namespace mage {

struct OrbTaps { int radius; int t[15]; };          // 8-bit fixed-point Gaussian taps (sum ~ 256), radius <= 7

struct OrbSelectArgs {
    const int2* raw; const int* tile_count;                   // per frame: n_tiles x tile_cap slots of (x | y << 16, response), entries per tile
    int n_tiles, tile_cap;
    unsigned long long* cand64; unsigned long long* key64; int* cell_start; int* cell_fill;   // scratch in HBM for oversized frames, per frame
    size_t scratch_cap;                                       // entries of cand64 / key64 per frame
    mage_keypoint* out_kp; int* out_count;
    int ncells, cells_x, cells_y;
    int nfeatures, max_num, fast_threshold, strong_response, capacity, patch_size;
    float feature_strength, min_robust, max_robust;
};

// wp = internal row pitch of the blurred image / raw score map (w rounded up to 4)
constexpr int ORB_MAX_LEVELS = 16;   // pyramid depth accepted by mage_orb_create
// FAST + NMS + border cull, one workgroup per image tile: the keypoints of tile t of a frame go to raw[(frame * n_tiles + t) * tile_cap ...],
// their number to tile_count[frame * n_tiles + t]; raw scores of frame 0 (optional, parity tests)
void orb_fast_tiling(int w, int h, int* tiles_x, int* tiles_y, int* tile_cap);
// With `taps` (only when orb_blur_fuses(taps): the 7-tap kernel) the same launch also writes the blurred image and orb_launch_blur is not needed.
bool orb_blur_fuses(const OrbTaps& taps);
// blur_tab / blur_c2 (orb_blur_mfma_table, 192 entries in device memory): the fused blur runs on the matrix cores; null: on the vector ALUs.
bool orb_blur_mfma_table(const OrbTaps& taps, unsigned long long* tab192, int* c2);
void orb_launch_fast(const uint8_t* img, int w, int h, int stride, size_t frame_stride, int n_frames, int threshold, int border, uint8_t* raw_frame0, int wp,
                     int2* raw, int* tile_count, const OrbTaps* taps, const unsigned long long* blur_tab, int blur_c2, uint8_t* blurred, hipStream_t st);
// cv::resize(INTER_LINEAR) of n_frames u8 images (OpenCV 3.4.0 fixed-point arithmetic) and the per-frame concatenation of a level's results
void orb_launch_resize(const uint8_t* src, int sw, int sh, int sstride, size_t sframe, uint8_t* dst, int dw, int dh, int dpitch, size_t dframe, int n_frames,
                       hipStream_t st);
void orb_launch_append_level(const mage_keypoint* kp_l, const uint8_t* desc_l, const int* count_l, int cap_l, mage_keypoint* kp, uint8_t* desc, int* count,
                             int capacity, int n_frames, float scale, float size, int level, hipStream_t st);
void orb_launch_select(const OrbSelectArgs& a, int n_frames, hipStream_t st);
void orb_launch_blur(const uint8_t* img, int w, int h, int stride, size_t frame_stride, int n_frames, const OrbTaps& taps, uint8_t* out, int wp, hipStream_t st);
struct OrbUmax { int half; int umax[18]; };         // row extents of the orientation disc, half <= 15 (OpenCVModified.cpp:672-688)
void orb_launch_angles(const uint8_t* img, int stride, size_t frame_stride, int n_frames, mage_keypoint* kps, const int* counts, int capacity,
                       const OrbUmax& um, hipStream_t st);
void orb_launch_brief(const uint8_t* blurred, int wp, int h, int n_frames, const mage_keypoint* kps, const int* counts, int capacity,
                      const signed char* pattern, int pattern_radius /* largest |coordinate| in the table */, uint8_t* desc, bool rotate_random, hipStream_t st);

// Hamming brute-force two-way matcher: one workgroup per pair.
void match_init_device();   // once per device: opt k_match into its LDS staging size
void match_launch(int n_pairs, const uint8_t* descA, const int* countsA, int capA, const uint8_t* descB, const int* countsB, int capB,
                  int max_dist, int min_diff, int* scratch, mage_dmatch* out, int cap_out, int* counts, int* done, hipStream_t st);

// RadiusMatch: one (query set, target set) problem per launch.
// cv::undistortPoints constants in float64 (camera matrix entries, 1/fx, 1/fy, k1 k2 p1 p2 k3 k4 k5 k6, P row-major)
struct UndistortConsts { double cx, cy, ifx, ify, k[8], RR[9]; };
void undistort_launch(mage_keypoint* kp, const int* counts, int n_frames, int capacity, int count_single, const UndistortConsts& U, hipStream_t st);

void radius_match_launch(const mage_keypoint* qk, int nq, const float2* qpos, const uint8_t* qmask, const uint8_t* qdesc, const mage_keypoint* tk,
                         int nt, const uint8_t* tmask, const uint8_t* tdesc, float radius, int max_dist, int min_diff, int* scratch,
                         mage_dmatch* out, int cap, int* count, hipStream_t st);
void indexed_match_launch(const uint8_t* descA, int nA, const uint8_t* maskA, const int* cb_off, const int* cb, const uint8_t* descB,
                          const uint8_t* maskB, const int* ca_off, const int* ca, int max_dist, int min_diff, mage_dmatch* out, int cap, int* count,
                          hipStream_t st, const int* leafA = nullptr, const int* leafB = nullptr);      // leafA / leafB: the lists are per vocabulary node, looked up through the descriptor's leaf
// one position of a vocabulary tree's concatenated child lists as the leaf lookup walks it (match_kernels.hip: k_bow_find_leaf)
struct BowWalkEntry { int child, k0, k1, pad; unsigned long long medoid[4]; };
static_assert(sizeof(BowWalkEntry) == 48, "walk entry");
void bow_walk_fill(const uint8_t* node_descriptors, const int32_t* child_offsets, const int32_t* children, int n_nodes, BowWalkEntry* dst);
void bow_find_leaf_launch(const BowWalkEntry* walk, int root_k0, int root_k1, const uint8_t* queries, int nq, int* leaf, hipStream_t st);

}  // namespace mage
