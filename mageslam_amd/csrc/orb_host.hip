// orb_host.hip -- host side of the ORB front-end and the Hamming matcher: the mage_orb_* / mage_match_* C ABI.
//
// Mirrors OrbDetector::DetectAndCompute (Core/MAGESLAM/Source/Image/OpenCVModified.cpp:771-886) as a fixed
// sequence of kernel launches per batch of frames: FAST score map -> NMS + raster compaction -> selection
// (RetainBestFeatures + ANMS) -> Gaussian blur -> BRIEF; and Match (Tracking/FeatureMatcher.cpp:61-190) as one
// launch per batch of pairs.  The host computes nothing but the blur taps and the rotated sampling pattern.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/mage_brief_patterns.h"
#include "mage_common.h"
#include "orb_kernels.h"

using namespace mage;

namespace {

template <typename F>
mage_status guarded(F&& f)
{
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(MAGE_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    catch (const std::exception& e) { return fail(MAGE_ERR_DEVICE, "unexpected exception: %s", e.what()); }
    catch (...) { return fail(MAGE_ERR_DEVICE, "unexpected exception"); }
}

// cvRound(256 * getGaussianKernel(k, 2, CV_32F)[i]) -- OpenCV 3.4.0's 8-bit separable path
OrbTaps gaussian_taps(unsigned ksize)
{
    OrbTaps t{};
    if (ksize <= 1) { t.radius = 0; t.t[0] = 256; return t; }
    float cf[15];
    double sum = 0;
    const double sigma = 2.0, scale2X = -0.5 / (sigma * sigma);
    for (unsigned i = 0; i < ksize; ++i) {
        const double x = i - (ksize - 1) * 0.5;
        cf[i] = (float)std::exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1.0 / sum;
    for (unsigned i = 0; i < ksize; ++i) { cf[i] = (float)(cf[i] * sum); t.t[i] = (int)std::nearbyint((double)cf[i] * 256.0); }
    t.radius = (int)ksize / 2;
    return t;
}

// rotation rows of the pre-rotated sampling pattern (include/mage_brief_patterns.h explains the rule)
void expand_pattern(unsigned patch, std::vector<signed char>& out)
{
    out.assign(MAGE_BRIEF_ROTATIONS * 1024, 0);
    if (patch != 15 && patch != 31) {
        // MakeRandomPattern (OpenCVModified.cpp:551-560): 512 points from cv::RNG(0x34985739), OpenCV's multiply-with-carry generator,
        // uniform(-patch/2, patch/2 + 1) for x then y.  Only the unrotated row is filled: orientation is refused with this pattern.
        uint64_t state = 0x34985739u;
        const int a = -(int)patch / 2, b = (int)patch / 2 + 1;
        for (int i = 0; i < 1024; ++i) {
            state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
            out[i] = (signed char)(a == b ? a : (int)((uint32_t)state % (uint32_t)(b - a) + (uint32_t)a));
        }
        return;
    }
    const signed char* base = patch == 31 ? MAGE_BRIEF_BASE_31 : MAGE_BRIEF_BASE_15;
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < MAGE_BRIEF_ROTATIONS; ++k) {
        const double a = k * 12.0 * pi / 180.0, c = std::cos(a), s = std::sin(a);
        for (int p = 0; p < 512; ++p) {
            const double bx = base[p * 2], by = base[p * 2 + 1];
            double v[2] = { bx * c - by * s, bx * s + by * c };
            for (int q = 0; q < 2; ++q) {
                const double hh = std::nearbyint(v[q] * 2.0) / 2.0;
                if (std::fabs(v[q] - hh) < 1e-9) v[q] = hh;
                out[k * 1024 + p * 2 + q] = (signed char)std::nearbyint(v[q]);
            }
        }
    }
}

}  // namespace

struct mage_orb {
    int device = 0;
    hipStream_t stream = nullptr;
    mage_orb_params P{};
    OrbTaps taps{};
    DevBuf<signed char> d_pattern;
    DevBuf<unsigned long long> d_blur_tab;   // tap matrices of the matrix-core blur (orb_blur_mfma_table); empty: the taps do not fit it
    int blur_c2 = 0;
    bool blur_on_matrix_cores = false;
    int pattern_radius = 0;         // largest |coordinate| of the sampling table
    DevBuf<uint8_t> d_img, d_rawscore, d_blur, d_desc;     // d_rawscore: FAST scores of frame 0 (parity tests)
    DevBuf<int> d_tile_count, d_cell_start, d_cell_fill, d_count;   // d_tile_count: keypoints per FAST tile, d_raw: their slots
    DevBuf<int2> d_raw;
    DevBuf<unsigned long long> d_cand, d_key;                  // scratch of k_select for frames whose working set exceeds LDS
    DevBuf<mage_keypoint> d_kp, d_undist, d_kp_lvl;
    DevBuf<uint8_t> d_pyr[2], d_blur_lvl, d_desc_lvl;      // pyramid levels >= 1 (ping-pong), their blurred image and per-level outputs
    DevBuf<int> d_count_lvl;
    hipEvent_t ev[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    mage_orb_profile prof{};
    bool profiling = false;          // stage events are recorded only on request: five event packets cost more than a one-frame batch's kernels are apart
    int last_w = 0, last_h = 0;
    ~mage_orb()
    {
        DeviceScope scope(device);
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        cached_stream_release(device, stream);
    }
};

MAGE_EXPORT mage_status mage_orb_default_params(mage_orb_params* p)
{
    if (!p) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
    // FeatureExtractorSettings defaults, Core/MAGESLAM/Source/MageSettings.h:151-167
    *p = mage_orb_params{ 7, 440, 1.5f, 1, 15, 4, 0, 1.5f, 0.9f, 20, 1.1f, 2.0f, 32, 32, -1 };
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_orb_create(const mage_orb_params* params, mage_orb** out)
{
    return guarded([&]() -> mage_status {
        if (!out || !params) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        *out = nullptr;
        const mage_orb_params& p = *params;
        if (p.nlevels < 1 || p.nlevels > ORB_MAX_LEVELS) return fail(MAGE_ERR_INVALID_ARGUMENT, "NumLevels = %u: 1 .. %d supported", p.nlevels, ORB_MAX_LEVELS);
        if (p.nlevels > 1 && !(p.scale_factor > 1.0f)) return fail(MAGE_ERR_INVALID_ARGUMENT, "a pyramid needs ScaleFactor > 1");
        if (p.patch_size != 15 && p.patch_size != 31) {
            // the random pattern of ComputeOrbDescriptors (:452-492, :551-560): with angle 0 its rotation is the identity; with
            // UseOrientation every keypoint rotates the points by its own angle (k_brief_rotated)
            if (p.patch_size < 2 || p.patch_size > 127) return fail(MAGE_ERR_INVALID_ARGUMENT, "patch size %u out of range 2 .. 127", p.patch_size);
        }
        if (p.gaussian_kernel_size > 15 || (p.gaussian_kernel_size > 1 && p.gaussian_kernel_size % 2 == 0))
            return fail(MAGE_ERR_INVALID_ARGUMENT, "Gaussian kernel size must be odd and <= 15");
        if (p.nfeatures < 2 || p.num_cells_x < 1 || p.num_cells_y < 1 || p.fast_threshold < 1)
            return fail(MAGE_ERR_INVALID_ARGUMENT, "nfeatures >= 2, cells >= 1 and FAST threshold >= 1 are required (asserts at OpenCVModified.cpp:177-178, :582)");
        int dev = 0;
        MAGE_TRY(select_device(p.device, &dev));
        std::unique_ptr<mage_orb> h(new mage_orb());
        h->device = dev; h->P = p;
        h->taps = gaussian_taps(p.gaussian_kernel_size);
        MAGE_DEVICE_SCOPE(dev);
        MAGE_TRY(cached_stream_acquire(dev, &h->stream));
        for (auto& e : h->ev) MAGE_HIP(hipEventCreate(&e));
        std::vector<signed char> pat;
        expand_pattern(p.patch_size, pat);
        // the table rows a keypoint can select: row 0 only without orientation (the patch then stays inside RunByImageBorder's margin)
        const size_t used = p.use_orientation ? pat.size() : std::min<size_t>(pat.size(), 1024);
        for (size_t i = 0; i < used; ++i) h->pattern_radius = std::max(h->pattern_radius, std::abs((int)pat[i]));
        MAGE_TRY(h->d_pattern.upload(pat.data(), pat.size(), h->stream));
        unsigned long long blur_tab[192];
        const char* blur_env = std::getenv("MAGE_ORB_BLUR");           // "valu": the vector-ALU form of the fused blur (A/B tests)
        if (!(blur_env && std::strcmp(blur_env, "valu") == 0) && orb_blur_mfma_table(h->taps, blur_tab, &h->blur_c2)) {
            MAGE_TRY(h->d_blur_tab.upload(blur_tab, 192, h->stream));
            h->blur_on_matrix_cores = true;
        }
        MAGE_HIP(hipStreamSynchronize(h->stream));
        *out = h.release();
        return MAGE_OK;
    });
}

MAGE_EXPORT void mage_orb_destroy(mage_orb* h) { delete h; }

namespace {

// Where one level's stages leave their results and with which quota.
struct LevelIO {
    int nfeatures;            // quota of the level (ComputeKeyPoints: nfeaturesPerLevel)
    int capacity;             // records per frame in kp / desc
    mage_keypoint* kp; uint8_t* desc; int* count;
    uint8_t* blur;            // blurred level image (pitch wp)
    uint8_t* raw_frame0;      // FAST scores of frame 0 (level 0 only; parity tests) or null
    bool record_events;
};

// runs the five stages on n_frames images of one pyramid level that are already in HBM
mage_status run_level(mage_orb* h, const uint8_t* d_images, int n_frames, int w, int h_img, int stride, size_t frame_stride, const LevelIO& io)
{
    const mage_orb_params& P = h->P;
    hipStream_t st = h->stream;
    const int wp = (w + 3) & ~3;                     // internal row pitch (score map, blurred image)
    int tiles_x, tiles_y, tile_cap;
    orb_fast_tiling(w, h_img, &tiles_x, &tiles_y, &tile_cap);
    const size_t n_tiles = (size_t)tiles_x * tiles_y;
    const size_t scratch_cap = (size_t)w * h_img / 4 + 16;      // a strict 3x3 maximum: at most one keypoint per 2x2 block
    const int ncells = P.num_cells_x * P.num_cells_y;
    const size_t nf = (size_t)std::max(n_frames, 1);
    MAGE_TRY(h->d_tile_count.reserve(nf * n_tiles));
    MAGE_TRY(h->d_raw.reserve(nf * n_tiles * tile_cap));
    MAGE_TRY(h->d_cand.reserve(nf * scratch_cap));
    MAGE_TRY(h->d_key.reserve(nf * scratch_cap));
    MAGE_TRY(h->d_cell_start.reserve(nf * (ncells + 1)));
    MAGE_TRY(h->d_cell_fill.reserve(nf * (ncells + 1)));

    if (io.record_events) MAGE_HIP(hipEventRecord(h->ev[0], st));
    // RunByImageBorder: half the patch, or its hypotenuse when the patch gets rotated (OpenCVModified.cpp:709-712)
    const int half_patch = (int)P.patch_size / 2;
    const int border = P.use_orientation ? (int)std::ceil((float)half_patch * std::sqrt(2.0f)) : half_patch;
    const bool fused_blur = orb_blur_fuses(h->taps);      // the 7-tap Gaussian rides in the FAST launch: the frame is read once for both
    orb_launch_fast(d_images, w, h_img, stride, frame_stride, n_frames, (int)std::min(P.fast_threshold, 255u), border, io.raw_frame0, wp,
                    h->d_raw.p, h->d_tile_count.p, fused_blur ? &h->taps : nullptr, h->blur_on_matrix_cores ? h->d_blur_tab.p : nullptr, h->blur_c2, io.blur, st);
    if (io.record_events) MAGE_HIP(hipEventRecord(h->ev[1], st));
    OrbSelectArgs a{};
    a.raw = h->d_raw.p; a.tile_count = h->d_tile_count.p; a.n_tiles = (int)n_tiles; a.tile_cap = tile_cap;
    a.cand64 = h->d_cand.p; a.key64 = h->d_key.p; a.scratch_cap = scratch_cap;
    a.cell_start = h->d_cell_start.p; a.cell_fill = h->d_cell_fill.p;
    a.out_kp = io.kp; a.out_count = io.count;
    a.ncells = ncells; a.cells_x = P.num_cells_x; a.cells_y = P.num_cells_y;
    a.nfeatures = io.nfeatures; a.max_num = (int)((float)io.nfeatures * P.feature_factor_anms);
    a.fast_threshold = (int)P.fast_threshold; a.strong_response = P.strong_response_anms; a.capacity = io.capacity; a.patch_size = (int)P.patch_size;
    a.feature_strength = P.feature_strength_anms; a.min_robust = P.min_robust_factor; a.max_robust = P.max_robust_factor;
    orb_launch_select(a, n_frames, st);
    if (P.use_orientation && io.capacity > 0) {          // ICAngles on the unblurred image (:745-748)
        OrbUmax um{};
        um.half = half_patch;
        const int vmax = (int)std::floor((float)half_patch * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil((float)half_patch * std::sqrt(2.f) / 2);
        for (int v = 0; v <= vmax; ++v) um.umax[v] = (int)std::nearbyint(std::sqrt((double)half_patch * half_patch - (double)v * v));
        for (int v = half_patch, v0 = 0; v >= vmin; --v) { while (um.umax[v0] == um.umax[v0 + 1]) ++v0; um.umax[v] = v0; ++v0; }
        orb_launch_angles(d_images, stride, frame_stride, n_frames, io.kp, io.count, io.capacity, um, st);
    }
    if (io.record_events) MAGE_HIP(hipEventRecord(h->ev[2], st));
    if (!fused_blur) orb_launch_blur(d_images, w, h_img, stride, frame_stride, n_frames, h->taps, io.blur, wp, st);
    if (io.record_events) MAGE_HIP(hipEventRecord(h->ev[3], st));
    if (io.capacity > 0) orb_launch_brief(io.blur, wp, h_img, n_frames, io.kp, io.count, io.capacity, h->d_pattern.p, h->pattern_radius, io.desc,
                                          P.use_orientation && P.patch_size != 15 && P.patch_size != 31, st);
    if (io.record_events) MAGE_HIP(hipEventRecord(h->ev[4], st));
    return MAGE_OK;
}

// DetectAndCompute on n_frames images that are already in HBM; leaves keypoints / descriptors / counts in HBM.
// One level: the stages write the final buffers directly.  A pyramid (OpenCVModified.cpp:793-841): level l is cv::resize of level
// l - 1, every level runs the same stages with its own quota into level buffers, and k_append_level concatenates the levels per
// frame (ImageData::Insert), scaling the coordinates by the level's scale (:756-760).
mage_status run_batch(mage_orb* h, const uint8_t* d_images, int n_frames, int w, int h_img, int stride, size_t frame_stride, int capacity)
{
    const mage_orb_params& P = h->P;
    if (w < 1 || h_img < 1 || w > 65535 || h_img > 32767) return fail(MAGE_ERR_INVALID_ARGUMENT, "image size %dx%d out of range", w, h_img);
    if (capacity < 0 || n_frames < 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "negative capacity / frame count");
    hipStream_t st = h->stream;
    const size_t nf = (size_t)std::max(n_frames, 1), cap = (size_t)std::max(capacity, 1);
    const int wp0 = (w + 3) & ~3;
    MAGE_TRY(h->d_blur.reserve(nf * (size_t)wp0 * h_img));
    MAGE_TRY(h->d_rawscore.reserve((size_t)wp0 * h_img));
    MAGE_TRY(h->d_kp.reserve(nf * cap));
    MAGE_TRY(h->d_desc.reserve(nf * cap * 32));
    MAGE_TRY(h->d_count.reserve(nf));
    h->last_w = w; h->last_h = h_img;
    h->prof = mage_orb_profile{};
    h->prof.n_frames = n_frames;
    if (n_frames == 0) return MAGE_OK;
    const int L = (int)P.nlevels;
    if (L == 1) {
        LevelIO io{ (int)P.nfeatures, capacity, h->d_kp.p, h->d_desc.p, h->d_count.p, h->d_blur.p, h->d_rawscore.p, h->profiling };
        return run_level(h, d_images, n_frames, w, h_img, stride, frame_stride, io);
    }
    // pyramid layout (:564-567, :797-799) and per-level quotas (:659-669), in the reference's float arithmetic
    int lw[ORB_MAX_LEVELS], lh[ORB_MAX_LEVELS], quota[ORB_MAX_LEVELS];
    float lscale[ORB_MAX_LEVELS];
    for (int l = 0; l < L; ++l) {
        lscale[l] = (float)std::pow((double)P.scale_factor, (double)l);
        lw[l] = (int)std::nearbyint((float)w / lscale[l]); lh[l] = (int)std::nearbyint((float)h_img / lscale[l]);
    }
    {
        const float factor = 1.0f / P.scale_factor;
        float ndesired = (float)P.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)L));
        int sum = 0;
        for (int l = 0; l < L - 1; ++l) { quota[l] = (int)std::nearbyint(ndesired); sum += quota[l]; ndesired *= factor; }
        quota[L - 1] = std::max((int)P.nfeatures - sum, 0);
    }
    if (h->profiling) MAGE_HIP(hipEventRecord(h->ev[0], st));
    MAGE_HIP(hipMemsetAsync(h->d_count.p, 0, sizeof(int) * nf, st));
    const uint8_t* src = d_images; int sw = w, sh = h_img, sstride = stride; size_t sfs = frame_stride;
    for (int l = 0; l < L; ++l) {
        const uint8_t* img_l = src; int stride_l = sstride; size_t fs_l = sfs;
        if (l > 0) {
            if (lw[l] < 1 || lh[l] < 1) break;
            DevBuf<uint8_t>& dst = h->d_pyr[l & 1];
            const int pitch = (lw[l] + 3) & ~3;
            MAGE_TRY(dst.reserve(nf * (size_t)pitch * lh[l]));
            orb_launch_resize(src, sw, sh, sstride, sfs, dst.p, lw[l], lh[l], pitch, (size_t)pitch * lh[l], n_frames, st);
            img_l = dst.p; stride_l = pitch; fs_l = (size_t)pitch * lh[l];
        }
        if (lw[l] >= 7 && lh[l] >= 7 && quota[l] >= 1) {     // a level without quota places nothing (the reference would assert in ANMS)
            const int cap_l = quota[l];
            const int wp = (lw[l] + 3) & ~3;
            MAGE_TRY(h->d_kp_lvl.reserve(nf * (size_t)cap_l));
            MAGE_TRY(h->d_desc_lvl.reserve(nf * (size_t)cap_l * 32));
            MAGE_TRY(h->d_count_lvl.reserve(nf));
            uint8_t* blur = h->d_blur.p;
            if (l > 0) { MAGE_TRY(h->d_blur_lvl.reserve(nf * (size_t)wp * lh[l])); blur = h->d_blur_lvl.p; }
            LevelIO io{ quota[l], quota[l], h->d_kp_lvl.p, h->d_desc_lvl.p, h->d_count_lvl.p, blur, l == 0 ? h->d_rawscore.p : nullptr, false };
            MAGE_TRY(run_level(h, img_l, n_frames, lw[l], lh[l], stride_l, fs_l, io));
            orb_launch_append_level(h->d_kp_lvl.p, h->d_desc_lvl.p, h->d_count_lvl.p, quota[l], h->d_kp.p, h->d_desc.p, h->d_count.p, capacity, n_frames,
                                    lscale[l], (float)P.patch_size * lscale[l], l, st);
        }
        src = img_l; sw = lw[l]; sh = lh[l]; sstride = stride_l; sfs = fs_l;
    }
    if (h->profiling) for (int e = 1; e <= 4; ++e) MAGE_HIP(hipEventRecord(h->ev[e], st));     // stage split is not recorded for pyramids: total only
    return MAGE_OK;
}

mage_status collect_profile(mage_orb* h)
{
    if (!h->profiling) return MAGE_OK;
    float ms = 0;
    MAGE_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1])); h->prof.fast_ms = ms;
    MAGE_HIP(hipEventElapsedTime(&ms, h->ev[1], h->ev[2])); h->prof.select_ms = ms;
    MAGE_HIP(hipEventElapsedTime(&ms, h->ev[2], h->ev[3])); h->prof.blur_ms = ms;
    MAGE_HIP(hipEventElapsedTime(&ms, h->ev[3], h->ev[4])); h->prof.brief_ms = ms;
    MAGE_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[4])); h->prof.total_ms = ms;
    if (h->P.nlevels > 1) h->prof.fast_ms = h->prof.select_ms = h->prof.blur_ms = h->prof.brief_ms = 0;     // pyramids record the total only
    return MAGE_OK;
}

}  // namespace

MAGE_EXPORT mage_status mage_orb_detect_batch(mage_orb* h, const uint8_t* images, int images_on_device, int n_frames, int width, int height,
                                              int stride, size_t frame_stride, mage_keypoint* keypoints, uint8_t* descriptors32, int capacity,
                                              int* counts)
{
    return guarded([&]() -> mage_status {
        if (!h || (n_frames > 0 && (!images || !counts))) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (capacity > 0 && (!keypoints || !descriptors32)) return fail(MAGE_ERR_INVALID_ARGUMENT, "null output buffer");
        if (stride < width) return fail(MAGE_ERR_INVALID_ARGUMENT, "stride < width");
        MAGE_DEVICE_SCOPE(h->device);
        const uint8_t* dimg = images;
        if (!images_on_device && n_frames > 0) {
            const size_t bytes = (size_t)(n_frames - 1) * frame_stride + (size_t)(height - 1) * stride + width;
            MAGE_TRY(h->d_img.reserve(bytes));
            MAGE_HIP(hipMemcpyAsync(h->d_img.p, images, bytes, hipMemcpyHostToDevice, h->stream));
            dimg = h->d_img.p;
        }
        MAGE_TRY(run_batch(h, dimg, n_frames, width, height, stride, frame_stride, capacity));
        if (n_frames == 0) return MAGE_OK;
        MAGE_HIP(hipMemcpyAsync(counts, h->d_count.p, sizeof(int) * (size_t)n_frames, hipMemcpyDeviceToHost, h->stream));
        if (capacity > 0) {
            MAGE_HIP(hipMemcpyAsync(keypoints, h->d_kp.p, sizeof(mage_keypoint) * (size_t)n_frames * capacity, hipMemcpyDeviceToHost, h->stream));
            MAGE_HIP(hipMemcpyAsync(descriptors32, h->d_desc.p, (size_t)n_frames * capacity * 32, hipMemcpyDeviceToHost, h->stream));
        }
        MAGE_HIP(wait_stream_briefly_spinning(h->stream, h->ev[5]));
        return collect_profile(h);
    });
}

MAGE_EXPORT mage_status mage_orb_detect(mage_orb* h, const uint8_t* image, int width, int height, int stride, mage_keypoint* keypoints,
                                        uint8_t* descriptors32, int capacity, int* count)
{
    return mage_orb_detect_batch(h, image, 0, 1, width, height, stride, (size_t)stride * (size_t)(height > 0 ? height : 0), keypoints,
                                 descriptors32, capacity, count);
}

MAGE_EXPORT mage_status mage_orb_detect_batch_device(mage_orb* h, const uint8_t* images_device, int n_frames, int width, int height, int stride,
                                                     size_t frame_stride, int capacity, const mage_keypoint** keypoints_device,
                                                     const uint8_t** descriptors_device, const int** counts_device)
{
    return guarded([&]() -> mage_status {
        if (!h || !images_device || !keypoints_device || !descriptors_device || !counts_device) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (stride < width) return fail(MAGE_ERR_INVALID_ARGUMENT, "stride < width");
        MAGE_DEVICE_SCOPE(h->device);
        MAGE_TRY(run_batch(h, images_device, n_frames, width, height, stride, frame_stride, capacity));
        MAGE_HIP(wait_stream_briefly_spinning(h->stream, h->ev[5]));
        *keypoints_device = h->d_kp.p; *descriptors_device = h->d_desc.p; *counts_device = h->d_count.p;
        return n_frames > 0 ? collect_profile(h) : MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_orb_enable_profile(mage_orb* h, int on)
{
    if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
    h->profiling = on != 0;
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_orb_debug_read(mage_orb* h, uint8_t* score_map, uint8_t* blurred)
{
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
        const size_t w = (size_t)h->last_w, rows = (size_t)h->last_h, wp = (w + 3) & ~(size_t)3;
        if (w * rows == 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "no frame has been processed yet");
        MAGE_DEVICE_SCOPE(h->device);
        if (score_map) MAGE_HIP(hipMemcpy2D(score_map, w, h->d_rawscore.p, wp, w, rows, hipMemcpyDeviceToHost));
        if (blurred) MAGE_HIP(hipMemcpy2D(blurred, w, h->d_blur.p, wp, w, rows, hipMemcpyDeviceToHost));
        return MAGE_OK;
    });
}

static mage_status undistort_consts(const mage_undistort_params* p, UndistortConsts* U)
{
    if (!p) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
    if (p->n_dist != 4 && p->n_dist != 5 && p->n_dist != 8)
        return fail(p->n_dist == 12 || p->n_dist == 14 ? MAGE_ERR_UNSUPPORTED : MAGE_ERR_INVALID_ARGUMENT,
                    "%d distortion coefficients: the reference's calibration models carry 5 (Poly3k) or 8 (Rational6k)", p->n_dist);
    const double fx = (double)p->camera_matrix[0], fy = (double)p->camera_matrix[4];
    if (!(fx != 0.0) || !(fy != 0.0)) return fail(MAGE_ERR_INVALID_ARGUMENT, "camera matrix has a zero focal length");
    U->cx = (double)p->camera_matrix[2]; U->cy = (double)p->camera_matrix[5];
    U->ifx = 1. / fx; U->ify = 1. / fy;
    for (int i = 0; i < 8; ++i) U->k[i] = i < p->n_dist ? (double)p->dist_coeffs[i] : 0.0;
    for (int i = 0; i < 9; ++i) U->RR[i] = (double)p->new_camera_matrix[i];          // P * R with R = identity
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_orb_undistort_keypoints(mage_orb* h, mage_keypoint* keypoints, int count, const mage_undistort_params* params)
{
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
        UndistortConsts U;
        MAGE_TRY(undistort_consts(params, &U));
        if (count < 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "negative count");
        if (count == 0) return MAGE_OK;                                              // OrbFeatureDetector.cpp:39-42
        if (!keypoints) return fail(MAGE_ERR_INVALID_ARGUMENT, "null buffer");
        MAGE_DEVICE_SCOPE(h->device);
        MAGE_TRY(h->d_undist.reserve((size_t)count));
        MAGE_HIP(hipMemcpyAsync(h->d_undist.p, keypoints, sizeof(mage_keypoint) * (size_t)count, hipMemcpyHostToDevice, h->stream));
        undistort_launch(h->d_undist.p, nullptr, 1, count, count, U, h->stream);
        MAGE_HIP(hipMemcpyAsync(keypoints, h->d_undist.p, sizeof(mage_keypoint) * (size_t)count, hipMemcpyDeviceToHost, h->stream));
        MAGE_HIP(hipStreamSynchronize(h->stream));
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_orb_undistort_keypoints_device(mage_orb* h, const mage_keypoint* keypoints_device, const int* counts_device, int n_frames,
                                                            int capacity, const mage_undistort_params* params)
{
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
        UndistortConsts U;
        MAGE_TRY(undistort_consts(params, &U));
        if (n_frames < 0 || capacity < 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "negative size");
        if (n_frames == 0 || capacity == 0) return MAGE_OK;
        if (!keypoints_device || !counts_device) return fail(MAGE_ERR_INVALID_ARGUMENT, "null buffer");
        MAGE_DEVICE_SCOPE(h->device);
        // the buffers belong to the handle (mage_orb_detect_batch_device hands them out as const views)
        undistort_launch(const_cast<mage_keypoint*>(keypoints_device), counts_device, n_frames, capacity, 0, U, h->stream);
        MAGE_HIP(hipGetLastError());
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_orb_get_profile(const mage_orb* h, mage_orb_profile* out)
{
    if (!h || !out) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
    *out = h->prof;
    return MAGE_OK;
}

// ================================================================================================
// matcher
// ================================================================================================
struct mage_matcher {
    int device = 0;
    hipStream_t stream = nullptr;
    DevBuf<uint8_t> d_A, d_B;
    DevBuf<int> d_cA, d_cB, d_scratch, d_counts, d_done;       // d_done: per-pair arrival counters of k_match_rows, zero between launches
    size_t done_zeroed = 0;
    DevBuf<mage_dmatch> d_out;
    DevBuf<uint8_t> d_tree;                    // mage_bow_set_tree: the validated tree as the lookup walks it (BowWalkEntry per child-list position)
    size_t tree_nodes = 0;
    int tree_root_k1 = 0;                      // the root's children are positions 0 .. tree_root_k1 - 1
    PinnedVec<uint8_t> h_io;                   // lookups with the tree kept on the device: [descriptors | leaf ids] in pinned memory the kernel reads and writes itself
    hipEvent_t e0 = nullptr, e1 = nullptr, e_wait = nullptr;
    double last_ms = 0;
    ~mage_matcher()
    {
        DeviceScope scope(device);
        if (stream) (void)hipStreamSynchronize(stream);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (e_wait) (void)hipEventDestroy(e_wait);
        cached_stream_release(device, stream);
    }
};

// FeatureMatcher.cpp:489-500 (8 x 32-bit SWAR popcount); one pair of descriptors is host work
MAGE_EXPORT int mage_hamming256(const uint8_t* d0, const uint8_t* d1)
{
    int result = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t a, b;
        std::memcpy(&a, d0 + 4 * i, 4); std::memcpy(&b, d1 + 4 * i, 4);
        uint32_t bits = a ^ b;
        bits = bits - ((bits >> 1) & 0x55555555u);
        bits = (bits & 0x33333333u) + ((bits >> 2) & 0x33333333u);
        result += (int)((((bits + (bits >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24);
    }
    return result;
}

MAGE_EXPORT mage_status mage_matcher_create(int device, mage_matcher** out)
{
    return guarded([&]() -> mage_status {
        if (!out) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        *out = nullptr;
        int dev = 0;
        MAGE_TRY(select_device(device, &dev));
        std::unique_ptr<mage_matcher> h(new mage_matcher());
        h->device = dev;
        MAGE_DEVICE_SCOPE(dev);
        MAGE_TRY(cached_stream_acquire(dev, &h->stream));
        MAGE_HIP(hipEventCreate(&h->e0)); MAGE_HIP(hipEventCreate(&h->e1)); MAGE_HIP(hipEventCreateWithFlags(&h->e_wait, hipEventDisableTiming));
        match_init_device();
        *out = h.release();
        return MAGE_OK;
    });
}

MAGE_EXPORT void mage_matcher_destroy(mage_matcher* h) { delete h; }

namespace {

// out_over / counts_over: where the kernel writes matches and counts instead of the handle's device buffers (pinned host memory the
// device can address: a small call then needs no copy back)
mage_status run_match(mage_matcher* h, int n_pairs, const uint8_t* dA, const int* dcA, int capA, const uint8_t* dB, const int* dcB, int capB,
                      int max_dist, int min_diff, int cap_out, mage_dmatch* out_over = nullptr, int* counts_over = nullptr)
{
    if (n_pairs < 0 || capA < 0 || capB < 0 || cap_out < 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "negative size");
    const size_t np = (size_t)std::max(n_pairs, 1);
    MAGE_TRY(h->d_scratch.reserve(np * (size_t)(capA + capB) * 2 + 4));
    MAGE_TRY(h->d_out.reserve(np * (size_t)std::max(cap_out, 1)));
    MAGE_TRY(h->d_counts.reserve(np));
    MAGE_TRY(h->d_done.reserve(np));
    if (h->done_zeroed != h->d_done.cap) {               // a fresh (recycled) allocation: zero once, the kernel keeps the counters at zero
        MAGE_HIP(hipMemsetAsync(h->d_done.p, 0, sizeof(int) * h->d_done.cap, h->stream));
        h->done_zeroed = h->d_done.cap;
    }
    if (n_pairs == 0) return MAGE_OK;
    MAGE_HIP(hipEventRecord(h->e0, h->stream));
    match_launch(n_pairs, dA, dcA, capA, dB, dcB, capB, max_dist, min_diff, h->d_scratch.p, out_over ? out_over : h->d_out.p, cap_out, counts_over ? counts_over : h->d_counts.p, h->d_done.p, h->stream);
    MAGE_HIP(hipEventRecord(h->e1, h->stream));
    return MAGE_OK;
}

}  // namespace

MAGE_EXPORT mage_status mage_match_bf_batch(mage_matcher* h, int n_pairs, const uint8_t* descA, const int* countsA, int capA,
                                            const uint8_t* descB, const int* countsB, int capB, int max_dist, int min_diff,
                                            mage_dmatch* out, int cap_out, int* counts)
{
    return guarded([&]() -> mage_status {
        if (!h || (n_pairs > 0 && (!countsA || !countsB || !counts))) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (n_pairs > 0 && ((capA > 0 && !descA) || (capB > 0 && !descB) || (cap_out > 0 && !out))) return fail(MAGE_ERR_INVALID_ARGUMENT, "null buffer");
        for (int p = 0; p < n_pairs; ++p)
            if (countsA[p] < 0 || countsA[p] > capA || countsB[p] < 0 || countsB[p] > capB) return fail(MAGE_ERR_INVALID_ARGUMENT, "pair %d: count exceeds capacity", p);
        MAGE_DEVICE_SCOPE(h->device);
        const size_t np = (size_t)std::max(n_pairs, 1);
        MAGE_TRY(h->d_A.reserve(np * (size_t)capA * 32 + 32)); MAGE_TRY(h->d_B.reserve(np * (size_t)capB * 32 + 32));
        MAGE_TRY(h->d_cA.reserve(np)); MAGE_TRY(h->d_cB.reserve(np));
        if (n_pairs == 0) return MAGE_OK;
        hipStream_t st = h->stream;
        // A SMALL call (the tracker's one pair per frame: 28 KB in, 7 KB out) is bound by its copy commands, not by the 16-us kernel: both
        // sets and the counts go up as ONE copy out of a pinned block, matches and counts are written by the kernel into that block
        const size_t bA = (size_t)n_pairs * capA * 32, bB = (size_t)n_pairs * capB * 32, bC = sizeof(int) * (size_t)n_pairs, bO = sizeof(mage_dmatch) * (size_t)n_pairs * cap_out;
        if (bA + bB + bO <= (size_t)256 * 1024) {
            auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
            const size_t o_b = al(bA), o_ca = al(o_b + bB), o_cb = al(o_ca + bC), up = al(o_cb + bC), o_cnt = up, o_out = al(o_cnt + bC), total = o_out + bO + 64;
            MAGE_TRY(h->h_io.resize_uninitialized(total));
            void* dio = nullptr;
            if (hipHostGetDevicePointer(&dio, h->h_io.data(), 0) == hipSuccess) {
                uint8_t* hp = h->h_io.data();
                if (bA) std::memcpy(hp, descA, bA);
                if (bB) std::memcpy(hp + o_b, descB, bB);
                std::memcpy(hp + o_ca, countsA, bC); std::memcpy(hp + o_cb, countsB, bC);
                MAGE_TRY(h->d_A.reserve(up + 64));
                MAGE_HIP(hipMemcpyAsync(h->d_A.p, hp, up, hipMemcpyHostToDevice, st));
                uint8_t* dp = static_cast<uint8_t*>(dio);
                MAGE_TRY(run_match(h, n_pairs, h->d_A.p, reinterpret_cast<const int*>(h->d_A.p + o_ca), capA, h->d_A.p + o_b, reinterpret_cast<const int*>(h->d_A.p + o_cb), capB,
                                   max_dist, min_diff, cap_out, reinterpret_cast<mage_dmatch*>(dp + o_out), reinterpret_cast<int*>(dp + o_cnt)));
                MAGE_HIP(wait_stream_briefly_spinning(st, h->e_wait));
                std::memcpy(counts, hp + o_cnt, bC);
                if (bO) std::memcpy(out, hp + o_out, bO);
                float ms = 0;
                MAGE_HIP(hipEventElapsedTime(&ms, h->e0, h->e1));
                h->last_ms = ms;
                return MAGE_OK;
            }
            (void)hipGetLastError();          // pinned memory this device cannot address: the copies below
        }
        if (capA) MAGE_HIP(hipMemcpyAsync(h->d_A.p, descA, (size_t)n_pairs * capA * 32, hipMemcpyHostToDevice, st));
        if (capB) MAGE_HIP(hipMemcpyAsync(h->d_B.p, descB, (size_t)n_pairs * capB * 32, hipMemcpyHostToDevice, st));
        MAGE_HIP(hipMemcpyAsync(h->d_cA.p, countsA, sizeof(int) * (size_t)n_pairs, hipMemcpyHostToDevice, st));
        MAGE_HIP(hipMemcpyAsync(h->d_cB.p, countsB, sizeof(int) * (size_t)n_pairs, hipMemcpyHostToDevice, st));
        MAGE_TRY(run_match(h, n_pairs, h->d_A.p, h->d_cA.p, capA, h->d_B.p, h->d_cB.p, capB, max_dist, min_diff, cap_out));
        MAGE_HIP(hipMemcpyAsync(counts, h->d_counts.p, sizeof(int) * (size_t)n_pairs, hipMemcpyDeviceToHost, st));
        if (cap_out) MAGE_HIP(hipMemcpyAsync(out, h->d_out.p, sizeof(mage_dmatch) * (size_t)n_pairs * cap_out, hipMemcpyDeviceToHost, st));
        MAGE_HIP(wait_stream_briefly_spinning(st, h->e_wait));
        float ms = 0;
        MAGE_HIP(hipEventElapsedTime(&ms, h->e0, h->e1));
        h->last_ms = ms;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_match_bf(mage_matcher* h, const uint8_t* descA, int nA, const uint8_t* descB, int nB, int max_dist, int min_diff,
                                      mage_dmatch* out, int capacity, int* count)
{
    if (!count) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
    // FeatureMatcher.cpp:72-77: an empty side yields no matches
    if (nA <= 0 || nB <= 0) { *count = 0; return h ? MAGE_OK : fail(MAGE_ERR_INVALID_ARGUMENT, "null handle"); }
    mage_status s = mage_match_bf_batch(h, 1, descA, &nA, nA, descB, &nB, nB, max_dist, min_diff, out, capacity, count);
    return s;
}

MAGE_EXPORT mage_status mage_match_masked(mage_matcher* h, const uint8_t* descA, int nDescA, const uint8_t* maskA, const uint8_t* descB,
                                          int nDescB, const uint8_t* maskB, int max_dist, int min_diff, mage_dmatch* out, int capacity,
                                          int* count)
{
    return guarded([&]() -> mage_status {
        if (!h || !count || (nDescA > 0 && !descA) || (nDescB > 0 && !descB)) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        // gather the unmasked descriptors in ascending original index (FeatureMatcher.cpp:88-110)
        std::vector<uint8_t> A, B;
        std::vector<int> ia, ib;
        for (int i = 0; i < nDescA; ++i) if (!maskA || maskA[i]) { ia.push_back(i); A.insert(A.end(), descA + (size_t)i * 32, descA + (size_t)i * 32 + 32); }
        for (int i = 0; i < nDescB; ++i) if (!maskB || maskB[i]) { ib.push_back(i); B.insert(B.end(), descB + (size_t)i * 32, descB + (size_t)i * 32 + 32); }
        *count = 0;
        if (ia.empty() || ib.empty()) return MAGE_OK;
        std::vector<mage_dmatch> tmp(ia.size());
        int n = 0;
        MAGE_TRY(mage_match_bf(h, A.data(), (int)ia.size(), B.data(), (int)ib.size(), max_dist, min_diff, tmp.data(), (int)tmp.size(), &n));
        for (int i = 0; i < n; ++i) {
            if (out && i < capacity) { out[i] = tmp[i]; out[i].queryIdx = ia[tmp[i].queryIdx]; out[i].trainIdx = ib[tmp[i].trainIdx]; }   // :160-163
        }
        *count = n;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_match_bf_batch_device(mage_matcher* h, int n_pairs, const uint8_t* descA_dev, const int* countsA_dev, int capA,
                                                   const uint8_t* descB_dev, const int* countsB_dev, int capB, int max_dist, int min_diff,
                                                   int cap_out, const mage_dmatch** out_dev, const int** counts_dev)
{
    return guarded([&]() -> mage_status {
        if (!h || !descA_dev || !descB_dev || !countsA_dev || !countsB_dev || !out_dev || !counts_dev) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        MAGE_DEVICE_SCOPE(h->device);
        MAGE_TRY(run_match(h, n_pairs, descA_dev, countsA_dev, capA, descB_dev, countsB_dev, capB, max_dist, min_diff, cap_out));
        MAGE_HIP(wait_stream_briefly_spinning(h->stream, h->e_wait));
        if (n_pairs > 0) { float ms = 0; MAGE_HIP(hipEventElapsedTime(&ms, h->e0, h->e1)); h->last_ms = ms; }
        *out_dev = h->d_out.p; *counts_dev = h->d_counts.p;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_match_radius(mage_matcher* h, const mage_keypoint* qk, int nQ, const float* qpos, const uint8_t* qmask,
                                          const uint8_t* qdesc, const mage_keypoint* tk, int nT, const uint8_t* tmask, const uint8_t* tdesc,
                                          float radius, int max_dist, int min_diff, mage_dmatch* out, int capacity, int* count)
{
    return guarded([&]() -> mage_status {
        if (!h || !count) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        *count = 0;
        if (nQ < 0 || nT < 0 || capacity < 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "negative size");
        if (nQ == 0 || nT == 0) return MAGE_OK;
        if (!qk || !qdesc || !tk || !tdesc || (capacity > 0 && !out)) return fail(MAGE_ERR_INVALID_ARGUMENT, "null buffer");
        MAGE_DEVICE_SCOPE(h->device);
        hipStream_t st = h->stream;
        // one staging buffer: [qk | tk | qpos | qdesc | tdesc | qmask | tmask], 16-byte aligned pieces
        auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
        const size_t o_qk = 0, o_tk = al(o_qk + sizeof(mage_keypoint) * nQ), o_qp = al(o_tk + sizeof(mage_keypoint) * nT),
                     o_qd = al(o_qp + 8 * (size_t)nQ), o_td = al(o_qd + 32 * (size_t)nQ), o_qm = al(o_td + 32 * (size_t)nT), o_tm = al(o_qm + nQ),
                     total = al(o_tm + nT);
        std::vector<uint8_t> stage(total, 0);
        std::memcpy(stage.data() + o_qk, qk, sizeof(mage_keypoint) * nQ); std::memcpy(stage.data() + o_tk, tk, sizeof(mage_keypoint) * nT);
        if (qpos) std::memcpy(stage.data() + o_qp, qpos, 8 * (size_t)nQ);
        std::memcpy(stage.data() + o_qd, qdesc, 32 * (size_t)nQ); std::memcpy(stage.data() + o_td, tdesc, 32 * (size_t)nT);
        if (qmask) std::memcpy(stage.data() + o_qm, qmask, nQ);
        if (tmask) std::memcpy(stage.data() + o_tm, tmask, nT);
        MAGE_TRY(h->d_A.reserve(total));
        MAGE_TRY(h->d_scratch.reserve(2 * (size_t)nQ + 2 * (size_t)nT + 4));
        MAGE_TRY(h->d_out.reserve((size_t)std::max(capacity, 1)));
        MAGE_TRY(h->d_counts.reserve(1));
        MAGE_HIP(hipMemcpyAsync(h->d_A.p, stage.data(), total, hipMemcpyHostToDevice, st));
        const uint8_t* d = h->d_A.p;
        MAGE_HIP(hipEventRecord(h->e0, st));
        radius_match_launch(reinterpret_cast<const mage_keypoint*>(d + o_qk), nQ, qpos ? reinterpret_cast<const float2*>(d + o_qp) : nullptr,
                            qmask ? d + o_qm : nullptr, d + o_qd, reinterpret_cast<const mage_keypoint*>(d + o_tk), nT, tmask ? d + o_tm : nullptr,
                            d + o_td, radius, max_dist, min_diff, h->d_scratch.p, h->d_out.p, capacity, h->d_counts.p, st);
        MAGE_HIP(hipEventRecord(h->e1, st));
        MAGE_HIP(hipMemcpyAsync(count, h->d_counts.p, sizeof(int), hipMemcpyDeviceToHost, st));
        MAGE_HIP(hipStreamSynchronize(st));
        const int n = std::min(*count, capacity);
        if (n > 0) MAGE_HIP(hipMemcpy(out, h->d_out.p, sizeof(mage_dmatch) * (size_t)n, hipMemcpyDeviceToHost));
        float ms = 0;
        MAGE_HIP(hipEventElapsedTime(&ms, h->e0, h->e1));
        h->last_ms = ms;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_match_indexed(mage_matcher* h, const uint8_t* descA, int nA, const uint8_t* maskA, const int32_t* cb_off, const int32_t* cb,
                                           const uint8_t* descB, int nB, const uint8_t* maskB, const int32_t* ca_off, const int32_t* ca,
                                           int max_dist, int min_diff, mage_dmatch* out, int capacity, int* count)
{
    return guarded([&]() -> mage_status {
        if (!h || !count) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        *count = 0;
        if (nA < 0 || nB < 0 || capacity < 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "negative size");
        if (nA == 0 || nB == 0) return MAGE_OK;
        if (!descA || !descB || !cb_off || !ca_off || (capacity > 0 && !out)) return fail(MAGE_ERR_INVALID_ARGUMENT, "null buffer");
        // FeatureMatcher.cpp:208: nothing to do when either mask is empty
        size_t cntA = 0, cntB = 0;
        for (int i = 0; i < nA; ++i) cntA += (!maskA || maskA[i]);
        for (int i = 0; i < nB; ++i) cntB += (!maskB || maskB[i]);
        if (cntA == 0 || cntB == 0) return MAGE_OK;
        // the candidate lists index into the other image: validate once on the host (the reference would read out of bounds)
        if (cb_off[0] != 0 || ca_off[0] != 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "candidate offsets must start at 0");
        for (int i = 0; i < nA; ++i) if (cb_off[i + 1] < cb_off[i]) return fail(MAGE_ERR_INVALID_ARGUMENT, "candidate offsets of A are not monotone");
        for (int i = 0; i < nB; ++i) if (ca_off[i + 1] < ca_off[i]) return fail(MAGE_ERR_INVALID_ARGUMENT, "candidate offsets of B are not monotone");
        const size_t ncb = (size_t)cb_off[nA], nca = (size_t)ca_off[nB];
        if ((ncb && !cb) || (nca && !ca)) return fail(MAGE_ERR_INVALID_ARGUMENT, "null candidate list");
        for (size_t k = 0; k < ncb; ++k) if (cb[k] < 0 || cb[k] >= nB) return fail(MAGE_ERR_INVALID_ARGUMENT, "candidate %zu of A is outside B", k);
        for (size_t k = 0; k < nca; ++k) if (ca[k] < 0 || ca[k] >= nA) return fail(MAGE_ERR_INVALID_ARGUMENT, "candidate %zu of B is outside A", k);
        MAGE_DEVICE_SCOPE(h->device);
        hipStream_t st = h->stream;
        // one staging buffer: [descA | descB | cb_off | ca_off | cb | ca | maskA | maskB], 16-byte aligned pieces
        auto al = [](size_t v) { return (v + 15) & ~(size_t)15; };
        const size_t o_da = 0, o_db = al(o_da + 32 * (size_t)nA), o_bo = al(o_db + 32 * (size_t)nB), o_ao = al(o_bo + 4 * ((size_t)nA + 1)),
                     o_cb = al(o_ao + 4 * ((size_t)nB + 1)), o_ca = al(o_cb + 4 * ncb), o_ma = al(o_ca + 4 * nca), o_mb = al(o_ma + nA), total = al(o_mb + nB);
        std::vector<uint8_t> stage(total, 0);
        std::memcpy(stage.data() + o_da, descA, 32 * (size_t)nA); std::memcpy(stage.data() + o_db, descB, 32 * (size_t)nB);
        std::memcpy(stage.data() + o_bo, cb_off, 4 * ((size_t)nA + 1)); std::memcpy(stage.data() + o_ao, ca_off, 4 * ((size_t)nB + 1));
        if (ncb) std::memcpy(stage.data() + o_cb, cb, 4 * ncb);
        if (nca) std::memcpy(stage.data() + o_ca, ca, 4 * nca);
        if (maskA) std::memcpy(stage.data() + o_ma, maskA, nA);
        if (maskB) std::memcpy(stage.data() + o_mb, maskB, nB);
        MAGE_TRY(h->d_A.reserve(total));
        MAGE_TRY(h->d_out.reserve((size_t)std::max(capacity, 1)));
        MAGE_TRY(h->d_counts.reserve(1));
        MAGE_HIP(hipMemcpyAsync(h->d_A.p, stage.data(), total, hipMemcpyHostToDevice, st));
        const uint8_t* d = h->d_A.p;
        MAGE_HIP(hipEventRecord(h->e0, st));
        indexed_match_launch(d + o_da, nA, maskA ? d + o_ma : nullptr, reinterpret_cast<const int*>(d + o_bo), reinterpret_cast<const int*>(d + o_cb),
                             d + o_db, maskB ? d + o_mb : nullptr, reinterpret_cast<const int*>(d + o_ao), reinterpret_cast<const int*>(d + o_ca),
                             max_dist, min_diff, h->d_out.p, capacity, h->d_counts.p, st);
        MAGE_HIP(hipEventRecord(h->e1, st));
        MAGE_HIP(hipMemcpyAsync(count, h->d_counts.p, sizeof(int), hipMemcpyDeviceToHost, st));
        MAGE_HIP(hipStreamSynchronize(st));
        const int n = std::min(*count, capacity);
        if (n > 0) MAGE_HIP(hipMemcpy(out, h->d_out.p, sizeof(mage_dmatch) * (size_t)n, hipMemcpyDeviceToHost));
        float ms = 0;
        MAGE_HIP(hipEventElapsedTime(&ms, h->e0, h->e1));
        h->last_ms = ms;
        return MAGE_OK;
    });
}

// ---- the vocabulary tree (mage_match.h): validation shared by the entry points; the tree goes up as its walk table (orb_kernels.h: BowWalkEntry)
namespace {
mage_status check_bow_tree(const mage_bow_tree* t)
{
    if (!t || !t->node_descriptors || !t->child_offsets) return fail(MAGE_ERR_INVALID_ARGUMENT, "null vocabulary tree");
    if (t->n_nodes < 1) return fail(MAGE_ERR_INVALID_ARGUMENT, "a vocabulary tree has at least its root");
    if (t->child_offsets[0] != 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "child offsets must start at 0");
    for (int n = 0; n < t->n_nodes; ++n) {
        if (t->child_offsets[n + 1] < t->child_offsets[n]) return fail(MAGE_ERR_INVALID_ARGUMENT, "child offsets are not monotone at node %d", n);
        if (t->child_offsets[n + 1] > t->child_offsets[n] && !t->children) return fail(MAGE_ERR_INVALID_ARGUMENT, "null child list");
        if (t->child_offsets[n + 1] - t->child_offsets[n] > 65535) return fail(MAGE_ERR_UNSUPPORTED, "node %d has more than 65535 children", n);
        // a child is created after its parent (OnlineBow's m_nodes grows by appending): child index > parent index, which also bounds every descent
        for (int k = t->child_offsets[n]; k < t->child_offsets[n + 1]; ++k)
            if (t->children[k] <= n || t->children[k] >= t->n_nodes) return fail(MAGE_ERR_INVALID_ARGUMENT, "child %d of node %d must lie in (%d, %d)", t->children[k], n, n, t->n_nodes);
    }
    return MAGE_OK;
}
mage_status check_leaf_lists(const int32_t* off, const int32_t* items, int n_nodes, int n_features, const char* which)
{
    if (!off) return fail(MAGE_ERR_INVALID_ARGUMENT, "null feature offsets of image %s", which);
    if (off[0] != 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "feature offsets of image %s must start at 0", which);
    for (int n = 0; n < n_nodes; ++n) if (off[n + 1] < off[n]) return fail(MAGE_ERR_INVALID_ARGUMENT, "feature offsets of image %s are not monotone", which);
    if (off[n_nodes] && !items) return fail(MAGE_ERR_INVALID_ARGUMENT, "null feature list of image %s", which);
    for (int k = 0; k < off[n_nodes]; ++k) if (items[k] < 0 || items[k] >= n_features) return fail(MAGE_ERR_INVALID_ARGUMENT, "feature %d filed for image %s is outside it", k, which);
    return MAGE_OK;
}
}  // namespace

MAGE_EXPORT mage_status mage_bow_set_tree(mage_matcher* h, const mage_bow_tree* tree)
{
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (!tree) { h->tree_nodes = 0; return MAGE_OK; }
        MAGE_TRY(check_bow_tree(tree));
        MAGE_DEVICE_SCOPE(h->device);
        const size_t nn = (size_t)tree->n_nodes, nch = (size_t)tree->child_offsets[nn], total = std::max<size_t>(nch, 1) * sizeof(BowWalkEntry);
        std::vector<uint8_t> stage(total, 0);
        bow_walk_fill(tree->node_descriptors, tree->child_offsets, tree->children, tree->n_nodes, reinterpret_cast<BowWalkEntry*>(stage.data()));
        h->tree_nodes = 0;
        MAGE_HIP(hipStreamSynchronize(h->stream));           // (a launch still reading the previous tree)
        MAGE_TRY(h->d_tree.reserve(total));
        MAGE_HIP(hipMemcpy(h->d_tree.p, stage.data(), total, hipMemcpyHostToDevice));
        h->tree_nodes = nn; h->tree_root_k1 = tree->child_offsets[1];
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_bow_find_leaf_batch(mage_matcher* h, const mage_bow_tree* tree, const uint8_t* descriptors, int n, int32_t* leaf_ids)
{
    if (h && !tree && h->tree_nodes) {
        // the tree kept by mage_bow_set_tree: only the descriptors travel
        return guarded([&]() -> mage_status {
            if (n < 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "negative size");
            if (n == 0) return MAGE_OK;
            if (!descriptors || !leaf_ids) return fail(MAGE_ERR_INVALID_ARGUMENT, "null buffer");
            MAGE_DEVICE_SCOPE(h->device);
            hipStream_t st = h->stream;
            // 28 KB in, 3.5 KB out for a frame pair: two copy commands and their bookkeeping cost more than the walk itself (round 6: 4.7 us
            // of kernel in a 50-us call).  The descriptors go into a pinned block with a host memcpy, the kernel reads them across PCIe
            // (a query is 32 bytes fetched once by its 16 lanes) and writes the leaf ids into the same block: one launch, one wait.
            const size_t o_leaf = (32 * (size_t)n + 63) & ~(size_t)63;
            MAGE_TRY(h->h_io.resize_uninitialized(o_leaf + sizeof(int) * (size_t)n));
            void* dio = nullptr;
            const bool zero_copy = hipHostGetDevicePointer(&dio, h->h_io.data(), 0) == hipSuccess;
            if (!zero_copy) (void)hipGetLastError();
            if (zero_copy) {
                std::memcpy(h->h_io.data(), descriptors, 32 * (size_t)n);
                MAGE_HIP(hipEventRecord(h->e0, st));
                bow_find_leaf_launch(reinterpret_cast<const BowWalkEntry*>(h->d_tree.p), 0, h->tree_root_k1, static_cast<const uint8_t*>(dio), n,
                                     reinterpret_cast<int*>(static_cast<uint8_t*>(dio) + o_leaf), st);
                MAGE_HIP(hipEventRecord(h->e1, st));
                MAGE_HIP(hipStreamSynchronize(st));
                std::memcpy(leaf_ids, h->h_io.data() + o_leaf, sizeof(int) * (size_t)n);
            } else {
                MAGE_TRY(h->d_A.reserve(32 * (size_t)n));
                MAGE_TRY(h->d_scratch.reserve((size_t)n));
                MAGE_HIP(hipMemcpyAsync(h->d_A.p, descriptors, 32 * (size_t)n, hipMemcpyHostToDevice, st));
                MAGE_HIP(hipEventRecord(h->e0, st));
                bow_find_leaf_launch(reinterpret_cast<const BowWalkEntry*>(h->d_tree.p), 0, h->tree_root_k1, h->d_A.p, n, h->d_scratch.p, st);
                MAGE_HIP(hipEventRecord(h->e1, st));
                MAGE_HIP(hipMemcpyAsync(leaf_ids, h->d_scratch.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
                MAGE_HIP(hipStreamSynchronize(st));
            }
            float ms = 0;
            MAGE_HIP(hipEventElapsedTime(&ms, h->e0, h->e1));
            h->last_ms = ms;
            return MAGE_OK;
        });
    }
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (n < 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "negative size");
        MAGE_TRY(check_bow_tree(tree));
        if (n == 0) return MAGE_OK;
        if (!descriptors || !leaf_ids) return fail(MAGE_ERR_INVALID_ARGUMENT, "null buffer");
        MAGE_DEVICE_SCOPE(h->device);
        hipStream_t st = h->stream;
        auto al = [](size_t v) { return (v + 31) & ~(size_t)31; };          // (the kernel reads queries as 32-byte vectors; the same alignment as mage_match_indexed_bow)
        const size_t nn = (size_t)tree->n_nodes, nch = (size_t)tree->child_offsets[nn];
        const size_t o_w = 0, o_q = al(o_w + sizeof(BowWalkEntry) * nch), total = al(o_q + 32 * (size_t)n);
        std::vector<uint8_t> stage(total, 0);
        bow_walk_fill(tree->node_descriptors, tree->child_offsets, tree->children, tree->n_nodes, reinterpret_cast<BowWalkEntry*>(stage.data() + o_w));
        std::memcpy(stage.data() + o_q, descriptors, 32 * (size_t)n);
        MAGE_TRY(h->d_A.reserve(total));
        MAGE_TRY(h->d_scratch.reserve((size_t)n));
        MAGE_HIP(hipMemcpyAsync(h->d_A.p, stage.data(), total, hipMemcpyHostToDevice, st));
        const uint8_t* d = h->d_A.p;
        MAGE_HIP(hipEventRecord(h->e0, st));
        bow_find_leaf_launch(reinterpret_cast<const BowWalkEntry*>(d + o_w), 0, tree->child_offsets[1], d + o_q, n, h->d_scratch.p, st);
        MAGE_HIP(hipEventRecord(h->e1, st));
        MAGE_HIP(hipMemcpyAsync(leaf_ids, h->d_scratch.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, st));
        MAGE_HIP(hipStreamSynchronize(st));
        float ms = 0;
        MAGE_HIP(hipEventElapsedTime(&ms, h->e0, h->e1));
        h->last_ms = ms;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_match_indexed_bow(mage_matcher* h, const mage_bow_tree* tree, const uint8_t* descA, int nA, const uint8_t* maskA,
                                               const int32_t* feat_a_off, const int32_t* feat_a, const uint8_t* descB, int nB, const uint8_t* maskB,
                                               const int32_t* feat_b_off, const int32_t* feat_b, int max_dist, int min_diff, mage_dmatch* out, int capacity, int* count)
{
    return guarded([&]() -> mage_status {
        if (!h || !count) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        *count = 0;
        if (nA < 0 || nB < 0 || capacity < 0) return fail(MAGE_ERR_INVALID_ARGUMENT, "negative size");
        const bool resident = !tree && h->tree_nodes;        // the tree kept by mage_bow_set_tree: it is not staged again
        if (!resident) MAGE_TRY(check_bow_tree(tree));
        const int n_nodes = resident ? (int)h->tree_nodes : tree->n_nodes;
        if (nA == 0 || nB == 0) return MAGE_OK;
        if (!descA || !descB || (capacity > 0 && !out)) return fail(MAGE_ERR_INVALID_ARGUMENT, "null buffer");
        MAGE_TRY(check_leaf_lists(feat_a_off, feat_a, n_nodes, nA, "A"));
        MAGE_TRY(check_leaf_lists(feat_b_off, feat_b, n_nodes, nB, "B"));
        size_t cntA = 0, cntB = 0;          // FeatureMatcher.cpp:208: nothing to do when either mask is empty
        for (int i = 0; i < nA; ++i) cntA += (!maskA || maskA[i]);
        for (int i = 0; i < nB; ++i) cntB += (!maskB || maskB[i]);
        if (cntA == 0 || cntB == 0) return MAGE_OK;
        MAGE_DEVICE_SCOPE(h->device);
        hipStream_t st = h->stream;
        // one staging buffer: [walk table of the tree | descA | descB | feat_b_off | feat_a_off | feat_b | feat_a | maskA | maskB]; descA and descB
        // are adjacent so that ONE launch finds the leaves of both images
        auto al = [](size_t v) { return (v + 31) & ~(size_t)31; };
        const size_t nn = (size_t)n_nodes, nch = resident ? 0 : (size_t)tree->child_offsets[nn], nfa = (size_t)feat_a_off[nn], nfb = (size_t)feat_b_off[nn];
        const size_t o_w = 0, o_da = resident ? 0 : al(o_w + sizeof(BowWalkEntry) * nch), o_db = o_da + 32 * (size_t)nA,
                     o_bo = al(o_db + 32 * (size_t)nB), o_ao = al(o_bo + 4 * (nn + 1)), o_fb = al(o_ao + 4 * (nn + 1)), o_fa = al(o_fb + 4 * nfb),
                     o_ma = al(o_fa + 4 * nfa), o_mb = al(o_ma + nA), total = al(o_mb + nB);
        // staged in PINNED memory (one real DMA instead of the runtime's copy of a pageable block), and the matches + their count written by
        // the kernel into the same block across PCIe (<= 7 KB): no copy back, one wait
        const size_t o_out = al(total), o_cnt = al(o_out + sizeof(mage_dmatch) * (size_t)std::max(capacity, 1)), pinned_total = o_cnt + 64;
        MAGE_TRY(h->h_io.resize_uninitialized(pinned_total));
        struct Stage { uint8_t* p; uint8_t* data() const { return p; } } stage{ h->h_io.data() };
        void* dio = nullptr;
        const bool zero_copy = hipHostGetDevicePointer(&dio, h->h_io.data(), 0) == hipSuccess;
        if (!zero_copy) (void)hipGetLastError();
        if (!resident) bow_walk_fill(tree->node_descriptors, tree->child_offsets, tree->children, tree->n_nodes, reinterpret_cast<BowWalkEntry*>(stage.data() + o_w));
        std::memcpy(stage.data() + o_da, descA, 32 * (size_t)nA); std::memcpy(stage.data() + o_db, descB, 32 * (size_t)nB);
        std::memcpy(stage.data() + o_bo, feat_b_off, 4 * (nn + 1)); std::memcpy(stage.data() + o_ao, feat_a_off, 4 * (nn + 1));
        if (nfb) std::memcpy(stage.data() + o_fb, feat_b, 4 * nfb);
        if (nfa) std::memcpy(stage.data() + o_fa, feat_a, 4 * nfa);
        if (maskA) std::memcpy(stage.data() + o_ma, maskA, nA);
        if (maskB) std::memcpy(stage.data() + o_mb, maskB, nB);
        MAGE_TRY(h->d_A.reserve(total));
        MAGE_TRY(h->d_scratch.reserve((size_t)nA + nB));
        MAGE_TRY(h->d_out.reserve((size_t)std::max(capacity, 1)));
        MAGE_TRY(h->d_counts.reserve(1));
        MAGE_HIP(hipMemcpyAsync(h->d_A.p, stage.data(), total, hipMemcpyHostToDevice, st));
        const uint8_t* d = h->d_A.p;
        int* leaf = h->d_scratch.p;          // [leaf of every A descriptor | leaf of every B descriptor]
        MAGE_HIP(hipEventRecord(h->e0, st));
        bow_find_leaf_launch(reinterpret_cast<const BowWalkEntry*>(resident ? h->d_tree.p : d + o_w), 0, resident ? h->tree_root_k1 : tree->child_offsets[1], d + o_da, nA + nB, leaf, st);
        indexed_match_launch(d + o_da, nA, maskA ? d + o_ma : nullptr, reinterpret_cast<const int*>(d + o_bo), reinterpret_cast<const int*>(d + o_fb),
                             d + o_db, maskB ? d + o_mb : nullptr, reinterpret_cast<const int*>(d + o_ao), reinterpret_cast<const int*>(d + o_fa),
                             max_dist, min_diff, zero_copy ? reinterpret_cast<mage_dmatch*>(static_cast<uint8_t*>(dio) + o_out) : h->d_out.p, capacity,
                             zero_copy ? reinterpret_cast<int*>(static_cast<uint8_t*>(dio) + o_cnt) : h->d_counts.p, st, leaf, leaf + nA);
        MAGE_HIP(hipEventRecord(h->e1, st));
        if (!zero_copy) MAGE_HIP(hipMemcpyAsync(count, h->d_counts.p, sizeof(int), hipMemcpyDeviceToHost, st));
        MAGE_HIP(hipStreamSynchronize(st));
        if (zero_copy) *count = *reinterpret_cast<const int*>(h->h_io.data() + o_cnt);
        const int n = std::min(*count, capacity);
        if (n > 0) {
            if (zero_copy) std::memcpy(out, h->h_io.data() + o_out, sizeof(mage_dmatch) * (size_t)n);
            else MAGE_HIP(hipMemcpy(out, h->d_out.p, sizeof(mage_dmatch) * (size_t)n, hipMemcpyDeviceToHost));
        }
        float ms = 0;
        MAGE_HIP(hipEventElapsedTime(&ms, h->e0, h->e1));
        h->last_ms = ms;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_matcher_last_kernel_ms(const mage_matcher* h, double* ms)
{
    if (!h || !ms) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
    *ms = h->last_ms;
    return MAGE_OK;
}
