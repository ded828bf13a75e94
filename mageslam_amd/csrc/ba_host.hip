// ba_host.hip -- host side of the bundle-adjustment back-end: the mage_ba_* C ABI (include/mage_ba.h).
//
// Mirrors the control flow of the reference facade and of the g2o pieces it drives:
//   * call protocol and StepOptimizer (dirty / useless / iteration)   BundlerLib.cpp:92-167, 198-309
//   * OptimizationAlgorithmLevenberg::solve (lambda policy, <=10 trials) SURVEY.md appendix A.4
//   * SparseOptimizer::initializeOptimization + index mapping          SURVEY.md appendix A.5
//   * StepBundleAdjustment post-pass (outliers, mean square error)     BundlerLib.cpp:364-447
// All arithmetic over observations, landmarks and the reduced camera system runs in the HIP kernels
// of ba_kernels.hip / chol_kernels.hip; the host only builds the (static between outlier removals)
// graph structure, sequences launches, and reads back three scalars per LM trial.
#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <functional>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <thread>
#include <queue>
#include <vector>

#include "ba_build.h"
#include "ba_kernels.h"
#include "chol_kernels.h"
#include "chol_dag.h"
#include "mage_common.h"

namespace mage {

std::string& last_error_ref()
{
    thread_local std::string s;
    return s;
}

namespace {
struct DeviceCache {
    static constexpr int MAX_DEVICES = 64;
    std::mutex m;
    std::multimap<size_t, void*> parked[MAX_DEVICES];
    std::vector<hipStream_t> streams[MAX_DEVICES];
    std::multimap<size_t, void*> pinned;
    size_t pinned_held = 0;
    size_t held[MAX_DEVICES] = {};
    size_t limit;
    DeviceCache()
    {
        const char* e = std::getenv("MAGE_DEVICE_CACHE_MB");
        limit = (size_t)(e ? std::max(0L, std::atol(e)) : 4096L) << 20;
    }
};
DeviceCache& device_cache()
{
    static DeviceCache* c = new DeviceCache();      // never destroyed: the HIP runtime may be gone by static-destructor time
    return *c;
}
size_t cache_round(size_t bytes)
{
    const size_t g = bytes < ((size_t)1 << 20) ? 4096 : ((size_t)1 << 20);
    return (std::max<size_t>(bytes, 1) + g - 1) / g * g;
}
}  // namespace

mage_status cached_device_alloc(void** p, size_t bytes, int* device, size_t* granted)
{
    int dev = 0;
    MAGE_HIP(hipGetDevice(&dev));
    const size_t want = cache_round(bytes);
    *device = dev; *granted = want; *p = nullptr;
    DeviceCache& c = device_cache();
    if (dev >= 0 && dev < DeviceCache::MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(c.m);
        auto it = c.parked[dev].lower_bound(want);
        if (it != c.parked[dev].end() && it->first <= want + want / 4) {       // a parked block no more than 25 % larger
            *p = it->second; *granted = it->first;
            c.held[dev] -= it->first;
            c.parked[dev].erase(it);
            return MAGE_OK;
        }
    }
    hipError_t e = hipMalloc(p, want);
    if (e == hipErrorOutOfMemory) {                 // give the parked memory back before reporting failure
        mage_release_cached_memory();
        e = hipMalloc(p, want);
    }
    MAGE_HIP(e);
    return MAGE_OK;
}

mage_status cached_pinned_alloc(void** p, size_t bytes, size_t* granted)
{
    const size_t want = cache_round(bytes);
    *granted = want; *p = nullptr;
    DeviceCache& c = device_cache();
    {
        std::lock_guard<std::mutex> lock(c.m);
        auto it = c.pinned.lower_bound(want);
        if (it != c.pinned.end() && it->first <= 2 * want) {
            *p = it->second; *granted = it->first;
            c.pinned_held -= it->first;
            c.pinned.erase(it);
            return MAGE_OK;
        }
    }
    MAGE_HIP(hipHostMalloc(p, want, hipHostMallocDefault));
    return MAGE_OK;
}

void cached_pinned_release(void* p, size_t bytes)
{
    if (!p) return;
    DeviceCache& c = device_cache();
    {
        std::lock_guard<std::mutex> lock(c.m);
        if (c.pinned_held + bytes <= c.limit / 4) {          // pinned host memory: a quarter of the device budget
            c.pinned.emplace(bytes, p);
            c.pinned_held += bytes;
            return;
        }
    }
    (void)hipHostFree(p);
}

mage_status cached_stream_acquire(int device, hipStream_t* out)
{
    DeviceCache& c = device_cache();
    if (device >= 0 && device < DeviceCache::MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(c.m);
        if (!c.streams[device].empty()) { *out = c.streams[device].back(); c.streams[device].pop_back(); return MAGE_OK; }
    }
    MAGE_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return MAGE_OK;
}

void cached_stream_release(int device, hipStream_t st)
{
    if (!st) return;
    DeviceCache& c = device_cache();
    if (c.limit > 0 && device >= 0 && device < DeviceCache::MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(c.m);
        if (c.streams[device].size() < 16) { c.streams[device].push_back(st); return; }
    }
    chol_forget_stream(st);
    (void)hipStreamDestroy(st);
}

void cached_device_release(void* p, size_t bytes, int device)
{
    if (!p) return;
    DeviceCache& c = device_cache();
    if (device >= 0 && device < DeviceCache::MAX_DEVICES) {
        std::lock_guard<std::mutex> lock(c.m);
        if (c.held[device] + bytes <= c.limit) {
            c.parked[device].emplace(bytes, p);
            c.held[device] += bytes;
            return;
        }
    }
    (void)hipFree(p);
}

mage_status select_device(int requested, int* chosen)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(MAGE_ERR_NO_DEVICE, "no HIP device visible (%s)", e != hipSuccess ? hipGetErrorString(e) : "count = 0");
    int dev = requested;
    if (dev < 0) MAGE_HIP(hipGetDevice(&dev));
    if (dev >= n) return fail(MAGE_ERR_INVALID_ARGUMENT, "device %d out of range (%d visible)", dev, n);
    hipDeviceProp_t prop;
    MAGE_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MAGE_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
    *chosen = dev;
    return MAGE_OK;
}

namespace {

struct HostCam {
    double q[4] = { 0, 0, 0, 1 }, t[3] = { 0, 0, 0 };
    double f = 1, cx = 0, cy = 0;
    uint8_t fixed = 0, set = 0;
};
struct HostTether {                 // one EdgeScaleConstraint / EdgeRotationConstraint / EdgeSE3Expmap
    uint32_t c0 = 0, c1 = 0;
    double q[4] = { 0, 0, 0, 1 }, t[3] = { 0, 0, 0 }, dist = 0, w = 0;
    uint8_t set = 0;
};
typedef ObsRecord HostObs;          // ba_build.h: kept in pinned memory, uploaded as it is when the structure is built on the device

// float32 rotation matrix (column-major) -> float32 quaternion -> normalise -> float64 -> SE3Quat
// normalisation; the chain BundlerLib.cpp:270-273 runs through Eigen.
void pose_from_f32(const float* Rcm, const float* t, HostCam& c)
{
    auto M = [&](int r, int col) { return Rcm[col * 3 + r]; };
    float q[4];
    float tr = M(0, 0) + M(1, 1) + M(2, 2);
    if (tr > 0.f) {
        float s = std::sqrt(tr + 1.0f);
        q[3] = 0.5f * s;
        s = 0.5f / s;
        q[0] = (M(2, 1) - M(1, 2)) * s; q[1] = (M(0, 2) - M(2, 0)) * s; q[2] = (M(1, 0) - M(0, 1)) * s;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        float s = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0f);
        q[i] = 0.5f * s;
        s = 0.5f / s;
        q[3] = (M(k, j) - M(j, k)) * s;
        q[j] = (M(j, i) + M(i, j)) * s;
        q[k] = (M(k, i) + M(i, k)) * s;
    }
    float nf = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double d[4] = { (double)(q[0] / nf), (double)(q[1] / nf), (double)(q[2] / nf), (double)(q[3] / nf) };
    if (d[3] < 0) { d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; d[3] = -d[3]; }
    double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
    for (int a = 0; a < 4; ++a) c.q[a] = d[a] / n;
    c.t[0] = t[0]; c.t[1] = t[1]; c.t[2] = t[2];
}

}  // namespace
}  // namespace mage

using namespace mage;

struct mage_ba {
    int device = 0;
    hipStream_t stream = nullptr;
    bool points_fixed = false;

    // ---- problem as set through the surface (host copy)
    std::vector<HostCam> cams;
    std::vector<double> pts;            // n x 3
    std::vector<uint8_t> pt_set;
    PinnedVec<HostObs> obs;
    bool obs_unfilled = false;          // AllocateObservations left the records uninitialised: a bulk setter that covers all of them need not
                                        // pay for clearing 24 bytes per record first (ensure_obs_filled clears them before anybody else looks)
    std::vector<HostTether> teth[3];    // FixedDistance / RelativeRotation / RelativeTransform constraints
    bool cams_allocated = false, pts_allocated = false, obs_allocated = false, teth_allocated[3] = { false, false, false };

    // ---- StepOptimizer / LM state
    bool dirty = true, useless = false;
    bool soft_dirty = false;            // observations were removed on the device since the last LM iteration
    long long n_active_remaining = 0;
    int iteration = 0;
    double lambda = -1.0, user_lambda = 0.0, ni = 2.0;

    // ---- where the truth about poses/points lives
    bool state_on_device = false;       // device buffers hold the current estimate
    mutable bool host_state_fresh = true;  // host copy equals the device estimate
    // small problems: the post-pass also writes the kept estimate behind the scalars, so that it rides their read-back (k_small_classify)
    size_t res_doubles = 0;                // poses x 8 + points x 4 of this build when that fits RES_CAP and the small path applies, else 0
    mutable bool result_in_mirror = false; // the pinned mirror holds the estimate of the last step: download_state needs no device operation

    // ---- device storage
    DevBuf<double> d_pose[2], d_pt[2], d_camK;
    int cur = 0;                        // index of the "current" state buffer
    DevBuf<int> d_cam2hc, d_hc2cam;
    DevBuf<float2> d_L_uv; DevBuf<float> d_L_info; DevBuf<uint32_t> d_L_cam, d_L_pt, d_L_edge; DevBuf<int> d_L_slot;
    DevBuf<int> d_lm_ptr, d_lm_pt, d_lm_wptr, d_w_hc, d_w_lm, d_camE_ptr, d_camE, d_camS_ptr, d_camS, d_blk_ptr;
    DevBuf<int2> d_blk_ij, d_con;
    // where the compact W records live (BaDeviceView::w_pos): built on the first LM iteration that uses the compact form
    DevBuf<int> d_w_pos, d_pos_lm, d_slot_order;
    DevBuf<ConPos> d_con_pos;
    DevBuf<int> d_stream_ptr, d_group_blocks, d_con_soa; DevBuf<int4> d_stream_blks; DevBuf<long long> d_stream_stamps;      // k_schur_stream's block lists (BaDeviceView::stream_blks)      // k_schur_stream's trip lists (BaDeviceView::trips)
    int n_cu = 0;
    bool schur_per_block = false;      // mage_ba_debug_schur_per_block / MAGE_BA_SCHUR_BLOCKS
    size_t n_con = 0;
    bool positions_valid = false;
    DevBuf<int> d_blk_order;
    bool dup_slots = false;            // some landmark is observed twice by one free camera
    DevBuf<double> d_errL, d_U, d_bc, d_V, d_bp, d_W, d_Dinv, d_db, d_S, d_y, d_xc, d_xl, d_partial, d_scal, d_Linv, d_camR;
    DevBuf<uint8_t> d_flagL, d_L_active;
    uint32_t* d_out_ids = nullptr;      // outliers of the last post-pass (original observation indices, unordered): they live BEHIND the scalars
                                        // in d_scal, so the first OUT_PREFIX of them come back in the scalars' read-back; cursor = the int behind the small-path counter
    int out_cursor = 0;                 // value of that cursor (it only grows between structure builds)
    size_t out_expect = 0;              // outliers of the previous post-pass: sizes the prefix that rides the next read-back
    DevBuf<int> d_T_kind, d_tc_hc, d_tc_ptr, d_tc_item, d_tp_ptr, d_tp_item;
    DevBuf<int2> d_T_cam, d_T_fixed, d_tp_ij;
    DevBuf<double> d_T_meas, d_T_w, d_T_out;
    int n_active_tethers = 0;
    DevBuf<int> d_queue;
    DevBuf<int> d_tile_env;            // the skyline of S by tile rows (k_zero_skyline)
    bool tile_env_valid = false;       // d_tile_env describes the CURRENT structure
    std::vector<int> tile_env_host;    // mage_ba_use_skyline: the same on the host (read back once per structure), handed to the dense solve's task-graph schedule
    bool use_skyline = false;
    bool S_outside_skyline_is_zero = false;   // every tile of S left of its row's envelope holds zeros (true after a clear + a factorisation that produced no NaN)
    // ---- device build of the structure (ba_build.h): the raw records and its scratch
    DevBuf<ObsRecord> d_obs_raw; DevBuf<uint8_t> d_cam_fixed; DevBuf<int> d_cam_extra;
    DevBuf<int> d_b_pt2lm, d_b_L_hc, d_b_L_lm, d_b_where, d_b_hist, d_b_w_end;
    DevBuf<unsigned long long> d_b_bucket, d_b_scan, d_b_row;
    DevBuf<unsigned char> d_b_zeroed;             // [BuildCounts | cam_deg | pt_deg]: cleared by one fill
    bool built_on_device = false;
    // ---- small problems built on the host: every list and every work array is a view into ONE device buffer filled by ONE copy of the
    // pinned arena they were built in (ImageStager below)
    DevBuf<unsigned char> d_image;
    struct ImageStager {
        bool on = false;
        size_t scratch = 0;                                               // device-only bytes behind the uploaded part
        std::vector<std::function<void(unsigned char* dev, const std::function<size_t(const void*)>& image_offset, size_t up_bytes)>> binds;      // image_offset: where a pinned staging address lies in the image
    } img;
    // ---- the tracker's per-frame problems (frame_step below): one image up, one launch, one record back; no structure stays on the device
    DevBuf<unsigned char> d_frame;
    void* h_frame = nullptr; size_t h_frame_bytes = 0;     // pinned (from the cache): the image and, behind it, the record that comes back
    unsigned long long edit_gen = 0;    // bumped by every edit of the graph through the surface (mark_edited)
    unsigned long long frame_gen = 0;   // edit_gen at the last frame_step
    bool frame_valid = false;           // the last step was a frame_step: `dirty` then only says "no structure on the device", not "graph edited"
    long long stall_retries_total = 0;  // trials re-run because a bounded hand-off of the dense solve timed out (diagnostic)
    PinnedArena build_arena;            // staging of the last structure build: released at the next completed read-back (no synchronisation of its own)
    void* h_pinned = nullptr; size_t h_pinned_bytes = 0;     // one pinned block (from the cache) holding the mirrors below
    uint32_t* h_out_ids = nullptr;      // first OUT_PREFIX outlier ids of the last post-pass: they ride the scalar read-back
    double* h_scal = nullptr;           // pinned mirror of d_scal
    DevBuf<PoseLmResult> d_pose_lm;     // pose-only problems: the record the one-launch solve leaves
    PoseLmResult* h_pose_lm = nullptr;  // pinned mirror
    BaDeviceView view{};
    std::vector<uint32_t> L_edge_host;  // landmark-order position -> observation index
    std::vector<uint8_t> flag_host;
    std::vector<uint32_t> last_outliers;   // full outlier list of the most recent mage_ba_step (mage_ba_get_outliers)

    // ---- device-resident pose exchange (mage_ba_bind_pose_exchange): camera index / block row lists, bound once
    DevBuf<uint32_t> d_x_exp_cam, d_x_exp_row, d_x_imp_cam, d_x_imp_row;
    size_t n_x_exp = 0, n_x_imp = 0;
    hipEvent_t ev_x[2] = { nullptr, nullptr };

    // ---- landmark-sharded map (mage_ba_set_landmark_shard): this handle holds the landmarks of one rank
    int shard_rank = 0, shard_ranks = 0;                    // 0 ranks: not sharded
    mage_ba_allreduce_fn shard_reduce = nullptr; void* shard_user = nullptr;
    DevBuf<double> d_xchg;                                  // lower tiles of S packed + y: what the ranks add per trial

    // ---- diagnostics
    std::vector<mage_ba_iter_stats> stats;
    bool profiling = false;             // stage events on: every stage of an LM iteration is bracketed
    bool profiling_factor = false;      // only the dense factorisation + solves is bracketed (two event records per trial)
    mage_ba_profile prof{};
    hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
    hipEvent_t ev_p[4] = { nullptr, nullptr, nullptr, nullptr };     // profiling only: linearise begin / end, update begin / end

    // The first asynchronous host-to-device copy of a PROCESS costs ~6.5 ms whatever it moves (the runtime sets its copy path up); the first
    // handle of a process makes it on a worker thread started at the top of mage_ba_create, beside the rest of the creation, instead of
    // inside its first step.  The first structure build joins the thread.
    std::thread warmup;
    void join_warmup() { if (warmup.joinable()) warmup.join(); }

    ~mage_ba()
    {
        join_warmup();
        DeviceScope scope(device);
        if (stream) (void)hipStreamSynchronize(stream);
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_p) if (e) (void)hipEventDestroy(e);
        for (auto& e : ev_x) if (e) (void)hipEventDestroy(e);
        if (h_pinned) cached_pinned_release(h_pinned, h_pinned_bytes);      // parked, not freed: hipHostFree costs ~200 us
        if (h_frame) cached_pinned_release(h_frame, h_frame_bytes);
        cached_stream_release(device, stream);
    }
};

namespace {

void ensure_obs_filled(mage_ba* h)
{
    if (!h->obs_unfilled) return;
    std::fill(h->obs.begin(), h->obs.end(), HostObs());
    h->obs_unfilled = false;
}

constexpr size_t OUT_PREFIX = 4096;      // outlier ids copied back together with the post-pass scalars (16 KB); a longer list takes a second copy
constexpr size_t RES_CAP = 4096;         // doubles of the pinned mirror reserved for the estimate of a small problem (32 KB: 12 keyframes + 1000 points)
constexpr size_t SC_PAD = 16;            // doubles reserved for the scalars in d_scal / the pinned mirror; the outlier ids follow
static_assert(SC_COUNT <= SC_PAD, "scalar block");

// ---- uploads and work arrays of a structure build.  Normally each is a device buffer of its own and a copy (or fill) of its own --
// forty small operations on the stream for a 2 000-observation local BA, ~5 us each whatever they move.  In IMAGE mode (h->img.on:
// small problems built on the host) nothing touches the device until the build is over: every list was built in the pinned arena,
// every work array is given an offset, and stage_commit() reserves ONE device buffer, makes each DevBuf a view into it and copies
// the arena's block with ONE DMA (arrays that must start as zeros / ones are filled in the arena and ride the same copy).
template <typename T>
mage_status stage_push(mage_ba* h, DevBuf<T>& d, const T* src_pinned, size_t count)
{
    if (h->img.on) {
        h->img.binds.push_back([&d, src_pinned, count](unsigned char* dev, const std::function<size_t(const void*)>& image_offset, size_t) {
            d.alias(reinterpret_cast<T*>(dev + image_offset(src_pinned)), count);
        });
        return MAGE_OK;
    }
    MAGE_TRY(d.reserve(count));
    if (count) MAGE_HIP(hipMemcpyAsync(d.p, src_pinned, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
    return MAGE_OK;
}
// work array of `count` elements; fill >= 0: every byte starts as `fill`
template <typename T>
mage_status stage_array(mage_ba* h, DevBuf<T>& d, size_t count, int fill = -1)
{
    if (h->img.on) {
        if (fill >= 0) {
            T* q = nullptr;
            MAGE_TRY(h->build_arena.take(count, &q));
            std::memset(q, fill, count * sizeof(T));
            return stage_push(h, d, q, count);
        }
        const size_t off = h->img.scratch;
        h->img.scratch += (count * sizeof(T) + 255) & ~(size_t)255;
        h->img.binds.push_back([&d, off, count](unsigned char* dev, const std::function<size_t(const void*)>&, size_t up_bytes) { d.alias(reinterpret_cast<T*>(dev + up_bytes + off), count); });
        return MAGE_OK;
    }
    MAGE_TRY(d.reserve(count));
    if (fill >= 0 && count) MAGE_HIP(hipMemsetAsync(d.p, fill, count * sizeof(T), h->stream));
    return MAGE_OK;
}
mage_status stage_commit(mage_ba* h)
{
    if (!h->img.on) return MAGE_OK;
    PinnedArena& A = h->build_arena;
    // The arena is normally ONE 32 MB block and the image one copy.  A problem whose lists outgrow it (many cameras sharing few
    // points: the contribution lists grow with the square of a landmark's cameras) spills into further blocks: they become consecutive
    // parts of the image, one copy each.
    std::vector<size_t> base(A.blocks.size() + 1, 0);
    for (size_t b = 0; b < A.blocks.size(); ++b) base[b + 1] = base[b] + ((A.blocks[b].used + 255) & ~(size_t)255);
    const size_t up = base[A.blocks.size()];
    const std::function<size_t(const void*)> image_offset = [&A, &base](const void* q) -> size_t {
        const char* c = static_cast<const char*>(q);
        for (size_t b = 0; b < A.blocks.size(); ++b) if (c >= A.blocks[b].p && c < A.blocks[b].p + A.blocks[b].cap) return base[b] + (size_t)(c - A.blocks[b].p);
        return 0;      // (unreachable: every staged array was taken from this arena)
    };
    // views of an earlier image are about to be re-pointed, memory an earlier (larger) build owned goes back to the cache: nothing of it may be in flight
    MAGE_HIP(hipStreamSynchronize(h->stream));
    MAGE_TRY(h->d_image.reserve(up + h->img.scratch + 256));
    for (auto& b : h->img.binds) b(h->d_image.p, image_offset, up);
    h->img.binds.clear();
    for (size_t b = 0; b < A.blocks.size(); ++b)
        if (A.blocks[b].used) MAGE_HIP(hipMemcpyAsync(h->d_image.p + base[b], A.blocks[b].p, A.blocks[b].used, hipMemcpyHostToDevice, h->stream));
    return MAGE_OK;
}

mage_status ensure_pinned_mirrors(mage_ba* h)
{
    if (h->h_pinned) return MAGE_OK;
    const size_t off = ((SC_PAD + RES_CAP) * sizeof(double) + OUT_PREFIX * sizeof(uint32_t) + 255) & ~(size_t)255;      // scalars, (the estimate of a small problem,) then the id prefix: one copy
    MAGE_TRY(cached_pinned_alloc(&h->h_pinned, off + sizeof(PoseLmResult), &h->h_pinned_bytes));
    h->h_scal = static_cast<double*>(h->h_pinned);
    h->h_out_ids = reinterpret_cast<uint32_t*>(h->h_scal + SC_PAD + h->res_doubles);
    h->h_pose_lm = reinterpret_cast<PoseLmResult*>(static_cast<char*>(h->h_pinned) + off);
    return MAGE_OK;
}

// events are created when first needed: the profiling ones only under mage_ba_enable_profiling, the exchange pair on the first
// device-resident export / import -- a bundler per frame (the tracker) must not pay for ten event create / destroy pairs
mage_status ensure_events(hipEvent_t* ev, int n, unsigned flags)
{
    for (int i = 0; i < n; ++i)
        if (!ev[i]) MAGE_HIP(hipEventCreateWithFlags(&ev[i], flags));
    return MAGE_OK;
}

mage_status download_state(const mage_ba* hc)
{
    mage_ba* h = const_cast<mage_ba*>(hc);
    if (!h->state_on_device || h->host_state_fresh) return MAGE_OK;
    if (h->result_in_mirror) {             // the estimate came back with the last step's scalars
        const double* pose = h->h_scal + SC_PAD;
        const double* pts = pose + h->cams.size() * 8;
        for (size_t i = 0; i < h->cams.size(); ++i) {
            for (int a = 0; a < 4; ++a) h->cams[i].q[a] = pose[i * 8 + a];
            for (int a = 0; a < 3; ++a) h->cams[i].t[a] = pose[i * 8 + 4 + a];
        }
        if (!h->points_fixed)
            for (size_t i = 0; i < h->pt_set.size(); ++i)
                for (int a = 0; a < 3; ++a) h->pts[i * 3 + a] = pts[i * 4 + a];
        h->host_state_fresh = true;
        return MAGE_OK;
    }
    MAGE_DEVICE_SCOPE(h->device);
    // fixed points never change on the device: only the poses come back then (the tracker's per-frame call).  Pinned staging and
    // a polled event: a copy into pageable memory goes through the runtime's own staging and a blocking synchronise (~50 us more
    // on the local-BA problem, where the caller reads the state back after every optimisation, BundleAdjust.cpp:318-347)
    const size_t n_pose = h->cams.size() * 8, n_pts = h->points_fixed ? 0 : h->pt_set.size() * 4;
    PinnedArena stage((n_pose + n_pts + 2) * sizeof(double) + 1024);     // sized to what comes back, not the build arena's 32 MB blocks
    double *pose = nullptr, *pts = nullptr;
    MAGE_TRY(stage.take(n_pose + 1, &pose)); MAGE_TRY(stage.take(n_pts + 1, &pts));
    if (n_pose) MAGE_HIP(hipMemcpyAsync(pose, h->d_pose[h->cur].p, n_pose * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (n_pts) MAGE_HIP(hipMemcpyAsync(pts, h->d_pt[h->cur].p, n_pts * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MAGE_HIP(hipEventRecord(h->ev[3], h->stream));
    for (;;) {
        const hipError_t e = hipEventQuery(h->ev[3]);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) MAGE_HIP(e);
    }
    for (size_t i = 0; i < h->cams.size(); ++i) {
        for (int a = 0; a < 4; ++a) h->cams[i].q[a] = pose[i * 8 + a];
        for (int a = 0; a < 3; ++a) h->cams[i].t[a] = pose[i * 8 + 4 + a];
    }
    for (size_t i = 0; i * 4 < n_pts; ++i)
        for (int a = 0; a < 3; ++a) h->pts[i * 3 + a] = pts[i * 4 + a];
    h->host_state_fresh = true;
    return MAGE_OK;
}

// A setter that arrives after stepping started: pull the estimate back so the host copy is the truth again.
inline void mark_edited(mage_ba* h) { h->dirty = true; ++h->edit_gen; }

mage_status before_host_edit(mage_ba* h)
{
    if (h->state_on_device) {
        MAGE_TRY(download_state(h));
        h->state_on_device = false;
    }
    mark_edited(h);
    return MAGE_OK;
}

// `arena` non-null: the staging copies live in the caller's pinned arena and the caller synchronises before releasing it (the
// structure build does, once, for all its uploads); null: staged locally and synchronised here.
mage_status upload_state(mage_ba* h, PinnedArena* arena = nullptr)
{
    const size_t nc = h->cams.size(), np = h->pt_set.size();
    PinnedArena local;
    PinnedArena& A = arena ? *arena : local;
    double *pose = nullptr, *K = nullptr, *pts = nullptr;
    MAGE_TRY(A.take(nc * 8 + 1, &pose)); MAGE_TRY(A.take(nc * 4 + 1, &K)); MAGE_TRY(A.take(np * 4 + 1, &pts));
    for (size_t i = 0; i < nc; ++i) {
        const HostCam& c = h->cams[i];
        for (int a = 0; a < 4; ++a) pose[i * 8 + a] = c.q[a];
        for (int a = 0; a < 3; ++a) pose[i * 8 + 4 + a] = c.t[a];
        pose[i * 8 + 7] = 0.0;
        K[i * 4] = c.f; K[i * 4 + 1] = c.cx; K[i * 4 + 2] = c.cy; K[i * 4 + 3] = 0.0;
    }
    for (size_t i = 0; i < np; ++i) {
        for (int a = 0; a < 3; ++a) pts[i * 4 + a] = h->pts[i * 3 + a];
        pts[i * 4 + 3] = 0.0;
    }
    if (h->img.on && arena) {                   // image mode: each device array is a view of ITS part of the arena -- the second state buffer needs its own copy
        double *pose2 = nullptr, *pts2 = nullptr;
        MAGE_TRY(A.take(nc * 8 + 1, &pose2)); MAGE_TRY(A.take(np * 4 + 1, &pts2));
        std::memcpy(pose2, pose, nc * 8 * sizeof(double)); std::memcpy(pts2, pts, np * 4 * sizeof(double));
        MAGE_TRY(stage_push(h, h->d_pose[0], pose, nc * 8)); MAGE_TRY(stage_push(h, h->d_pose[1], pose2, nc * 8));
        MAGE_TRY(stage_push(h, h->d_pt[0], pts, np * 4)); MAGE_TRY(stage_push(h, h->d_pt[1], pts2, np * 4));
        MAGE_TRY(stage_push(h, h->d_camK, K, nc * 4));
    } else {
        for (int b = 0; b < 2; ++b) {
            MAGE_TRY(h->d_pose[b].upload(pose, nc * 8, h->stream));
            MAGE_TRY(h->d_pt[b].upload(pts, np * 4, h->stream));
        }
        MAGE_TRY(h->d_camK.upload(K, nc * 4, h->stream));
    }
    if (!arena) MAGE_HIP(hipStreamSynchronize(h->stream));
    h->cur = 0;
    h->state_on_device = true;
    h->host_state_fresh = true;
    return MAGE_OK;
}

void refresh_view_state(mage_ba* h)
{
    h->view.pose_cur = h->d_pose[h->cur].p; h->view.pose_trial = h->d_pose[h->cur ^ 1].p;
    h->view.pt_cur = h->d_pt[h->cur].p; h->view.pt_trial = h->d_pt[h->cur ^ 1].p;
}

// SparseOptimizer::initializeOptimization + BlockSolver::buildStructure, re-expressed as flat CSR arrays.
// MAGE_BA_TIMING=1 prints the host phases of the structure build to stderr (diagnostics only).
// The structure build is host work between two GPU phases; its independent loops run on a few short-lived threads
// (MAGE_HOST_THREADS, default min(8, hardware threads); 1 = the calling thread only).
int host_threads()
{
    static const int n = [] {
        const char* e = std::getenv("MAGE_HOST_THREADS");
        int t = e ? std::atoi(e) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1, std::min(t, 64));
    }();
    return n;
}

// fn(begin, end, part) over [0, n) split into equal contiguous parts; part boundaries are a function of (n, parts) only.
// Small problems (the tracking thread's pose-only solves, local BA) stay on the calling thread.
inline int parts_for(int n, int grain) { return std::max(1, std::min(host_threads(), n / std::max(grain, 1))); }

template <typename F>
void parallel_ranges(int n, int parts, F&& fn)
{
    parts = std::max(1, std::min(parts, n));
    if (parts == 1) { fn(0, n, 0); return; }
    std::vector<std::thread> th;
    th.reserve(parts - 1);
    for (int p = 1; p < parts; ++p)
        th.emplace_back([&, p] { fn((int)((int64_t)n * p / parts), (int)((int64_t)n * (p + 1) / parts), p); });
    fn(0, (int)((int64_t)n / parts), 0);
    for (auto& t : th) t.join();
}

struct PhaseTimer {
    bool on; std::chrono::steady_clock::time_point t0;
    PhaseTimer() : on(std::getenv("MAGE_BA_TIMING") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mage_ba structure] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

struct ListSizes {
    int nL = 0, nfc = 0, nlm = 0, nw = 0, nblk = 0, n_blk_slots = 0;
    size_t ncon = 0;
    bool dup_slots = false;
};

// Which twin builds the lists (ba_build.h).  MAGE_BA_BUILD=host|device forces one (A/B, tests); by default the device builds
// everything except the tracker's per-frame pose-only problems (a few hundred observations of fixed points: the three short host
// loops cost less than the two read-backs of the device build).
// MAGE_BA_CONSERVATIVE=1 takes every fall-back this file has for a device or a runtime that lacks something the fast paths rely on -- the
// host-built lists as per-buffer uploads instead of one image, read-back copies instead of the kernels writing the pinned mirror, the
// pose-only solve with its arrays in HBM behind an upload, the outlier pass as a call of its own instead of queued behind the last trial --
// so that the tests can hold them against the defaults (same results, tests/test_ba_gpu.py).
bool conservative_paths()
{
    static const bool on = std::getenv("MAGE_BA_CONSERVATIVE") != nullptr;
    return on;
}

bool use_device_build(const mage_ba* h, size_t n_obs)
{
    const char* e = std::getenv("MAGE_BA_BUILD");
    if (e && e[0] == 'h') return false;
    if (e && e[0] == 'd') return true;
    // (round 4: with the host-built lists going up as ONE image -- ImageStager -- the host build wins up to a few thousand observations:
    // what a small problem pays for is the NUMBER of operations on the stream, ~40 for the device build's kernels, fills and read-backs)
    constexpr size_t host_max = 10000;      // measured (one-iteration bundler, create -> destroy): 2 000 observations 0.17 against 0.26 ms, 4 000: 0.27 / 0.36, 8 000: 0.38 / 0.40, 16 000: 0.60 / 0.52
    return !h->points_fixed && n_obs >= host_max;
}

// The lists on the host (the A/B twin of ba_build.hip): SparseOptimizer::initializeOptimization + BlockSolver::buildStructure,
// re-expressed as flat CSR arrays.
mage_status build_lists_host(mage_ba* h, PinnedArena& arena, const std::vector<int>& cam_extra_deg, PhaseTimer& tm, ListSizes& Z,
                             std::vector<int>& cam2hc, std::vector<uint32_t>& L_edge)
{
    const int nc = (int)h->cams.size(), np = (int)h->pt_set.size();
    const size_t no = h->obs.size();
    // active observations: set, not removed, not (camera fixed and points fixed)
    std::vector<uint32_t> active;
    active.reserve(no);
    std::vector<int> cam_deg(cam_extra_deg), pt_deg(np, 0);
    for (size_t e = 0; e < no; ++e) {
        const HostObs& o = h->obs[e];
        if (!o.set || o.removed) continue;
        if (h->cams[o.cam].fixed && h->points_fixed) continue;
        active.push_back((uint32_t)e);
        cam_deg[o.cam]++; pt_deg[o.pt]++;
    }
    const int nL = (int)active.size();
    std::vector<int> hc2cam;
    cam2hc.assign(nc, -1);
    for (int i = 0; i < nc; ++i)
        if ((cam_deg[i] > 0 || h->shard_ranks > 0) && !h->cams[i].fixed) { cam2hc[i] = (int)hc2cam.size(); hc2cam.push_back(i); }   // sharded: the ranks' systems must have one shape
    const int nfc = (int)hc2cam.size();
    std::vector<int> pt2lm(np, -1), lm_pt;
    for (int i = 0; i < np; ++i)
        if (pt_deg[i] > 0) { pt2lm[i] = (int)lm_pt.size(); lm_pt.push_back(i); }
    const int nlm = (int)lm_pt.size();
    const bool points_free = !h->points_fixed;

    tm.mark("active sets");
    auto push = [&](auto& dbuf, const auto* src, size_t count) -> mage_status { return stage_push(h, dbuf, src, count); };
    auto push_vec = [&](auto& dbuf, const auto& vec) -> mage_status {
        typename std::remove_reference<decltype(vec)>::type::value_type* q = nullptr;
        MAGE_TRY(arena.take(vec.size(), &q));
        if (!vec.empty()) std::memcpy(q, vec.data(), vec.size() * sizeof(*q));
        return push(dbuf, q, vec.size());
    };
    MAGE_TRY(push_vec(h->d_cam2hc, cam2hc));
    MAGE_TRY(push_vec(h->d_hc2cam, hc2cam));

    // landmark-ordered observation list
    std::vector<int> lm_ptr(nlm + 1, 0);
    for (int a = 0; a < nL; ++a) lm_ptr[pt2lm[h->obs[active[a]].pt] + 1]++;
    for (int l = 0; l < nlm; ++l) lm_ptr[l + 1] += lm_ptr[l];
    L_edge.assign(nL, 0);
    {
        std::vector<int> fill(lm_ptr.begin(), lm_ptr.end() - 1);
        for (int a = 0; a < nL; ++a) L_edge[fill[pt2lm[h->obs[active[a]].pt]]++] = active[a];
    }
    // inside a landmark: free cameras ascending (ties by observation index), then fixed cameras
    auto key = [&](uint32_t e) -> uint64_t {
        int hcv = cam2hc[h->obs[e].cam];
        return ((uint64_t)(hcv < 0 ? 0x7fffffff : hcv) << 32) | e;
    };
    float2* L_uv = nullptr; float* L_info = nullptr; uint32_t *L_cam = nullptr, *L_pt = nullptr; int *L_slot = nullptr, *w_hc = nullptr, *w_lm = nullptr;
    MAGE_TRY(arena.take(nL, &L_uv)); MAGE_TRY(arena.take(nL, &L_info)); MAGE_TRY(arena.take(nL, &L_cam)); MAGE_TRY(arena.take(nL, &L_pt));
    MAGE_TRY(arena.take(nL, &L_slot)); MAGE_TRY(arena.take(nL, &w_hc)); MAGE_TRY(arena.take(nL, &w_lm));        // slots <= observations
    std::vector<int> lm_wptr(nlm + 1, 0);
    const int lm_parts = parts_for(nlm, 8192);
    // order inside each landmark, and its number of slots (distinct free cameras)
    parallel_ranges(nlm, lm_parts, [&](int l0, int l1, int) {
        std::vector<uint64_t> keys;
        for (int l = l0; l < l1; ++l) {
            const int b = lm_ptr[l], k = lm_ptr[l + 1] - b;
            bool sorted = true;
            keys.resize(k);
            for (int i = 0; i < k; ++i) { keys[i] = key(L_edge[b + i]); if (i && keys[i] < keys[i - 1]) sorted = false; }
            if (!sorted) {
                std::sort(keys.begin(), keys.end());
                for (int i = 0; i < k; ++i) L_edge[b + i] = (uint32_t)(keys[i] & 0xffffffffu);
            }
            int slots = 0;
            if (points_free) {
                uint32_t prev = 0xffffffffu;
                for (int i = 0; i < k; ++i) {
                    const uint32_t hcv = (uint32_t)(keys[i] >> 32);
                    if (hcv != 0x7fffffffu && hcv != prev) { ++slots; prev = hcv; }
                }
            }
            lm_wptr[l + 1] = slots;
        }
    });
    for (int l = 0; l < nlm; ++l) lm_wptr[l + 1] += lm_wptr[l];
    const int nw = lm_wptr[nlm];
    parallel_ranges(nlm, lm_parts, [&](int l0, int l1, int) {
        for (int l = l0; l < l1; ++l) {
            int prev = -1, w = lm_wptr[l];
            for (int i = lm_ptr[l]; i < lm_ptr[l + 1]; ++i) {
                const HostObs& o = h->obs[L_edge[i]];
                L_uv[i] = make_float2(o.u, o.v); L_info[i] = o.info; L_cam[i] = o.cam; L_pt[i] = o.pt;
                const int hcv = cam2hc[o.cam];
                int slot = -1;
                if (points_free && hcv >= 0) {
                    if (hcv != prev) { w_hc[w] = hcv; w_lm[w] = l; ++w; prev = hcv; }
                    slot = w - 1;
                }
                L_slot[i] = slot;
            }
        }
    });
    // several observations of one landmark by one free camera share a W slot (their blocks are summed in order); the small-problem
    // linearisation gives every observation its own lane only when that never happens
    {
        std::vector<long long> part_count(lm_parts, 0);
        parallel_ranges(nlm, lm_parts, [&](int l0, int l1, int part) {
            long long c = 0;
            for (int i = lm_ptr[l0]; i < lm_ptr[l1]; ++i) c += L_slot[i] >= 0 ? 1 : 0;
            part_count[part] = c;
        });
        long long slot_obs = 0;
        for (long long c : part_count) slot_obs += c;
        Z.dup_slots = slot_obs != (long long)nw;
    }
    MAGE_TRY(push(h->d_L_uv, L_uv, nL)); MAGE_TRY(push(h->d_L_info, L_info, nL)); MAGE_TRY(push(h->d_L_cam, L_cam, nL));
    MAGE_TRY(push(h->d_L_pt, L_pt, nL)); MAGE_TRY(push(h->d_L_slot, L_slot, nL));
    MAGE_TRY(push(h->d_w_hc, w_hc, nw)); MAGE_TRY(push(h->d_w_lm, w_lm, nw));
    MAGE_TRY(push_vec(h->d_L_edge, L_edge));
    MAGE_TRY(push_vec(h->d_lm_ptr, lm_ptr)); MAGE_TRY(push_vec(h->d_lm_pt, lm_pt)); MAGE_TRY(push_vec(h->d_lm_wptr, lm_wptr));

    tm.mark("landmark lists");
    // per-camera lists
    // Both are counting sorts by camera that keep the input order (observation index for camE, slot index for camS); each
    // thread counts a contiguous range of the input, the offsets are the camera's base plus the counts of the ranges before it.
    std::vector<int> camE_ptr(nfc + 1, 0), camS_ptr(nfc + 1, 0);
    int *camE = nullptr, *camS = nullptr;
    {
        const int parts = parts_for(nL, 65536);
        std::vector<int> where(no, -1);                 // observation index -> landmark-order position
        parallel_ranges(nL, parts, [&](int i0, int i1, int) { for (int i = i0; i < i1; ++i) where[L_edge[i]] = i; });
        auto stable_by_camera = [&](int count, auto&& camera_of, auto&& item_of, std::vector<int>& ptr, int** out) -> mage_status {
            const int np_ = std::max(1, std::min(parts, count));
            std::vector<std::vector<int>> cnt(np_, std::vector<int>(nfc, 0));
            parallel_ranges(count, np_, [&](int a0, int a1, int part) {
                for (int a = a0; a < a1; ++a) { const int c = camera_of(a); if (c >= 0) cnt[part][c]++; }
            });
            for (int c = 0; c < nfc; ++c) {
                int base = ptr[c];
                for (int part = 0; part < np_; ++part) { const int k = cnt[part][c]; cnt[part][c] = base; base += k; }
                ptr[c + 1] = base;
            }
            MAGE_TRY(arena.take((size_t)ptr[nfc], out));
            int* dst = *out;
            parallel_ranges(count, np_, [&](int a0, int a1, int part) {
                for (int a = a0; a < a1; ++a) { const int c = camera_of(a); if (c >= 0) dst[cnt[part][c]++] = item_of(a); }
            });
            return MAGE_OK;
        };
        // a camera's observations in ascending observation index: `active` is ascending and L_edge is a permutation of it
        MAGE_TRY(stable_by_camera(nL, [&](int a) { return cam2hc[L_cam[where[active[a]]]]; }, [&](int a) { return where[active[a]]; }, camE_ptr, &camE));
        MAGE_TRY(stable_by_camera(nw, [&](int s2) { return w_hc[s2]; }, [&](int s2) { return s2; }, camS_ptr, &camS));
    }
    MAGE_TRY(push(h->d_camE, camE, (size_t)camE_ptr[nfc])); MAGE_TRY(push(h->d_camS, camS, (size_t)nw));
    MAGE_TRY(push_vec(h->d_camE_ptr, camE_ptr)); MAGE_TRY(push_vec(h->d_camS_ptr, camS_ptr));

    tm.mark("camera lists");
    // reduced-camera-matrix blocks: contributions (slot_a, slot_b), a <= b inside a landmark, ordered by
    // (i, j) with landmark order preserved inside a block.
    size_t ncon = 0;
    for (int l = 0; l < nlm; ++l) { size_t k = (size_t)(lm_wptr[l + 1] - lm_wptr[l]); ncon += k * (k + 1) / 2; }
    if (ncon > (size_t)0x7fffffff)      // block offsets and contribution indices are 32-bit on the device
        return fail(MAGE_ERR_UNSUPPORTED, "%zu Schur contributions exceed the 32-bit index range of the block lists (tracks too long)", ncon);
    // Row i of the block structure is built from the slots of camera i (camS: ascending slot = ascending landmark): every
    // later slot b >= a of the same landmark is a camera j >= i.  A counting sort over j inside the row (two passes over a
    // few thousand contributions, cache resident) replaces a global sort of all contributions.
    std::vector<int> blk_ptr; std::vector<int2> blk_ij;
    int2* con = nullptr;
    MAGE_TRY(arena.take(ncon, &con));
    {
        // pass 1 (parallel over rows): the blocks of each row and their sizes
        const int parts = ncon >= 262144 ? parts_for(nfc, 8) : 1;
        std::vector<std::vector<int2>> row_blocks(nfc);                 // (j, count) ascending j
        parallel_ranges(nfc, parts, [&](int r0, int r1, int) {
            std::vector<int> cnt(nfc, 0), touched;
            for (int i = r0; i < r1; ++i) {
                touched.clear();
                touched.push_back(i);    // the diagonal block of every free camera exists even with no landmark contribution
                for (int k = camS_ptr[i]; k < camS_ptr[i + 1]; ++k) {
                    const int a = camS[k], end = lm_wptr[w_lm[a] + 1];
                    for (int b = a; b < end; ++b) { const int j = w_hc[b]; if (cnt[j]++ == 0 && j != i) touched.push_back(j); }
                }
                std::sort(touched.begin(), touched.end());
                row_blocks[i].reserve(touched.size());
                for (int j : touched) { row_blocks[i].push_back(make_int2(j, cnt[j])); cnt[j] = 0; }
            }
        });
        // block offsets (serial prefix sum)
        size_t nb = 0;
        for (int i = 0; i < nfc; ++i) nb += row_blocks[i].size();
        blk_ptr.reserve(nb + 1); blk_ij.reserve(nb);
        std::vector<size_t> row_first(nfc + 1, 0);
        size_t q = 0;
        for (int i = 0; i < nfc; ++i) {
            row_first[i] = blk_ptr.size();
            for (const int2& jc : row_blocks[i]) { blk_ptr.push_back((int)q); blk_ij.push_back(make_int2(i, jc.x)); q += (size_t)jc.y; }
        }
        row_first[nfc] = blk_ptr.size();
        blk_ptr.push_back((int)ncon);
        // pass 2 (parallel over rows): scatter the contributions of a row into its blocks
        parallel_ranges(nfc, parts, [&](int r0, int r1, int) {
            std::vector<size_t> start(nfc, 0);
            for (int i = r0; i < r1; ++i) {
                for (size_t k = row_first[i]; k < row_first[i + 1]; ++k) start[blk_ij[k].y] = (size_t)blk_ptr[k];
                for (int k = camS_ptr[i]; k < camS_ptr[i + 1]; ++k) {
                    const int a = camS[k], end = lm_wptr[w_lm[a] + 1];
                    for (int b = a; b < end; ++b) con[start[w_hc[b]]++] = make_int2(a, b);
                }
            }
        });
    }
    const int nblk = (int)blk_ij.size();
    MAGE_TRY(push(h->d_con, con, ncon));
    MAGE_TRY(push_vec(h->d_blk_ptr, blk_ptr)); MAGE_TRY(push_vec(h->d_blk_ij, blk_ij));
    // slot -> block table of k_schur_block: workgroup w (SCHUR_WAVES wavefront slots) lands on XCD w % 8 and takes the next blocks of that XCD's run
    std::vector<int> blk_order;
    {
        constexpr int XCD = 8;
        std::vector<std::vector<int>> per(XCD);
        // contiguous runs of rows per XCD, cut at equal shares of the contributions: blocks whose cameras are close see the
        // same landmarks, so their W blocks meet in one L2
        for (int b2 = 0; b2 < nblk; ++b2) {
            const size_t mid = ((size_t)blk_ptr[b2] + (size_t)blk_ptr[b2 + 1]) / 2;
            const int x = ncon ? (int)std::min<size_t>(XCD - 1, mid * XCD / ncon) : 0;
            per[x].push_back(b2);
        }
        size_t longest = 0;
        for (auto& v2 : per) longest = std::max(longest, v2.size());
        constexpr size_t G = SCHUR_WAVES;
        const size_t groups = (longest + G - 1) / G;
        blk_order.assign(groups * XCD * G, -1);
        for (int x = 0; x < XCD; ++x)
            for (size_t q2 = 0; q2 < per[x].size(); ++q2) blk_order[((q2 / G) * XCD + x) * G + (q2 % G)] = per[x][q2];
    }
    MAGE_TRY(push_vec(h->d_blk_order, blk_order));
    tm.mark("schur contributions");
    Z.nL = nL; Z.nfc = nfc; Z.nlm = nlm; Z.nw = nw; Z.nblk = nblk; Z.n_blk_slots = (int)blk_order.size(); Z.ncon = ncon;
    return MAGE_OK;
}

// Small device -> host read-back in the middle of a build: copy, then poll (a blocking synchronise costs 20-30 us of wake-up).
mage_status read_back(mage_ba* h, void* dst_pinned, const void* src_device, size_t bytes)
{
    MAGE_HIP(hipMemcpyAsync(dst_pinned, src_device, bytes, hipMemcpyDeviceToHost, h->stream));
    MAGE_HIP(hipEventRecord(h->ev[3], h->stream));
    for (;;) {
        const hipError_t e = hipEventQuery(h->ev[3]);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) MAGE_HIP(e);
    }
    return MAGE_OK;
}

// The lists on the device (ba_build.hip): the records go up as the setters left them, two small read-backs bring the sizes back.
mage_status build_lists_device(mage_ba* h, PinnedArena& arena, const std::vector<int>& cam_extra_deg, bool have_tethers, PhaseTimer& tm,
                               ListSizes& Z, std::vector<int>& cam2hc)
{
    static_assert(SCHUR_WAVES == 1, "the device build lays the slot -> block table out for one wavefront per workgroup");
    const int nc = (int)h->cams.size(), np = (int)h->pt_set.size();
    const size_t no = h->obs.size();
    hipStream_t st = h->stream;
    BuildArgs a{};
    // ---- inputs
    MAGE_TRY(h->d_obs_raw.reserve(no + 1));
    if (no) MAGE_HIP(hipMemcpyAsync(h->d_obs_raw.p, h->obs.data(), no * sizeof(ObsRecord), hipMemcpyHostToDevice, st));   // pinned: one DMA
    uint8_t* fixed = nullptr;
    MAGE_TRY(arena.take((size_t)nc + 1, &fixed));
    for (int i = 0; i < nc; ++i) fixed[i] = h->cams[i].fixed;
    MAGE_TRY(h->d_cam_fixed.upload(fixed, (size_t)nc + 1, st));
    if (have_tethers) {
        int* extra = nullptr;
        MAGE_TRY(arena.take((size_t)nc + 1, &extra));
        std::memcpy(extra, cam_extra_deg.data(), (size_t)nc * sizeof(int));
        MAGE_TRY(h->d_cam_extra.upload(extra, (size_t)nc + 1, st));
    }
    a.obs = h->d_obs_raw.p; a.n_obs = (int)no;
    a.cam_fixed = h->d_cam_fixed.p; a.n_cams = nc; a.cam_extra_deg = have_tethers ? h->d_cam_extra.p : nullptr;
    a.n_pts = np; a.points_fixed = h->points_fixed ? 1 : 0; a.keep_all_free_cameras = h->shard_ranks > 0 ? 1 : 0;
    // ---- outputs and scratch of phase 1, sized by what was allocated through the surface
    int n_fc_max = 0;
    for (int i = 0; i < nc; ++i) n_fc_max += h->cams[i].fixed ? 0 : 1;
    if (n_fc_max * 6 > CHOL_MAX_ORDER && h->shard_ranks == 0) n_fc_max = std::min(n_fc_max, nc);      // (refused below once the exact count is known)
    const size_t lm_max = std::min<size_t>((size_t)np, no);
    MAGE_TRY(h->d_cam2hc.reserve((size_t)nc + 1)); MAGE_TRY(h->d_hc2cam.reserve((size_t)nc + 1));
    MAGE_TRY(h->d_L_uv.reserve(no + 1)); MAGE_TRY(h->d_L_info.reserve(no + 1)); MAGE_TRY(h->d_L_cam.reserve(no + 1)); MAGE_TRY(h->d_L_pt.reserve(no + 1));
    MAGE_TRY(h->d_L_slot.reserve(no + 1)); MAGE_TRY(h->d_L_edge.reserve(no + 1));
    MAGE_TRY(h->d_lm_ptr.reserve(lm_max + 2)); MAGE_TRY(h->d_lm_pt.reserve(lm_max + 1)); MAGE_TRY(h->d_lm_wptr.reserve(lm_max + 2));
    MAGE_TRY(h->d_w_hc.reserve(no + 1 + BUILD_ROW_KC)); MAGE_TRY(h->d_w_lm.reserve(no + 1)); MAGE_TRY(h->d_b_w_end.reserve(no + 1));
    MAGE_TRY(h->d_camE.reserve(no + 1)); MAGE_TRY(h->d_camS.reserve(no + 1));
    MAGE_TRY(h->d_camE_ptr.reserve((size_t)n_fc_max + 2)); MAGE_TRY(h->d_camS_ptr.reserve((size_t)n_fc_max + 2));
    MAGE_TRY(h->d_b_zeroed.reserve(build_zeroed_bytes(nc, np))); MAGE_TRY(h->d_b_pt2lm.reserve((size_t)np + 1));
    MAGE_TRY(h->d_b_bucket.reserve(no + 1)); MAGE_TRY(h->d_b_L_hc.reserve(no + 1)); MAGE_TRY(h->d_b_L_lm.reserve(no + 1)); MAGE_TRY(h->d_b_where.reserve(no + 1));
    MAGE_TRY(h->d_b_scan.reserve(build_scan_tmp_elems(std::max<size_t>(std::max<size_t>(no, (size_t)np), (size_t)nc))));
    MAGE_TRY(h->d_b_hist.reserve(build_hist_ints((int)no, n_fc_max)));
    MAGE_TRY(h->d_b_row.reserve((size_t)n_fc_max + 2));
    a.cam2hc = h->d_cam2hc.p; a.hc2cam = h->d_hc2cam.p;
    a.L_uv = h->d_L_uv.p; a.L_info = h->d_L_info.p; a.L_cam = h->d_L_cam.p; a.L_pt = h->d_L_pt.p; a.L_slot = h->d_L_slot.p; a.L_edge = h->d_L_edge.p;
    a.lm_ptr = h->d_lm_ptr.p; a.lm_pt = h->d_lm_pt.p; a.lm_wptr = h->d_lm_wptr.p; a.w_hc = h->d_w_hc.p; a.w_lm = h->d_w_lm.p;
    a.camE_ptr = h->d_camE_ptr.p; a.camE = h->d_camE.p; a.camS_ptr = h->d_camS_ptr.p; a.camS = h->d_camS.p;
    a.counts = reinterpret_cast<BuildCounts*>(h->d_b_zeroed.p);
    a.cam_deg = reinterpret_cast<int*>(a.counts + 1); a.pt_deg = a.cam_deg + nc + 1;
    a.pt2lm = h->d_b_pt2lm.p;
    a.bucket = h->d_b_bucket.p; a.L_hc = h->d_b_L_hc.p; a.L_lm = h->d_b_L_lm.p; a.where = h->d_b_where.p; a.w_end = h->d_b_w_end.p;
    a.scan_tmp = h->d_b_scan.p; a.hist = h->d_b_hist.p; a.row = h->d_b_row.p;
    tm.mark("device: uploads + reserves");
    build_launch_phase1(a, n_fc_max, st);
    tm.mark("device: phase 1 enqueued");
    BuildCounts* hc = nullptr;
    MAGE_TRY(arena.take(1, &hc));
    MAGE_TRY(read_back(h, hc, a.counts, sizeof(BuildCounts)));
    tm.mark("device: maps, landmark order, slots, camera views, rows counted");
    Z.nL = hc->n_L; Z.nfc = hc->n_fc; Z.nlm = hc->n_lm; Z.nw = hc->n_w; Z.ncon = (size_t)hc->n_con; Z.nblk = hc->n_blk;
    Z.dup_slots = hc->slot_obs != hc->n_w;
    if (Z.ncon > (size_t)0x7fffffff)
        return fail(MAGE_ERR_UNSUPPORTED, "%zu Schur contributions exceed the 32-bit index range of the block lists (tracks too long)", Z.ncon);
    if (Z.nfc * 6 > CHOL_MAX_ORDER) return fail(MAGE_ERR_UNSUPPORTED, "reduced camera system of order %d exceeds %d (one resident workgroup per tile column)", Z.nfc * 6, CHOL_MAX_ORDER);
    if (have_tethers) {                      // the tether gather lists are built on the host from the index map (a few entries)
        int* c2h = nullptr;
        MAGE_TRY(arena.take((size_t)nc + 1, &c2h));
        MAGE_TRY(read_back(h, c2h, a.cam2hc, (size_t)nc * sizeof(int)));
        cam2hc.assign(c2h, c2h + nc);
    }
    // ---- the block lists of S
    const int nfc = Z.nfc;
    MAGE_TRY(h->d_con.reserve(Z.ncon + 1));
    MAGE_TRY(h->d_blk_ptr.reserve((size_t)Z.nblk + 2)); MAGE_TRY(h->d_blk_ij.reserve((size_t)Z.nblk + 1));
    a.con = h->d_con.p; a.blk_ptr = h->d_blk_ptr.p; a.blk_ij = h->d_blk_ij.p;
    // the slot -> block table is k_schur_block's (large problems): the small-problem path walks the blocks in order
    // (same predicate as ba_small_applies; a forced build mode is a test comparing every list)
    const bool small_path = h->shard_ranks == 0 && ba_small_shape_applies(nfc, have_tethers ? 1 : 0, Z.nL);      // lm_solve's own predicate (a sharded map never takes the small path)
    const bool want_xcd = !small_path || std::getenv("MAGE_BA_BUILD") != nullptr;
    if (nfc > 0) {
        build_launch_row_fill(a, nfc, want_xcd, st);
        if (want_xcd) {
            MAGE_TRY(read_back(h, hc, a.counts, sizeof(BuildCounts)));
            Z.n_blk_slots = hc->xcd_longest * 8;
            MAGE_TRY(h->d_blk_order.reserve((size_t)Z.n_blk_slots + 1));
            a.blk_order = h->d_blk_order.p;
            build_launch_blk_order(a, Z.n_blk_slots, st);
        } else {
            MAGE_TRY(h->d_blk_order.reserve(1));
            Z.n_blk_slots = 0;
        }
    } else {
        // no free camera: empty camera views (the offset arrays are still read)
        MAGE_HIP(hipMemsetAsync(h->d_camE_ptr.p, 0, 2 * sizeof(int), st)); MAGE_HIP(hipMemsetAsync(h->d_camS_ptr.p, 0, 2 * sizeof(int), st));
        MAGE_HIP(hipMemsetAsync(h->d_blk_ptr.p, 0, 2 * sizeof(int), st));
        MAGE_TRY(h->d_blk_order.reserve(1));
        Z.nblk = 0; Z.n_blk_slots = 0;
    }
    MAGE_HIP(hipGetLastError());
    tm.mark("device: camera views, block lists");
    return MAGE_OK;
}

mage_status initialize_optimization(mage_ba* h)
{
    PhaseTimer tm;
    MAGE_DEVICE_SCOPE(h->device);
    const int nc = (int)h->cams.size(), np = (int)h->pt_set.size();
    // Lists that go to the device are built in pinned memory and copied as soon as they are complete, so the DMA overlaps
    // the rest of the build; small ones are staged through the same arena.  The arena is the handle's and goes back to the cache
    // when the first LM trial's scalars have come back (read_scalars): the build ends without a synchronisation of its own, so the
    // first iteration's launches queue up behind the last build kernels.
    ensure_obs_filled(h);
    PinnedArena& arena = h->build_arena;
    if (!arena.blocks.empty()) { MAGE_HIP(hipStreamSynchronize(h->stream)); arena.release(); }
    h->join_warmup();
    const bool on_device = use_device_build(h, h->obs.size());
    // A new pinned block costs ~0.37 ms per MB in a process that has none parked (5.5 ms to allocate 32 MB, 6.5 ms more when the first copy
    // maps it for the device): the device build stages only the state and a few short lists through the arena (the observation records go
    // up from where the setters wrote them), so its first block is sized for that, not for a host build's lists.
    arena.min_block = on_device ? ((size_t)1 << 20) + ((size_t)nc * 12 + (size_t)np * 4) * sizeof(double) : (size_t)32 << 20;
    // small problems built on the host go to the device as ONE image (stage_push / stage_array / stage_commit above)
    h->img.on = !on_device && !conservative_paths() && !h->state_on_device && h->shard_ranks == 0 && h->obs.size() <= 65536 && nc <= 4096 && np <= 65536;
    h->img.scratch = 0; h->img.binds.clear();
    struct ImageOff { mage_ba* h; ~ImageOff() { h->img.on = false; h->img.binds.clear(); } } image_off_at_exit{ h };      // every exit leaves the mode off
    if (!h->state_on_device) MAGE_TRY(upload_state(h, &arena));
    else {
        // Trials only write the entities that are in the system, and accepting a trial swaps the two
        // state buffers; an entity that just left the system (all its observations removed) must
        // therefore hold its kept estimate in BOTH buffers.
        if (nc) MAGE_HIP(hipMemcpyAsync(h->d_pose[h->cur ^ 1].p, h->d_pose[h->cur].p, (size_t)nc * 8 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        if (np) MAGE_HIP(hipMemcpyAsync(h->d_pt[h->cur ^ 1].p, h->d_pt[h->cur].p, (size_t)np * 4 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    }
    tm.mark("state upload");

    // tether edges: active unless both endpoints are fixed (OptimizableGraph::Edge::allVerticesFixed); they keep a
    // free camera in the system even when it has no observation
    std::vector<int> T_kind; std::vector<int2> T_cam, T_fixed; std::vector<double> T_meas, T_w;
    std::vector<int> cam_extra_deg(nc, 0);
    for (int k = 0; k < 3; ++k)
        for (const HostTether& t : h->teth[k]) {
            if (!t.set) continue;
            if (h->cams[t.c0].fixed && h->cams[t.c1].fixed) continue;
            T_kind.push_back(k);
            T_cam.push_back(make_int2((int)t.c0, (int)t.c1));
            T_fixed.push_back(make_int2(h->cams[t.c0].fixed ? 1 : 0, h->cams[t.c1].fixed ? 1 : 0));
            for (int a = 0; a < 4; ++a) T_meas.push_back(t.q[a]);
            for (int a = 0; a < 3; ++a) T_meas.push_back(t.t[a]);
            T_meas.push_back(t.dist);
            T_w.push_back(t.w);
            cam_extra_deg[t.c0]++; cam_extra_deg[t.c1]++;
        }
    const int nT = (int)T_kind.size();

    ListSizes Z;
    std::vector<int> cam2hc;
    std::vector<uint32_t> L_edge;
    if (on_device) MAGE_TRY(build_lists_device(h, arena, cam_extra_deg, nT > 0, tm, Z, cam2hc));
    else MAGE_TRY(build_lists_host(h, arena, cam_extra_deg, tm, Z, cam2hc, L_edge));
    const int nL = Z.nL, nfc = Z.nfc, nlm = Z.nlm, nw = Z.nw, nblk = Z.nblk;
    const size_t ncon = Z.ncon;
    h->dup_slots = Z.dup_slots;
    const bool points_free = !h->points_fixed;
    h->useless = (nfc + (points_free ? nlm : 0)) == 0;
    if (h->shard_ranks > 0) {
        if (nfc == 0) return fail(MAGE_ERR_UNSUPPORTED, "a landmark-sharded map needs a free camera (without one the landmarks are independent: solve them unsharded)");
        h->useless = false;              // a rank without landmarks still takes part in every exchange
    }
    hipStream_t st = h->stream;
    auto push = [&](auto& dbuf, const auto* src, size_t count) -> mage_status { return stage_push(h, dbuf, src, count); };
    auto push_vec = [&](auto& dbuf, const auto& vec) -> mage_status {
        typename std::remove_reference<decltype(vec)>::type::value_type* q = nullptr;
        MAGE_TRY(arena.take(vec.size(), &q));
        if (!vec.empty()) std::memcpy(q, vec.data(), vec.size() * sizeof(*q));
        return push(dbuf, q, vec.size());
    };

    // tether gather lists: per camera (tether, side) and per free-camera pair i < j (tether, transposed), tether order kept
    std::vector<int> tc_hc, tc_ptr{ 0 }, tc_item, tp_ptr{ 0 }, tp_item; std::vector<int2> tp_ij;
    if (nT > 0) {
        std::vector<std::pair<int, int>> ci;                    // (hc, item)
        std::vector<std::pair<uint64_t, int>> pi;               // ((i << 32) | j, item)
        for (int t = 0; t < nT; ++t) {
            const int h0 = cam2hc[T_cam[t].x], h1 = cam2hc[T_cam[t].y];
            if (h0 >= 0) ci.push_back({ h0, t * 2 });
            if (h1 >= 0) ci.push_back({ h1, t * 2 + 1 });
            if (h0 >= 0 && h1 >= 0) {
                if (h0 < h1) pi.push_back({ ((uint64_t)h0 << 32) | (uint32_t)h1, t * 2 });
                else pi.push_back({ ((uint64_t)h1 << 32) | (uint32_t)h0, t * 2 + 1 });
            }
        }
        std::stable_sort(ci.begin(), ci.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first < b.first; });
        std::stable_sort(pi.begin(), pi.end(), [](const std::pair<uint64_t, int>& a, const std::pair<uint64_t, int>& b) { return a.first < b.first; });
        for (size_t k = 0; k < ci.size(); ++k) {
            if (k == 0 || ci[k].first != ci[k - 1].first) { if (k) tc_ptr.push_back((int)k); tc_hc.push_back(ci[k].first); }
            tc_item.push_back(ci[k].second);
        }
        if (!ci.empty()) tc_ptr.push_back((int)ci.size());
        for (size_t k = 0; k < pi.size(); ++k) {
            if (k == 0 || pi[k].first != pi[k - 1].first) {
                if (k) tp_ptr.push_back((int)k);
                tp_ij.push_back(make_int2((int)(pi[k].first >> 32), (int)(pi[k].first & 0xffffffffu)));
            }
            tp_item.push_back(pi[k].second);
        }
        if (!pi.empty()) tp_ptr.push_back((int)pi.size());
    }

    const int n = nfc * 6;
    const int n_pad = std::max(CHOL_TILE, ((n + CHOL_TILE - 1) / CHOL_TILE) * CHOL_TILE);

    tm.mark("tethers");
    MAGE_TRY(push_vec(h->d_T_kind, T_kind));
    MAGE_TRY(push_vec(h->d_T_cam, T_cam));
    MAGE_TRY(push_vec(h->d_T_fixed, T_fixed));
    MAGE_TRY(push_vec(h->d_T_meas, T_meas));
    MAGE_TRY(push_vec(h->d_T_w, T_w));
    MAGE_TRY(push_vec(h->d_tc_hc, tc_hc));
    MAGE_TRY(push_vec(h->d_tc_ptr, tc_ptr));
    MAGE_TRY(push_vec(h->d_tc_item, tc_item));
    MAGE_TRY(push_vec(h->d_tp_ij, tp_ij));
    MAGE_TRY(push_vec(h->d_tp_ptr, tp_ptr));
    MAGE_TRY(push_vec(h->d_tp_item, tp_item));
    MAGE_TRY(stage_array(h, h->d_T_out, (size_t)nT * TETHER_OUT_STRIDE + 1));

    tm.mark("uploads queued");
    if (n_pad > CHOL_MAX_ORDER) return fail(MAGE_ERR_UNSUPPORTED, "reduced camera system of order %d exceeds %d (one resident workgroup per tile column)", n_pad, CHOL_MAX_ORDER);
    chol_dag_prefetch(n_pad);          // the dense solve's task lists for this order: a worker thread builds them while the structure is built
    const int nb_l = (nlm + 255) / 256, nb_c = (nfc + 255) / 256;
    MAGE_TRY(stage_array(h, h->d_errL, (size_t)nL * 2 + 2, 0));                // residuals of never-evaluated edges are 0
    MAGE_TRY(stage_array(h, h->d_U, (size_t)nfc * 36 + 1));
    MAGE_TRY(stage_array(h, h->d_bc, (size_t)nfc * 6 + 1));
    MAGE_TRY(stage_array(h, h->d_camR, (size_t)nfc * 12 + 2));
    MAGE_TRY(stage_array(h, h->d_V, (size_t)nlm * 6 + 1));
    MAGE_TRY(stage_array(h, h->d_bp, (size_t)nlm * 4 + 1));
    MAGE_TRY(stage_array(h, h->d_W, (size_t)nw * 18 + 1));
    if (points_free && !h->dup_slots && (nfc * 6 > 128 || nT > 0 || h->shard_ranks > 0) && ba_compact_w_enabled()) {      // the compact form will be used: its position maps (lm_solve fills them)
        MAGE_TRY(stage_array(h, h->d_w_pos, (size_t)nw + 1)); MAGE_TRY(stage_array(h, h->d_pos_lm, (size_t)nw + 1)); MAGE_TRY(stage_array(h, h->d_con_pos, ncon + 1));
        MAGE_TRY(stage_array(h, h->d_slot_order, (size_t)Z.n_blk_slots + 8));
    } else { h->d_w_pos.drop_alias(); h->d_pos_lm.drop_alias(); h->d_con_pos.drop_alias(); h->d_slot_order.drop_alias(); }      // (views of an earlier image must not outlive it)
    MAGE_TRY(stage_array(h, h->d_Dinv, (size_t)nlm * 6 + 1));
    MAGE_TRY(stage_array(h, h->d_db, (size_t)nlm * 4 + 1));
    MAGE_TRY(stage_array(h, h->d_S, (size_t)n_pad * n_pad));
    MAGE_TRY(stage_array(h, h->d_y, (size_t)n_pad));
    if (h->shard_ranks > 0) MAGE_TRY(h->d_xchg.reserve(ba_packed_doubles(n_pad)));
    MAGE_TRY(stage_array(h, h->d_xc, (size_t)n_pad));
    MAGE_TRY(stage_array(h, h->d_xl, (size_t)nlm * 4 + 1));
    // two-level reductions: one double per block of the widest launch; the small-problem linearisation (8 lanes per landmark, 4 blocks
    // per camera) also parks the cameras' partial (U, b_c) sums behind its chi2 partials
    MAGE_TRY(stage_array(h, h->d_partial, std::max<size_t>(4 * 1024, std::max<size_t>((size_t)nb_l + nb_c, (size_t)nlm * 16 / 256 + (size_t)nfc * 4 * 29 + 8)) + 16));          // (the fused large-problem form needs nlm / 32 + nfc entries: covered)
    {   // small problems that free their points: the kept estimate rides the post-pass read-back (k_small_classify)
        const bool off = conservative_paths();
        const size_t want = (size_t)nc * 8 + (size_t)np * 4;
        h->res_doubles = (!off && points_free && h->shard_ranks == 0 && want <= RES_CAP && ba_small_shape_applies(nfc, nT, nL)) ? want : 0;
        h->result_in_mirror = false;
    }
    MAGE_TRY(stage_array(h, h->d_scal, SC_PAD + h->res_doubles + ((size_t)nL + 2) / 2 + 1));          // scalars, (the estimate,) then up to n_L outlier ids (32-bit)
    h->out_cursor = 0;                                                        // d_queue (with the cursor) starts as zeros
    MAGE_TRY(stage_array(h, h->d_Linv, chol_workspace_doubles(n_pad)));
    MAGE_TRY(stage_array(h, h->d_queue, chol_sync_ints(n_pad) + 64, 0));      // + the small-path counter + the outlier cursor; recycled memory arrives dirty
    MAGE_TRY(stage_array(h, h->d_flagL, (size_t)nL + 1));
    MAGE_TRY(stage_array(h, h->d_L_active, (size_t)nL + 1, 1));
    MAGE_TRY(ensure_pinned_mirrors(h));
    MAGE_TRY(stage_commit(h));                                                // image mode: one buffer, one copy; every DevBuf above is a view from here on
    h->d_out_ids = reinterpret_cast<uint32_t*>(h->d_scal.p + SC_PAD + h->res_doubles);
    h->h_out_ids = reinterpret_cast<uint32_t*>(h->h_scal + SC_PAD + h->res_doubles);

    tm.mark("reserve");
    BaDeviceView& v = h->view;
    v.n_cams = nc; v.n_pts = np; v.n_L = nL; v.n_lm = nlm; v.n_fc = nfc; v.n_w = nw; v.n_blk = nblk; v.dup_slots = h->dup_slots ? 1 : 0;
    v.points_free = points_free ? 1 : 0; v.n_pad = n_pad;
    v.camK = h->d_camK.p; v.cam2hc = h->d_cam2hc.p; v.hc2cam = h->d_hc2cam.p;
    v.L_uv = h->d_L_uv.p; v.L_info = h->d_L_info.p; v.L_cam = h->d_L_cam.p; v.L_pt = h->d_L_pt.p; v.L_slot = h->d_L_slot.p; v.L_edge = h->d_L_edge.p; v.L_active = h->d_L_active.p;
    v.lm_ptr = h->d_lm_ptr.p; v.lm_pt = h->d_lm_pt.p; v.lm_wptr = h->d_lm_wptr.p; v.w_hc = h->d_w_hc.p; v.w_lm = h->d_w_lm.p;
    v.camE_ptr = h->d_camE_ptr.p; v.camE = h->d_camE.p; v.camS_ptr = h->d_camS_ptr.p; v.camS = h->d_camS.p;
    v.blk_ptr = h->d_blk_ptr.p; v.blk_ij = h->d_blk_ij.p; v.con = h->d_con.p;
    v.blk_order = h->d_blk_order.p; v.n_blk_slots = Z.n_blk_slots;
    v.n_T = nT; v.n_tc = (int)tc_hc.size(); v.n_tp = (int)tp_ij.size();
    v.T_kind = h->d_T_kind.p; v.T_cam = h->d_T_cam.p; v.T_fixed = h->d_T_fixed.p; v.T_meas = h->d_T_meas.p; v.T_w = h->d_T_w.p; v.T_out = h->d_T_out.p;
    v.tc_hc = h->d_tc_hc.p; v.tc_ptr = h->d_tc_ptr.p; v.tc_item = h->d_tc_item.p;
    v.tp_ij = h->d_tp_ij.p; v.tp_ptr = h->d_tp_ptr.p; v.tp_item = h->d_tp_item.p;
    h->n_active_tethers = nT;
    v.errL = h->d_errL.p; v.U = h->d_U.p; v.bc = h->d_bc.p; v.V = h->d_V.p; v.bp = h->d_bp.p; v.W = h->d_W.p;
    v.compact = 0; v.camR = h->d_camR.p;
    v.w_pos = nullptr; v.pos_lm = nullptr; v.con_pos = nullptr; v.slot_order = nullptr; v.stream_ptr = nullptr; v.stream_blks = nullptr; v.n_stream_groups = 0; v.stream_stamps = nullptr; v.con_soa = nullptr; v.con_soa_pitch = 0;
    h->positions_valid = false; h->n_con = ncon;
    v.Dinv = h->d_Dinv.p; v.db = h->d_db.p; v.S = h->d_S.p; v.y = h->d_y.p; v.xc = h->d_xc.p; v.xl = h->d_xl.p;
    v.partial = h->d_partial.p; v.scal = h->d_scal.p;
    refresh_view_state(h);
    // the skyline of S (large systems that are not sharded: a rank does not know the other ranks' blocks); S itself may be a recycled
    // buffer, so the first trial clears all of it
    v.tile_env = nullptr;
    h->S_outside_skyline_is_zero = false;
    if (n_pad >= 1024 && h->shard_ranks == 0) {
        MAGE_TRY(h->d_tile_env.reserve((size_t)n_pad / CHOL_TILE + 1));
        ba_launch_tile_envelope(v, h->d_tile_env.p, st);
        h->tile_env_valid = true;
        h->tile_env_host.clear();
        if (h->use_skyline) {
            // the solve is to skip the tiles left of the skyline (they are zero and stay zero in the factor): its schedule is built on
            // the host from the skyline, so the skyline comes back once per structure (n_pad / 128 ints)
            h->tile_env_host.resize((size_t)n_pad / CHOL_TILE);
            MAGE_HIP(hipMemcpyAsync(h->tile_env_host.data(), h->d_tile_env.p, h->tile_env_host.size() * sizeof(int), hipMemcpyDeviceToHost, st));
            MAGE_HIP(hipStreamSynchronize(st));
            for (size_t R = 0; R < h->tile_env_host.size(); ++R) h->tile_env_host[R] = std::max(0, std::min(h->tile_env_host[R], (int)R));
            chol_dag_prefetch(n_pad, h->tile_env_host.data());
        }
    } else { h->tile_env_valid = false; h->tile_env_host.clear(); }
    h->L_edge_host.swap(L_edge);          // device build: empty, fetched if the pose-only path ever needs it (mage_ba_step)
    h->built_on_device = on_device;
    h->prof.system_order = n; h->prof.padded_order = n_pad;
    h->prof.factor_flops_each = (double)n * n * n / 3.0;      // algorithmic: the system's order, not the padded one
    {   // algorithmic bytes of the HBM-bound stages for this problem (each array counted once per stage)
        const double dL = nL, dW = nw, dP = nlm, dC = nfc;
        // a W slot is 144 bytes materialised, 32 in the compact form lm_solve selects for large problems without shared slots (ba_kernels.h)
        const bool compact_w = points_free && !h->dup_slots && (nfc * 6 > 128 || nT > 0 || h->shard_ranks > 0) && h->shard_ranks >= 0 && ba_compact_w_enabled();
        const double bW = compact_w ? 32.0 : 144.0;
        h->prof.linearize_bytes_each = dL * (3 * 24 + 16 + 4) + dW * bW + dP * (80 + 3 * 32) + dC * 336;
        // (the zero-fill of S counts only where the whole lower triangle is cleared: with the skyline clear -- unsharded systems of >= 1024
        // rows -- a steady trial clears the few tiles inside the envelope, whose number only the device knows: left out rather than overstated)
        const bool full_clear = !(n_pad >= 1024 && h->shard_ranks == 0);
        h->prof.schur_bytes_each = dP * (160 + 48 + 32) + 2 * bW * dW + 8.0 * (double)ncon + 288.0 * nblk + (full_clear ? 4.0 * (double)n_pad * n_pad : 0.0);
        h->prof.update_bytes_each = bW * dW + dP * (32 + 48 + 32 + 64 + 32) + dL * 40;
    }
    h->iteration = 0;
    h->dirty = false;
    h->soft_dirty = false;
    h->n_active_remaining = nL;
    return MAGE_OK;
}

// The LM control flow needs three scalars on the host per trial; the GPU idles while they travel.  A blocking
// hipStreamSynchronize parks the thread and costs 20-30 us of wake-up latency per read, so the host polls the event instead
// (the step is a few milliseconds: spinning that long is the cheaper side of the trade).
// published: the last launch wrote the mirror itself (k_small_classify<true>, ba_kernels.h): nothing to copy, only its end to wait for
mage_status read_scalars(mage_ba* h, size_t outlier_prefix = 0, bool published = false)
{
    const size_t bytes = outlier_prefix ? (SC_PAD + h->res_doubles) * sizeof(double) + outlier_prefix * sizeof(uint32_t) : SC_COUNT * sizeof(double);
    if (!published) MAGE_HIP(hipMemcpyAsync(h->h_scal, h->d_scal.p, bytes, hipMemcpyDeviceToHost, h->stream));
    MAGE_HIP(hipEventRecord(h->ev[3], h->stream));
    for (;;) {
        const hipError_t e = hipEventQuery(h->ev[3]);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) MAGE_HIP(e);
    }
    if (!h->build_arena.blocks.empty()) h->build_arena.release();      // everything staged for the structure build has been consumed
    return MAGE_OK;
}

enum { LM_OK = 0, LM_TERMINATE = 1, LM_FAIL = 2 };

// What mage_ba_step tells the LM solve about the call it is part of, so that the outlier pass can be queued behind the last
// trial instead of waiting for the host (ba_launch_classify_after_trial): one host round trip per StepBundleAdjustment call.
struct PostPassPlan {
    bool last_iteration = false;     // in: no LM iteration follows in this call
    double max_err_sq = 0;           // in
    size_t prefix = 0;               // in: outlier ids that ride the scalars' read-back
    bool done = false;               // out: the post-pass ran on the device and its sums / ids are in the pinned mirror
};

// OptimizationAlgorithmLevenberg::solve (appendix A.4); scalars only cross PCIe.
mage_status lm_solve(mage_ba* h, double huber, int* result, PostPassPlan* plan = nullptr)
{
    hipStream_t st = h->stream;
    BaDeviceView& v = h->view;
    mage_ba_iter_stats tr{};
    const bool sharded = h->shard_ranks > 0;
    const bool small = !sharded && ba_small_applies(v);
    // landmark-sharded map: rank 0 alone damps the camera blocks and pads the diagonal, the ranks' systems are added
    const bool adds_damping = !sharded || h->shard_rank == 0;
    auto all_reduce = [&](double* buf, size_t count, int op) -> mage_status {
        if (h->shard_reduce(h->shard_user, buf, count, op, (void*)st) != 0) return fail(MAGE_ERR_DEVICE, "landmark-sharded map: the all-reduce callback failed");
        return MAGE_OK;
    };
    int* counter = h->d_queue.p + chol_sync_ints(v.n_pad);       // one int behind the factorisation's counters, zero between launches
    if (h->profiling) MAGE_HIP(hipEventRecord(h->ev_p[0], st));
    // W in its compact form (x/z, y/z, 1/z, weight per slot: ba_kernels.h) whenever the fused linearisation writes it
    v.compact = (!small && v.points_free && ba_fused_linearize_applies(v) && ba_compact_w_enabled()) ? 1 : 0;
    if (v.compact && !h->positions_valid) {
        MAGE_TRY(h->d_w_pos.reserve((size_t)v.n_w + 1)); MAGE_TRY(h->d_pos_lm.reserve((size_t)v.n_w + 1)); MAGE_TRY(h->d_con_pos.reserve(h->n_con + 1));
        const bool lpt = v.n_blk_slots > 0;          // every XCD's run of blocks longest first (slot_order)
        if (lpt) MAGE_TRY(h->d_slot_order.reserve((size_t)v.n_blk_slots + 8));
        // the Schur blocks as streams of trips taken by resident wavefronts (k_schur_stream); MAGE_BA_SCHUR_BLOCKS=1: one wavefront per block (A/B)
        const bool stream = lpt && !h->schur_per_block && v.n_blk > SCHUR_SPLIT_BLOCKS_BELOW && ba_schur_stream_groups(h->n_cu) >= 8;
        const int soa_pitch = (int)((h->n_con + 63) & ~(size_t)63);
        if (stream) MAGE_TRY(h->d_con_soa.reserve((size_t)soa_pitch * 3 + 64));
        ba_launch_build_positions(v, h->d_w_pos.p, h->d_pos_lm.p, h->d_con_pos.p, lpt ? h->d_slot_order.p : nullptr, stream ? h->d_con_soa.p : nullptr, soa_pitch, st);
        v.w_pos = h->d_w_pos.p; v.pos_lm = h->d_pos_lm.p; v.con_pos = h->d_con_pos.p; v.slot_order = lpt ? h->d_slot_order.p : nullptr;
        v.con_soa = stream ? h->d_con_soa.p : nullptr; v.con_soa_pitch = soa_pitch;
        v.stream_ptr = nullptr; v.stream_blks = nullptr; v.n_stream_groups = 0; v.stream_stamps = nullptr;
        if (stream) {
            const int n_groups = ba_schur_stream_groups(h->n_cu);
            MAGE_TRY(h->d_stream_ptr.reserve((size_t)n_groups + 2)); MAGE_TRY(h->d_stream_blks.reserve((size_t)v.n_blk + 8));
            MAGE_TRY(h->d_group_blocks.reserve((size_t)n_groups * ba_schur_stream_rounds(v.n_blk_slots, n_groups) + 8));
            ba_launch_build_stream_lists(v, n_groups, h->d_group_blocks.p, h->d_stream_ptr.p, h->d_stream_blks.p, st);
            v.stream_ptr = h->d_stream_ptr.p; v.stream_blks = h->d_stream_blks.p; v.n_stream_groups = n_groups;
            static const bool stamps = std::getenv("MAGE_BA_SCHUR_TRACE") != nullptr;      // per wavefront: start, end (100 MHz clock), hardware id, trips | blocks << 32 (mage_ba_debug_structure "stream_stamps")
            if (stamps) { MAGE_TRY(h->d_stream_stamps.reserve((size_t)n_groups * 8 * 4)); v.stream_stamps = h->d_stream_stamps.p; }
        }
        h->positions_valid = true;
    }
    int chi_partials = 0;        // > 0: the linearisation left its chi2 partials for the first trial's Schur launch to add (ba_fused_linearize)
    if (small) ba_small_linearize(v, huber, h->iteration == 0, counter, st);
    else if (ba_fused_linearize_applies(v)) chi_partials = ba_fused_linearize(v, huber, counter, st, /*defer_chi_fold=*/!sharded && !(h->iteration == 0 && !(h->user_lambda > 0)));      // (iteration 0 without a user lambda: max |diag| is reduced through v.partial before the Schur launch)
    else {
        ba_launch_error(v, false, huber, st);
        ba_launch_linearize(v, huber, st);
    }
    if (h->profiling) MAGE_HIP(hipEventRecord(h->ev_p[1], st));
    bool lin_timed = false;
    // The chi2 of the current estimate is only needed on the host together with the first trial's (rho); it has its own
    // scalar slot, so after the first iteration of a run no host round trip separates linearisation from the solve.
    // Iteration 0 needs max |diag| on the host to seed lambda.
    if (sharded) MAGE_TRY(all_reduce(v.scal + SC_CHI, 1, 0));
    // lambda of a (re-)initialised optimiser (iteration 0): the user's value when set (BundlerLib.cpp:123-130), else g2o's
    // tau * max |diag|.  The small-problem path seeds it ON THE DEVICE -- its trial kernels take "lambda < 0" as "1e-5 * SC_MAXDIAG" --
    // and the host learns the value from the first trial's read-back: no round trip between linearisation and the first trial (a
    // one-iteration local BA is ~0.45 ms, a round trip 15 us of it).  With a user lambda nothing needs reading at all.
    bool seed_on_device = false;
    if (h->iteration == 0 && h->user_lambda > 0 && !sharded) {
        h->lambda = h->user_lambda;
        h->ni = 2;
    } else if (h->iteration == 0 && small) {
        seed_on_device = true;
        h->ni = 2;
    } else if (h->iteration == 0) {
        if (sharded) {
            // max |diag| of the WHOLE map's Hessian: U's diagonal is a sum over the ranks, V's is a rank's own
            ba_launch_gather_udiag(v, h->d_xchg.p, st);
            MAGE_TRY(all_reduce(h->d_xchg.p, (size_t)v.n_fc * 6, 0));
            ba_launch_maxdiag(v, st, h->d_xchg.p);
            MAGE_TRY(all_reduce(v.scal + SC_MAXDIAG, 1, 1));
        } else if (!small) ba_launch_maxdiag(v, st);
        MAGE_TRY(read_scalars(h));
        h->lambda = h->user_lambda > 0 ? h->user_lambda : 1e-5 * h->h_scal[SC_MAXDIAG];
        h->ni = 2;
    }
    double currentChi = 0;
    bool have_chi = false;
    double rho = 0;
    int qmax = 0;
    bool speculated = false;
    int stall_retries = 0;
    CholWorkspace ws{ h->d_Linv.p, h->d_queue.p, nullptr, v.scal + SC_CHOL_STALL };
    if (h->use_skyline && h->tile_env_valid && !sharded && !h->tile_env_host.empty()) ws.env_host = h->tile_env_host.data();
    bool again = false;
    do {
        again = false;
        bool published = false;
        const double lambda = seed_on_device ? -1.0 : h->lambda;
        if (h->profiling) MAGE_HIP(hipEventRecord(h->ev[0], st));
        if (small) {
            ba_small_solve_trial(v, lambda, huber, h->d_Linv.p, counter, st);
            if (h->profiling) { MAGE_HIP(hipEventRecord(h->ev[1], st)); MAGE_HIP(hipEventRecord(h->ev[2], st)); MAGE_HIP(hipEventRecord(h->ev_p[3], st)); }
            if (plan) {      // the outlier pass rides behind the trial (see the large-problem branch)
                ClassifyAfterTrial c{};
                c.chi_ref = currentChi; c.chi_on_device = have_chi ? 0 : 1; c.trials_done = qmax + 1; c.last_iteration = plan->last_iteration ? 1 : 0;
                // (the launch writes the pinned mirror itself; MAGE_BA_CONSERVATIVE=1 or a mirror the device cannot address: a read-back copy behind it)
                const bool no_publish = conservative_paths();
                double* mirror = nullptr;
                if (!no_publish) {
                    void* dp = nullptr;
                    if (hipHostGetDevicePointer(&dp, h->h_scal, 0) == hipSuccess) mirror = static_cast<double*>(dp);
                    else (void)hipGetLastError();                 // a mirror this device cannot address: the read-back copy as before
                }
                ba_small_classify_after_trial(v, c, plan->max_err_sq, h->d_out_ids, counter + 1, h->out_cursor, counter, h->res_doubles ? h->d_scal.p + SC_PAD : nullptr,
                                              mirror, (int)SC_COUNT, (int)(SC_PAD + h->res_doubles), (int)plan->prefix, st);
                published = mirror != nullptr;
                speculated = true;
            }
        } else {
            v.tile_env = (h->S_outside_skyline_is_zero && h->tile_env_valid && !sharded) ? h->d_tile_env.p : nullptr;
            ba_launch_schur(v, lambda, adds_damping ? lambda : 0.0, adds_damping ? 1.0 : 0.0, chi_partials, st);
            chi_partials = 0;          // (added once: the later trials' launches overwrite the partials)
            h->S_outside_skyline_is_zero = false;          // until this trial's factorisation is known to have produced no NaN (below)
            if (sharded) {
                ba_launch_pack_lower(v, h->d_xchg.p, true, st);
                MAGE_TRY(all_reduce(h->d_xchg.p, ba_packed_doubles(v.n_pad), 0));
                ba_launch_pack_lower(v, h->d_xchg.p, false, st);
            }
            if (h->profiling || h->profiling_factor) MAGE_HIP(hipEventRecord(h->ev[1], st));
            chol_factor_solve(v.S, v.y, v.xc, v.n_pad, ws, v.scal + SC_CHOL_OK, st);
            if (h->profiling || h->profiling_factor) MAGE_HIP(hipEventRecord(h->ev[2], st));
            if (ba_update_and_trial_error_fuses(v)) ba_launch_update_and_trial_error(v, lambda, adds_damping ? lambda : 0.0, huber, st);
            else {
                ba_launch_update(v, lambda, adds_damping ? lambda : 0.0, st);
                ba_launch_error(v, true, huber, st);
            }
            if (h->profiling) MAGE_HIP(hipEventRecord(h->ev_p[3], st));        // end of the update stage (before the queued outlier pass)
            if (plan && !sharded) {
                // the outlier pass rides behind the trial: it runs only if this turns out to be the call's last trial
                ClassifyAfterTrial c{};
                c.chi_ref = currentChi; c.chi_on_device = have_chi ? 0 : 1; c.trials_done = qmax + 1; c.last_iteration = plan->last_iteration ? 1 : 0;
                int* small_counter = h->d_queue.p + chol_sync_ints(v.n_pad);
                ba_launch_classify_after_trial(v, c, plan->max_err_sq, h->d_out_ids, small_counter + 1, h->out_cursor, st);
                speculated = true;
            }
            if (sharded) {
                MAGE_TRY(all_reduce(v.scal + SC_SCALE, 1, 0));
                // chi2 of the trial and, in the same call, the "hand-off timed out" flag of the dense solve (adjacent scalars): a
                // rank whose solve stalled makes EVERY rank return MAGE_ERR_DEVICE below instead of leaving the others in the
                // next all-reduce
                static_assert(SC_CHOL_STALL == SC_CHI_TRIAL + 1, "the trial's chi2 and the stall flag travel together");
                MAGE_TRY(all_reduce(v.scal + SC_CHI_TRIAL, 2, 0));
            }
        }
        MAGE_TRY(read_scalars(h, speculated ? plan->prefix : 0, published));
        if (h->profiling) {
            float ms = 0;
            if (!lin_timed) {
                MAGE_HIP(hipEventElapsedTime(&ms, h->ev_p[0], h->ev_p[1]));
                h->prof.linearize_ms_total += ms; h->prof.linearize_launches++;
                lin_timed = true;
            }
            MAGE_HIP(hipEventElapsedTime(&ms, h->ev[2], h->ev_p[3]));
            h->prof.update_ms_total += ms; h->prof.update_launches++;
            MAGE_HIP(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
            h->prof.schur_ms_total += ms; h->prof.schur_launches++;
        }
        if (h->profiling || (h->profiling_factor && !small)) {
            float ms = 0;
            MAGE_HIP(hipEventElapsedTime(&ms, h->ev[1], h->ev[2]));
            h->prof.factor_ms_total += ms; h->prof.n_factorizations++;
        }
        if (h->h_scal[SC_CHOL_STALL] != 0.0) {
            // A bounded wait between workgroups of the dense solve ran out: nothing of this trial was used (the queued outlier pass
            // skipped itself).  That is a scheduling accident -- seen when SEVERAL PROCESSES oversubscribe one GPU with large
            // problems and the hardware scheduler time-slices them, never with the handles of one process -- not a property of the
            // system, and the linearisation it came from is untouched: the trial is simply run again (Schur build, factorisation,
            // update: the same inputs, hence the same bits as an undisturbed run).  Only a repeated stall is an error.
            static const bool log_stalls = std::getenv("MAGE_BA_STALL_LOG") != nullptr;
            // (a landmark-sharded map all-reduces the flag with SUM beside the trial's chi2: the value is then the sum of the ranks' codes and
            // names no wait any more -- any stall on any rank is treated as the merged panel solve's, the one that has a remedy, on EVERY rank)
            const int stall_code = h->shard_ranks > 0 ? 2 : (int)h->h_scal[SC_CHOL_STALL];
            if (log_stalls) std::fprintf(stderr, "[mage_ba] dense solve: wait %d timed out (1 split diagonal tile, 2 merged panel solve, 3 backward solve%s); trial re-run\n", stall_code,
                                         h->shard_ranks > 0 ? "; sharded map: some rank's wait, reported as 2" : "");
            chol_report_stall(stall_code);       // a stalled merged panel solve switches this process to separate panel-solve launches
            if (++stall_retries <= 3) { h->stall_retries_total++; again = true; continue; }       // lambda unchanged, qmax not advanced
            return fail(MAGE_ERR_DEVICE, "dense solve: a cross-workgroup hand-off timed out four times in a row (device stalled or oversubscribed); the trial was not evaluated");
        }
        if (seed_on_device) { h->lambda = 1e-5 * h->h_scal[SC_MAXDIAG]; seed_on_device = false; }      // the value the device used (the same product)
        const bool ok2 = h->h_scal[SC_CHOL_OK] != 0.0;
        if (!small && ok2) h->S_outside_skyline_is_zero = true;      // S was cleared (at least its skyline, over zeros elsewhere) and factored without a NaN: the zeros outside the skyline survived
        if (!have_chi) { currentChi = h->h_scal[SC_CHI]; tr.chi2_before = currentChi; have_chi = true; }
        double tempChi = h->h_scal[SC_CHI_TRIAL];
        if (!ok2) { tempChi = DBL_MAX; rho = -1.0; }       // the reference's failed-solve branch: always rejected
        else {
            const double scale = h->h_scal[SC_SCALE] + 1e-3;
            rho = (currentChi - tempChi) / scale;
        }
        if (ok2 && rho > 0 && std::isfinite(tempChi)) {
            double alpha = 1. - std::pow((2 * rho - 1), 3);
            alpha = std::min(alpha, 2. / 3.);
            h->lambda *= std::max(1. / 3., alpha);
            h->ni = 2;
            currentChi = tempChi;
            h->cur ^= 1;                 // discardTop: the trial becomes the estimate
            refresh_view_state(h);
        } else {
            h->lambda *= h->ni;
            h->ni *= 2;                  // pop: the estimate buffers were never touched
        }
        qmax++;
    } while (again || (rho < 0 && qmax < 10));
    h->host_state_fresh = false;
    tr.chi2_after = currentChi; tr.lambda = h->lambda; tr.trials = qmax;
    tr.code = (qmax == 10 || rho == 0) ? LM_TERMINATE : LM_OK;
    if (h->stats.size() < 64) h->stats.push_back(tr);
    *result = tr.code;
    // the queued outlier pass ran iff the device took the same decision as the loop above: "over, and (last iteration or Terminate)"
    if (plan) plan->done = speculated && (plan->last_iteration || tr.code == LM_TERMINATE) && h->h_scal[SC_SPEC_DONE] == 1.0;
    return MAGE_OK;
}

// StepOptimizer::Step  (BundlerLib.cpp:132-149)
mage_status step_optimizer(mage_ba* h, double huber, bool* cont, PostPassPlan* plan = nullptr)
{
    if (h->dirty) MAGE_TRY(initialize_optimization(h));
    else if (h->soft_dirty) {
        // Outliers were removed since the last iteration.  The reference re-initialises the optimiser here
        // (BundlerLib.cpp:135-138, 156-166): iteration 0 again, hence lambda re-seeded.  The graph arrays are NOT rebuilt:
        // removed observations are masked on the device (L_active) and contribute nothing; a vertex left without
        // observations keeps a lambda-only diagonal block and a zero right-hand side, i.e. it does not move, exactly
        // as if it had been dropped from the index map.
        h->iteration = 0;
        h->soft_dirty = false;
        if (h->n_active_remaining <= 0 && h->n_active_tethers == 0 && h->shard_ranks == 0) h->useless = true;
    }
    if (h->useless) { *cont = false; return MAGE_OK; }
    int r = LM_OK;
    PhaseTimer tm;
    const bool first = h->iteration == 0;
    MAGE_TRY(lm_solve(h, huber, &r, plan));
    if (first) tm.mark("first LM iteration");
    h->iteration++;
    *cont = (r == LM_OK);
    return MAGE_OK;
}

// =================================================================================================
// The tracker's per-frame problems ("frame path"): BundlerParameters::ArePointsFixed, a handful of cameras, a few hundred
// observations, ONE StepBundleAdjustment per bundler (Tracking/TrackLocalMap.cpp:421-501 builds, steps once, reads the pose and
// destroys -- twice per frame).  The general path costs such a call ~40 device allocations from the cache, ~30 small uploads, the
// launch and two read-backs (0.2 ms where one CPU core needs 0.04).  Here the whole problem is ONE image built in pinned memory in
// the layout k_pose_lm reads (the lists in the order build_lists_host gives them: same sums, same bits), ONE upload, ONE launch
// (every LM iteration with its trials and the outlier pass: ba_kernels.hip "POSE-ONLY problems"), ONE record back that also carries
// both pose buffers -- GetPose then reads the host copy.  Nothing stays on the device: the estimate lives on the host afterwards
// (state_on_device = false), `dirty` stays set for the general path, and whether the GRAPH was edited since is edit_gen vs
// frame_gen (the LM state -- iteration, lambda, ni -- carries over exactly as StepOptimizer's does, BundlerLib.cpp:132-149).
// MAGE_BA_NO_FRAME_PATH=1 (or MAGE_BA_BUILD=host|device, which the structure tests set) sends these problems down the general path.
// =================================================================================================
bool frame_path_enabled()
{
    static const bool on = std::getenv("MAGE_BA_NO_FRAME_PATH") == nullptr && std::getenv("MAGE_BA_BUILD") == nullptr && std::getenv("MAGE_BA_NO_SMALL_PATH") == nullptr;
    return on;
}

bool frame_applies(const mage_ba* h, size_t n_iter)
{
    if (!frame_path_enabled() || !h->points_fixed || h->shard_ranks > 0 || h->state_on_device || !h->dirty) return false;
    if (n_iter < 1 || n_iter > (size_t)POSE_LM_MAX_ITERS || h->profiling || h->profiling_factor) return false;
    if (h->cams.empty() || h->cams.size() > 64 || h->obs.size() > 16384 || h->obs.size() == 0 || h->pt_set.size() > 16384) return false;
    for (int k = 0; k < 3; ++k) if (!h->teth[k].empty()) return false;
    return true;
}

// `*taken` = false: the problem has no free camera with an active observation (the general path knows what to return then)
mage_status frame_step(mage_ba* h, const float* huber, size_t n_iter, float max_err_sq, bool* taken, double* err_sum, double* err_cnt, size_t* n_out)
{
    *taken = false;
    ensure_obs_filled(h);
    const int nc = (int)h->cams.size(), np = (int)h->pt_set.size();
    const size_t no = h->obs.size();
    // ---- the lists, in build_lists_host's order: active observations by landmark (point index ascending), inside a landmark free
    // cameras ascending (ties: observation index) and fixed cameras last; a camera's observations in ascending observation index
    std::vector<uint32_t> active;
    active.reserve(no);
    std::vector<int> cam_deg(nc, 0), pt_deg(np, 0);
    for (size_t e = 0; e < no; ++e) {
        const HostObs& o = h->obs[e];
        if (!o.set || o.removed || h->cams[o.cam].fixed) continue;     // points are fixed: an observation of a fixed camera is not an edge of the problem
        active.push_back((uint32_t)e);
        cam_deg[o.cam]++; pt_deg[o.pt]++;
    }
    const int nL = (int)active.size();
    std::vector<int> hc2cam, cam2hc(nc, -1);
    for (int i = 0; i < nc; ++i)
        if (cam_deg[i] > 0 && !h->cams[i].fixed) { cam2hc[i] = (int)hc2cam.size(); hc2cam.push_back(i); }
    const int nfc = (int)hc2cam.size();
    if (nL == 0 || nfc == 0) return MAGE_OK;
    std::vector<int> lm_ptr(np + 1, 0);
    for (int a = 0; a < nL; ++a) lm_ptr[h->obs[active[a]].pt + 1]++;
    for (int l = 0; l < np; ++l) lm_ptr[l + 1] += lm_ptr[l];
    std::vector<uint32_t> L_edge(nL);
    {
        std::vector<int> fill(lm_ptr.begin(), lm_ptr.end() - 1);
        for (int a = 0; a < nL; ++a) L_edge[fill[h->obs[active[a]].pt]++] = active[a];
    }
    for (int l = 0; l < np; ++l) {
        const int b = lm_ptr[l], k = lm_ptr[l + 1] - b;
        if (k > 1) std::sort(L_edge.begin() + b, L_edge.begin() + b + k, [&](uint32_t x, uint32_t y) {
            const int hx = cam2hc[h->obs[x].cam], hy = cam2hc[h->obs[y].cam];
            return hx != hy ? hx < hy : x < y;
        });
    }
    std::vector<int> where(no, -1), camE_ptr(nfc + 1, 0);
    for (int i = 0; i < nL; ++i) where[L_edge[i]] = i;
    for (int a = 0; a < nL; ++a) camE_ptr[cam2hc[h->obs[active[a]].cam] + 1]++;
    for (int c = 0; c < nfc; ++c) camE_ptr[c + 1] += camE_ptr[c];

    // ---- the image: [up: pose0 | pose1 | camK | pt | L_uv | L_info | L_cam | L_pt | camE | camE_ptr | hc2cam | L_active][back: record | flags | pose0' | pose1'][scratch]
    // `up` is the head of the staged kernel's LDS image, in its order and packed (k_pose_lm copies it as ONE flat run); nothing in it is
    // written by the solve, everything the solve returns is in `back`.
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    auto al16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_p0 = 0, o_p1 = o_p0 + (size_t)nc * 64, o_K = o_p1 + (size_t)nc * 64, o_pt = o_K + (size_t)nc * 32;
    const size_t o_uv = al16(o_pt + (size_t)np * 32), o_info = o_uv + (size_t)nL * 8, o_cam = o_info + (size_t)nL * 4, o_lpt = o_cam + (size_t)nL * 4,
                 o_ce = o_lpt + (size_t)nL * 4, o_cep = o_ce + (size_t)nL * 4, o_hc = o_cep + (size_t)(nfc + 1) * 4, o_act = o_hc + (size_t)nfc * 4;
    const size_t up_bytes = al(o_act + (size_t)nL + 16);                       // (+16: the second run is copied in whole 16-byte pieces)
    const size_t o_res = up_bytes, o_flag = o_res + al16(sizeof(PoseLmResult)), o_q0 = al16(o_flag + (size_t)nL), o_q1 = o_q0 + (size_t)nc * 64;
    size_t off = al(o_q1 + (size_t)nc * 64);
    const size_t back_bytes = off - up_bytes;
    auto take = [&](size_t bytes) { const size_t o = off; off = al(off + bytes); return o; };
    const size_t o_err = take((size_t)nL * 16), o_U = take((size_t)nfc * 36 * 8), o_bc = take((size_t)nfc * 48), o_xc = take((size_t)nfc * 48);
    const size_t dev_bytes = off;
    if (h->h_frame_bytes < up_bytes + back_bytes) {
        if (h->h_frame) { cached_pinned_release(h->h_frame, h->h_frame_bytes); h->h_frame = nullptr; h->h_frame_bytes = 0; }
        MAGE_TRY(cached_pinned_alloc(&h->h_frame, up_bytes + back_bytes, &h->h_frame_bytes));
    }
    MAGE_TRY(h->d_frame.reserve(dev_bytes));
    char* img = static_cast<char*>(h->h_frame);
    char* back = img + up_bytes;
    {
        double* K = reinterpret_cast<double*>(img + o_K);
        double *p0 = reinterpret_cast<double*>(img + o_p0), *p1 = reinterpret_cast<double*>(img + o_p1);
        for (int i = 0; i < nc; ++i) {
            const HostCam& c = h->cams[i];
            K[i * 4] = c.f; K[i * 4 + 1] = c.cx; K[i * 4 + 2] = c.cy; K[i * 4 + 3] = 0.0;
            for (int a = 0; a < 4; ++a) p0[i * 8 + a] = c.q[a];
            for (int a = 0; a < 3; ++a) p0[i * 8 + 4 + a] = c.t[a];
            p0[i * 8 + 7] = 0.0;
        }
        std::memcpy(p1, p0, (size_t)nc * 64);
        double* P = reinterpret_cast<double*>(img + o_pt);
        for (int i = 0; i < np; ++i) { P[i * 4] = h->pts[(size_t)i * 3]; P[i * 4 + 1] = h->pts[(size_t)i * 3 + 1]; P[i * 4 + 2] = h->pts[(size_t)i * 3 + 2]; P[i * 4 + 3] = 0.0; }
        std::memcpy(img + o_hc, hc2cam.data(), (size_t)nfc * 4);
        std::memcpy(img + o_cep, camE_ptr.data(), (size_t)(nfc + 1) * 4);
        int* camE = reinterpret_cast<int*>(img + o_ce);
        {
            std::vector<int> fill(camE_ptr.begin(), camE_ptr.end() - 1);
            for (int a = 0; a < nL; ++a) camE[fill[cam2hc[h->obs[active[a]].cam]]++] = where[active[a]];      // ascending observation index per camera
        }
        float2* uv = reinterpret_cast<float2*>(img + o_uv);
        float* info = reinterpret_cast<float*>(img + o_info);
        uint32_t *Lc = reinterpret_cast<uint32_t*>(img + o_cam), *Lp = reinterpret_cast<uint32_t*>(img + o_lpt);
        for (int i = 0; i < nL; ++i) {
            const HostObs& o = h->obs[L_edge[i]];
            uv[i] = make_float2(o.u, o.v); info[i] = o.info; Lc[i] = o.cam; Lp[i] = o.pt;
        }
        std::memset(img + o_act, 1, (size_t)nL);
    }
    // ---- StepOptimizer's entry (BundlerLib.cpp:132-149): an edited graph, or observations removed by the last call, start the optimiser over
    const bool graph_changed = !(h->frame_valid && h->frame_gen == h->edit_gen);
    if (graph_changed || h->soft_dirty) h->iteration = 0;
    h->soft_dirty = false;
    unsigned char* D = h->d_frame.p;
    BaDeviceView v{};
    v.n_cams = nc; v.n_pts = np; v.n_L = nL; v.n_fc = nfc; v.points_free = 0; v.n_pad = CHOL_TILE;
    // DIRECT (round 4): the staged kernel reads every input exactly once (into LDS) and writes its record, the flags and the two pose
    // buffers exactly once -- so it takes them from / leaves them in the pinned image itself, across PCIe, and the two copy commands
    // (a blit launch and a dependency each, ~10 us of a 65 us call) are not queued at all.  MAGE_BA_CONSERVATIVE=1: upload + read-back, arrays in HBM.
    const bool staged_off = conservative_paths();
    bool direct = !staged_off && ba_pose_lm_staged_fits(v);
    if (direct) {
        void* dev_img = nullptr;
        if (hipHostGetDevicePointer(&dev_img, img, 0) == hipSuccess && dev_img) D = static_cast<unsigned char*>(dev_img);
        else { (void)hipGetLastError(); direct = false; }      // pinned memory this device cannot address: upload + read-back as before
    }
    v.camK = reinterpret_cast<const double*>(D + o_K); v.pt_cur = v.pt_trial = reinterpret_cast<double*>(D + o_pt);
    v.hc2cam = reinterpret_cast<const int*>(D + o_hc); v.camE_ptr = reinterpret_cast<const int*>(D + o_cep); v.camE = reinterpret_cast<const int*>(D + o_ce);
    v.L_uv = reinterpret_cast<const float2*>(D + o_uv); v.L_info = reinterpret_cast<const float*>(D + o_info);
    v.L_cam = reinterpret_cast<const uint32_t*>(D + o_cam); v.L_pt = reinterpret_cast<const uint32_t*>(D + o_lpt); v.L_active = D + o_act;
    v.pose_cur = reinterpret_cast<double*>(D + o_p0); v.pose_trial = reinterpret_cast<double*>(D + o_p1);
    v.errL = reinterpret_cast<double*>(D + o_err); v.U = reinterpret_cast<double*>(D + o_U); v.bc = reinterpret_cast<double*>(D + o_bc); v.xc = reinterpret_cast<double*>(D + o_xc);
    PoseLmArgs a{};
    a.n_huber = (int)n_iter;
    for (size_t it = 0; it < n_iter; ++it) a.huber[it] = huber[it];
    a.max_err_sq = (double)max_err_sq; a.lambda = h->lambda; a.user_lambda = h->user_lambda; a.ni = h->ni; a.iteration = h->iteration;
    // (D is the pinned image itself when `direct`: record, flags and the two pose buffers then land in its `back` part with no copy queued)
    PoseLmResult* d_res = reinterpret_cast<PoseLmResult*>(D + o_res);
    double* d_q = reinterpret_cast<double*>(D + o_q0);
    if (direct) {
        if (!ba_launch_pose_lm_staged(v, a, d_res, D + o_flag, d_q, h->stream)) return fail(MAGE_ERR_DEVICE, "pose-only solve: the staged launch was refused");
    } else {
        MAGE_HIP(hipMemcpyAsync(D, img, up_bytes, hipMemcpyHostToDevice, h->stream));
        if (staged_off || !ba_launch_pose_lm_staged(v, a, d_res, D + o_flag, d_q, h->stream)) ba_launch_pose_lm(v, a, d_res, D + o_flag, d_q, h->stream);
        MAGE_HIP(hipMemcpyAsync(back, D + o_res, back_bytes, hipMemcpyDeviceToHost, h->stream));
    }
    MAGE_HIP(hipGetLastError());          // a refused launch must not be followed by reading stale poses out of `back` behind a completed event
    MAGE_HIP(hipEventRecord(h->ev[3], h->stream));
    for (;;) {
        const hipError_t e = hipEventQuery(h->ev[3]);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) MAGE_HIP(e);
    }
    const PoseLmResult& r = *reinterpret_cast<const PoseLmResult*>(back);
    const double* kept = reinterpret_cast<const double*>(back + (((r.flips & 1) ? o_q1 : o_q0) - o_res));
    for (int hc = 0; hc < nfc; ++hc) {                    // only the cameras of the system move
        HostCam& c = h->cams[hc2cam[hc]];
        for (int q = 0; q < 4; ++q) c.q[q] = kept[hc2cam[hc] * 8 + q];
        for (int q = 0; q < 3; ++q) c.t[q] = kept[hc2cam[hc] * 8 + 4 + q];
    }
    h->lambda = r.lambda; h->ni = r.ni; h->iteration = r.iteration;
    for (int i = 0; i < r.n_stats && i < POSE_LM_MAX_ITERS; ++i) {
        mage_ba_iter_stats tr{};
        tr.code = r.stats[i].code; tr.trials = r.stats[i].trials; tr.chi2_before = r.stats[i].chi2_before;
        tr.chi2_after = r.stats[i].chi2_after; tr.lambda = r.stats[i].lambda;
        if (h->stats.size() < 64) h->stats.push_back(tr);
    }
    *err_sum = r.err_sum; *err_cnt = r.err_cnt; *n_out = (size_t)r.n_out;
    if (r.n_out > 0) {
        const unsigned char* flag = reinterpret_cast<const unsigned char*>(back + (o_flag - o_res));
        std::vector<uint32_t>& ids = h->last_outliers;
        ids.reserve((size_t)r.n_out);
        for (int i = 0; i < nL; ++i) if (flag[i]) ids.push_back(L_edge[i]);
        std::sort(ids.begin(), ids.end());
        for (uint32_t id : ids) h->obs[id].removed = 1;      // removeEdge
        h->soft_dirty = true;
    }
    h->n_active_remaining = (long long)nL - (long long)r.n_out;
    h->useless = false;
    h->host_state_fresh = true; h->state_on_device = false;
    h->dirty = true;                                           // no structure on the device (edit_gen is NOT bumped: the graph is the caller's)
    h->frame_valid = true; h->frame_gen = h->edit_gen;
    *taken = true;
    return MAGE_OK;
}

template <typename F>
mage_status guarded(F&& f)
{
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(MAGE_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    catch (const std::exception& e) { return fail(MAGE_ERR_DEVICE, "unexpected exception: %s", e.what()); }
    catch (...) { return fail(MAGE_ERR_DEVICE, "unexpected exception"); }
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
MAGE_EXPORT const char* mage_last_error(void) { return last_error_ref().c_str(); }

MAGE_EXPORT void mage_release_cached_memory(void)
{
    DeviceCache& c = device_cache();
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    std::lock_guard<std::mutex> lock(c.m);
    for (int d = 0; d < DeviceCache::MAX_DEVICES; ++d) {
        if (c.parked[d].empty() && c.streams[d].empty()) continue;
        (void)hipSetDevice(d);
        for (auto& kv : c.parked[d]) (void)hipFree(kv.second);
        c.parked[d].clear();
        c.held[d] = 0;
        for (hipStream_t st : c.streams[d]) { chol_forget_stream(st); (void)hipStreamDestroy(st); }
        c.streams[d].clear();
    }
    for (auto& kv : c.pinned) (void)hipHostFree(kv.second);
    c.pinned.clear();
    c.pinned_held = 0;
    if (have_cur) (void)hipSetDevice(cur);
}

// Blocks until the dense solve's task lists for a reduced system of padded order n_pad are on the current device (1) or reports that
// this order is served by the column launches (0); *build_ms = what the last build took on its worker thread.  bench.py calls it in
// front of the timed region and reports the figure in extra.cold_start; a deployment never needs it.
MAGE_EXPORT int mage_debug_chol_wait_schedule(int device, int n_pad, double* build_ms)
{
    int dev = 0;
    if (select_device(device, &dev) != MAGE_OK) return 0;
    MAGE_DEVICE_SCOPE(dev);
    chol_init_device();
    return chol_dag_wait_schedule(n_pad, build_ms) ? 1 : 0;
}

static mage_status debug_dense_solve(int device, int n, const double* A_colmajor, const double* b, double* x, int* ok, const int* tile_env);
// The same with the skyline of A by tile rows (tile_env[R] = first tile column of tile row R that holds a non-zero; n_pad / 128 entries):
// the task-graph schedule skips every tile left of it.  The caller vouches for the zeros (tests/test_chol_gpu.py compares with the dense solve).
MAGE_EXPORT mage_status mage_debug_dense_solve_skyline(int device, int n, const double* A_colmajor, const double* b, double* x, int* ok, const int* tile_env)
{
    return debug_dense_solve(device, n, A_colmajor, b, x, ok, tile_env);
}
MAGE_EXPORT mage_status mage_debug_dense_solve(int device, int n, const double* A_colmajor, const double* b, double* x, int* ok)
{
    return debug_dense_solve(device, n, A_colmajor, b, x, ok, nullptr);
}
static mage_status debug_dense_solve(int device, int n, const double* A_colmajor, const double* b, double* x, int* ok, const int* tile_env)
{
    return guarded([&]() -> mage_status {
        if (n <= 0 || !A_colmajor || !b || !x || !ok) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument or n <= 0");
        int dev = 0;
        MAGE_TRY(select_device(device, &dev));
        MAGE_DEVICE_SCOPE(dev);
        chol_init_device();
        const int n_pad = std::max(CHOL_TILE, ((n + CHOL_TILE - 1) / CHOL_TILE) * CHOL_TILE);
        if (n_pad > CHOL_MAX_ORDER) return fail(MAGE_ERR_UNSUPPORTED, "order %d exceeds %d", n_pad, CHOL_MAX_ORDER);
        (void)chol_dag_wait_schedule(n_pad, nullptr, tile_env);          // (a test of the solver: the schedule this size has by default, not the one it starts with)
        // the lower triangle, padded with an identity block exactly as the bundle adjustment pads its reduced camera system
        std::vector<double> S((size_t)n_pad * n_pad, 0.0), y(n_pad, 0.0);
        for (int c = 0; c < n; ++c)
            for (int r = c; r < n; ++r) S[(size_t)c * n_pad + r] = A_colmajor[(size_t)c * n + r];
        for (int i = n; i < n_pad; ++i) S[(size_t)i * n_pad + i] = 1.0;
        for (int i = 0; i < n; ++i) y[i] = b[i];
        DevBuf<double> dS, dy, dx, dLinv, dok; DevBuf<int> dsync;
        hipStream_t st = nullptr;
        MAGE_TRY(cached_stream_acquire(dev, &st));
        mage_status rc = [&]() -> mage_status {
            MAGE_TRY(dS.upload(S.data(), S.size(), st)); MAGE_TRY(dy.upload(y.data(), y.size(), st));
            MAGE_TRY(dx.reserve(n_pad)); MAGE_TRY(dLinv.reserve(chol_workspace_doubles(n_pad))); MAGE_TRY(dok.reserve(2));
            MAGE_TRY(dsync.reserve(chol_sync_ints(n_pad)));
            CholWorkspace ws{ dLinv.p, dsync.p };
            ws.env_host = tile_env;
            chol_factor_solve(dS.p, dy.p, dx.p, n_pad, ws, dok.p, st);
            std::vector<double> xs(n_pad);
            double okv[2] = { 0, 0 };
            MAGE_HIP(hipMemcpyAsync(xs.data(), dx.p, n_pad * sizeof(double), hipMemcpyDeviceToHost, st));
            MAGE_HIP(hipMemcpyAsync(okv, dok.p, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
            MAGE_HIP(hipStreamSynchronize(st));
            if (okv[1] != 0.0) return fail(MAGE_ERR_DEVICE, "dense solve: a cross-workgroup hand-off timed out");
            for (int i = 0; i < n; ++i) x[i] = xs[i];
            *ok = okv[0] != 0.0 ? 1 : 0;
            return MAGE_OK;
        }();
        (void)hipStreamSynchronize(st);
        cached_stream_release(dev, st);
        return rc;
    });
}

MAGE_EXPORT mage_status mage_ba_create(const mage_ba_params* params, mage_ba** out)
{
    return guarded([&]() -> mage_status {
        if (!out) return fail(MAGE_ERR_INVALID_ARGUMENT, "out is null");
        *out = nullptr;
        int dev = 0;
        MAGE_TRY(select_device(params ? params->device : -1, &dev));
        std::unique_ptr<mage_ba> h(new mage_ba());
        h->device = dev;
        { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256; h->n_cu = cu; }
        h->points_fixed = params ? params->are_points_fixed != 0 : false;
        { static const bool pb = std::getenv("MAGE_BA_SCHUR_BLOCKS") != nullptr; h->schur_per_block = pb; }
        { static const bool sky = std::getenv("MAGE_BA_SKYLINE") != nullptr; h->use_skyline = sky; }          // process-wide default of mage_ba_use_skyline
        MAGE_DEVICE_SCOPE(dev);
        MAGE_TRY(cached_stream_acquire(dev, &h->stream));
        MAGE_HIP(hipEventCreateWithFlags(&h->ev[3], hipEventDisableTiming));       // the scalar read-back's event; the others: ensure_events
        static std::atomic<bool> first_copy_made{ false };
        if (!first_copy_made.exchange(true) && std::getenv("MAGE_BA_NO_WARMUP") == nullptr) {
            hipStream_t st = h->stream;
            h->warmup = std::thread([dev, st]() {
                if (hipSetDevice(dev) != hipSuccess) return;
                void* q = nullptr; size_t got = 0;
                if (cached_pinned_alloc(&q, (size_t)1 << 20, &got) != MAGE_OK) return;
                void* d = nullptr; size_t dg = 0; int dv = dev;
                if (cached_device_alloc(&d, (size_t)1 << 20, &dv, &dg) == MAGE_OK) {
                    std::memset(q, 0, (size_t)256 << 10);
                    if (hipMemcpyAsync(d, q, (size_t)256 << 10, hipMemcpyHostToDevice, st) == hipSuccess) (void)hipStreamSynchronize(st);
                    cached_device_release(d, dg, dv);
                }
                (void)hipGetLastError();
                cached_pinned_release(q, got);
            });
        }
        chol_init_device();
        ba_small_init_device();
        build_init_device();
        *out = h.release();
        return MAGE_OK;
    });
}

MAGE_EXPORT void mage_ba_destroy(mage_ba* h) { delete h; }

MAGE_EXPORT mage_status mage_ba_alloc_cameras(mage_ba* h, size_t count)
{
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
        if (h->cams_allocated) return fail(MAGE_ERR_INVALID_ARGUMENT, "cameras can only be allocated once");   // BundlerLib.cpp:200
        h->cams.assign(count, HostCam());
        h->cams_allocated = true;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_set_camera(mage_ba* h, size_t idx, const float position[3], const float R_colmajor[9],
                                           const float K[4], int is_fixed)
{
    return guarded([&]() -> mage_status {
        if (!h || !position || !R_colmajor || !K) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (idx >= h->cams.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "camera index %zu out of range (%zu)", idx, h->cams.size());
        MAGE_TRY(before_host_edit(h));
        HostCam& c = h->cams[idx];
        pose_from_f32(R_colmajor, position, c);
        c.f = K[2]; c.cx = K[0]; c.cy = K[1];        // fy (K[3]) unused, BundlerLib.cpp:266
        c.fixed = is_fixed ? 1 : 0; c.set = 1;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_set_cameras_bulk(mage_ba* h, size_t count, const float* positions3, const float* R9,
                                                 const float* K4, const uint8_t* is_fixed)
{
    return guarded([&]() -> mage_status {
        if (!h || !positions3 || !R9 || !K4) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (count > h->cams.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "count %zu exceeds allocated cameras %zu", count, h->cams.size());
        MAGE_TRY(before_host_edit(h));
        for (size_t i = 0; i < count; ++i) {
            HostCam& c = h->cams[i];
            pose_from_f32(R9 + i * 9, positions3 + i * 3, c);
            c.f = K4[i * 4 + 2]; c.cx = K4[i * 4]; c.cy = K4[i * 4 + 1];
            c.fixed = (is_fixed && is_fixed[i]) ? 1 : 0; c.set = 1;
        }
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_fix_camera(mage_ba* h, size_t idx, int is_fixed)
{
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
        if (idx >= h->cams.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "camera index %zu out of range", idx);
        const uint8_t nv = is_fixed ? 1 : 0;
        if (h->cams[idx].fixed != nv) { h->cams[idx].fixed = nv; mark_edited(h); }
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_update_camera_poses(mage_ba* h, size_t count, const uint32_t* indices, const float* positions3,
                                                    const float* R_colmajor9)
{
    return guarded([&]() -> mage_status {
        if (!h || (count && (!indices || !positions3 || !R_colmajor9))) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        for (size_t k = 0; k < count; ++k) {
            if (indices[k] >= h->cams.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "camera index %u out of range (%zu)", indices[k], h->cams.size());
            if (!h->cams[indices[k]].set) return fail(MAGE_ERR_INVALID_ARGUMENT, "camera %u was never set: use mage_ba_set_camera first", indices[k]);
        }
        if (count == 0) return MAGE_OK;
        // The estimate may live on the device (stepping has started): the host copy of every other entity stays as stale or
        // as fresh as it was; the edited cameras are written to both places, and to BOTH device state buffers (a fixed camera
        // is never written by a trial, so the buffer that becomes current after an accepted trial must hold it too).
        MAGE_DEVICE_SCOPE(h->device);
        std::vector<double> recs;
        for (size_t k = 0; k < count; ++k) {
            HostCam& c = h->cams[indices[k]];
            pose_from_f32(R_colmajor9 + 9 * k, positions3 + 3 * k, c);
            if (h->state_on_device) {
                const double rec[8] = { c.q[0], c.q[1], c.q[2], c.q[3], c.t[0], c.t[1], c.t[2], 0.0 };
                recs.insert(recs.end(), rec, rec + 8);
            }
        }
        if (h->state_on_device) {
            // one staging copy, then a scatter on the device into both state buffers (instead of a synchronised copy per camera)
            DevBuf<double> stage; DevBuf<uint32_t> idx;
            MAGE_TRY(stage.upload(recs.data(), recs.size(), h->stream));
            MAGE_TRY(idx.upload(indices, count, h->stream));
            ba_launch_import_poses(h->d_pose[0].p, h->d_pose[1].p, idx.p, nullptr, count, stage.p, h->stream);
            MAGE_HIP(hipStreamSynchronize(h->stream));      // the staging buffers go back to the cache
        }
        h->iteration = 0;          // a different linear system: the optimiser starts over (lambda re-seeded), the graph is unchanged
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_alloc_points(mage_ba* h, size_t count)
{
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
        if (h->pts_allocated) return fail(MAGE_ERR_INVALID_ARGUMENT, "map points can only be allocated once");
        h->pts.assign(count * 3, 0.0);
        h->pt_set.assign(count, 0);
        h->pts_allocated = true;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_set_point(mage_ba* h, size_t idx, const float xyz[3])
{
    return guarded([&]() -> mage_status {
        if (!h || !xyz) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (idx >= h->pt_set.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "point index %zu out of range (%zu)", idx, h->pt_set.size());
        MAGE_TRY(before_host_edit(h));
        for (int a = 0; a < 3; ++a) h->pts[idx * 3 + a] = xyz[a];
        h->pt_set[idx] = 1;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_set_points_bulk(mage_ba* h, size_t count, const float* xyz3)
{
    return guarded([&]() -> mage_status {
        if (!h || !xyz3) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (count > h->pt_set.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "count %zu exceeds allocated points %zu", count, h->pt_set.size());
        MAGE_TRY(before_host_edit(h));
        for (size_t i = 0; i < count * 3; ++i) h->pts[i] = xyz3[i];
        std::fill(h->pt_set.begin(), h->pt_set.begin() + count, 1);
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_alloc_observations(mage_ba* h, size_t count)
{
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
        if (h->obs_allocated) return fail(MAGE_ERR_INVALID_ARGUMENT, "observations can only be allocated once");
        if (count > 0x7fffffffull) return fail(MAGE_ERR_INVALID_ARGUMENT, "too many observations");
        MAGE_TRY(h->obs.resize_uninitialized(count));      // cleared lazily: see obs_unfilled
        h->obs_unfilled = true;
        h->obs_allocated = true;
        return MAGE_OK;
    });
}

static mage_status set_obs(mage_ba* h, size_t idx, float u, float v, uint64_t cam, uint64_t pt, float info)
{
    if (cam >= h->cams.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "observation %zu: camera index %llu out of range", idx, (unsigned long long)cam);
    if (pt >= h->pt_set.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "observation %zu: point index %llu out of range", idx, (unsigned long long)pt);
    HostObs& o = h->obs[idx];
    o.u = u; o.v = v; o.info = info; o.cam = (uint32_t)cam; o.pt = (uint32_t)pt; o.set = 1; o.removed = 0;
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_ba_set_observation(mage_ba* h, size_t idx, const float uv[2], uint64_t cam, uint64_t pt, float info)
{
    return guarded([&]() -> mage_status {
        if (!h || !uv) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (idx >= h->obs.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "observation index %zu out of range (%zu)", idx, h->obs.size());
        mark_edited(h);
        ensure_obs_filled(h);
        return set_obs(h, idx, uv[0], uv[1], cam, pt, info);
    });
}

MAGE_EXPORT mage_status mage_ba_set_observations_bulk(mage_ba* h, size_t count, const float* uv2, const uint32_t* cam,
                                                      const uint32_t* pt, const float* info)
{
    return guarded([&]() -> mage_status {
        if (!h || !uv2 || !cam || !pt || !info) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (count > h->obs.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "count %zu exceeds allocated observations %zu", count, h->obs.size());
        mark_edited(h);
        // AllocateObservations left the records uninitialised (obs_unfilled): when this call covers all of them there is nothing to
        // clear first -- unless an index is out of range, in which case the records from the offender on are cleared before the error
        // is returned ("never set")
        const bool covers_all = h->obs_unfilled && count == h->obs.size();
        if (!covers_all) ensure_obs_filled(h);
        auto clear_from = [&](size_t first) { std::fill(h->obs.begin() + first, h->obs.end(), HostObs()); h->obs_unfilled = false; };
        // a million records are 24 MB written and 20 MB read: a few host threads, each on its own range (large maps only)
        const int parts = parts_for((int)std::min<size_t>(count, 0x7fffffff), 131072);
        if (parts <= 1) {
            for (size_t i = 0; i < count; ++i) {
                const mage_status st = set_obs(h, i, uv2[i * 2], uv2[i * 2 + 1], cam[i], pt[i], info[i]);
                if (st != MAGE_OK) { if (covers_all) clear_from(i); return st; }
            }
            h->obs_unfilled = false;
            return MAGE_OK;
        }
        const size_t nc = h->cams.size(), np = h->pt_set.size();
        // the indices are validated FIRST (a parallel read-only pass): a bad index must leave the same state behind whatever the
        // problem size -- the sequential loop's: the records before the first offender written, nothing after it
        std::vector<long long> bad(parts, -1);
        parallel_ranges((int)count, parts, [&](int i0, int i1, int part) {
            for (int i = i0; i < i1; ++i) if (cam[i] >= nc || pt[i] >= np) { bad[part] = i; break; }
        });
        for (long long b : bad)
            if (b >= 0) {                      // the first offender of the lowest range = the first offender
                for (size_t i = 0; i < (size_t)b; ++i) (void)set_obs(h, i, uv2[i * 2], uv2[i * 2 + 1], cam[i], pt[i], info[i]);
                if (covers_all) clear_from((size_t)b);
                return set_obs(h, (size_t)b, uv2[b * 2], uv2[b * 2 + 1], cam[b], pt[b], info[b]);
            }
        HostObs* obs = h->obs.data();
        parallel_ranges((int)count, parts, [&](int i0, int i1, int) {
            for (int i = i0; i < i1; ++i) {
                HostObs& o = obs[i];
                o.u = uv2[(size_t)i * 2]; o.v = uv2[(size_t)i * 2 + 1]; o.info = info[i]; o.cam = cam[i]; o.pt = pt[i]; o.set = 1; o.removed = 0;
            }
        });
        h->obs_unfilled = false;
        return MAGE_OK;
    });
}

static mage_status tether_alloc(mage_ba* h, size_t count, int kind)
{
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
        if (h->teth_allocated[kind]) return fail(MAGE_ERR_INVALID_ARGUMENT, "constraints of this kind can only be allocated once");   // BundlerLib.cpp:233-258
        h->teth[kind].assign(count, HostTether());
        h->teth_allocated[kind] = true;
        return MAGE_OK;
    });
}
MAGE_EXPORT mage_status mage_ba_alloc_fixed_distance_constraints(mage_ba* h, size_t n) { return tether_alloc(h, n, TETHER_DISTANCE); }
MAGE_EXPORT mage_status mage_ba_alloc_relative_rotation_constraints(mage_ba* h, size_t n) { return tether_alloc(h, n, TETHER_ROTATION); }
MAGE_EXPORT mage_status mage_ba_alloc_relative_transform_constraints(mage_ba* h, size_t n) { return tether_alloc(h, n, TETHER_TRANSFORM); }

static mage_status tether_slot(mage_ba* h, int kind, size_t idx, size_t c1, size_t c2, HostTether** out)
{
    if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
    if (idx >= h->teth[kind].size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "constraint index %zu out of range (%zu allocated)", idx, h->teth[kind].size());
    if (c1 >= h->cams.size() || c2 >= h->cams.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "constraint %zu: camera index out of range", idx);
    if (c1 == c2) return fail(MAGE_ERR_INVALID_ARGUMENT, "constraint %zu joins camera %zu to itself", idx, c1);
    MAGE_TRY(before_host_edit(h));
    HostTether& t = h->teth[kind][idx];
    t = HostTether();
    t.c0 = (uint32_t)c1; t.c1 = (uint32_t)c2; t.set = 1;
    *out = &t;
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_ba_set_fixed_distance_constraint(mage_ba* h, size_t idx, size_t c1, size_t c2, float distance, float weight)
{
    return guarded([&]() -> mage_status {
        HostTether* t = nullptr;
        MAGE_TRY(tether_slot(h, TETHER_DISTANCE, idx, c1, c2, &t));
        t->dist = (double)distance; t->w = (double)weight;             // BundlerLib.cpp:315-319
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_set_relative_rotation_constraint(mage_ba* h, size_t idx, size_t c1, size_t c2, const float* q, float weight)
{
    return guarded([&]() -> mage_status {
        if (!q) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        HostTether* t = nullptr;
        MAGE_TRY(tether_slot(h, TETHER_ROTATION, idx, c1, c2, &t));
        for (int a = 0; a < 4; ++a) t->q[a] = (double)q[a];            // deltaRotation.cast<number_t>(), not normalised (BundlerLib.cpp:333)
        t->w = (double)weight;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_set_relative_transform_constraint(mage_ba* h, size_t idx, size_t c1, size_t c2, const float* p,
                                                                  const float* q, float weight)
{
    return guarded([&]() -> mage_status {
        if (!q || !p) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        HostTether* t = nullptr;
        MAGE_TRY(tether_slot(h, TETHER_TRANSFORM, idx, c1, c2, &t));
        // setMeasurement({q, p}) builds an SE3Quat, whose constructor normalises the rotation (BundlerLib.cpp:348)
        double d[4] = { (double)q[0], (double)q[1], (double)q[2], (double)q[3] };
        if (d[3] < 0) { d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; d[3] = -d[3]; }
        const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3]);
        for (int a = 0; a < 4; ++a) t->q[a] = d[a] / n;
        for (int a = 0; a < 3; ++a) t->t[a] = (double)p[a];
        t->w = (double)weight;                                         // information = Identity * weight
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_set_lambda(mage_ba* h, float user_lambda)
{
    if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
    h->iteration = 0;                       // BundlerLib.cpp:123-130
    h->user_lambda = (double)user_lambda;
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_ba_get_lambda(const mage_ba* h, float* out)
{
    if (!h || !out) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
    *out = (float)h->lambda;
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_ba_step(mage_ba* h, const float* huber, size_t n_iter, float max_err_sq, uint32_t* outliers,
                                     size_t capacity, size_t* n_outliers, float* mean_sq_err)
{
    return guarded([&]() -> mage_status {
        if (!h || (n_iter && !huber)) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (n_outliers) *n_outliers = 0;
        if (mean_sq_err) *mean_sq_err = NAN;
        MAGE_DEVICE_SCOPE(h->device);
        h->result_in_mirror = false;
        // The structure build's pinned staging goes back to the cache when the first trial's scalars have arrived; a call that leaves
        // before any read-back (a problem with nothing to optimise, a failed build) must not keep >= 32 MB pinned per handle.
        struct ArenaGuard {
            mage_ba* h;
            ~ArenaGuard() { if (!h->build_arena.blocks.empty()) { (void)hipStreamSynchronize(h->stream); h->build_arena.release(); } }
        } arena_guard{ h };
        h->stats.clear();
        h->last_outliers.clear();
        for (size_t it = 0; it < n_iter; ++it)
            if (huber[it] < 0.f) return fail(MAGE_ERR_INVALID_ARGUMENT, "Huber widths must be nonnegative");
        const BaDeviceView& v = h->view;
        double err_sum = 0, cnt = 0;
        size_t nout = 0;
        bool in_one_launch = false, ids_from_device = false;
        size_t out_prefix = 0;
        const bool sharded = h->shard_ranks > 0;
        if (sharded && !h->shard_reduce) return fail(MAGE_ERR_INVALID_ARGUMENT, "landmark-sharded map without an all-reduce callback");
        // the tracker's per-frame problems: one upload, one launch, one record back (frame_step above)
        if (frame_applies(h, n_iter)) {
            bool taken = false;
            MAGE_TRY(frame_step(h, huber, n_iter, max_err_sq, &taken, &err_sum, &cnt, &nout));
            if (taken) {
                if (mean_sq_err) *mean_sq_err = (float)(err_sum / cnt);
                for (size_t i = 0; i < h->last_outliers.size() && outliers && i < capacity; ++i) outliers[i] = h->last_outliers[i];
                if (n_outliers) *n_outliers = nout;
                return MAGE_OK;
            }
        }
        // a handle that was stepped on the frame path has no structure on the device (`dirty`); when its graph was not edited since,
        // the general path must not restart the optimiser (iteration 0 re-seeds lambda) just because it builds the structure now
        const bool keep_lm = h->frame_valid && h->dirty && h->frame_gen == h->edit_gen && !h->soft_dirty;
        const int keep_iteration = h->iteration;
        h->frame_valid = false;
        {
            // StepOptimizer's entry conditions (BundlerLib.cpp:132-149).  A landmark-sharded map enters collectively: the ranks
            // agree ONCE per step (a) that every rank has its structure -- a rank whose build failed (out of memory, an
            // unsupported shape) must not leave the others waiting in the next all-reduce, so the flag carries the failure and
            // ALL ranks abandon the step -- and (b) whether the optimiser starts over: the reference does when anything changed
            // (BundlerLib.cpp:135-138), and here "anything" includes another rank's observations.
            const bool reinit = h->dirty || h->soft_dirty;
            mage_status init_rc = MAGE_OK;
            std::string init_err;
            if (h->dirty && (n_iter > 0 || sharded)) {          // (sharded, no iteration: the post-pass below is collective too)
                init_rc = initialize_optimization(h);
                if (init_rc == MAGE_OK && keep_lm) h->iteration = keep_iteration;
                if (init_rc != MAGE_OK) {
                    if (!sharded) return init_rc;
                    init_err = last_error_ref();
                }
            }
            if (sharded) {
                if (init_rc != MAGE_OK) {                        // whatever the build got to: the flag needs the scalar block and its mirror
                    if (h->d_scal.reserve(SC_PAD + 2) != MAGE_OK || ensure_pinned_mirrors(h) != MAGE_OK) { last_error_ref() = init_err; return init_rc; }
                }
                h->h_scal[SC_SHARD_FLAG] = init_rc != MAGE_OK ? 2.0 : (reinit ? 1.0 : 0.0);
                MAGE_HIP(hipMemcpyAsync(h->d_scal.p + SC_SHARD_FLAG, h->h_scal + SC_SHARD_FLAG, sizeof(double), hipMemcpyHostToDevice, h->stream));
                if (h->shard_reduce(h->shard_user, h->d_scal.p + SC_SHARD_FLAG, 1, 1, (void*)h->stream) != 0)
                    return fail(MAGE_ERR_DEVICE, "landmark-sharded map: the all-reduce callback failed");
                MAGE_TRY(read_scalars(h));
                if (h->h_scal[SC_SHARD_FLAG] >= 2.0) {
                    if (init_rc != MAGE_OK) { last_error_ref() = init_err; return init_rc; }
                    return fail(MAGE_ERR_DEVICE, "landmark-sharded map: another rank could not build its part of the problem; the step was abandoned on every rank");
                }
                if (h->h_scal[SC_SHARD_FLAG] != 0.0 && n_iter > 0) { h->iteration = 0; h->soft_dirty = false; }
            }
        }
        if (n_iter > 0) {
            if (!sharded && !h->dirty && !h->useless && v.n_L > 0 && ba_pose_lm_applies(v, n_iter)) {
                if (h->soft_dirty) {
                    h->iteration = 0; h->soft_dirty = false;
                    if (h->n_active_remaining <= 0 && h->n_active_tethers == 0) h->useless = true;
                }
                if (!h->useless) {
                    PoseLmArgs a{};
                    a.n_huber = (int)n_iter;
                    for (size_t it = 0; it < n_iter; ++it) a.huber[it] = huber[it];
                    a.max_err_sq = (double)max_err_sq; a.lambda = h->lambda; a.user_lambda = h->user_lambda; a.ni = h->ni; a.iteration = h->iteration;
                    MAGE_TRY(h->d_pose_lm.reserve(1));
                    MAGE_TRY(ensure_pinned_mirrors(h));
                    ba_launch_pose_lm(v, a, h->d_pose_lm.p, h->d_flagL.p, nullptr, h->stream);
                    MAGE_HIP(hipMemcpyAsync(h->h_pose_lm, h->d_pose_lm.p, sizeof(PoseLmResult), hipMemcpyDeviceToHost, h->stream));
                    MAGE_HIP(hipEventRecord(h->ev[3], h->stream));
                    for (;;) {
                        const hipError_t e = hipEventQuery(h->ev[3]);
                        if (e == hipSuccess) break;
                        if (e != hipErrorNotReady) MAGE_HIP(e);
                    }
                    if (!h->build_arena.blocks.empty()) h->build_arena.release();
                    const PoseLmResult& r = *h->h_pose_lm;
                    h->lambda = r.lambda; h->ni = r.ni; h->iteration = r.iteration;
                    if (r.flips & 1) { h->cur ^= 1; refresh_view_state(h); }
                    h->host_state_fresh = false;
                    for (int i = 0; i < r.n_stats && i < POSE_LM_MAX_ITERS; ++i) {
                        mage_ba_iter_stats tr{};
                        tr.code = r.stats[i].code; tr.trials = r.stats[i].trials; tr.chi2_before = r.stats[i].chi2_before;
                        tr.chi2_after = r.stats[i].chi2_after; tr.lambda = r.stats[i].lambda;
                        if (h->stats.size() < 64) h->stats.push_back(tr);
                    }
                    err_sum = r.err_sum; cnt = r.err_cnt; nout = (size_t)r.n_out;
                    in_one_launch = true;
                }
            }
        }
        if (!in_one_launch) {
            // the first OUT_PREFIX ids of the outlier list ride the same read-back as the three sums (usually that is the whole list:
            // a run that removed outliers last time will again -- twice that many, at least 64, at most OUT_PREFIX; a 16 KB copy is ~8 us
            // slower than a 100-byte one, so a run without outliers does not pay for it)
            const size_t out_expect = h->out_expect;
            auto prefix_now = [&]() { return std::min<size_t>(std::min<size_t>(OUT_PREFIX, (size_t)h->view.n_L), std::max<size_t>(64, 2 * out_expect)); };
            const bool no_spec = conservative_paths();
            bool post_done = false;
            size_t prefix = 0;
            for (size_t it = 0; it < n_iter; ++it) {
                bool cont = true;
                PostPassPlan plan;
                // unsharded maps (both the large- and the small-problem path): the outlier pass is queued behind every trial that may be the call's last
                // (the structure must exist: a dirty graph is built first, inside step_optimizer, and then has no queued pass yet --
                // its first trial is queued only on the next iteration; the classic pass below covers it)
                // (only behind the trials of the call's LAST iteration: an earlier iteration ends the call only when it returns Terminate,
                // which the classic pass below covers -- queueing behind every trial of a 25-iteration call would be 50 idle launches)
                const bool can_plan = !no_spec && !sharded && !h->dirty && !h->useless && v.n_L > 0 && it + 1 == n_iter;
                if (can_plan) { plan.last_iteration = it + 1 == n_iter; plan.max_err_sq = (double)max_err_sq; plan.prefix = prefix_now(); }
                MAGE_TRY(step_optimizer(h, (double)huber[it], &cont, can_plan ? &plan : nullptr));
                if (can_plan && plan.done) { post_done = true; prefix = plan.prefix; }
                if (!cont) break;
            }
            // post-pass over the active observations of the last initialisation
            if (!sharded && v.n_L == 0) return MAGE_OK;     // count == 0 -> NaN
            int* small_counter = h->d_queue.p + chol_sync_ints(v.n_pad);
            ids_from_device = true;
            if (!post_done) {
                if (!sharded && ba_small_applies(v)) ba_small_classify(v, (double)max_err_sq, h->d_out_ids, small_counter + 1, h->out_cursor, small_counter, h->res_doubles ? h->d_scal.p + SC_PAD : nullptr, h->stream);
                else ba_launch_classify(v, (double)max_err_sq, h->d_out_ids, small_counter + 1, h->out_cursor, h->stream);
                prefix = prefix_now();
            }
            if (sharded) {
                // the mean error is the map's, the list a rank's own; whether anything was removed ANYWHERE decides the re-initialisation
                MAGE_HIP(hipMemcpyAsync(h->d_scal.p + SC_NOUT_OWN, h->d_scal.p + SC_NOUT, sizeof(double), hipMemcpyDeviceToDevice, h->stream));
                if (h->shard_reduce(h->shard_user, h->d_scal.p + SC_ERRSUM, 3, 0, (void*)h->stream) != 0)
                    return fail(MAGE_ERR_DEVICE, "landmark-sharded map: the all-reduce callback failed");
            }
            if (!post_done) MAGE_TRY(read_scalars(h, prefix));
            h->result_in_mirror = h->res_doubles > 0 && !sharded && ba_small_applies(v);      // (post_done: the kernel ran, i.e. found the call finished -- plan.done)
            err_sum = h->h_scal[SC_ERRSUM]; cnt = h->h_scal[SC_ERRCNT];
            nout = (size_t)h->h_scal[sharded ? SC_NOUT_OWN : SC_NOUT];
            if (sharded && h->h_scal[SC_NOUT] > 0) h->soft_dirty = true;
            h->out_cursor += (int)nout;
            h->out_expect = nout;
            out_prefix = prefix;
        }
        if (mean_sq_err) *mean_sq_err = (float)(err_sum / cnt);
        if (nout > 0) {
            // The whole list stays in the handle (mage_ba_get_outliers): a caller whose buffer was too small loses nothing --
            // the reference appends to a std::vector and its callers drop these associations from the map (BundleAdjust.cpp:316-320).
            std::vector<uint32_t>& ids = h->last_outliers;
            if (ids_from_device) {
                ids.resize(nout);
                const size_t first = std::min<size_t>(nout, out_prefix);
                std::copy(h->h_out_ids, h->h_out_ids + first, ids.begin());
                if (nout > first) {
                    MAGE_HIP(hipMemcpyAsync(ids.data() + first, h->d_out_ids + first, (nout - first) * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
                    MAGE_HIP(hipStreamSynchronize(h->stream));
                }
            } else {
                h->flag_host.resize(v.n_L);
                MAGE_HIP(hipMemcpyAsync(h->flag_host.data(), h->d_flagL.p, (size_t)v.n_L, hipMemcpyDeviceToHost, h->stream));
                if (h->L_edge_host.size() != (size_t)v.n_L) {          // lists built on the device: the position -> observation map comes back once
                    h->L_edge_host.resize(v.n_L);
                    MAGE_HIP(hipMemcpyAsync(h->L_edge_host.data(), h->d_L_edge.p, (size_t)v.n_L * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
                }
                MAGE_HIP(hipStreamSynchronize(h->stream));
                ids.reserve(nout);
                for (int i = 0; i < v.n_L; ++i) if (h->flag_host[i]) ids.push_back(h->L_edge_host[i]);
            }
            std::sort(ids.begin(), ids.end());
            for (size_t i = 0; i < ids.size(); ++i) {
                h->obs[ids[i]].removed = 1;                                  // removeEdge
                if (outliers && i < capacity) outliers[i] = ids[i];
            }
            h->soft_dirty = true;
            h->n_active_remaining -= (long long)ids.size();
        }
        if (n_outliers) *n_outliers = nout;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_get_outliers(const mage_ba* h, uint32_t* outliers, size_t capacity, size_t* count)
{
    if (!h || !count) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
    *count = h->last_outliers.size();
    for (size_t i = 0; outliers && i < h->last_outliers.size() && i < capacity; ++i) outliers[i] = h->last_outliers[i];
    return MAGE_OK;
}

// ---- landmark-sharded maps (include/mage_ba.h)
MAGE_EXPORT mage_status mage_ba_set_landmark_shard(mage_ba* h, int rank, int n_ranks, mage_ba_allreduce_fn allreduce, void* user)
{
    if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
    if (n_ranks < 0 || (n_ranks > 0 && (rank < 0 || rank >= n_ranks || !allreduce))) return fail(MAGE_ERR_INVALID_ARGUMENT, "rank %d of %d ranks, callback %p", rank, n_ranks, (void*)allreduce);
    if ((h->shard_ranks > 0) != (n_ranks > 0)) mark_edited(h);            // which cameras are in the system depends on it
    h->shard_rank = n_ranks > 0 ? rank : 0; h->shard_ranks = n_ranks;
    h->shard_reduce = n_ranks > 0 ? allreduce : nullptr; h->shard_user = user;
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_ba_partition_landmarks(size_t n_points, size_t n_observations, const uint32_t* point_index, int n_ranks, int32_t* owner)
{
    return guarded([&]() -> mage_status {
        if (n_ranks < 1 || (n_observations && !point_index) || (n_points && !owner)) return fail(MAGE_ERR_INVALID_ARGUMENT, "bad argument");
        std::vector<uint64_t> k(n_points, 0);
        for (size_t e = 0; e < n_observations; ++e) {
            if (point_index[e] >= n_points) return fail(MAGE_ERR_INVALID_ARGUMENT, "observation %zu refers to point %u of %zu", e, point_index[e], n_points);
            k[point_index[e]]++;
        }
        // heaviest point first (k (k + 1) / 2 blocks of S; ties: lower index) onto the lightest rank (ties: lower rank)
        std::vector<uint32_t> order(n_points);
        for (size_t i = 0; i < n_points; ++i) { order[i] = (uint32_t)i; k[i] = k[i] * (k[i] + 1) / 2; }
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return k[a] > k[b]; });
        typedef std::pair<uint64_t, int> Load;
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
        for (int r = 0; r < n_ranks; ++r) heap.push({ 0, r });
        for (uint32_t p : order) {
            Load l = heap.top(); heap.pop();
            owner[p] = l.second;
            heap.push({ l.first + k[p], l.second });
        }
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_device_allreduce_local(double* const* bufs_device, int n_bufs, size_t count, int op, void* stream)
{
    return guarded([&]() -> mage_status {
        if (!bufs_device || n_bufs < 1 || (op != 0 && op != 1)) return fail(MAGE_ERR_INVALID_ARGUMENT, "bad argument");
        for (int k = 0; k < n_bufs; ++k) if (!bufs_device[k] && count) return fail(MAGE_ERR_INVALID_ARGUMENT, "null buffer %d", k);
        if (!ba_launch_allreduce_local(bufs_device, n_bufs, count, op, (hipStream_t)stream)) return fail(MAGE_ERR_UNSUPPORTED, "at most 16 buffers");
        MAGE_HIP(hipGetLastError());
        return MAGE_OK;
    });
}

// ---- device-resident pose exchange (window-sharded maps; include/mage_ba.h)
MAGE_EXPORT mage_status mage_ba_bind_pose_exchange(mage_ba* h, size_t n_export, const uint32_t* export_cameras, const uint32_t* export_rows,
                                                   size_t n_import, const uint32_t* import_cameras, const uint32_t* import_rows)
{
    return guarded([&]() -> mage_status {
        if (!h || (n_export && (!export_cameras || !export_rows)) || (n_import && (!import_cameras || !import_rows)))
            return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        for (size_t k = 0; k < n_export; ++k)
            if (export_cameras[k] >= h->cams.size() || !h->cams[export_cameras[k]].set) return fail(MAGE_ERR_INVALID_ARGUMENT, "export camera %u is not a set camera", export_cameras[k]);
        for (size_t k = 0; k < n_import; ++k)
            if (import_cameras[k] >= h->cams.size() || !h->cams[import_cameras[k]].set) return fail(MAGE_ERR_INVALID_ARGUMENT, "import camera %u is not a set camera", import_cameras[k]);
        MAGE_DEVICE_SCOPE(h->device);
        MAGE_HIP(hipStreamSynchronize(h->stream));          // a previous export / import may still read the old lists
        MAGE_TRY(h->d_x_exp_cam.upload(export_cameras, n_export, h->stream)); MAGE_TRY(h->d_x_exp_row.upload(export_rows, n_export, h->stream));
        MAGE_TRY(h->d_x_imp_cam.upload(import_cameras, n_import, h->stream)); MAGE_TRY(h->d_x_imp_row.upload(import_rows, n_import, h->stream));
        MAGE_HIP(hipStreamSynchronize(h->stream));          // the sources are the caller's pageable arrays
        h->n_x_exp = n_export; h->n_x_imp = n_import;
        return MAGE_OK;
    });
}

// Both calls are ordered AS IF ENQUEUED ON `stream`: the handle's stream first waits for what `stream` holds (the caller's
// zero-fill before an export, its all-reduce before an import), and `stream` then waits for the kernel (so the caller's next
// zero-fill cannot overtake an import that still reads the block).  Two events, no host synchronisation.
static mage_status join_before(mage_ba* h, void* stream)
{
    MAGE_TRY(ensure_events(h->ev_x, 2, hipEventDisableTiming));
    if (stream != (void*)h->stream) {          // NULL is the device's null stream: the handle's stream is non-blocking, so it must be joined explicitly too
        MAGE_HIP(hipEventRecord(h->ev_x[0], static_cast<hipStream_t>(stream)));
        MAGE_HIP(hipStreamWaitEvent(h->stream, h->ev_x[0], 0));
    }
    return MAGE_OK;
}
static mage_status join_after(mage_ba* h, void* stream)
{
    if (stream != (void*)h->stream) {
        MAGE_HIP(hipEventRecord(h->ev_x[1], h->stream));
        MAGE_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(stream), h->ev_x[1], 0));
    }
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_ba_export_poses_device(mage_ba* h, double* block, void* stream)
{
    return guarded([&]() -> mage_status {
        if (!h || !block) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        MAGE_DEVICE_SCOPE(h->device);
        if (!h->state_on_device) MAGE_TRY(upload_state(h));
        MAGE_TRY(join_before(h, stream));
        if (h->n_x_exp) ba_launch_export_poses(h->d_pose[h->cur].p, h->d_x_exp_cam.p, h->d_x_exp_row.p, h->n_x_exp, block, h->stream);
        return join_after(h, stream);
    });
}

MAGE_EXPORT mage_status mage_ba_import_poses_device(mage_ba* h, const double* block, void* stream)
{
    return guarded([&]() -> mage_status {
        if (!h || !block) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        MAGE_DEVICE_SCOPE(h->device);
        if (!h->state_on_device) MAGE_TRY(upload_state(h));
        MAGE_TRY(join_before(h, stream));
        if (h->n_x_imp) {
            ba_launch_import_poses(h->d_pose[0].p, h->d_pose[1].p, h->d_x_imp_cam.p, h->d_x_imp_row.p, h->n_x_imp, block, h->stream);
            h->host_state_fresh = false; h->result_in_mirror = false;
            h->iteration = 0;          // a different linear system: the optimiser starts over, the graph is unchanged
        }
        return join_after(h, stream);
    });
}

MAGE_EXPORT mage_status mage_ba_synchronize(mage_ba* h)
{
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
        MAGE_DEVICE_SCOPE(h->device);
        MAGE_HIP(hipStreamSynchronize(h->stream));
        return MAGE_OK;
    });
}

static void pose_to_f32(const HostCam& c, float t[3], float Rcm[9])
{
    // q.normalized().toRotationMatrix() in float64, then cast (BundlerLib.cpp:463-464)
    double n = std::sqrt(c.q[0] * c.q[0] + c.q[1] * c.q[1] + c.q[2] * c.q[2] + c.q[3] * c.q[3]);
    double x = c.q[0] / n, y = c.q[1] / n, z = c.q[2] / n, w = c.q[3] / n;
    double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    double R[9] = { 1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy) };
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Rcm[cc * 3 + r] = (float)R[r * 3 + cc];
    t[0] = (float)c.t[0]; t[1] = (float)c.t[1]; t[2] = (float)c.t[2];
}

MAGE_EXPORT mage_status mage_ba_get_pose(const mage_ba* h, size_t idx, float position[3], float R_colmajor[9])
{
    return guarded([&]() -> mage_status {
        if (!h || !position || !R_colmajor) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (idx >= h->cams.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "camera index %zu out of range", idx);
        MAGE_TRY(download_state(h));
        pose_to_f32(h->cams[idx], position, R_colmajor);
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_get_point(const mage_ba* h, size_t idx, float xyz[3])
{
    return guarded([&]() -> mage_status {
        if (!h || !xyz) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (idx >= h->pt_set.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "point index %zu out of range", idx);
        MAGE_TRY(download_state(h));
        for (int a = 0; a < 3; ++a) xyz[a] = (float)h->pts[idx * 3 + a];
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_get_poses_bulk(const mage_ba* h, size_t count, float* positions3, float* R9)
{
    return guarded([&]() -> mage_status {
        if (!h || !positions3 || !R9) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (count > h->cams.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "count out of range");
        MAGE_TRY(download_state(h));
        for (size_t i = 0; i < count; ++i) pose_to_f32(h->cams[i], positions3 + i * 3, R9 + i * 9);
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_get_points_bulk(const mage_ba* h, size_t count, float* xyz3)
{
    return guarded([&]() -> mage_status {
        if (!h || !xyz3) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (count > h->pt_set.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "count out of range");
        MAGE_TRY(download_state(h));
        for (size_t i = 0; i < count * 3; ++i) xyz3[i] = (float)h->pts[i];
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_get_state_f64(const mage_ba* h, double* poses7, double* points3)
{
    return guarded([&]() -> mage_status {
        if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
        MAGE_TRY(download_state(h));
        if (poses7)
            for (size_t i = 0; i < h->cams.size(); ++i) {
                for (int a = 0; a < 4; ++a) poses7[i * 7 + a] = h->cams[i].q[a];
                for (int a = 0; a < 3; ++a) poses7[i * 7 + 4 + a] = h->cams[i].t[a];
            }
        if (points3) std::copy(h->pts.begin(), h->pts.end(), points3);
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_get_iter_stats(const mage_ba* h, mage_ba_iter_stats* out, size_t capacity, size_t* count)
{
    if (!h || !count) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
    *count = h->stats.size();
    for (size_t i = 0; out && i < h->stats.size() && i < capacity; ++i) out[i] = h->stats[i];
    return MAGE_OK;
}

// DIAGNOSTIC: one list of the graph structure as it sits on the device (tests compare the host and the device build with it).
MAGE_EXPORT mage_status mage_ba_debug_structure(mage_ba* h, const char* name, void* out, size_t capacity_bytes, size_t* bytes)
{
    return guarded([&]() -> mage_status {
        if (!h || !name || !bytes) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        MAGE_DEVICE_SCOPE(h->device);
        if (h->dirty && h->frame_valid && h->frame_gen == h->edit_gen) {      // stepped on the frame path: the structure was never put on the device
            const int keep_iteration = h->iteration;
            const bool keep_soft = h->soft_dirty;
            MAGE_TRY(initialize_optimization(h));
            h->iteration = keep_iteration; h->soft_dirty = keep_soft;
            h->frame_valid = false;
        }
        if (h->dirty) return fail(MAGE_ERR_INVALID_ARGUMENT, "the structure is built by the first step");
        const BaDeviceView& v = h->view;
        const std::string n(name);
        const void* src = nullptr; size_t nb = 0;
        int ncon = 0;
        if (v.n_blk > 0) {
            MAGE_HIP(hipMemcpyAsync(&ncon, v.blk_ptr + v.n_blk, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            MAGE_HIP(hipStreamSynchronize(h->stream));
        }
        const int sizes[12] = { v.n_L, v.n_lm, v.n_fc, v.n_w, v.n_blk, v.n_blk_slots, ncon, v.dup_slots, v.n_pad, h->built_on_device ? 1 : 0, 0, 0 };
        if (n == "sizes") { src = nullptr; nb = sizeof(sizes); }
        else if (n == "cam2hc") { src = v.cam2hc; nb = (size_t)v.n_cams * 4; }
        else if (n == "hc2cam") { src = v.hc2cam; nb = (size_t)v.n_fc * 4; }
        else if (n == "L_edge") { src = v.L_edge; nb = (size_t)v.n_L * 4; }
        else if (n == "L_uv") { src = v.L_uv; nb = (size_t)v.n_L * 8; }
        else if (n == "L_info") { src = v.L_info; nb = (size_t)v.n_L * 4; }
        else if (n == "L_cam") { src = v.L_cam; nb = (size_t)v.n_L * 4; }
        else if (n == "L_pt") { src = v.L_pt; nb = (size_t)v.n_L * 4; }
        else if (n == "L_slot") { src = v.L_slot; nb = (size_t)v.n_L * 4; }
        else if (n == "lm_ptr") { src = v.lm_ptr; nb = (size_t)(v.n_lm + 1) * 4; }
        else if (n == "lm_pt") { src = v.lm_pt; nb = (size_t)v.n_lm * 4; }
        else if (n == "lm_wptr") { src = v.lm_wptr; nb = (size_t)(v.n_lm + 1) * 4; }
        else if (n == "w_hc") { src = v.w_hc; nb = (size_t)v.n_w * 4; }
        else if (n == "w_lm") { src = v.w_lm; nb = (size_t)v.n_w * 4; }
        else if (n == "camE_ptr") { src = v.camE_ptr; nb = (size_t)(v.n_fc + 1) * 4; }
        else if (n == "camS_ptr") { src = v.camS_ptr; nb = (size_t)(v.n_fc + 1) * 4; }
        else if (n == "camS") { src = v.camS; nb = (size_t)v.n_w * 4; }
        else if (n == "blk_ptr") { src = v.blk_ptr; nb = (size_t)(v.n_blk + 1) * 4; }
        else if (n == "blk_ij") { src = v.blk_ij; nb = (size_t)v.n_blk * 8; }
        else if (n == "con") { src = v.con; nb = (size_t)ncon * 8; }
        else if (n == "blk_order") { src = v.blk_order; nb = (size_t)v.n_blk_slots * 4; }
        else if (n == "stream_ptr") { src = v.stream_ptr; nb = v.stream_ptr ? (size_t)(v.n_stream_groups + 1) * 4 : 0; }
        else if (n == "stream_blks") { src = v.stream_blks; nb = v.stream_blks ? (size_t)v.n_blk * 16 : 0; }
        else if (n == "stream_stamps") { src = v.stream_stamps; nb = v.stream_stamps ? (size_t)v.n_stream_groups * 8 * 32 : 0; }
        else if (n == "camE") {
            int ne = 0;
            if (v.n_fc > 0) {
                MAGE_HIP(hipMemcpyAsync(&ne, v.camE_ptr + v.n_fc, sizeof(int), hipMemcpyDeviceToHost, h->stream));
                MAGE_HIP(hipStreamSynchronize(h->stream));
            }
            src = v.camE; nb = (size_t)ne * 4;
        }
        else return fail(MAGE_ERR_INVALID_ARGUMENT, "unknown list '%s'", name);
        *bytes = nb;
        if (!out || capacity_bytes < nb) return MAGE_OK;
        if (n == "sizes") { std::memcpy(out, sizes, nb); return MAGE_OK; }
        if (nb) {
            MAGE_HIP(hipMemcpyAsync(out, src, nb, hipMemcpyDeviceToHost, h->stream));
            MAGE_HIP(hipStreamSynchronize(h->stream));
        }
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_ba_enable_profiling(mage_ba* h, int enable)
{
    if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
    if (enable) {
        DeviceScope scope(h->device);
        if (ensure_events(h->ev, 3, hipEventDefault) != MAGE_OK || ensure_events(h->ev_p, 4, hipEventDefault) != MAGE_OK) return MAGE_ERR_DEVICE;
    }
    h->profiling = enable == 1;
    h->profiling_factor = enable == 2;
    h->prof.n_factorizations = 0; h->prof.factor_ms_total = 0; h->prof.schur_launches = 0; h->prof.schur_ms_total = 0;
    h->prof.linearize_launches = 0; h->prof.linearize_ms_total = 0; h->prof.update_launches = 0; h->prof.update_ms_total = 0;
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_ba_use_skyline(mage_ba* h, int enable)
{
    if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
    if (h->use_skyline != (enable != 0)) { h->use_skyline = enable != 0; h->dirty = true; }          // (the skyline is read back with the structure)
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_ba_debug_schur_per_block(mage_ba* h, int enable)
{
    if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
    if (h->schur_per_block != (enable != 0)) { h->schur_per_block = enable != 0; h->positions_valid = false; }
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_ba_get_profile(const mage_ba* h, mage_ba_profile* out)
{
    if (!h || !out) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
    *out = h->prof;
    out->trials_rerun_after_stall = (uint64_t)h->stall_retries_total;
    out->fallback_to_separate_launches = chol_merge_fallback_active() ? 1 : 0;
    return MAGE_OK;
}
