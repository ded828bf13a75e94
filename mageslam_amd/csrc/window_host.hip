// window_host.hip -- the window-sharded map driver behind include/mage_window.h: host C++ over the mage_ba_* C ABI.
//
// What a window is follows the reference's local bundle adjustment (Map/ThreadSafeMap.cpp:868-971: the window's keyframes are
// free, all map points they observe are in the problem, every other keyframe observing one of those points is a FIXED camera,
// :939), and how a problem is built and stepped follows BundleAdjust.cpp:25-193, 281-354.  The exchange between windows --
// the one step of this path that crosses GPUs -- never leaves HBM: see mage_ba_export_poses_device / mage_ba_import_poses_device.
#include <atomic>
#include <cmath>
#include <memory>
#include <thread>
#include <vector>

#include "mage_common.h"
#include "../../include/mage_window.h"

using namespace mage;

namespace {

struct Window {
    std::vector<uint32_t> own, cams, pts;        // global ids: own keyframes; own + overlap + halo; points (ascending)
    size_t n_obs = 0;
    bool mine = false;
    mage_ba* ba = nullptr;
    // result of the last step
    float mse = NAN;
    size_t n_outliers = 0;
    size_t n_removed = 0;                        // observations removed by the outlier passes of EARLIER steps (they no longer count in mse)
    mage_status status = MAGE_OK;
    std::string error;
};

template <typename F>
mage_status guarded_w(F&& f)
{
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(MAGE_ERR_OUT_OF_MEMORY, "host allocation failed"); }
    catch (const std::exception& e) { return fail(MAGE_ERR_DEVICE, "unexpected exception: %s", e.what()); }
    catch (...) { return fail(MAGE_ERR_DEVICE, "unexpected exception"); }
}

}  // namespace

struct mage_wmap {
    mage_wmap_params P{};
    int device = 0;
    size_t n_cams = 0;
    std::vector<Window> windows;
    std::vector<int> mine;
    hipStream_t xstream = nullptr;
    double* block = nullptr;                     // n_cams x 8 f64, device
    mage_allreduce_fn allreduce = nullptr;
    void* allreduce_ctx = nullptr;
    bool exchanged = false;                      // the block holds the map only after the first exchange

    ~mage_wmap()
    {
        DeviceScope scope(device);
        for (Window& w : windows) if (w.ba) mage_ba_destroy(w.ba);
        if (xstream) { (void)hipStreamSynchronize(xstream); (void)hipStreamDestroy(xstream); }
        if (block) (void)hipFree(block);
    }
};

MAGE_EXPORT mage_status mage_wmap_create(const mage_wmap_params* params, size_t n_cams, const float* t3, const float* R9, const float* K4,
                                         const uint8_t* is_fixed, size_t n_pts, const float* xyz3, size_t n_obs, const float* uv2,
                                         const uint32_t* obs_cam, const uint32_t* obs_pt, const float* info, mage_wmap** out)
{
    return guarded_w([&]() -> mage_status {
        if (!params || !out || !t3 || !R9 || !K4 || !xyz3 || (n_obs && (!uv2 || !obs_cam || !obs_pt || !info)))
            return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        *out = nullptr;
        const int W = params->n_windows, world = params->world, rank = params->rank;
        if (W < 1 || world < 1 || rank < 0 || rank >= world || params->overlap < 0 || params->threads < 1)
            return fail(MAGE_ERR_INVALID_ARGUMENT, "bad window parameters (n_windows %d, overlap %d, rank %d of %d, threads %d)", W, params->overlap, rank, world, params->threads);
        if (n_cams == 0 || n_cams > 0x7fffffffull || n_pts > 0x7fffffffull || n_obs > 0x7fffffffull) return fail(MAGE_ERR_INVALID_ARGUMENT, "bad sizes");
        for (size_t e = 0; e < n_obs; ++e)
            if (obs_cam[e] >= n_cams || obs_pt[e] >= n_pts) return fail(MAGE_ERR_INVALID_ARGUMENT, "observation %zu refers to a camera or point out of range", e);
        int dev = 0;
        MAGE_TRY(select_device(params->device, &dev));
        std::unique_ptr<mage_wmap> h(new mage_wmap());
        h->P = *params; h->device = dev; h->n_cams = n_cams;
        MAGE_DEVICE_SCOPE(dev);
        MAGE_HIP(hipStreamCreateWithFlags(&h->xstream, hipStreamNonBlocking));
        MAGE_HIP(hipMalloc(reinterpret_cast<void**>(&h->block), n_cams * 8 * sizeof(double)));
        MAGE_HIP(hipMemsetAsync(h->block, 0, n_cams * 8 * sizeof(double), h->xstream));

        const int nc = (int)n_cams, ov = params->overlap;
        h->windows.resize(W);
        std::vector<uint8_t> pt_sel(n_pts), cam_sel(n_cams);
        std::vector<int> cam_l(n_cams), pt_l(n_pts);
        for (int w = 0; w < W; ++w) {
            Window& win = h->windows[w];
            const int lo = (int)((int64_t)w * nc / W), hi = (int)((int64_t)(w + 1) * nc / W);
            const int flo = std::max(0, lo - ov), fhi = std::min(nc, hi + ov);         // free keyframes: own + overlap
            win.mine = (int)((int64_t)w * world / W) == rank;
            std::fill(pt_sel.begin(), pt_sel.end(), 0);
            std::fill(cam_sel.begin(), cam_sel.end(), 0);
            for (size_t e = 0; e < n_obs; ++e)
                if ((int)obs_cam[e] >= flo && (int)obs_cam[e] < fhi) pt_sel[obs_pt[e]] = 1;
            std::vector<uint32_t> obs_idx;                                                 // every observation of those points, map order
            for (size_t e = 0; e < n_obs; ++e)
                if (pt_sel[obs_pt[e]]) { obs_idx.push_back((uint32_t)e); cam_sel[obs_cam[e]] = 1; }
            for (int c = lo; c < hi; ++c) win.own.push_back((uint32_t)c);
            win.cams = win.own;
            for (int c = flo; c < lo; ++c) win.cams.push_back((uint32_t)c);                // overlap: free here, owned elsewhere
            for (int c = hi; c < fhi; ++c) win.cams.push_back((uint32_t)c);
            const size_t n_free = win.cams.size();
            for (int c = 0; c < nc; ++c)
                if (cam_sel[c] && !(c >= flo && c < fhi)) win.cams.push_back((uint32_t)c); // halo: fixed
            for (size_t p = 0; p < n_pts; ++p) if (pt_sel[p]) win.pts.push_back((uint32_t)p);
            win.n_obs = obs_idx.size();
            if (!win.mine) continue;
            h->mine.push_back(w);
            // the sub-problem in local indices, through the bulk setters (BuildDataForG2O's call order)
            for (size_t k = 0; k < win.cams.size(); ++k) cam_l[win.cams[k]] = (int)k;
            for (size_t k = 0; k < win.pts.size(); ++k) pt_l[win.pts[k]] = (int)k;
            const size_t lc = win.cams.size(), lp = win.pts.size(), lo_n = obs_idx.size();
            std::vector<float> ct(lc * 3), cR(lc * 9), cK(lc * 4), px(lp * 3), uv(lo_n * 2), inf(lo_n);
            std::vector<uint8_t> fx(lc);
            std::vector<uint32_t> oc(lo_n), op(lo_n);
            for (size_t k = 0; k < lc; ++k) {
                const size_t g = win.cams[k];
                for (int a = 0; a < 3; ++a) ct[k * 3 + a] = t3[g * 3 + a];
                for (int a = 0; a < 9; ++a) cR[k * 9 + a] = R9[g * 9 + a];
                for (int a = 0; a < 4; ++a) cK[k * 4 + a] = K4[g * 4 + a];
                fx[k] = k < n_free ? (uint8_t)((is_fixed && is_fixed[g]) ? 1 : 0) : (uint8_t)1;
            }
            for (size_t k = 0; k < lp; ++k)
                for (int a = 0; a < 3; ++a) px[k * 3 + a] = xyz3[(size_t)win.pts[k] * 3 + a];
            for (size_t k = 0; k < lo_n; ++k) {
                const size_t e = obs_idx[k];
                uv[k * 2] = uv2[e * 2]; uv[k * 2 + 1] = uv2[e * 2 + 1]; inf[k] = info[e];
                oc[k] = (uint32_t)cam_l[obs_cam[e]]; op[k] = (uint32_t)pt_l[obs_pt[e]];
            }
            mage_ba_params bp{ 0, dev };
            MAGE_TRY(mage_ba_create(&bp, &win.ba));
            MAGE_TRY(mage_ba_alloc_cameras(win.ba, lc));
            MAGE_TRY(mage_ba_set_cameras_bulk(win.ba, lc, ct.data(), cR.data(), cK.data(), fx.data()));
            MAGE_TRY(mage_ba_alloc_points(win.ba, lp));
            if (lp) MAGE_TRY(mage_ba_set_points_bulk(win.ba, lp, px.data()));
            MAGE_TRY(mage_ba_alloc_observations(win.ba, lo_n));
            if (lo_n) MAGE_TRY(mage_ba_set_observations_bulk(win.ba, lo_n, uv.data(), oc.data(), op.data(), inf.data()));
            // exchange lists: the own keyframes are published, everything else (overlap + halo) is re-seeded from the block
            const size_t k_own = win.own.size();
            std::vector<uint32_t> ec(k_own), ic(lc - k_own);
            for (size_t k = 0; k < k_own; ++k) ec[k] = (uint32_t)k;
            for (size_t k = k_own; k < lc; ++k) ic[k - k_own] = (uint32_t)k;
            MAGE_TRY(mage_ba_bind_pose_exchange(win.ba, k_own, ec.data(), win.cams.data(), lc - k_own, ic.data(), win.cams.data() + k_own));
        }
        MAGE_HIP(hipStreamSynchronize(h->xstream));
        *out = h.release();
        return MAGE_OK;
    });
}

MAGE_EXPORT void mage_wmap_destroy(mage_wmap* h) { delete h; }

MAGE_EXPORT mage_status mage_wmap_set_allreduce(mage_wmap* h, mage_allreduce_fn fn, void* ctx)
{
    if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
    h->allreduce = fn; h->allreduce_ctx = ctx;
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_wmap_outer_iteration(mage_wmap* h, float huber, float max_err_sq, int inner, double* mean_sq_err)
{
    return guarded_w([&]() -> mage_status {
        if (!h || inner < 1) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle or inner < 1");
        if (mean_sq_err) *mean_sq_err = NAN;
        // Without the exchange the block rows of the other ranks' windows would stay zero and be imported as poses
        // (zero quaternion, zero translation) into every overlap / halo camera: refuse instead of corrupting the map.
        if (h->P.world > 1 && !h->allreduce)
            return fail(MAGE_ERR_INVALID_ARGUMENT, "a map sharded over %d ranks needs an all-reduce callback (mage_wmap_set_allreduce) before the first outer iteration", h->P.world);
        MAGE_DEVICE_SCOPE(h->device);
        // 1. the owned windows, independent between exchanges: `threads` host threads take them from a shared counter
        const std::vector<float> widths((size_t)inner, huber);
        auto step = [&](int w) {
            Window& win = h->windows[w];
            size_t n = 0; float mse = NAN;
            win.status = mage_ba_step(win.ba, widths.data(), widths.size(), max_err_sq, nullptr, 0, &n, &mse);
            if (win.status != MAGE_OK) win.error = mage_last_error();
            win.mse = mse; win.n_outliers = n;
        };
        const int nt = std::min<int>(h->P.threads, (int)h->mine.size());
        if (nt > 1) {
            std::atomic<int> next{ 0 };
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([&] {
                    DeviceScope scope(h->device);
                    for (int i = next.fetch_add(1); i < (int)h->mine.size(); i = next.fetch_add(1)) step(h->mine[i]);
                });
            for (auto& t : th) t.join();
        } else {
            for (int w : h->mine) step(w);
        }
        double err_sum = 0; size_t n_sum = 0;
        for (int w : h->mine) {
            const Window& win = h->windows[w];
            if (win.status != MAGE_OK) return fail(win.status, "window %d: %s", w, win.error.c_str());
            // the step's mean is over the observations that were active when it ran and were not classified as outliers by it:
            // everything removed by earlier steps is out of the graph (BundlerLib.cpp:386-446)
            const size_t gone = win.n_removed + win.n_outliers;
            const size_t n = win.n_obs > gone ? win.n_obs - gone : 0;
            if (n > 0 && std::isfinite(win.mse)) { err_sum += (double)win.mse * (double)n; n_sum += n; }
        }
        for (int w : h->mine) h->windows[w].n_removed += h->windows[w].n_outliers;
        if (mean_sq_err && n_sum) *mean_sq_err = err_sum / (double)n_sum;
        // 2. the exchange: zero, publish the owned rows, sum over the ranks, re-seed what each window does not own -- all in
        //    stream order on the device (every export / import is ordered as if enqueued on xstream)
        MAGE_HIP(hipMemsetAsync(h->block, 0, h->n_cams * 8 * sizeof(double), h->xstream));
        for (int w : h->mine) MAGE_TRY(mage_ba_export_poses_device(h->windows[w].ba, h->block, h->xstream));
        if (h->allreduce && h->allreduce(h->allreduce_ctx, h->block, h->n_cams * 8, h->xstream) != 0)
            return fail(MAGE_ERR_DEVICE, "the all-reduce callback reported a failure");
        for (int w : h->mine) {
            mage_ba* ba = h->windows[w].ba;
            float lam = 0;
            MAGE_TRY(mage_ba_get_lambda(ba, &lam));
            MAGE_TRY(mage_ba_import_poses_device(ba, h->block, h->xstream));
            if (lam > 0) MAGE_TRY(mage_ba_set_lambda(ba, lam));      // the damping carries over, as MappingWorker carries it from one BA to the next
        }
        h->exchanged = true;
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_wmap_get_pose_block(mage_wmap* h, double* poses8)
{
    return guarded_w([&]() -> mage_status {
        if (!h || !poses8) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument");
        if (!h->exchanged) return fail(MAGE_ERR_INVALID_ARGUMENT, "the pose block is defined after the first outer iteration (nothing has been published yet)");
        MAGE_DEVICE_SCOPE(h->device);
        MAGE_HIP(hipMemcpyAsync(poses8, h->block, h->n_cams * 8 * sizeof(double), hipMemcpyDeviceToHost, h->xstream));
        MAGE_HIP(hipStreamSynchronize(h->xstream));
        return MAGE_OK;
    });
}

MAGE_EXPORT mage_status mage_wmap_pose_block_device(mage_wmap* h, double** block_device, void** hip_stream)
{
    if (!h) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle");
    if (block_device) *block_device = h->block;
    if (hip_stream) *hip_stream = (void*)h->xstream;
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_wmap_window_info(const mage_wmap* h, int w, size_t* n_own, size_t* n_cameras, size_t* n_points, size_t* n_observations, int* owned)
{
    if (!h || w < 0 || w >= (int)h->windows.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "null handle or window out of range");
    const Window& win = h->windows[w];
    if (n_own) *n_own = win.own.size();
    if (n_cameras) *n_cameras = win.cams.size();
    if (n_points) *n_points = win.pts.size();
    if (n_observations) *n_observations = win.n_obs;
    if (owned) *owned = win.mine ? 1 : 0;
    return MAGE_OK;
}

MAGE_EXPORT mage_status mage_wmap_window_handle(const mage_wmap* h, int w, mage_ba** out)
{
    if (!h || !out || w < 0 || w >= (int)h->windows.size()) return fail(MAGE_ERR_INVALID_ARGUMENT, "null argument or window out of range");
    if (!h->windows[w].mine) return fail(MAGE_ERR_INVALID_ARGUMENT, "window %d is not owned by this rank", w);
    *out = h->windows[w].ba;
    return MAGE_OK;
}
