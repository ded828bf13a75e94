// orb_kernels.hip -- HIP kernels (gfx950) of the ORB front-end; byte/integer HBM-bound work, no MFMA.
//
// Reference code replaced (Core/MAGESLAM/Source/Image/OpenCVModified.cpp):
//   k_fast_nms       FAST_t<16> segment test + cornerScore<16> + 3x3 NMS  :926-1071, :1224-1510; RunByImageBorder :619-639
//   k_nms_emit       raster-order keypoint list from the kept map         :1499-1510
//   k_select         RetainBestFeatures + AdaptiveNonMaximalSuppresion    :571-617, :144-360
//   k_blur           cv::GaussianBlur(k x k, sigma 2, REFLECT_101) on u8  :853-865 (OpenCV 3.4.0 fixed-point path)
//   k_brief          ComputeOrbDescriptorsPrerotated                      :502-549
// Frames are independent: every kernel takes the frame index from blockIdx.y (batched over frames).
// All outputs are integers and must equal the CPU oracle bit for bit; float arithmetic that feeds comparisons
// (ANMS robustness factor, cv::fastAtan2) is written with plain operators under `#pragma clang fp contract(off)`: the
// __f*_rn helpers of this toolchain are ordinary operators inside header functions and do get fused into FMAs.
#include "orb_kernels.h"

namespace mage {
namespace {

// ---------------------------------------------------------------------------------------------
// FAST-9/16 score map.  32x8 pixel tile per workgroup, LDS tile with a 3-pixel halo.
// score = max over the 16 arcs of 9 contiguous ring pixels of the arc's minimum margin, for darker and for
// brighter rings; a pixel is a corner iff that maximum exceeds the threshold, and its score is maximum - 1
// (identical to the reference's threshold-table pre-test + min/max ladder, which computes the same quantity).
// ---------------------------------------------------------------------------------------------
constexpr int FT_W = 64, FT_H = ORB_BAND_ROWS;   // 3072 pixels per workgroup, 12 per thread: fat workgroups amortise the halo, the LDS clears and five barriers (480 rows = 10 bands)

// The ring differences fit 16 bits, so the score runs on PACKED pairs: register k holds (d[k], d[k + 8]) -- a ring pixel and its
// opposite.  A rotation of the ring by one position is "next register", and crossing position 7 -> 8 is a swap of the halves,
// so every windowed minimum / maximum (v_pk_min_i16 / v_pk_max_i16) serves two ring positions at once: ~100 operations where
// the 32-bit form needs ~180.
typedef short short2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ short2_t swap16(short2_t v) { return __builtin_shufflevector(v, v, 1, 0); }
__device__ __forceinline__ short2_t pmin(short2_t a, short2_t b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ short2_t pmax(short2_t a, short2_t b) { return __builtin_elementwise_max(a, b); }

__device__ __forceinline__ int fast_score_at(const uint8_t* __restrict__ c, int TP, int t)
{
    const int v = c[0];
    const int ox[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
    const int oy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };
    short2_t P[8], Q[8];                                  // P[k] = (d[k], d[k+8]),  Q[k] = swapped = (d[k+8], d[k])
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int lo = v - (int)c[oy[k] * TP + ox[k]], hi = v - (int)c[oy[k + 8] * TP + ox[k + 8]];
        P[k] = (short2_t){ (short)lo, (short)hi };
        Q[k] = (short2_t){ (short)hi, (short)lo };
    }
    // high-speed pre-test (a 9-arc always contains one pixel of every opposite pair): every pair needs a member > t (darker
    // ring) or every pair a member < -t (brighter ring)
    short2_t mx = pmax(P[0], Q[0]), mn = pmin(P[0], Q[0]);          // both halves equal: max / min of the pair
    short2_t all_dark = mx, all_bright = mn;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        all_dark = pmin(all_dark, pmax(P[k], Q[k]));                 // smallest pair-maximum
        all_bright = pmax(all_bright, pmin(P[k], Q[k]));             // largest pair-minimum
    }
    if (!((int)all_dark.x > t || (int)all_bright.x < -t)) return 0;
    // windowed minima / maxima by doubling: windows of 2, 4, 8 ring positions, then 9
    short2_t A2[8], B2[8], A4[8], B4[8], A8[8], B8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const short2_t nx = k < 7 ? P[k + 1] : Q[0];                 // positions k+1 and k+9
        A2[k] = pmin(P[k], nx); B2[k] = pmax(P[k], nx);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const short2_t na = k < 6 ? A2[k + 2] : swap16(A2[k - 6]), nb = k < 6 ? B2[k + 2] : swap16(B2[k - 6]);
        A4[k] = pmin(A2[k], na); B4[k] = pmax(B2[k], nb);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const short2_t na = k < 4 ? A4[k + 4] : swap16(A4[k - 4]), nb = k < 4 ? B4[k + 4] : swap16(B4[k - 4]);
        A8[k] = pmin(A4[k], na); B8[k] = pmax(B4[k], nb);
    }
    // arcs of 9: window of 8 starting at k plus position k + 8 (the opposite pixel = the swapped pair)
    short2_t dark = pmin(A8[0], Q[0]), bright = pmax(B8[0], Q[0]);
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        dark = pmax(dark, pmin(A8[k], Q[k]));          // darker ring: largest arc minimum of (v - ring)
        bright = pmin(bright, pmax(B8[k], Q[k]));      // brighter ring: smallest arc maximum of (v - ring)
    }
    const int m = max(max((int)dark.x, (int)dark.y), -min((int)bright.x, (int)bright.y));
    return m > t ? m - 1 : 0;
}

// Stage a (TH x TW)-byte window of the source image, top-left at (gx0, gy0) with gx0 a multiple of 4, into LDS with
// 32-bit loads when the source allows it (row pitch and base 4-byte aligned), bytes otherwise.  `fill(gx, gy)` maps
// out-of-image coordinates (zero padding for FAST, reflection for the blur).
template <int TW, int TH, int TP, bool REFLECT>
__device__ __forceinline__ void stage_window(uint8_t* __restrict__ tile, const uint8_t* __restrict__ I, int w, int h, int stride, int gx0, int gy0,
                                             int tw, int th)
{
    auto rx = [&](int p, int n) { if (n == 1) return 0; while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p; return p; };
    const bool aligned = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(I) & 3) == 0);
    const int nq = tw / 4;                              // tw is a multiple of 4
    for (int e = threadIdx.x; e < th * nq; e += 256) {
        const int ty = e / nq, tq = e % nq;
        const int gx = gx0 + 4 * tq;
        int gy = gy0 + ty;
        uint32_t v = 0;
        const bool row_ok = REFLECT || (gy >= 0 && gy < h);
        if (REFLECT) gy = rx(gy, h);
        if (row_ok) {
            if (aligned && gx >= 0 && gx + 3 < w) v = *reinterpret_cast<const uint32_t*>(I + (size_t)gy * stride + gx);
            else {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    int x = gx + b;
                    uint32_t px = 0;
                    if (REFLECT) px = I[(size_t)gy * stride + rx(x, w)];
                    else if (x >= 0 && x < w) px = I[(size_t)gy * stride + x];
                    v |= px << (8 * b);
                }
            }
        }
        *reinterpret_cast<uint32_t*>(tile + ty * TP + 4 * tq) = v;
    }
}

// Internal images (score map, blurred image) use a row pitch wp = w rounded up to 4 so every kernel below moves 4 pixels
// per 32-bit access; pixels in [w, wp) are written as 0.
// FAST score + 3x3 non-maximum suppression + border cull + response histogram, one 64x32 tile per workgroup.
// Scores are computed for the tile and a one-pixel ring around it (66x34), so the suppression of every tile pixel is decided
// here from LDS and the full-image count pass is gone; what reaches HBM is the KEPT map (score where the pixel survives, else
// 0), the per-frame histogram and the per-band (32 image rows = one tile row) keypoint counts for the raster-order emit pass.
constexpr int FH = 4;                                   // image halo of the staged window: 3 (ring) + 1 (score ring)
constexpr int SCW = FT_W + 2, SCH = FT_H + 2, SCP = 72; // score region and its LDS pitch
constexpr int SC_OFF = 3;                               // region pixel rx sits at byte rx + 3 of its row: the tile's quads are dword-aligned

__global__ __launch_bounds__(256) void k_fast_nms(const uint8_t* __restrict__ img, int w, int h, int stride, size_t frame_stride,
                                                  int threshold, int border, uint8_t* __restrict__ kept, uint8_t* __restrict__ raw_frame0, int wp,
                                                  int* __restrict__ hist, int* __restrict__ band_count)
{
    constexpr int TW = FT_W + 8, TH = FT_H + 2 * FH, TP = TW;        // window starts 4 pixels left of the tile (aligned)
    __shared__ __attribute__((aligned(4))) uint8_t tile[TH * TP + 8];
    __shared__ __attribute__((aligned(4))) uint8_t sc[SCH * SCP];     // scores of the tile and its ring
    __shared__ uint16_t cand[SCH * SCP];                              // pixels that survive the compass test
    __shared__ int lh[256];
    __shared__ int n_cand, n_kept;
    const int f = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const uint8_t* I = img + (size_t)f * frame_stride;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
    if (tid == 0) { n_cand = 0; n_kept = 0; }
    lh[tid] = 0;
    for (int e = tid; e < SCH * SCP / 4; e += 256) reinterpret_cast<uint32_t*>(sc)[e] = 0u;
    stage_window<TW, TH, TP, false>(tile, I, w, h, stride, x0 - 4, y0 - FH, TW, TH);
    __syncthreads();
    // Phase 1, every pixel of the 66x34 region: the compass test.  Any arc of 9 contiguous ring pixels contains two
    // NEIGHBOURING compass points (ring positions 0, 4, 8, 12), so a corner needs two neighbouring compass pixels both darker
    // or both brighter than the centre by more than the threshold -- a necessary condition costing 4 ring pixels.  Survivors
    // (a few per cent to ~20 %) are appended to an LDS list; the full 16-pixel score is then computed on the compacted list,
    // where every lane of a wavefront has real work.  Region pixel (rx, ry) = image (x0 - 1 + rx, y0 - 1 + ry) = tile byte
    // (rx + 3, ry + 3); a thread takes four consecutive rx (17 groups per row, the last one half empty).
    for (int qi = tid; qi < 17 * SCH; qi += 256) {
        const int ry = qi / 17, q = qi % 17;
        const int y = y0 - 1 + ry;
        const bool row_ok = y >= 3 && y < h - 3;
        uint32_t m0 = 0, m1 = 0, m2 = 0, u0 = 0, u1 = 0, d0_ = 0, d1_ = 0;
        if (row_ok) {
            const uint32_t* mp = reinterpret_cast<const uint32_t*>(&tile[(ry + 3) * TP + 4 * q]);
            m0 = mp[0]; m1 = mp[1]; m2 = mp[2];
            const uint32_t* up = reinterpret_cast<const uint32_t*>(&tile[(ry + 6) * TP + 4 * q]);   // row y + 3 (ring 0)
            const uint32_t* dn = reinterpret_cast<const uint32_t*>(&tile[(ry + 0) * TP + 4 * q]);   // row y - 3 (ring 8)
            u0 = up[0]; u1 = up[1]; d0_ = dn[0]; d1_ = dn[1];
        }
        // the four centres are bytes 3..6 of (m0, m1, m2); N / S are bytes 3..6 of the rows three above / below; E and W are the
        // centre row three bytes on.  Two pixels at a time in packed 16-bit arithmetic: a corner needs two neighbouring compass
        // points both darker (min of the pair of differences > t) or both brighter (max of the pair < -t).
        unsigned passbits = 0;
        if (row_ok) {
            const short2_t T1 = (short2_t){ (short)(threshold + 1), (short)(threshold + 1) }, T = (short2_t){ (short)threshold, (short)threshold };
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
                // two adjacent bytes j, j + 1 (0 <= j <= 6) of the 8 bytes hi:lo, zero-extended to 16 bits each (v_perm_b32:
                // selector 0..3 = bytes of lo, 4..7 = bytes of hi, 0x0c = zero)
                auto pair16 = [&](uint32_t hi, uint32_t lo, int j) -> short2_t {
                    const uint32_t sel = 0x0c000c00u | (uint32_t)j | ((uint32_t)(j + 1) << 16);
                    const uint32_t r = __builtin_amdgcn_perm(hi, lo, sel);
                    short2_t o; __builtin_memcpy(&o, &r, 4); return o;
                };
                // pixels 2 hp, 2 hp + 1 of the group: centre bytes 3 + 2 hp .., east 6 + 2 hp .., west 2 hp .. of (m0, m1, m2)
                const short2_t V = hp == 0 ? pair16(m1, m0, 3) : pair16(m2, m1, 1);
                const short2_t Ee = hp == 0 ? pair16(m2, m1, 2) : pair16(0u, m2, 0);
                const short2_t Ww = hp == 0 ? pair16(m1, m0, 0) : pair16(m1, m0, 2);
                const short2_t Nn = hp == 0 ? pair16(u1, u0, 3) : pair16(0u, u1, 1);
                const short2_t Ss = hp == 0 ? pair16(d1_, d0_, 3) : pair16(0u, d1_, 1);
                const short2_t dN = V - Nn, dE = V - Ee, dS = V - Ss, dW = V - Ww;
                const short2_t dark = pmax(pmax(pmin(dN, dE), pmin(dE, dS)), pmax(pmin(dS, dW), pmin(dW, dN)));
                const short2_t bright = pmin(pmin(pmax(dN, dE), pmax(dE, dS)), pmin(pmax(dS, dW), pmax(dW, dN)));
                const short2_t xd = dark - T1, xb = bright + T;                       // dark > t  <=>  xd >= 0 ; bright < -t  <=>  xb < 0
                uint32_t ud, ub;
                __builtin_memcpy(&ud, &xd, 4); __builtin_memcpy(&ub, &xb, 4);
                const uint32_t sg = (~ud | ub) & 0x80008000u;
                passbits |= ((sg >> 15) & 1u) << (2 * hp);
                passbits |= ((sg >> 31) & 1u) << (2 * hp + 1);
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int rx = 4 * q + b, x = x0 - 1 + rx;
            const bool pass = ((passbits >> b) & 1u) && rx < SCW && x >= 3 && x < w - 3;
            const unsigned long long bal = __ballot(pass);
            if (bal) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&n_cand, __popcll(bal));
                base = __shfl(base, 0, 64);
                if (pass) cand[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)(ry * SCP + rx);
            }
        }
    }
    __syncthreads();
    // Phase 2: exact score of the survivors
    const int nc = n_cand;
    for (int c = tid; c < nc; c += 256) {
        const int p = cand[c];
        const int ry = p / SCP, rx = p % SCP;
        sc[p + SC_OFF] = (uint8_t)fast_score_at(&tile[(ry + 3) * TP + rx + 3], TP, threshold);
    }
    __syncthreads();
    // Phase 3: strict 3x3 maximum among the raw scores + RunByImageBorder; one 32-bit store per quad of the kept map
    const int lo = border > 3 ? border : 3;
    uint8_t* K = kept + (size_t)f * wp * h;
    int mine = 0;
#pragma unroll
    for (int i = 0; i < FT_W * FT_H / 4 / 256; ++i) {
        const int qi = tid + 256 * i;
        const int ly = qi / (FT_W / 4), lq = qi % (FT_W / 4);
        const int y = y0 + ly, xq = x0 + 4 * lq;
        if (y >= h || xq >= wp) continue;
        const uint8_t* c0 = &sc[(ly + 1) * SCP + 4 * lq + 1 + SC_OFF];           // dword-aligned: the four scores in one LDS read
        const uint32_t raw4 = *reinterpret_cast<const uint32_t*>(c0);
        uint32_t kept4 = 0;
        if (raw4 != 0u && y >= lo && y < h - lo) {                              // most quads hold no corner at all
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int s = (int)((raw4 >> (8 * b)) & 0xffu);
                const int x = xq + b;
                if (s == 0 || x < lo || x >= w - lo) continue;
                const uint8_t* p = c0 + b;
                const bool keep = s > p[-1] && s > p[1] && s > p[-SCP - 1] && s > p[-SCP] && s > p[-SCP + 1] && s > p[SCP - 1] && s > p[SCP] && s > p[SCP + 1];
                if (keep) { kept4 |= (uint32_t)s << (8 * b); ++mine; atomicAdd(&lh[s], 1); }
            }
        }
        *reinterpret_cast<uint32_t*>(K + (size_t)y * wp + xq) = kept4;
        if (f == 0 && raw_frame0) *reinterpret_cast<uint32_t*>(raw_frame0 + (size_t)y * wp + xq) = raw4;
    }
    if (mine) atomicAdd(&n_kept, mine);
    __syncthreads();
    if (lh[tid]) atomicAdd(&hist[f * 256 + tid], lh[tid]);
    if (tid == 0 && n_kept) atomicAdd(&band_count[f * gridDim.y + blockIdx.y], n_kept);
}

// ---------------------------------------------------------------------------------------------
// NMS + border cull + raster-order compaction.  A workgroup owns NMS_ROWS full image rows (a contiguous
// raster segment); pass 1 counts (and builds the response histogram), pass 2 writes at the scanned offsets.
// A thread examines 4 adjacent pixels from nine 32-bit loads.
// ---------------------------------------------------------------------------------------------
// exclusive scan of the per-workgroup counts of one frame (single wavefront per frame; n_wg is small)
__global__ __launch_bounds__(64) void k_scan_counts(const int* __restrict__ wg_count, int n_wg, int* __restrict__ wg_off, int* __restrict__ n_raw)
{
    const int f = blockIdx.x, lane = threadIdx.x;
    int base = 0;
    for (int b = 0; b < n_wg; b += 64) {
        const int i = b + lane;
        int v = i < n_wg ? wg_count[f * n_wg + i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (i < n_wg) wg_off[f * n_wg + i] = base + incl - v;
        base += __shfl(incl, 63, 64);
    }
    if (lane == 0) n_raw[f] = base;
}

__global__ __launch_bounds__(256) void k_nms_emit(const uint8_t* __restrict__ score, int w, int h, int wp, int border, int rows_per_wg,
                                                  const int* __restrict__ wg_off, int2* __restrict__ raw, size_t raw_cap)   // score = the KEPT map of k_fast_nms
{
    __shared__ int wave_cnt[4];
    __shared__ int base_s;
    const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t* Sc = score + (size_t)f * wp * h;
    int2* out = raw + (size_t)f * raw_cap;
    if (tid == 0) base_s = wg_off[f * gridDim.x + blockIdx.x];
    __syncthreads();
    const int y0 = blockIdx.x * rows_per_wg;
    const int y1 = min(y0 + rows_per_wg, h);
    const int qpr = wp / 4;
    const int q_end = y1 * qpr;
    for (int qb = y0 * qpr; qb < q_end; qb += 1024) {
        const int q0 = qb + 4 * tid;                 // this thread's four consecutive quads (thread order = raster order)
        uint32_t c4[4] = { 0, 0, 0, 0 };
        if (q0 + 3 < q_end) { const uint4 v = *reinterpret_cast<const uint4*>(Sc + (size_t)q0 * 4); c4[0] = v.x; c4[1] = v.y; c4[2] = v.z; c4[3] = v.w; }
        else for (int j = 0; j < 4; ++j) if (q0 + j < q_end) c4[j] = *reinterpret_cast<const uint32_t*>(Sc + (size_t)(q0 + j) * 4);
        int m[4] = { 0, 0, 0, 0 }, resp[4][4];
        int mine = 0;
        if ((c4[0] | c4[1] | c4[2] | c4[3]) != 0u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (c4[j] == 0u) continue;
                // every non-zero byte of the kept map is a keypoint; its value is the response
#pragma unroll
                for (int b = 0; b < 4; ++b) { resp[j][b] = (int)((c4[j] >> (8 * b)) & 0xffu); if (resp[j][b]) m[j] |= 1 << b; }
                mine += __popc(m[j]);
            }
        }
        // exclusive prefix of `mine` over the wavefront
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wave_cnt[wave] = incl;
        __syncthreads();
        int off = base_s + incl - mine;
        for (int w2 = 0; w2 < wave; ++w2) off += wave_cnt[w2];
        if (mine) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!m[j]) continue;
                const int q = q0 + j, xq = 4 * (q % qpr), y = q / qpr;
#pragma unroll
                for (int b = 0; b < 4; ++b) if (m[j] & (1 << b)) out[off++] = make_int2((xq + b) | (y << 16), resp[j][b]);
            }
        }
        __syncthreads();
        if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// selection: one workgroup (1024 threads) per frame.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_scan_excl(int v, int* sh /* 17 ints */, int& total)
{
    // exclusive scan over 1024 threads (16 waves) in thread order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    __syncthreads();
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
    for (int q = 0; q < 16; ++q) { if (q < wave) woff += sh[q]; tot += sh[q]; }
    total = tot;
    return woff + incl - v;
}

__global__ __launch_bounds__(1024) void k_select(OrbSelectArgs a)
{
#pragma clang fp contract(off)          // float results feed comparisons that must match the CPU restatement: plain IEEE operations, never an FMA
    __shared__ int sh[32];
    __shared__ int lhist[256];
    __shared__ int s_cut;
    __shared__ int s_minX, s_maxX, s_minY, s_maxY, s_minS;
    const int f = blockIdx.x, tid = threadIdx.x;
    // Working set of the suppression (candidates, cell index, radii): in LDS whenever it fits -- the ring search chases
    // cell -> member -> candidate, three dependent reads per visited point, and is pure latency when they go to L2.
    constexpr int SEL_CAP = 2048, SEL_CELLS = 1024;
    __shared__ int2 l_cand[SEL_CAP];
    __shared__ int l_cellcnt[SEL_CELLS + 1], l_cellfill[SEL_CELLS + 1], l_cellmem[SEL_CAP], l_rad[SEL_CAP];
    __shared__ int s_M;
    const int2* raw = a.raw + (size_t)f * a.raw_cap;
    int2* cand = a.cand + (size_t)f * a.raw_cap;
    int* cellcnt = a.cell_start + (size_t)f * (a.ncells + 1);
    int* cellfill = a.cell_fill + (size_t)f * (a.ncells + 1);
    int* cellmem = a.cell_members + (size_t)f * a.raw_cap;
    int* rad = a.radius + (size_t)f * a.raw_cap;
    mage_keypoint* okp = a.out_kp + (size_t)f * a.capacity;
    const int n_raw = a.n_raw[f];
    const float size_f = (float)a.patch_size * 1.0f;
    if (n_raw <= a.nfeatures) {
        const int n = min(n_raw, a.capacity);
        for (int i = tid; i < n; i += 1024) {
            const int2 r = raw[i];
            mage_keypoint k = { (float)(r.x & 0xffff), (float)(r.x >> 16), size_f, 0.0f, (float)r.y, 0, -1 };
            okp[i] = k;
        }
        if (tid == 0) a.out_count[f] = n;
        return;
    }
    // ---- RetainBestFeatures: whole histogram bins from 255 downwards.  suffix[i] = number of responses >= i, by a parallel
    // scan (a single thread walking the 256 bins three times cost ~45 us of dependent LDS reads per frame); the two
    // thresholds of OpenCVModified.cpp:571-617 are then "largest bin whose suffix count reaches the quota".
    __shared__ int s_mnt;
    if (tid < 256) lhist[tid] = a.hist[f * 256 + tid];
    if (tid == 0) { s_mnt = -1; s_cut = -1; s_minX = 1 << 30; s_maxX = -1; s_minY = 1 << 30; s_maxY = -1; s_minS = 1 << 30; }
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                      // Hillis-Steele suffix sum over 256 bins
        int v = 0;
        if (tid < 256) v = lhist[tid] + (tid + o < 256 ? lhist[tid + o] : 0);
        __syncthreads();
        if (tid < 256) lhist[tid] = v;
        __syncthreads();
    }
    const int min_thr = a.fast_threshold;
    if (tid < 256 && tid >= min_thr && lhist[tid] >= a.nfeatures) atomicMax(&s_mnt, tid);
    __syncthreads();
    const int mnt = s_mnt >= 0 ? s_mnt : min_thr;
    const int lower = max((int)((float)mnt * a.feature_strength), min_thr);
    if (tid < 256 && tid >= lower && lhist[tid] >= a.max_num) atomicMax(&s_cut, tid);
    __syncthreads();
    if (tid == 0) {
        if (s_cut < 0) s_cut = lower;
        s_M = s_cut < 256 ? lhist[s_cut] : 0;                // how many candidates the compaction below will keep
    }
    __syncthreads();
    const int cut = s_cut;
    if (s_M <= SEL_CAP && a.ncells <= SEL_CELLS) { cand = l_cand; cellcnt = l_cellcnt; cellfill = l_cellfill; cellmem = l_cellmem; rad = l_rad; }
    int mbase = 0;
    for (int i0 = 0; i0 < n_raw; i0 += 1024) {          // ordered compaction (raster order is kept: choice C1)
        const int i = i0 + tid;
        int2 r = make_int2(0, -1);
        if (i < n_raw) r = raw[i];
        const int keep = (i < n_raw && r.y >= cut) ? 1 : 0;
        int tot;
        const int off = block_scan_excl(keep, sh, tot);
        if (keep) cand[mbase + off] = r;
        {   // bounding box and weakest response: reduce inside the wavefront first, one LDS atomic per wavefront and quantity
            int mnx = keep ? (r.x & 0xffff) : (1 << 30), mxx = keep ? (r.x & 0xffff) : -1;
            int mny = keep ? (r.x >> 16) : (1 << 30), mxy = keep ? (r.x >> 16) : -1, mns = keep ? r.y : (1 << 30);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                mnx = min(mnx, __shfl_xor(mnx, o, 64)); mxx = max(mxx, __shfl_xor(mxx, o, 64));
                mny = min(mny, __shfl_xor(mny, o, 64)); mxy = max(mxy, __shfl_xor(mxy, o, 64)); mns = min(mns, __shfl_xor(mns, o, 64));
            }
            if ((tid & 63) == 0 && mxx >= 0) {
                atomicMin(&s_minX, mnx); atomicMax(&s_maxX, mxx); atomicMin(&s_minY, mny); atomicMax(&s_maxY, mxy); atomicMin(&s_minS, mns);
            }
        }
        mbase += tot;
        __syncthreads();
    }
    const int M = mbase;
    const int N = a.nfeatures;
    __threadfence_block();
    __syncthreads();
    if (N > M) {                                         // ANMS returns early: keep all (cannot happen: M >= N by construction)
        const int n = min(M, a.capacity);
        for (int i = tid; i < n; i += 1024) {
            const int2 r = cand[i];
            mage_keypoint k = { (float)(r.x & 0xffff), (float)(r.x >> 16), size_f, 0.0f, (float)r.y, 0, -1 };
            okp[i] = k;
        }
        if (tid == 0) a.out_count[f] = n;
        return;
    }
    // The rest runs twice in the source: once on the LDS arrays (address space known to the compiler: ds_read / ds_write) and
    // once, for oversized inputs, on the global scratch.
    const bool in_lds = cand == l_cand;
    auto suppress_and_rank = [&](int2* cand, int* cellcnt, int* cellfill, int* cellmem, int* rad) {
        // ---- AdaptiveNonMaximalSuppresion
        const int minX = s_minX, maxX = s_maxX, minY = s_minY, maxY = s_maxY;
        const int numX = a.cells_x, numY = a.cells_y, thr = a.fast_threshold;
        float rf;
        {
            const float hi = (float)a.strong_response - (float)thr;
            float val = (float)s_minS - (float)thr;
            val = val < 0.0f ? 0.0f : (val > hi ? hi : val);
            float range = a.max_robust - a.min_robust;
            if (range < 0.0f) range = 0.0f;
            rf = a.max_robust - (val / (float)(a.strong_response - thr)) * range;
        }
        for (int c = tid; c <= a.ncells; c += 1024) { cellcnt[c] = 0; cellfill[c] = 0; }
        __syncthreads();
        for (int i = tid; i < M; i += 1024) {
            const int2 r = cand[i];
            const int cx = ((r.x & 0xffff) - minX) * numX / (maxX + 1 - minX), cy = ((r.x >> 16) - minY) * numY / (maxY + 1 - minY);
            atomicAdd(&cellcnt[cy * numX + cx + 1], 1);
        }
        __syncthreads();
        {   // cell_start = inclusive scan of the counts, in place: thread t owns cells [t*per, (t+1)*per)
            const int per = (a.ncells + 1023) / 1024;
            const int c0 = tid * per, c1 = min(c0 + per, a.ncells);
            int local = 0;
            for (int c = c0; c < c1; ++c) local += cellcnt[c + 1];
            int tot;
            int run = block_scan_excl(local, sh, tot);
            for (int c = c0; c < c1; ++c) { run += cellcnt[c + 1]; cellcnt[c + 1] = run; }
        }
        __threadfence_block();
        __syncthreads();
        for (int i = tid; i < M; i += 1024) {
            const int2 r = cand[i];
            const int cx = ((r.x & 0xffff) - minX) * numX / (maxX + 1 - minX), cy = ((r.x >> 16) - minY) * numY / (maxY + 1 - minY);
            const int c = cy * numX + cx;
            cellmem[cellcnt[c] + atomicAdd(&cellfill[c], 1)] = i;     // member order inside a cell does not affect a minimum
        }
        __threadfence_block();
        __syncthreads();
        const int globalMaxR2 = (int)(((double)(maxX - minX)) * ((double)(maxY - minY)) / (double)N);
        int minCellDelta2;
        {
            const int dx = max((maxX - minX) / numX, 1), dy = max((maxY - minY) / numY, 1);
            const int m = min(dx, dy);
            minCellDelta2 = m * m;
        }
        for (int i = tid; i < M; i += 1024) {
            const int2 r = cand[i];
            const int x = r.x & 0xffff, y = r.x >> 16;
            const int cx = (x - minX) * numX / (maxX + 1 - minX), cy = (y - minY) * numY / (maxY + 1 - minY);
            const float strength = (float)r.y;
            const float s = strength * rf + 0.002f;                              // strength >= 0 always (FAST score); no FMA: contraction is off in this kernel
            int minR2 = globalMaxR2;
            for (int d = 0; max(0, d - 1) * max(0, d - 1) * minCellDelta2 < minR2; ++d)
                for (int yy = -d; yy <= d; ++yy) {
                    const int cYY = yy + cy;
                    if (cYY < 0 || cYY >= numY) continue;
                    for (int xx = -d; xx <= d; ++xx) {
                        const int cXX = xx + cx;
                        if (cXX < 0 || cXX >= numX || max(abs(xx), abs(yy)) != d) continue;
                        const int c = cYY * numX + cXX;
                        for (int q = cellcnt[c]; q < cellcnt[c + 1]; ++q) {
                            const int2 o = cand[cellmem[q]];
                            if ((float)o.y > s) {
                                const int ddx = x - (o.x & 0xffff), ddy = y - (o.x >> 16);
                                const int rr = ddx * ddx + ddy * ddy;
                                if (rr < minR2) minR2 = rr;
                            }
                        }
                    }
                }
            rad[i] = minR2;
        }
        __threadfence_block();
        __syncthreads();
        // ---- keep the N first in the total order (radius desc, strength desc, index asc); rank = output position (choice C2)
        // The comparator is a lexicographic order on (radius desc, strength desc, index asc): pack it into one 64-bit key per
        // candidate (LDS path) so that the O(M^2) rank is one wide LDS read and one compare per pair, 8 pairs in flight.
        __shared__ unsigned long long l_key[SEL_CAP];
        if (in_lds) {
            for (int i = tid; i < M; i += 1024)
                l_key[i] = ((unsigned long long)(unsigned)rad[i] << 32) | ((unsigned long long)(unsigned)l_cand[i].y << 16) | (unsigned long long)(0xFFFF - i);
            __syncthreads();
        }
        for (int i = tid; i < M; i += 1024) {
            int rank = 0;
            if (in_lds) {
                const unsigned long long ki = l_key[i];
                int j = 0;
                for (; j + 8 <= M; j += 8) {
    #pragma unroll
                    for (int u = 0; u < 8; ++u) rank += l_key[j + u] > ki;
                }
                for (; j < M; ++j) rank += l_key[j] > ki;
            } else {
                const int ri = rad[i], si = cand[i].y;
                for (int j = 0; j < M; ++j) {
                    const int rj = rad[j], sj = cand[j].y;
                    rank += (rj > ri) || (rj == ri && (sj > si || (sj == si && j < i)));
                }
            }
            if (rank < N && rank < a.capacity) {
                const int2 r = cand[i];
                mage_keypoint k = { (float)(r.x & 0xffff), (float)(r.x >> 16), size_f, 0.0f, (float)r.y, 0, -1 };
                okp[rank] = k;
            }
        }

    };
    if (in_lds) suppress_and_rank(l_cand, l_cellcnt, l_cellfill, l_cellmem, l_rad);
    else suppress_and_rank(cand, cellcnt, cellfill, cellmem, rad);
    if (tid == 0) a.out_count[f] = min(N, a.capacity);
}

// ---------------------------------------------------------------------------------------------
// separable integer Gaussian, REFLECT_101, 64x16 output tile per workgroup
// ---------------------------------------------------------------------------------------------
constexpr int BT_W = 64, BT_H = 64, MAXR = 7;

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

__global__ __launch_bounds__(256) void k_blur(const uint8_t* __restrict__ img, int w, int h, int stride, size_t frame_stride,
                                              OrbTaps taps, uint8_t* __restrict__ out, int wp)
{
    constexpr int SW = BT_W + 16, SH = BT_H + 2 * MAXR, SP = SW;       // window starts 8 pixels left of the tile (aligned), radius <= 7 used
    __shared__ __attribute__((aligned(4))) uint8_t src[SH * SP];
    __shared__ __attribute__((aligned(16))) int hrow[SH * BT_W];
    const int f = blockIdx.z, r = taps.radius;
    const uint8_t* I = img + (size_t)f * frame_stride;
    uint8_t* O = out + (size_t)f * wp * h;
    const int x0 = blockIdx.x * BT_W, y0 = blockIdx.y * BT_H;
    const int sh = BT_H + 2 * r;
    stage_window<SW, SH, SP, true>(src, I, w, h, stride, x0 - 8, y0 - r, SW, sh);
    __syncthreads();
    // taps into registers; products fit 24 bits (tap <= 256, pixel <= 255, row sum < 2^17), so the full-rate 24-bit
    // multiply-add is exact
    int tp[2 * MAXR + 1];
#pragma unroll
    for (int t = 0; t < 2 * MAXR + 1; ++t) tp[t] = t <= 2 * r ? taps.t[t] : 0;
    // horizontal pass for every staged row
    if (r == 3 && taps.t[3] < 256) {
        // 7 taps (all <= 255): four outputs per thread from three aligned 32-bit LDS reads; each output is two 4-way byte dot
        // products (v_dot4_u32_u8) over windows cut out of the 12 bytes with v_alignbyte -- 17 operations for 4 pixels
        // where the byte-at-a-time form issued 28 LDS reads and 56 multiply-adds.
        const uint32_t T0 = (uint32_t)tp[0] | ((uint32_t)tp[1] << 8) | ((uint32_t)tp[2] << 16) | ((uint32_t)tp[3] << 24);
        const uint32_t T1 = (uint32_t)tp[4] | ((uint32_t)tp[5] << 8) | ((uint32_t)tp[6] << 16);
        for (int e = threadIdx.x; e < sh * (BT_W / 4); e += 256) {
            const int ty = e / (BT_W / 4), tq = e % (BT_W / 4);
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(&src[ty * SP + 4 + 4 * tq]);      // bytes o .. o+11, o = 4 + 4 tq; output j starts at o + 1 + j
            const uint32_t A = sp[0], B = sp[1], C = sp[2];
            int4 o;
            o.x = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 1), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 1), T0, 0u, false), false);
            o.y = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 2), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 2), T0, 0u, false), false);
            o.z = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 3), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 3), T0, 0u, false), false);
            o.w = (int)__builtin_amdgcn_udot4(C, T1, __builtin_amdgcn_udot4(B, T0, 0u, false), false);
            *reinterpret_cast<int4*>(&hrow[ty * BT_W + 4 * tq]) = o;
        }
    } else {
        for (int e = threadIdx.x; e < sh * BT_W; e += 256) {
            const int ty = e / BT_W, tx = e % BT_W;
            const uint8_t* sp = &src[ty * SP + 8 - r + tx];
            int acc = 0;
            for (int t = 0; t <= 2 * r; ++t) acc = (__mul24(taps.t[t], (int)sp[t]) + acc);
            hrow[ty * BT_W + tx] = acc;
        }
    }
    __syncthreads();
    if (r == 3) {
        // vertical pass, 7 taps: a thread owns four pixel columns and four consecutive rows, reads the ten row sums they need once
        // (10 LDS reads for 4 outputs where the row-at-a-time form below issues 28) and packs one 32-bit store per row
        static_assert(BT_H * (BT_W / 4) == 4 * 256, "one 4 x 4 pixel patch per thread");
        const int tq = threadIdx.x % (BT_W / 4), strip = threadIdx.x / (BT_W / 4);
        const int xq = x0 + 4 * tq;
        if (xq < wp) {
            int4 hv[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) hv[i] = *reinterpret_cast<const int4*>(&hrow[(strip * 4 + i) * BT_W + 4 * tq]);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int y = y0 + strip * 4 + o;
                if (y >= h) break;
                int acc4[4] = { 0, 0, 0, 0 };
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    acc4[0] = (__mul24(tp[t], hv[o + t].x) + acc4[0]); acc4[1] = (__mul24(tp[t], hv[o + t].y) + acc4[1]);
                    acc4[2] = (__mul24(tp[t], hv[o + t].z) + acc4[2]); acc4[3] = (__mul24(tp[t], hv[o + t].w) + acc4[3]);
                }
                uint32_t packed = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (xq + b >= w) continue;
                    const int v = (acc4[b] + (1 << 15)) >> 16;
                    packed |= (uint32_t)(v > 255 ? 255 : v) << (8 * b);
                }
                *reinterpret_cast<uint32_t*>(O + (size_t)y * wp + xq) = packed;
            }
        }
        return;
    }
    // vertical pass, 4 pixels per thread, one 32-bit store
    for (int e = threadIdx.x; e < BT_H * (BT_W / 4); e += 256) {
        const int ty = e / (BT_W / 4), tq = e % (BT_W / 4);
        const int xq = x0 + 4 * tq, y = y0 + ty;
        if (xq >= wp || y >= h) continue;
        int acc4[4] = { 0, 0, 0, 0 };
        if (r == 3) {
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                const int4 hv = *reinterpret_cast<const int4*>(&hrow[(ty + t) * BT_W + 4 * tq]);
                acc4[0] = (__mul24(tp[t], hv.x) + acc4[0]); acc4[1] = (__mul24(tp[t], hv.y) + acc4[1]);
                acc4[2] = (__mul24(tp[t], hv.z) + acc4[2]); acc4[3] = (__mul24(tp[t], hv.w) + acc4[3]);
            }
        } else {
            for (int t = 0; t <= 2 * r; ++t) {
                const int4 hv = *reinterpret_cast<const int4*>(&hrow[(ty + t) * BT_W + 4 * tq]);
                acc4[0] = (__mul24(taps.t[t], hv.x) + acc4[0]); acc4[1] = (__mul24(taps.t[t], hv.y) + acc4[1]);
                acc4[2] = (__mul24(taps.t[t], hv.z) + acc4[2]); acc4[3] = (__mul24(taps.t[t], hv.w) + acc4[3]);
            }
        }
        uint32_t packed = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (xq + b >= w) continue;
            const int v = (acc4[b] + (1 << 15)) >> 16;
            packed |= (uint32_t)(v > 255 ? 255 : v) << (8 * b);
        }
        *reinterpret_cast<uint32_t*>(O + (size_t)y * wp + xq) = packed;
    }
}

// plain copy when gaussian_kernel_size <= 1
__global__ void k_copy_image(const uint8_t* __restrict__ img, int w, int h, int stride, size_t frame_stride, uint8_t* __restrict__ out, int wp)
{
    const int f = blockIdx.y;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < wp * h; p += gridDim.x * blockDim.x) {
        const int x = p % wp, y = p / wp;
        out[(size_t)f * wp * h + p] = x < w ? img[(size_t)f * frame_stride + (size_t)y * stride + x] : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Pyramid levels >= 1: cv::resize(prev, level, INTER_LINEAR) on CV_8UC1 as OpenCV 3.4.0 computes it (resize.cpp, HResizeLinear /
// VResizeLinear<uchar, int, short>): source position and weights in float from a double scale, weights rounded to 11-bit fixed
// point, horizontal pass in int, vertical pass (((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2.  Four output pixels per
// thread (one 32-bit store); contraction off, the position arithmetic must round like the CPU restatement.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int sat_short_dev(float v)
{
    const int r = (int)rintf(v);
    return r < -32768 ? -32768 : (r > 32767 ? 32767 : r);
}

__global__ __launch_bounds__(64) void k_resize_linear(const uint8_t* __restrict__ src, int sw, int sh, int sstride, size_t sframe,
                                                      uint8_t* __restrict__ dst, int dw, int dh, int dpitch, size_t dframe)
{
#pragma clang fp contract(off)
    const int f = blockIdx.z, dy = blockIdx.y, q = blockIdx.x * 64 + threadIdx.x;
    if (4 * q >= dpitch) return;
    const uint8_t* S = src + (size_t)f * sframe;
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int b0 = sat_short_dev((1.f - fy) * 2048), b1 = sat_short_dev(fy * 2048);
    const int r0 = min(max(sy, 0), sh - 1), r1 = min(max(sy + 1, 0), sh - 1);
    const uint8_t* S0r = S + (size_t)r0 * sstride; const uint8_t* S1r = S + (size_t)r1 * sstride;
    uint32_t packed = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int dx = 4 * q + b;
        if (dx >= dw) continue;
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        const int a0 = sat_short_dev((1.f - fx) * 2048), a1 = sat_short_dev(fx * 2048);
        const int sx1 = min(sx + 1, sw - 1);
        const int H0 = (int)S0r[sx] * a0 + (int)S0r[sx1] * a1, H1 = (int)S1r[sx] * a0 + (int)S1r[sx1] * a1;
        const int v = (((b0 * (H0 >> 4)) >> 16) + ((b1 * (H1 >> 4)) >> 16) + 2) >> 2;
        packed |= (uint32_t)(v & 0xff) << (8 * b);
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)f * dframe + (size_t)dy * dpitch + 4 * q) = packed;
}

// ImageData::Insert of one level's keypoints behind those of the coarser... finer levels (ImageData.h:65-70, OpenCVModified.cpp:731-737):
// copy what still fits, coordinates scaled to the full-resolution frame (:756-760), octave and size stamped (:713-717).
__global__ __launch_bounds__(256) void k_append_level(const mage_keypoint* __restrict__ kp_l, const uint8_t* __restrict__ desc_l, const int* __restrict__ count_l,
                                                      int cap_l, mage_keypoint* __restrict__ kp, uint8_t* __restrict__ desc, int* __restrict__ count, int capacity,
                                                      float scale, float size, int level)
{
    const int f = blockIdx.x, tid = threadIdx.x;
    const int n = min(count_l[f], cap_l), base = count[f];
    const int take = min(n, max(capacity - base, 0));
    for (int i = tid; i < take; i += 256) {
        mage_keypoint k = kp_l[(size_t)f * cap_l + i];
        k.x = k.x * scale; k.y = k.y * scale; k.size = size; k.octave = level;
        kp[(size_t)f * capacity + base + i] = k;
    }
    const uint32_t* dsrc = reinterpret_cast<const uint32_t*>(desc_l + (size_t)f * cap_l * 32);
    uint32_t* ddst = reinterpret_cast<uint32_t*>(desc + ((size_t)f * capacity + base) * 32);
    for (int i = tid; i < take * 8; i += 256) ddst[i] = dsrc[i];
    __syncthreads();
    if (tid == 0) count[f] = base + take;
}

// ---------------------------------------------------------------------------------------------
// ICAngles (OpenCVModified.cpp:399-437), only with UseOrientation: first moments of the UNBLURRED image over the discretised
// disc of radius half (row v spans |u| <= umax[v]); one wavefront per keypoint, lanes over the (2 half + 1)^2 window, exact
// integer sums (order-free), then cv::fastAtan2 in float32 with contraction off.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
#pragma clang fp contract(off)          // plain IEEE operations in source order: bit-identical to the CPU restatement (tools/atan_check.hip)
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale, p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    const float eps = (float)2.2204460492503131e-16;
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + eps); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + eps); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

__global__ __launch_bounds__(256) void k_ic_angles(const uint8_t* __restrict__ img, int stride, size_t frame_stride, mage_keypoint* __restrict__ kps,
                                                   const int* __restrict__ counts, int capacity, OrbUmax um)
{
    const int f = blockIdx.y;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= counts[f]) return;
    mage_keypoint* kp = kps + (size_t)f * capacity + k;
    const int cx = (int)rintf(kp->x), cy = (int)rintf(kp->y);
    const uint8_t* c = img + (size_t)f * frame_stride + (size_t)cy * stride + cx;
    const int half = um.half, side = 2 * half + 1;
    int m01 = 0, m10 = 0;
    for (int e = lane; e < side * side; e += 64) {
        const int v = e / side - half, u = e % side - half;
        if (abs(u) <= um.umax[abs(v)]) {
            const int val = c[v * stride + u];
            m10 += u * val; m01 += v * val;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { m01 += __shfl_xor(m01, o, 64); m10 += __shfl_xor(m10, o, 64); }
    if (lane == 0) kp->angle = fast_atan2_deg((float)m01, (float)m10);
}

// ---------------------------------------------------------------------------------------------
// BRIEF-256: one wavefront per keypoint; lane l evaluates pairs 4l..4l+3, two lanes make a byte.
// ---------------------------------------------------------------------------------------------
constexpr int BRIEF_KPW = 4;      // keypoints per wavefront: their 32 pixel loads are in flight together, the pattern row is fetched once
// ComputeOrbDescriptors (OpenCVModified.cpp:452-492): patch sizes without a pre-rotated table together with UseOrientation --
// every keypoint rotates the 512 random points by its own angle: a = (float)cos(angle), b = (float)sin(angle) (taken in double and
// rounded to float: the pinned choice of oracle/orb_oracle.c), x' = cvRound(x a - y b), y' = cvRound(x b + y a) in float, no FMA.
// One wavefront per keypoint, four pairs per lane as in k_brief.
__global__ __launch_bounds__(256) void k_brief_rotated(const uint8_t* __restrict__ blurred, int wp, int h, const mage_keypoint* __restrict__ kps,
                                                       const int* __restrict__ counts, int capacity, const signed char* __restrict__ pattern,
                                                       uint8_t* __restrict__ desc)
{
#pragma clang fp contract(off)
    const int f = blockIdx.y, k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= counts[f]) return;
    const mage_keypoint kp = kps[(size_t)f * capacity + k];
    const uint8_t* c = blurred + (size_t)f * wp * h + (size_t)((int)rintf(kp.y)) * wp + (int)rintf(kp.x);
    float ang = kp.angle;
    ang = ang * (float)(3.14159265358979323846 / 180.0);
    const float a = (float)cos((double)ang), b = (float)sin((double)ang);
    const int4 pr = *reinterpret_cast<const int4*>(pattern + lane * 16);
    const int w4[4] = { pr.x, pr.y, pr.z, pr.w };
    int nib = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = (float)(int)(signed char)(w4[q] & 0xff), y0 = (float)(int)(signed char)((w4[q] >> 8) & 0xff);
        const float x1 = (float)(int)(signed char)((w4[q] >> 16) & 0xff), y1 = (float)(int)(signed char)((w4[q] >> 24) & 0xff);
        const float rx0 = x0 * a - y0 * b, ry0 = x0 * b + y0 * a, rx1 = x1 * a - y1 * b, ry1 = x1 * b + y1 * a;
        const int t0 = c[(int)rintf(ry0) * wp + (int)rintf(rx0)];
        const int t1 = c[(int)rintf(ry1) * wp + (int)rintf(rx1)];
        nib |= (t0 < t1) << q;
    }
    const int hi = __shfl_down(nib, 1, 64);
    if ((lane & 1) == 0) desc[((size_t)f * capacity + k) * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
}

__global__ __launch_bounds__(256) void k_brief(const uint8_t* __restrict__ blurred, int wp, int h, const mage_keypoint* __restrict__ kps,
                                               const int* __restrict__ counts, int capacity, const signed char* __restrict__ pattern,
                                               uint8_t* __restrict__ desc)
{
    const int f = blockIdx.y;
    const int k0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * BRIEF_KPW, lane = threadIdx.x & 63;
    const int n = counts[f];
    if (k0 >= n) return;
    const uint8_t* frame = blurred + (size_t)f * wp * h;
    // cvRound of an integer-valued float; angleIncrement = cvRound(angle / 12) % 30 picks the pre-rotated table row
    // (OpenCVModified.cpp:523-532; angle is 0 unless UseOrientation)
    const uint8_t* c[BRIEF_KPW];
    int inc[BRIEF_KPW];
#pragma unroll
    for (int q = 0; q < BRIEF_KPW; ++q) {
        const int k = min(k0 + q, n - 1);
        const mage_keypoint kp = kps[(size_t)f * capacity + k];
        c[q] = frame + (size_t)((int)rintf(kp.y)) * wp + (int)rintf(kp.x);
        inc[q] = (int)rintf(kp.angle / 12.0f) % 30;
    }
    // this lane's four pairs of the table row: 16 signed bytes, one 128-bit load (rows are 1024 bytes, lane * 16 is aligned)
    int4 pr = *reinterpret_cast<const int4*>(pattern + inc[0] * 1024 + lane * 16);
    int t[BRIEF_KPW][8];
#pragma unroll
    for (int q = 0; q < BRIEF_KPW; ++q) {
        if (q > 0 && inc[q] != inc[q - 1]) pr = *reinterpret_cast<const int4*>(pattern + inc[q] * 1024 + lane * 16);
        const int w4[4] = { pr.x, pr.y, pr.z, pr.w };
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int x0 = (int)(signed char)(w4[b] & 0xff), y0 = (int)(signed char)((w4[b] >> 8) & 0xff);
            const int x1 = (int)(signed char)((w4[b] >> 16) & 0xff), y1 = (int)(signed char)((w4[b] >> 24) & 0xff);
            t[q][2 * b] = c[q][y0 * wp + x0];
            t[q][2 * b + 1] = c[q][y1 * wp + x1];
        }
    }
#pragma unroll
    for (int q = 0; q < BRIEF_KPW; ++q) {
        int nib = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) nib |= (t[q][2 * b] < t[q][2 * b + 1]) << b;
        const int hi = __shfl_down(nib, 1, 64);
        if ((lane & 1) == 0 && k0 + q < n) desc[((size_t)f * capacity + k0 + q) * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
    }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

void orb_launch_fast(const uint8_t* img, int w, int h, int stride, size_t frame_stride, int n_frames, int threshold, int border, uint8_t* kept,
                     uint8_t* raw_frame0, int wp, int* hist, int* band_count, int n_bands, hipStream_t st)
{
    // one tile row of k_fast_nms = one band of the emit pass: FT_H rows (orb_host.hip sizes n_bands with the same constant)
    (void)hipMemsetAsync(hist, 0, sizeof(int) * 256 * (size_t)n_frames, st);
    (void)hipMemsetAsync(band_count, 0, sizeof(int) * (size_t)n_bands * (size_t)n_frames, st);
    hipLaunchKernelGGL(k_fast_nms, dim3(cdiv(wp, FT_W), cdiv(h, FT_H), n_frames), dim3(256), 0, st, img, w, h, stride, frame_stride, threshold, border,
                       kept, raw_frame0, wp, hist, band_count);
}

void orb_launch_collect(const uint8_t* kept, int w, int h, int wp, int n_frames, int border, int rows_per_wg, int n_wg, int* wg_count, int* wg_off,
                        int* n_raw, int2* raw, size_t raw_cap, hipStream_t st)
{
    hipLaunchKernelGGL(k_scan_counts, dim3(n_frames), dim3(64), 0, st, wg_count, n_wg, wg_off, n_raw);
    hipLaunchKernelGGL(k_nms_emit, dim3(n_wg, n_frames), dim3(256), 0, st, kept, w, h, wp, border, rows_per_wg, wg_off, raw, raw_cap);
}

void orb_launch_select(const OrbSelectArgs& a, int n_frames, hipStream_t st)
{
    hipLaunchKernelGGL(k_select, dim3(n_frames), dim3(1024), 0, st, a);
}

void orb_launch_blur(const uint8_t* img, int w, int h, int stride, size_t frame_stride, int n_frames, const OrbTaps& taps, uint8_t* out, int wp, hipStream_t st)
{
    if (taps.radius == 0) hipLaunchKernelGGL(k_copy_image, dim3(cdiv(wp * h, 256 * 8), n_frames), dim3(256), 0, st, img, w, h, stride, frame_stride, out, wp);
    else hipLaunchKernelGGL(k_blur, dim3(cdiv(wp, BT_W), cdiv(h, BT_H), n_frames), dim3(256), 0, st, img, w, h, stride, frame_stride, taps, out, wp);
}

void orb_launch_resize(const uint8_t* src, int sw, int sh, int sstride, size_t sframe, uint8_t* dst, int dw, int dh, int dpitch, size_t dframe, int n_frames,
                       hipStream_t st)
{
    hipLaunchKernelGGL(k_resize_linear, dim3(cdiv(dpitch / 4, 64), dh, n_frames), dim3(64), 0, st, src, sw, sh, sstride, sframe, dst, dw, dh, dpitch, dframe);
}

void orb_launch_append_level(const mage_keypoint* kp_l, const uint8_t* desc_l, const int* count_l, int cap_l, mage_keypoint* kp, uint8_t* desc, int* count,
                             int capacity, int n_frames, float scale, float size, int level, hipStream_t st)
{
    hipLaunchKernelGGL(k_append_level, dim3(n_frames), dim3(256), 0, st, kp_l, desc_l, count_l, cap_l, kp, desc, count, capacity, scale, size, level);
}

void orb_launch_angles(const uint8_t* img, int stride, size_t frame_stride, int n_frames, mage_keypoint* kps, const int* counts, int capacity,
                       const OrbUmax& um, hipStream_t st)
{
    hipLaunchKernelGGL(k_ic_angles, dim3(cdiv(capacity, 4), n_frames), dim3(256), 0, st, img, stride, frame_stride, kps, counts, capacity, um);
}

void orb_launch_brief(const uint8_t* blurred, int wp, int h, int n_frames, const mage_keypoint* kps, const int* counts, int capacity,
                      const signed char* pattern, uint8_t* desc, bool rotate_random, hipStream_t st)
{
    if (rotate_random) hipLaunchKernelGGL(k_brief_rotated, dim3(cdiv(capacity, 4), n_frames), dim3(256), 0, st, blurred, wp, h, kps, counts, capacity, pattern, desc);
    else hipLaunchKernelGGL(k_brief, dim3(cdiv(capacity, 4 * BRIEF_KPW), n_frames), dim3(256), 0, st, blurred, wp, h, kps, counts, capacity, pattern, desc);
}

}  // namespace mage
