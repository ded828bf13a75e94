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
// Development probe (tools/orb_phase_probe.py builds a second library with -DMAGE_ORB_CLOCKS): shader-clock time per kernel
// phase, summed over workgroups by thread 0.  Compiled out of the product library.
#ifdef MAGE_ORB_CLOCKS
__device__ unsigned long long g_orb_clk[32];
// one workgroup in 64 is sampled, so the probe's own atomics do not queue up behind each other
#define ORB_CLK_BEGIN() const bool clk_on = threadIdx.x == 0 && ((blockIdx.x + 7 * blockIdx.y + 13 * blockIdx.z) & 63) == 0; \
    unsigned long long clk_prev = clk_on ? __builtin_amdgcn_s_memtime() : 0ull; if (clk_on) atomicAdd(&g_orb_clk[31], 1ull)
#define ORB_CLK(i) do { if (clk_on) { const unsigned long long clk_now = __builtin_amdgcn_s_memtime(); atomicAdd(&g_orb_clk[i], clk_now - clk_prev); clk_prev = clk_now; } } while (0)
#define ORB_CLK_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define ORB_CLK_WAIT() do { } while (0)
#define ORB_CLK_BEGIN() do { } while (0)
#define ORB_CLK(i) do { } while (0)
#endif

// Development only (tools/orb_ablate.py): bit 0 runs the compass arithmetic twice, bit 1 the even-position test, bit 2 the exact
// score, bit 3 launches the kernel without the blur, bit 4 takes the blur's tap matrices from the lane number instead of
// memory, bit 5 leaves its stores out, bit 10 leaves the 3 x 3 maximum out, bit 12 phases 2a / 2b (empty lists), bit 13 the
// append of phase 1 -- what a phase costs where it stands.  0 in the product.
#ifndef MAGE_ORB_ABLATE
#define MAGE_ORB_ABLATE 0
#endif
#define ORB_LAUNDER(x) asm volatile("" : "+v"(x))

constexpr int FT_W = 64, FT_H = 24;   // output tile of k_fast_keypoints: 640 x 480 = 10 x 20 workgroups per frame

// The ring differences fit 16 bits, so the score runs on PACKED pairs: register k holds (d[k], d[k + 8]) -- a ring pixel and its
// opposite.  A rotation of the ring by one position is "next register", and crossing position 7 -> 8 is a swap of the halves,
// so every windowed minimum / maximum (v_pk_min_i16 / v_pk_max_i16) serves two ring positions at once: ~100 operations where
// the 32-bit form needs ~180.
typedef short short2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ short2_t swap16(short2_t v) { return __builtin_shufflevector(v, v, 1, 0); }
__device__ __forceinline__ short2_t pmin(short2_t a, short2_t b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ short2_t pmax(short2_t a, short2_t b) { return __builtin_elementwise_max(a, b); }

// Second screen, on the EVEN ring positions (the compass points and the four diagonals): 9 contiguous ring pixels contain at
// least 4 consecutive even positions, so a corner needs 4 consecutive even pixels all darker than the centre by more than t, or
// all brighter.  Nine byte reads; ~11 % of the pixels of a textured frame pass (the opposite-pair test on all 16 pixels passes
// ~8 % but costs twice as much on the ~21 % it is run on).  Register j holds even positions j and j + 4 (a ring pixel and its
// opposite) in its 16-bit halves, the comparisons are the guard-bit subtractions of phase 1 (flag = bit 15 of each half), and "four
// consecutive flags" is bitwise logic on the registers and their half-swapped copies: window e (low half) and e + 4 (high half)
// of { D0 D1 D2 D3 }, { D1 D2 D3 S0 }, { D2 D3 S0 S1 }, { D3 S0 S1 S2 }.  2-cycle operations but for the three swaps, where the
// packed min / max form took ~45 4-cycle ones.
// Returns bit 0 = a darker run exists, bit 1 = a brighter run exists (0 = not a corner).
__device__ __forceinline__ int fast_even4_test(const uint8_t* __restrict__ c, int TP, uint32_t kd, uint32_t kb)
{
    // even positions 0, 2, 4, 6 and their opposites 8, 10, 12, 14: (0, 3) (2, 2) (3, 0) (2, -2) / (0, -3) (-2, -2) (-3, 0) (-2, 2)
    const int ex[4] = { 0, 2, 3, 2 }, ey[4] = { 3, 2, 0, -2 };
    const uint32_t v2 = (uint32_t)c[0] * 0x00010001u, Vd = v2 + kd, Vb = v2 + kb;
    uint32_t D[4], B[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t P = (uint32_t)c[ey[j] * TP + ex[j]] | ((uint32_t)c[-ey[j] * TP - ex[j]] << 16);
        D[j] = Vd - P;           // flag: darker than v - t
        B[j] = Vb - P;           // flag: NOT brighter than v + t
    }
    auto swap_halves = [](uint32_t x) -> uint32_t { return __builtin_amdgcn_alignbit(x, x, 16); };
    const uint32_t S0 = swap_halves(D[0]), S1 = swap_halves(D[1]), S2 = swap_halves(D[2]);
    const uint32_t d23 = D[2] & D[3], s01 = S0 & S1;
    const uint32_t dark = (D[0] & D[1] & d23) | (D[1] & d23 & S0) | (d23 & s01) | (D[3] & s01 & S2);
    const uint32_t T0 = swap_halves(B[0]), T1 = swap_halves(B[1]), T2 = swap_halves(B[2]);
    const uint32_t b23 = B[2] | B[3], t01 = T0 | T1;
    const uint32_t not_bright = (B[0] | B[1] | b23) & (B[1] | b23 | T0) & (b23 | t01) & (B[3] | t01 | T2);   // a window is all brighter iff none of its flags is set
    return ((dark & 0x80008000u) ? 1 : 0) | ((~not_bright & 0x80008000u) ? 2 : 0);
}

// The corner score for ONE polarity: D[k] = v - ring[k] (darker ring) or ring[k] - v (brighter ring), m = max over the 16 arcs of 9
// contiguous ring pixels of the arc's minimum; the pixel is a corner of that polarity iff m > t, with score m - 1 (identical to the
// reference's threshold-table pre-test + min/max ladder, which computes max(m_darker, m_brighter)).  A pixel cannot hold a darker
// and a brighter 9-arc at once (18 > 16 ring pixels), so only the polarity the even-position screen left open can exceed the
// threshold: one min ladder on packed pairs (register k = ring pixel k and its opposite; a rotation of the ring is "next
// register", crossing position 7 -> 8 a swap of the halves) instead of a min and a max ladder.
__device__ __forceinline__ int fast_score_polar(const uint8_t* __restrict__ c, int TP, bool brighter)
{
    const int v = c[0];
    const int ox[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
    const int oy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };
    short2_t P[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int a = (int)c[oy[k] * TP + ox[k]], b = (int)c[oy[k + 8] * TP + ox[k + 8]];
        const uint32_t ring = (uint32_t)a | ((uint32_t)b << 16), cen = (uint32_t)v * 0x00010001u;
        const uint32_t lhs = brighter ? ring : cen, rhs = brighter ? cen : ring;
        short2_t L, R; __builtin_memcpy(&L, &lhs, 4); __builtin_memcpy(&R, &rhs, 4);
        P[k] = L - R;
    }
    short2_t A2[8], A4[8], A8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) A2[k] = pmin(P[k], k < 7 ? P[k + 1] : swap16(P[0]));
#pragma unroll
    for (int k = 0; k < 8; ++k) A4[k] = pmin(A2[k], k < 6 ? A2[k + 2] : swap16(A2[k - 6]));
#pragma unroll
    for (int k = 0; k < 8; ++k) A8[k] = pmin(A4[k], k < 4 ? A4[k + 4] : swap16(A4[k - 4]));
    short2_t best = pmin(A8[0], swap16(P[0]));
#pragma unroll
    for (int k = 1; k < 8; ++k) best = pmax(best, pmin(A8[k], swap16(P[k])));
    return max((int)best.x, (int)best.y);
}

__device__ __forceinline__ int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}
// the same for -n < p < 2 n - 1 (one reflection at most): two selects instead of a loop
__device__ __forceinline__ int reflect101_once(int p, int n)
{
    p = p < 0 ? -p : p;
    return p >= n ? 2 * (n - 1) - p : p;
}

// Stage a (TH x TW)-byte window of the source image, top-left at (gx0, gy0) with gx0 a multiple of 4, into LDS with
// 32-bit loads when the source allows it (row pitch and base 4-byte aligned), bytes otherwise.  Out-of-image coordinates are
// zero (FAST) or reflected (the blur).
template <int TW, int TH, int TP, bool REFLECT>
__device__ __forceinline__ void stage_window(uint8_t* __restrict__ tile, const uint8_t* __restrict__ I, int w, int h, int stride, int gx0, int gy0,
                                             int tw, int th)
{
    const bool aligned = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(I) & 3) == 0);
    const int nq = tw / 4;                              // tw is a multiple of 4
    for (int e = threadIdx.x; e < th * nq; e += 256) {
        const int ty = e / nq, tq = e % nq;
        const int gx = gx0 + 4 * tq;
        int gy = gy0 + ty;
        uint32_t v = 0;
        const bool row_ok = REFLECT || (gy >= 0 && gy < h);
        if (REFLECT) gy = reflect101(gy, h);
        if (row_ok) {
            if (aligned && gx >= 0 && gx + 3 < w) v = *reinterpret_cast<const uint32_t*>(I + (size_t)gy * stride + gx);
            else {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    int x = gx + b;
                    uint32_t px = 0;
                    if (REFLECT) px = I[(size_t)gy * stride + reflect101(x, w)];
                    else if (x >= 0 && x < w) px = I[(size_t)gy * stride + x];
                    v |= px << (8 * b);
                }
            }
        }
        *reinterpret_cast<uint32_t*>(tile + ty * TP + 4 * tq) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// FAST-9/16 + 3x3 non-maximum suppression + border cull, one 64 x 24 pixel tile per workgroup; what reaches HBM is the tile's
// keypoints, (x | y << 16, response), in the tile's own F_MAXKP slots of the frame's list plus their count, in NO particular order
// inside a tile: everything downstream is a function of the SET (k_select ranks with the raster position inside its key), so the
// score map, the raster-order emit pass and the per-band counts of the first version are gone.
//   stage    the tile and a 4-pixel halo (3 ring + 1 score ring), one 128-bit load per thread
//   phase 1  compass test on the tile and a one-pixel ring (66 x 26): any 9-arc contains two NEIGHBOURING compass points (ring
//            positions 0, 4, 8, 12), so a corner needs two neighbouring compass pixels both darker or both brighter than the
//            centre by more than the threshold: (N or S) and (E or W).  Survivors (~20 %) go to an LDS list.
//   phase 2a the same idea on the eight even ring positions (4 consecutive ones) on the list; survivors (~11 %) go to a second list
//   phase 2b the exact score on the second list, where every lane of a wavefront has real work
//   phase 3  strict 3x3 maximum among the raw scores + RunByImageBorder, append
// ---------------------------------------------------------------------------------------------
constexpr int FH = 4;                                   // image halo of the staged window: 3 (ring) + 1 (score ring)
constexpr int FWX = 16;                                 // the window starts 16 pixels left of the tile: 128-bit aligned rows
constexpr int FTW = FT_W + 2 * FWX, FTH = FT_H + 2 * FH; // staged window
constexpr int SCW = FT_W + 2, SCH = FT_H + 2, SCP = 72; // score region and its LDS pitch
constexpr int SC_OFF = 3;                               // region pixel rx sits at byte rx + 3 of its row: the tile's quads are dword-aligned
constexpr int F_MAXKP = FT_W * FT_H / 4;                // a strict 3x3 maximum leaves at most one keypoint per 2x2 block

// wave-aggregated append to an LDS list: lane l contributes the positions pos0 + j of the set bits j of its 8-bit mask.  The
// exclusive prefix of the lanes' counts is a DPP scan along the wavefront, one atomic per wavefront reserves the
// range, and every lane issues its eight stores unconditionally -- the unset ones go to a dump slot -- so there is no divergent
// code at all.
__device__ __forceinline__ void append8(unsigned bits, int pos0, uint16_t* __restrict__ list, int dump, int* __restrict__ counter, int lane)
{
    const int c = __popc(bits);
    // inclusive prefix sum of the lanes' counts along the wavefront: four row_shr steps inside each row of 16 lanes, then the row
    // totals through row_bcast:15 / row_bcast:31 -- six v_add_u32_dpp where the ballot form needed ~35 operations
    int v = c;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    const int tot = __builtin_amdgcn_readlane(v, 63);
    if (tot == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(counter, tot);
    // byte offsets inside the list; a set bit stores at `off` and advances it, an unset one stores at the dump slot.  Bit j of the
    // mask as a one-bit signed field is -1 or 0; twice that is the step of the offset and (both offsets being even) the select mask
    unsigned off = 2u * (unsigned)(__builtin_amdgcn_readfirstlane(base) + v - c);
    const unsigned dump2 = 2u * (unsigned)dump;
    uint8_t* const lb = reinterpret_cast<uint8_t*>(list);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)bits, j, 1), m2 = m + m;
        *reinterpret_cast<uint16_t*>(lb + ((off & m2) | (dump2 & ~m2))) = (uint16_t)(pos0 + j);
        off -= m2;
    }
}

// BLUR: the 7-tap Gaussian of the same tile (what k_blur computes) from the same staged window -- the image is read from HBM once
// for both.  Out-of-image window pixels are then REFLECTED instead of zero; FAST never looks at them (its centres keep 3 pixels
// from the image edge).
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
constexpr int BLUR_NONE = 0, BLUR_VALU = 1, BLUR_MFMA = 2;

template <int BLUR>
__global__ __launch_bounds__(256) void k_fast_keypoints(const uint8_t* __restrict__ img, int w, int h, int stride, size_t frame_stride,
                                                        int threshold, int border, uint8_t* __restrict__ raw_frame0, int wp,
                                                        int2* __restrict__ raw, int* __restrict__ tile_count, OrbTaps taps, uint8_t* __restrict__ blurred,
                                                        const unsigned long long* __restrict__ blur_tab, int blur_c2)
{
    constexpr int TP = FTW;
    constexpr int HROWS = FT_H + 6;                                   // row sums the vertical pass of the blur needs
    constexpr int NCAND = SCH * SCP > HROWS * FT_W ? SCH * SCP : HROWS * FT_W;   // the two candidate lists (2 x 16 bit) share their bytes with the row sums (32 bit)
    __shared__ __attribute__((aligned(16))) uint8_t tile[FTH * TP + 16];
    __shared__ __attribute__((aligned(16))) uint8_t sc[SCH * SCP];    // scores of the tile and its ring
    __shared__ __attribute__((aligned(16))) uint16_t cand_both[2 * NCAND];
    uint16_t* const cand = cand_both;                                 // pixels that survive the compass test
    uint16_t* const cand2 = cand_both + NCAND;                        // ... and the even-position test
    int* const hrow = reinterpret_cast<int*>(cand_both);              // BLUR, after phase 2b: horizontal 7-tap sums of window rows 1 .. HROWS
    __shared__ int2 kl[F_MAXKP];
    __shared__ uint16_t corners[FT_W * FT_H];
    __shared__ int n_cand, n_cand2, n_kept, n_corner;
    const int f = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const uint8_t* I = img + (size_t)f * frame_stride;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
    ORB_CLK_BEGIN();
    // window rows y0 - 4 .. y0 + FT_H + 3, columns x0 - 16 .. x0 + 79: six 16-byte pieces per row, one per thread.  (Measured and
    // left out: 2 / 4 / 5 / 10 vertically adjacent tiles per workgroup with the next window fetched into registers while the
    // current tile is worked on -- a workgroup spends 28 % of its life waiting for its window -- 1.76 / 1.77 / 1.76 / 1.78 /
    // 1.80 ms: with eight wavefronts per SIMD resident the wait is already covered by the other workgroups.)
    static_assert(FTH * (FTW / 16) <= 256, "one 128-bit load per thread");
    const bool aligned16 = ((stride & 15) == 0) && ((reinterpret_cast<uintptr_t>(I) & 15) == 0);
    const int wty = tid / (FTW / 16), wtq = tid % (FTW / 16);
    auto load_window = [&](int wy0) -> uint4 {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (tid < FTH * (FTW / 16)) {
            const int gx = x0 - FWX + 16 * wtq, gy = wy0 - FH + wty;
            if (BLUR || (gy >= 0 && gy < h)) {
                const uint8_t* rowp = I + (size_t)(BLUR ? (h > FH ? reflect101_once(gy, h) : reflect101(gy, h)) : gy) * stride;   // gy in [-4, h + 3]
                if (aligned16 && gx >= 0 && gx + 15 < w) v = *reinterpret_cast<const uint4*>(rowp + gx);
                else if (BLUR && (gx + 15 < 0 ? -gx < w : gx >= w && 2 * (w - 1) - gx < w && 2 * (w - 1) - gx - 15 >= 0)) {
                    // a piece entirely beyond the left or the right edge, reflected once: sixteen consecutive pixels in reverse
                    // order (every edge tile has such pieces; byte by byte they cost ~150 instructions per wavefront)
                    typedef uint32_t __attribute__((aligned(1))) unaligned_u32;
                    const unaligned_u32* src = reinterpret_cast<const unaligned_u32*>(rowp + (gx + 15 < 0 ? -gx - 15 : 2 * (w - 1) - gx - 15));
                    v.w = __builtin_bswap32(src[0]); v.z = __builtin_bswap32(src[1]); v.y = __builtin_bswap32(src[2]); v.x = __builtin_bswap32(src[3]);
                }
                else if (BLUR || (gx + 15 >= 0 && gx < w)) {
                    // the frame's edge columns or an unaligned frame: byte by byte, one dword at a time (a rolled loop: this path
                    // must not set the register count of the kernel)
#pragma unroll 1
                    for (int q = 0; q < 4; ++q) {
                        uint32_t dq = 0;
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const int x = gx + 4 * q + b;
                            if (BLUR) dq |= (uint32_t)rowp[reflect101(x, w)] << (8 * b);
                            else if (x >= 0 && x < w) dq |= (uint32_t)rowp[x] << (8 * b);
                        }
                        if (q == 0) v.x = dq; else if (q == 1) v.y = dq; else if (q == 2) v.z = dq; else v.w = dq;
                    }
                }
            }
        }
        return v;
    };
    const uint4 window = load_window(y0);
    if (tid == 0) { n_cand = 0; n_cand2 = 0; n_kept = 0; n_corner = 0; }
    for (int e = tid; e < SCH * SCP / 16; e += 256) reinterpret_cast<uint4*>(sc)[e] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < FTH * (FTW / 16)) *reinterpret_cast<uint4*>(tile + wty * TP + 16 * wtq) = window;
    __syncthreads();
    ORB_CLK(0);
    // The comparisons of phases 1 and 2a are 32-bit subtractions and bitwise logic on TWO 16-bit fields per register (the
    // operations that issue in 2 cycles on this part; packed 16-bit min / max take 4, tools/valu_rate.hip).  With a guard bit per
    // field no borrow crosses it: bit 15 of (V + 0x8000 - t - 1) - R is set iff R < V - t (R darker than the centre by more than
    // t), bit 15 of (V + 0x8000 + t) - R iff R <= V + t (R NOT brighter by more than t).
    const uint32_t tc = (uint32_t)min(max(threshold, 0), 255);        // beyond 255 nothing passes, as at 255
    const uint32_t kd = (0x8000u - tc - 1u) * 0x00010001u, kb = (0x8000u + tc) * 0x00010001u;
    // Phase 1.  Region pixel (rx, ry) = image (x0 - 1 + rx, y0 - 1 + ry) = tile byte (rx + 15, ry + 3); a thread takes EIGHT
    // consecutive rx: 9 groups per region row (SCP = 72 = 9 x 8, so the list position of pixel j of group e is 8 e + j) and
    // 9 x SCH groups in all -- one pass of the workgroup.
    static_assert(SCP == 8 * ((SCW + 7) / 8) && (SCP / 8) * SCH <= 256, "one group of eight region pixels per thread");
    {
        const int e = tid, ry = e / (SCP / 8), o = e - ry * (SCP / 8);
        const int y = y0 - 1 + ry;
        unsigned passbits = 0;
        if (e < (SCP / 8) * SCH && y >= 3 && y < h - 3) {
            // bytes 8 o + 12 .. 8 o + 27 of the centre row: centre j is byte 3 + j, its W neighbour byte j, its E neighbour byte 6 + j;
            // N / S are bytes 3 + j of the rows three above / below
            const uint32_t* mp = reinterpret_cast<const uint32_t*>(&tile[(ry + 3) * TP + 8 * o + 12]);
            const uint32_t* up = reinterpret_cast<const uint32_t*>(&tile[(ry + 6) * TP + 8 * o + 12]);   // row y + 3 (ring 0)
            const uint32_t* dn = reinterpret_cast<const uint32_t*>(&tile[(ry + 0) * TP + 8 * o + 12]);   // row y - 3 (ring 8)
            const uint32_t mc[4] = { mp[0], mp[1], mp[2], mp[3] }, mu[3] = { up[0], up[1], up[2] }, md[3] = { dn[0], dn[1], dn[2] };
            // two pixels per register, one in each 16-bit half (kd, kb above)
            uint32_t signs = 0;
#pragma unroll
            for (int twice = 0; twice < ((MAGE_ORB_ABLATE & 1) ? 2 : 1); ++twice)
#pragma unroll
            for (int hp = 0; hp < 4; ++hp) {
                if (twice) { uint32_t* q = const_cast<uint32_t*>(mc); ORB_LAUNDER(q[0]); ORB_LAUNDER(q[1]); ORB_LAUNDER(q[2]); ORB_LAUNDER(q[3]); }
                // bytes i, i + 1 of a group of dwords, zero-extended to 16 bits each (v_perm_b32: selector 0..3 = bytes of the
                // second operand, 4..7 = bytes of the first, 0x0c = zero)
                auto pair16 = [&](const uint32_t* g, int i) -> uint32_t {
                    const int d = i >> 2, k = i & 3;
                    return k < 3 ? __builtin_amdgcn_perm(0u, g[d], 0x0c000c00u | (uint32_t)k | ((uint32_t)(k + 1) << 16))
                                 : __builtin_amdgcn_perm(g[d + 1], g[d], 0x0c040c03u);
                };
                const uint32_t V = pair16(mc, 3 + 2 * hp), Ee = pair16(mc, 6 + 2 * hp), Ww = pair16(mc, 2 * hp);
                const uint32_t Nn = pair16(mu, 3 + 2 * hp), Ss = pair16(md, 3 + 2 * hp);
                const uint32_t Vd = V + kd, Vb = V + kb;
                // two neighbouring compass points both darker, or both brighter: (N or S) and (E or W)
                const uint32_t dark = ((Vd - Nn) | (Vd - Ss)) & ((Vd - Ee) | (Vd - Ww));
                const uint32_t not_bright = ((Vb - Nn) & (Vb - Ss)) | ((Vb - Ee) & (Vb - Ww));
                signs |= ((dark | ~not_bright) & 0x80008000u) >> (15 - 2 * hp);      // pixel 2 hp -> bit 2 hp, pixel 2 hp + 1 -> bit 16 + 2 hp
            }
            passbits = (signs | (signs >> 15)) & 0xffu;
            // pixels of the group that are region pixels and FAST centres: rx < SCW, 3 <= x < w - 3
            const int xs = x0 - 1 + 8 * o;
            const int jl = min(max(3 - xs, 0), 8), jh = min(max(min(SCW - 8 * o, w - 3 - xs), 0), 8);
            passbits &= ((1u << jh) - 1u) & ~((1u << jl) - 1u);
        }
        if (!(MAGE_ORB_ABLATE & 8192)) append8(passbits, 8 * e, cand, NCAND - 1, &n_cand, lane);
        else if (passbits == 0x5a5a5a5au) n_cand = 1;
    }
    __syncthreads();
    ORB_CLK(1);
    // Phase 2a: even-position test.  A survivor is listed once per polarity that is still possible (bit 12 = brighter ring), so
    // that phase 2b runs ONE min ladder per entry; the few pixels with both polarities open get two entries.  (Measured and left
    // out: the EXACT FAST-9 decision here as bit arithmetic -- sixteen guard-bit subtractions per polarity, the flags gathered into a
    // ring mask, three shift-and-and steps; bit-exact, and phase 2b shrinks to the ~3 % that are corners -- but it reads 17 ring
    // bytes per candidate where this test reads 9, and the byte gathers from LDS, not the arithmetic, are what a candidate costs:
    // FAST 1.83 -> 2.02 ms.)
    const int nc = (MAGE_ORB_ABLATE & 4096) ? 0 : n_cand;
    for (int c0 = 0; c0 < nc; c0 += 256) {
        const int c = c0 + tid;
        int p = 0, polar = 0;
        if (c < nc) {
            p = cand[c];
            const int ry = p / SCP, rx = p % SCP;
            polar = fast_even4_test(&tile[(ry + 3) * TP + rx + 15], TP, kd, kb);
            if (MAGE_ORB_ABLATE & 2) { int off = (ry + 3) * TP + rx + 15; ORB_LAUNDER(off); polar |= fast_even4_test(&tile[off], TP, kd, kb); }
        }
        const unsigned long long bal0 = __ballot(polar & 1), bal1 = __ballot(polar & 2);
        if (bal0 | bal1) {
            int base = 0;
            const int n0 = __popcll(bal0);
            if (lane == 0) base = atomicAdd(&n_cand2, n0 + __popcll(bal1));
            base = __shfl(base, 0, 64);
            const unsigned long long below = (1ull << lane) - 1ull;
            if (polar & 1) cand2[base + __popcll(bal0 & below)] = (uint16_t)p;
            if (polar & 2) cand2[base + n0 + __popcll(bal1 & below)] = (uint16_t)(p | 0x1000);
        }
    }
    __syncthreads();
    // Phase 2b: exact score of the survivors.  Only a corner is written (the tile of scores starts at zero), and a pixel cannot be a
    // corner in both polarities, so its two entries never write both.
    const int nc2 = (MAGE_ORB_ABLATE & 4096) ? 0 : n_cand2;
    for (int c = tid; c < nc2; c += 256) {
        const int pc = cand2[c], p = pc & 0xfff;
        const int ry = p / SCP, rx = p % SCP;
        int m = fast_score_polar(&tile[(ry + 3) * TP + rx + 15], TP, (pc & 0x1000) != 0);
        if (MAGE_ORB_ABLATE & 4) { int off = (ry + 3) * TP + rx + 15; ORB_LAUNDER(off); m = max(m, fast_score_polar(&tile[off], TP, (pc & 0x1000) != 0)); }
        const bool corner = m > threshold;
        if (corner) sc[p + SC_OFF] = (uint8_t)(m - 1);
        // the corners of the tile proper (not of its ring) are what phase 3 has to look at: a list, ~3 % of the pixels
        const unsigned long long bal = __ballot(corner && rx >= 1 && rx <= FT_W && ry >= 1 && ry <= FT_H);
        if (bal) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&n_corner, __popcll(bal));
            base = __shfl(base, 0, 64);
            if ((bal >> lane) & 1ull) corners[base + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)p;
        }
    }
    __syncthreads();
    ORB_CLK(2);
    const int lo = border > 3 ? border : 3;
    if (f == 0 && raw_frame0) {                          // parity tests read the raw score map of frame 0
        for (int qi = tid; qi < FT_W * FT_H / 4; qi += 256) {
            const int ly = qi / (FT_W / 4), lq = qi % (FT_W / 4);
            const int y = y0 + ly, xq = x0 + 4 * lq;
            if (y < h && xq < wp)
                *reinterpret_cast<uint32_t*>(raw_frame0 + (size_t)y * wp + xq) = *reinterpret_cast<const uint32_t*>(&sc[(ly + 1) * SCP + 4 * lq + 1 + SC_OFF]);
        }
    }
    // Phase 3: strict 3x3 maximum among the raw scores + RunByImageBorder, one lane per listed corner (~3 % of the pixels: mostly
    // the first wavefront), its eight neighbours in eight independent byte reads
    {
        const int ncn = (MAGE_ORB_ABLATE & 1024) ? 0 : n_corner;
        for (int c = tid; c < ncn; c += 256) {
            const int p = corners[c];
            const int ry = p / SCP, rx = p % SCP;
            const int x = x0 - 1 + rx, y = y0 - 1 + ry;
            const uint8_t* q = &sc[p + SC_OFF];
            const int s = q[0];
            const int m = max(max(max((int)q[-SCP - 1], (int)q[-SCP]), max((int)q[-SCP + 1], (int)q[-1])),
                              max(max((int)q[1], (int)q[SCP - 1]), max((int)q[SCP], (int)q[SCP + 1])));
            if (s > m && x >= lo && x < w - lo && y >= lo && y < h - lo) kl[atomicAdd(&n_kept, 1)] = make_int2(x | (y << 16), s);
        }
    }
    if (BLUR == BLUR_MFMA) {
        // The 7-tap separable Gaussian as two banded matrix products on the otherwise idle matrix cores (v_mfma_i32_16x16x32_i8):
        // the vector ALUs, which bound this kernel, are left with the operand shuffles.  Wavefront J owns pixel columns
        // 16 J .. 16 J + 15 of the tile.
        //   pass 1   Hs[row][col] = sum_k (pixel[row][k] - 128) * B1[k][col] + 128  for the 32 window rows: A = eight bytes of a window
        //            row per lane (one ds_read_b64, xor 0x80: i8 operands), B1 = the taps as a banded matrix (from blur_tab)
        //   split    Hs fits 16 bits (taps sum <= 257): Hs = 256 hi + (lo_s + 128) with hi = Hs >> 8 and lo_s = (Hs & 255) - 128,
        //            both i8.  The result registers of pass 1 -- lane = column, four consecutive rows each -- ARE the A operand
        //            layout of pass 2 (lane = column, eight K slots = eight window rows), so the split is byte permutes in place.
        //   pass 2   out[col][row] = (256 * sum_k hi[k][col] B2[k][row] + sum_k lo_s[k][col] B2[k][row] + c2) >> 16, c2 = 128 T^2 +
        //            2^15: two products per 16 output rows, lane = output row, four consecutive columns per lane -> one
        //            32-bit store.  Exact integer arithmetic throughout, the same value the two-pass form computes.
        const int J = tid >> 6, g = lane >> 4, j = lane & 15;
        const long B1 = (MAGE_ORB_ABLATE & 16) ? (long)lane * 0x0101010101010101l : (long)blur_tab[lane];
        const long B2q[2] = { (MAGE_ORB_ABLATE & 16) ? (long)lane * 0x0102010201020102l : (long)blur_tab[64 + lane], (MAGE_ORB_ABLATE & 16) ? (long)lane * 0x0301030103010301l : (long)blur_tab[128 + lane] };
        const long flip = (long)0x8080808080808080ull;
        const long a0 = *reinterpret_cast<const long*>(&tile[j * TP + 16 * J + 8 + 8 * g]) ^ flip;
        const long a1 = *reinterpret_cast<const long*>(&tile[(16 + j) * TP + 16 * J + 8 + 8 * g]) ^ flip;
        const v4i_t c128 = { 128, 128, 128, 128 };
        const v4i_t h0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a0, B1, c128, 0, 0, 0);
        const v4i_t h1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a1, B1, c128, 0, 0, 0);
        // 16-bit pairs, then the low bytes (xor 0x80: lo_s) and the high bytes of eight rows
        const uint32_t p0 = __builtin_amdgcn_perm((uint32_t)h0[1], (uint32_t)h0[0], 0x05040100u), p1 = __builtin_amdgcn_perm((uint32_t)h0[3], (uint32_t)h0[2], 0x05040100u);
        const uint32_t p2 = __builtin_amdgcn_perm((uint32_t)h1[1], (uint32_t)h1[0], 0x05040100u), p3 = __builtin_amdgcn_perm((uint32_t)h1[3], (uint32_t)h1[2], 0x05040100u);
        const uint32_t lo_a = __builtin_amdgcn_perm(p1, p0, 0x06040200u) ^ 0x80808080u, lo_b = __builtin_amdgcn_perm(p3, p2, 0x06040200u) ^ 0x80808080u;
        const uint32_t hi_a = __builtin_amdgcn_perm(p1, p0, 0x07050301u), hi_b = __builtin_amdgcn_perm(p3, p2, 0x07050301u);
        const long A_lo = (long)((unsigned long long)lo_a | ((unsigned long long)lo_b << 32)), A_hi = (long)((unsigned long long)hi_a | ((unsigned long long)hi_b << 32));
        uint8_t* O = blurred + (size_t)f * wp * h;
        const int xq = x0 + 16 * J + 4 * g;
        const v4i_t cc2 = { blur_c2, blur_c2, blur_c2, blur_c2 }, zero4 = { 0, 0, 0, 0 };
#pragma unroll
        for (int Q = 0; Q < 2; ++Q) {
            const v4i_t dl = __builtin_amdgcn_mfma_i32_16x16x32_i8(A_lo, B2q[Q], cc2, 0, 0, 0);
            const v4i_t dh = __builtin_amdgcn_mfma_i32_16x16x32_i8(A_hi, B2q[Q], zero4, 0, 0, 0);
            uint32_t wv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) wv[r] = ((uint32_t)dh[r] << 8) + (uint32_t)dl[r];
            // bits 16 .. 31 of each value (<= 257), saturated to a byte
            const uint32_t q01 = __builtin_amdgcn_perm(wv[1], wv[0], 0x07060302u), q23 = __builtin_amdgcn_perm(wv[3], wv[2], 0x07060302u);
            ushort2_t s01, s23; __builtin_memcpy(&s01, &q01, 4); __builtin_memcpy(&s23, &q23, 4);
            const ushort2_t cap = { 255, 255 };
            s01 = __builtin_elementwise_min(s01, cap); s23 = __builtin_elementwise_min(s23, cap);
            uint32_t c01, c23; __builtin_memcpy(&c01, &s01, 4); __builtin_memcpy(&c23, &s23, 4);
            uint32_t packed = __builtin_amdgcn_perm(c23, c01, 0x06040200u);
            const int orow = 16 * Q + j, y = y0 + orow;
            if ((MAGE_ORB_ABLATE & 32) ? packed == 0x12345678u && orow < FT_H : orow < FT_H && y < h && xq < wp) {
                if (xq + 4 > w) packed &= xq < w ? (1u << (8 * (w - xq))) - 1u : 0u;        // bytes beyond the image width stay zero
                *reinterpret_cast<uint32_t*>(O + (size_t)y * wp + xq) = packed;
            }
        }
    }
    if (BLUR == BLUR_VALU) {
        // horizontal pass (the candidate lists are dead): four outputs per thread from three aligned 32-bit LDS reads; each output
        // is two 4-way byte dot products (v_dot4_u32_u8) over windows cut out of the 12 bytes with v_alignbyte
        const uint32_t T0 = (uint32_t)taps.t[0] | ((uint32_t)taps.t[1] << 8) | ((uint32_t)taps.t[2] << 16) | ((uint32_t)taps.t[3] << 24);
        const uint32_t T1 = (uint32_t)taps.t[4] | ((uint32_t)taps.t[5] << 8) | ((uint32_t)taps.t[6] << 16);
        for (int e = tid; e < HROWS * (FT_W / 4); e += 256) {
            const int ty = e / (FT_W / 4), tq = e % (FT_W / 4);
            const uint32_t* sp = reinterpret_cast<const uint32_t*>(&tile[(ty + 1) * TP + FWX - 4 + 4 * tq]);   // bytes o .. o + 11, o = image x - 4; output j starts at o + 1 + j
            const uint32_t A = sp[0], B = sp[1], C = sp[2];
            int4 o;
            o.x = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 1), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 1), T0, 0u, false), false);
            o.y = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 2), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 2), T0, 0u, false), false);
            o.z = (int)__builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(C, B, 3), T1, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(B, A, 3), T0, 0u, false), false);
            o.w = (int)__builtin_amdgcn_udot4(C, T1, __builtin_amdgcn_udot4(B, T0, 0u, false), false);
            *reinterpret_cast<int4*>(&hrow[ty * FT_W + 4 * tq]) = o;
        }
        __syncthreads();
        ORB_CLK(3);
        // vertical pass: a thread owns four pixel columns and two consecutive rows, reads the eight row sums they need once and
        // packs one 32-bit store per row; products fit 24 bits (tap <= 255, row sum < 2^16)
        static_assert(FT_H % 2 == 0 && (FT_H / 2) * (FT_W / 4) <= 192, "one 4 x 2 pixel patch per thread of the first three wavefronts");
        uint8_t* O = blurred + (size_t)f * wp * h;
        const int tq = tid % (FT_W / 4), strip = tid / (FT_W / 4);
        const int xq = x0 + 4 * tq;
        if (strip < FT_H / 2 && xq < wp) {
            int4 hv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) hv[i] = *reinterpret_cast<const int4*>(&hrow[(strip * 2 + i) * FT_W + 4 * tq]);
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int y = y0 + strip * 2 + o;
                if (y >= h) break;
                int acc4[4] = { 0, 0, 0, 0 };
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    acc4[0] = (__mul24(taps.t[k], hv[o + k].x) + acc4[0]); acc4[1] = (__mul24(taps.t[k], hv[o + k].y) + acc4[1]);
                    acc4[2] = (__mul24(taps.t[k], hv[o + k].z) + acc4[2]); acc4[3] = (__mul24(taps.t[k], hv[o + k].w) + acc4[3]);
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
    }
    __syncthreads();
    // the tile's keypoints go to the tile's own slots of the frame's list: no cursor, no atomic, nothing to clear between launches
    const int nk = n_kept;
    const size_t t = ((size_t)f * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (tid == 0) tile_count[t] = nk;
    int2* out = raw + t * F_MAXKP;
    for (int i = tid; i < nk; i += 256) out[i] = kl[i];
    ORB_CLK(4);
}

// ---------------------------------------------------------------------------------------------
// selection: one workgroup (1024 threads) per frame.  The input list has no order; every result below is a function of the
// set (histogram, bounding box, cell lists, minima), and the output position is the rank in a TOTAL order whose last tie-break
// is the raster position -- what "index in raster order" was when the list was emitted row by row.
// ---------------------------------------------------------------------------------------------
constexpr int SEL_T = 1024;                     // threads
constexpr int SEL_CAP = 2048, SEL_CELLS = 1024; // LDS working set: candidates after RetainBest, grid cells
constexpr int SEL_ROUNDS = 7;                   // tiles per half wavefront whose first 64 entries stay in registers (32 x 7 = 224 tiles)
constexpr int SEL_HCOPIES = 16;                 // interleaved copies of the response histogram (same-bin lanes spread over banks)

// inclusive prefix sum along the wavefront: four row_shr steps inside each row of 16 lanes, then the row totals through
// row_bcast:15 / row_bcast:31 (six v_add_u32_dpp; the shuffle form was six LDS-crossbar round trips)
__device__ __forceinline__ int wave_scan_incl(int v, int)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// reduction along the wavefront with an idempotent operation (min / max), result in lane 63: the same six DPP steps; a lane without a
// source keeps its own value
template <class Op>
__device__ __forceinline__ int wave_reduce_to_last(int v, Op op)
{
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return v;
}

// exclusive scan over the 1024 threads in thread order; sh = 16 ints
__device__ __forceinline__ int block_scan_excl(int v, int* sh, int& total)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int incl = wave_scan_incl(v, lane);
    __syncthreads();
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int c = sh[q]; if (q < wave) woff += c; tot += c; }
    total = tot;
    return woff + incl - v;
}

__device__ __forceinline__ mage_keypoint make_kp(int2 r, float size_f)
{
    mage_keypoint k = { (float)(r.x & 0xffff), (float)(r.x >> 16), size_f, 0.0f, (float)r.y, 0, -1 };
    return k;
}

// isqrt for the ring bound (values < 2^31)
__device__ __forceinline__ int isqrt_floor(int v)
{
    int r = (int)sqrtf((float)v);
    while (r * r > v) --r;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}

__global__ __launch_bounds__(SEL_T) void k_select(OrbSelectArgs a)
{
#pragma clang fp contract(off)          // float results feed comparisons that must match the CPU restatement: plain IEEE operations, never an FMA
    __shared__ int sh[16];
    __shared__ int lhist[256 * SEL_HCOPIES];
    __shared__ __attribute__((aligned(16))) int suffix[260];
    __shared__ int s_n, s_uni[8];
    __shared__ int s_minX, s_maxX, s_minY, s_maxY, s_minS;
    __shared__ __attribute__((aligned(16))) unsigned long long l_cand[SEL_CAP];      // candidates in cell order: pos | response << 32
    __shared__ __attribute__((aligned(16))) unsigned long long l_key[SEL_CAP];
    __shared__ int l_cellstart[SEL_CELLS + 1], l_cellfill[SEL_CELLS];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int2* raw = a.raw + (size_t)f * a.n_tiles * a.tile_cap;
    const int* tcount = a.tile_count + (size_t)f * a.n_tiles;
    mage_keypoint* okp = a.out_kp + (size_t)f * a.capacity;
    const float size_f = (float)a.patch_size * 1.0f;
    const int N = a.nfeatures;
    ORB_CLK_BEGIN();
    for (int e = tid; e < 256 * SEL_HCOPIES; e += SEL_T) lhist[e] = 0;
    for (int c = tid; c <= SEL_CELLS; c += SEL_T) { l_cellstart[c] = 0; if (c < SEL_CELLS) l_cellfill[c] = 0; }     // (here rather than where they are used: one barrier less on the way)
    if (tid == 0) { s_n = 0; s_minX = 1 << 30; s_maxX = -1; s_minY = 1 << 30; s_maxY = -1; s_minS = 1 << 30; }
    // The frame's list is ragged: tile t holds tcount[t] entries in slots t * tile_cap ...  A HALF wavefront owns tile hw, hw + 32, ...
    // and keeps the first 64 entries of its first SEL_ROUNDS tiles in registers (224 tiles; 640 x 480 has 200 with ~20 entries each,
    // rarely 40).  The slots are fetched SPECULATIVELY, together with the counts that say which of them are entries: one trip to
    // memory where "counts -> prefix sums -> binary search per entry -> slots" took two and ~25 LDS round trips (11 k of the kernel's
    // 75 k cycles on one frame, tools/orb_phase_probe.py).  Entries beyond 64 per tile and tiles beyond 224 are read again from HBM
    // whenever the set is walked (for_each_entry).
    const int hw = tid >> 5, l32 = tid & 31;
    int cnt[SEL_ROUNDS];
    int2 ra[SEL_ROUNDS], rb[SEL_ROUNDS];
    const int capm1 = a.tile_cap - 1;
#pragma unroll
    for (int r = 0; r < SEL_ROUNDS; ++r) {
        const int t = hw + 32 * r;
        const bool ok = t < a.n_tiles;
        const int2* slot = raw + (size_t)(ok ? t : 0) * a.tile_cap;
        cnt[r] = ok ? tcount[t] : 0;
        ra[r] = slot[min(l32, capm1)];
        rb[r] = slot[min(32 + l32, capm1)];
    }
    int n_raw;
    bool overflow = a.n_tiles > 32 * SEL_ROUNDS;
    {
        int local = 0;
        if (l32 == 0) {
#pragma unroll
            for (int r = 0; r < SEL_ROUNDS; ++r) local += cnt[r];
            for (int t = hw + 32 * SEL_ROUNDS; t < a.n_tiles; t += 32) local += tcount[t];
        }
#pragma unroll
        for (int r = 0; r < SEL_ROUNDS; ++r) overflow = overflow || cnt[r] > 64;
        overflow = __any(overflow);                      // per wavefront: the walk below branches on it uniformly
        const int incl = wave_scan_incl(local, lane);
        if (lane == 63) sh[tid >> 6] = incl;
        __syncthreads();
        int tot = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += sh[q];
        n_raw = tot;
    }
    ORB_CLK_WAIT();
    ORB_CLK(16);
    // every entry of the frame's list exactly once, in no particular order (everything below is a function of the SET)
    auto for_each_entry = [&](auto fn) {
#pragma unroll
        for (int r = 0; r < SEL_ROUNDS; ++r) {
            if (l32 < cnt[r]) fn(ra[r]);
            if (32 + l32 < cnt[r]) fn(rb[r]);
        }
        if (overflow) {
#pragma unroll 1
            for (int r = 0; r < SEL_ROUNDS; ++r)
                for (int k = 64 + l32; k < cnt[r]; k += 32) fn(raw[(size_t)(hw + 32 * r) * a.tile_cap + k]);
            for (int t = hw + 32 * SEL_ROUNDS; t < a.n_tiles; t += 32) {
                const int c = tcount[t];
                for (int k = l32; k < c; k += 32) fn(raw[(size_t)t * a.tile_cap + k]);
            }
        }
    };

    // Keeps every entry as the reference's early returns do (fewer detections than the quota): output in raster order = rank by
    // position.  `get(i)` reads entry i of the set, n of them.
    auto emit_all_by_position = [&](auto get, int n) {
        for (int i = tid; i < n; i += SEL_T) {
            const int2 r = get(i);
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += (unsigned)get(j).x < (unsigned)r.x ? 1 : 0;      // pos = x | y << 16 is the raster order
            if (rank < a.capacity) okp[rank] = make_kp(r, size_f);
        }
        if (tid == 0) a.out_count[f] = min(n, a.capacity);
    };
    // the entries that pass `keep`, gathered (in no order) in LDS when they fit and in the frame's scratch otherwise, then emitted by position
    auto gather_and_emit = [&](auto keep, int upper_bound) {
        int2* scratch = reinterpret_cast<int2*>(a.cand64 + (size_t)f * a.scratch_cap);
        const bool fits = upper_bound <= SEL_CAP;
        for_each_entry([&](int2 r) {
            if (keep(r)) {
                const int at = atomicAdd(&s_n, 1);
                if (fits) l_cand[at] = (unsigned long long)(unsigned)r.x | ((unsigned long long)(unsigned)r.y << 32);
                else scratch[at] = r;
            }
        });
        __threadfence_block();
        __syncthreads();
        if (fits) emit_all_by_position([&](int i) { const unsigned long long v = l_cand[i]; return make_int2((int)(unsigned)v, (int)(v >> 32)); }, s_n);
        else emit_all_by_position([&](int i) { return scratch[i]; }, s_n);
    };
    if (n_raw <= N) {
        gather_and_emit([](int2) { return true; }, n_raw);
        return;
    }
    // ---- RetainBestFeatures (OpenCVModified.cpp:571-617): whole histogram bins from 255 downwards.  Histogram of the responses
    // with LDS atomics on 16 interleaved copies (most responses sit in a handful of low bins), suffix[i] = number of
    // responses >= i, and the two thresholds are "largest bin whose suffix count reaches the quota".
    for_each_entry([&](int2 r) { atomicAdd(&lhist[(r.y & 255) * SEL_HCOPIES + (lane & (SEL_HCOPIES - 1))], 1); });
    __syncthreads();
    ORB_CLK(17);
    {
        int v = 0;
        if (tid < 256) {
            const int4* hp = reinterpret_cast<const int4*>(&lhist[tid * SEL_HCOPIES]);
#pragma unroll
            for (int k = 0; k < SEL_HCOPIES / 4; ++k) { const int4 c = hp[k]; v += c.x + c.y + c.z + c.w; }
        }
        // suffix sum over the 256 bins = total - exclusive prefix
        int tot;
        const int excl = block_scan_excl(v, sh, tot);
        if (tid < 256) suffix[tid] = tot - excl;
        if (tid == 0) suffix[256] = 0;
    }
    __syncthreads();
    ORB_CLK(18);
    // suffix[] does not increase with the bin, so "the largest bin whose count reaches q" is (number of bins that reach q) - 1: four bins
    // per lane, four ballots, no atomic and no barrier (256 lanes raising one LDS word with atomicMax took 2.6 k cycles per threshold)
    const int min_thr = a.fast_threshold;
    const int4 sfx = *reinterpret_cast<const int4*>(&suffix[4 * lane]);
    auto bins_reaching = [&](int q) {
        return __popcll(__ballot(sfx.x >= q)) + __popcll(__ballot(sfx.y >= q)) + __popcll(__ballot(sfx.z >= q)) + __popcll(__ballot(sfx.w >= q));
    };
    const int top_n = bins_reaching(N) - 1;
    const int mnt = top_n >= min_thr ? top_n : min_thr;
    const int lower = max((int)((float)mnt * a.feature_strength), min_thr);
    const int top_m = bins_reaching(a.max_num) - 1;
    const int cut = top_m >= lower ? top_m : lower;
    const int M = cut < 256 ? suffix[cut] : 0;          // how many candidates RetainBest keeps
    ORB_CLK(10);
    // bounding box and weakest response of the kept set
    {
        int mnx = 1 << 30, mxx = -1, mny = 1 << 30, mxy = -1, mns = 1 << 30;
        for_each_entry([&](int2 r) {
            if (r.y >= cut) {
                const int x = r.x & 0xffff, y = r.x >> 16;
                mnx = min(mnx, x); mxx = max(mxx, x); mny = min(mny, y); mxy = max(mxy, y); mns = min(mns, r.y);
            }
        });
        auto lo = [](int p, int q) { return min(p, q); };
        auto hi = [](int p, int q) { return max(p, q); };
        mnx = wave_reduce_to_last(mnx, lo); mxx = wave_reduce_to_last(mxx, hi);
        mny = wave_reduce_to_last(mny, lo); mxy = wave_reduce_to_last(mxy, hi); mns = wave_reduce_to_last(mns, lo);
        if (lane == 63 && mxx >= 0) {
            atomicMin(&s_minX, mnx); atomicMax(&s_maxX, mxx); atomicMin(&s_minY, mny); atomicMax(&s_maxY, mxy); atomicMin(&s_minS, mns);
        }
    }
    __syncthreads();
    ORB_CLK(11);
    if (N > M) {
        // AdaptiveNonMaximalSuppresion returns its input when there is nothing to suppress (feature_strength > 1 can cut below
        // the quota): the kept set in raster order.
        gather_and_emit([&](int2 r) { return r.y >= cut; }, M);
        return;
    }
    // ---- AdaptiveNonMaximalSuppresion (OpenCVModified.cpp:144-360)
    const int minX = s_minX, maxX = s_maxX, minY = s_minY, maxY = s_maxY;
    const int numX = a.cells_x, numY = a.cells_y, thr = a.fast_threshold;
    // Division by the bounding box's extents (uniform per frame, non-negative dividends below 2^31): multiply-high by a magic number
    // made once -- the compiler's general 32-bit division is ~40 vector instructions, and every candidate is divided six times.
    // q = floor(n / d) exactly for 0 <= n < 2^31: M = floor(2^32 (2^s - d) / d) + 1 with s = ceil(log2 d), t = mulhi(n, M), q = (t + ((n - t) >> 1)) >> (s - 1)
    struct UDiv { unsigned M; int s; };
    // (the magic number by long division in two 16-bit steps: d <= 65536 -- coordinates are 16 bits -- so e = 2^s - d < d and both partial
    // dividends fit 32 bits; the 64-bit division this replaces is ~1.5 k cycles of library code, twice per frame)
    auto make_udiv = [](unsigned d) -> UDiv {
        const int sft = d <= 1u ? 0 : 32 - __clz((int)(d - 1u));      // smallest s with 2^s >= d
        const unsigned e = (1u << sft) - d;
        const unsigned t1 = (e << 16) / d, r1 = (e << 16) - t1 * d;
        const unsigned t2 = (r1 << 16) / d;
        return UDiv{ (t1 << 16) + t2 + 1u, sft };
    };
    auto udiv = [](unsigned n, const UDiv& D) -> int {
        if (D.s == 0) return (int)n;                    // d == 1
        const unsigned t = __umulhi(n, D.M);
        return (int)((t + ((n - t) >> 1)) >> (D.s - 1));
    };
    // The per-frame constants are ~350 instructions of divisions.  Sixteen wavefronts each working them out is 4 x 350 issue slots
    // per SIMD (5.6 k cycles, the largest single item of the "cell counts" phase); four wavefronts on four SIMDs take a quarter each
    // and hand the results over through LDS.
    {
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
        if (w == 0) {
            const UDiv d = make_udiv((unsigned)(maxX + 1 - minX));
            if (lane == 0) { s_uni[0] = (int)d.M; s_uni[1] = d.s; }
        } else if (w == 1) {
            const UDiv d = make_udiv((unsigned)(maxY + 1 - minY));
            if (lane == 0) { s_uni[2] = (int)d.M; s_uni[3] = d.s; }
        } else if (w == 2) {
            const float hi = (float)a.strong_response - (float)thr;
            float val = (float)s_minS - (float)thr;
            val = val < 0.0f ? 0.0f : (val > hi ? hi : val);
            float range = a.max_robust - a.min_robust;
            if (range < 0.0f) range = 0.0f;
            const float rf_ = a.max_robust - (val / (float)(a.strong_response - thr)) * range;
            if (lane == 0) s_uni[4] = __float_as_int(rf_);
        } else if (w == 3) {
            const int g = (int)(((double)(maxX - minX)) * ((double)(maxY - minY)) / (double)N);
            const int dx = max((maxX - minX) / numX, 1), dy = max((maxY - minY) / numY, 1);
            const int m = min(dx, dy);
            if (lane == 0) { s_uni[5] = g; s_uni[6] = m * m; }
        }
        __syncthreads();
    }
    const UDiv divX = { (unsigned)s_uni[0], s_uni[1] }, divY = { (unsigned)s_uni[2], s_uni[3] };
    const float rf = __int_as_float(s_uni[4]);
    const int globalMaxR2 = s_uni[5], minCellDelta2 = s_uni[6];
    const bool narrow_box = (maxX - minX) < numX || (maxY - minY) < numY;
    auto cell_of = [&](int pos) { return udiv((unsigned)(((pos >> 16) - minY) * numY), divY) * numX + udiv((unsigned)(((pos & 0xffff) - minX) * numX), divX); };
    // The working set (candidates in cell order, cell starts, keys) lives in LDS whenever it fits; oversized inputs run the same
    // code on the per-frame scratch in HBM.
    const bool in_lds = M <= SEL_CAP && a.ncells <= SEL_CELLS;
    auto suppress_and_rank = [&](unsigned long long* cand, int* cellstart, int* cellfill, unsigned long long* key) {
        // counting sort of the kept set by grid cell: cellstart[c] .. cellstart[c + 1] are the members of cell c, and the cells of
        // one grid row are consecutive, so a row of the search window is ONE contiguous range
        if (cellstart != l_cellstart) {
            for (int c = tid; c <= a.ncells; c += SEL_T) { cellstart[c] = 0; if (c < a.ncells) cellfill[c] = 0; }
            __syncthreads();
        }
        for_each_entry([&](int2 r) { if (r.y >= cut) atomicAdd(&cellstart[cell_of(r.x) + 1], 1); });
        __syncthreads();
        ORB_CLK(19);
        {   // inclusive scan of the counts, in place: thread t owns cells [t*per, (t+1)*per)
            const int per = (a.ncells + SEL_T - 1) / SEL_T;
            const int c0 = min(tid * per, a.ncells), c1 = min(c0 + per, a.ncells);
            int local = 0;
            for (int c = c0; c < c1; ++c) local += cellstart[c + 1];
            int tot;
            int run = block_scan_excl(local, sh, tot);
            for (int c = c0; c < c1; ++c) { run += cellstart[c + 1]; cellstart[c + 1] = run; }
        }
        __threadfence_block();
        __syncthreads();
        ORB_CLK(20);
        for_each_entry([&](int2 r) {
            if (r.y >= cut) {
                const int c = cell_of(r.x);
                cand[cellstart[c] + atomicAdd(&cellfill[c], 1)] = (unsigned long long)(unsigned)r.x | ((unsigned long long)(unsigned)r.y << 32);
            }
        });
        __threadfence_block();
        __syncthreads();
        ORB_CLK(12);
        // Suppression radius of candidate i = min(globalMaxR2, min over the candidates j with response_j > response_i * rf + 0.002 of
        // |p_i - p_j|^2).  The reference walks square rings of cells outwards and stops at ring d once (d - 1)^2 * minCellDelta2
        // reaches the running minimum: every candidate it has not visited by then is at least that far away, so the result is
        // exactly this minimum whatever the order of the walk.  Here: the rows of the (2 D + 1)^2 window, nearest first, each row
        // one contiguous member range, D shrinking with the running minimum.
        for (int i = tid; i < M; i += SEL_T) {
            const unsigned long long me = cand[i];
            const int pos = (int)(unsigned)me, x = pos & 0xffff, y = pos >> 16;
            const int cx = udiv((unsigned)((x - minX) * numX), divX), cy = udiv((unsigned)((y - minY) * numY), divY);
            const float s = (float)(int)(me >> 32) * rf + 0.002f;               // response >= 0 always (FAST score); no FMA: contraction is off in this kernel
            int minR2 = globalMaxR2;
            auto ring_max = [&](int r2) { return r2 <= 0 ? -1 : 1 + isqrt_floor((r2 - 1) / minCellDelta2); };   // largest d with max(0, d - 1)^2 * minCellDelta2 < r2
            int D = ring_max(minR2);
            if (narrow_box) {
                // a bounding box narrower than the grid: cells are narrower than the one pixel minCellDelta2 assumes, the ring
                // bound is no lower bound any more and the result depends on the walk -- walk exactly as the reference does
                D = -1;
                for (int d = 0; max(0, d - 1) * max(0, d - 1) * minCellDelta2 < minR2; ++d)
                    for (int yy = -d; yy <= d; ++yy) {
                        const int cYY = yy + cy;
                        if (cYY < 0 || cYY >= numY) continue;
                        for (int xx = -d; xx <= d; ++xx) {
                            const int cXX = xx + cx;
                            if (cXX < 0 || cXX >= numX || max(abs(xx), abs(yy)) != d) continue;
                            const int c = cYY * numX + cXX;
                            for (int q = cellstart[c]; q < cellstart[c + 1]; ++q) {
                                const unsigned long long o = cand[q];
                                if ((float)(int)(o >> 32) > s) {
                                    const int ddx = x - (int)((unsigned)o & 0xffffu), ddy = y - (int)((unsigned)o >> 16);
                                    const int r2 = ddx * ddx + ddy * ddy;
                                    if (r2 < minR2) minR2 = r2;
                                }
                            }
                        }
                    }
            }
            for (int k = 0; k <= 2 * D; ++k) {                                  // row offsets 0, -1, +1, -2, +2, ...
                const int ay = (k + 1) >> 1;
                if (ay > D) break;
                const int cYY = cy + ((k & 1) ? -ay : ay);
                if (cYY < 0 || cYY >= numY) continue;
                const int xl = max(cx - D, 0), xh = min(cx + D, numX - 1);
                const int q1 = cellstart[cYY * numX + xh + 1];
                // four members per trip, fetched together (the loop is a chain of LDS round trips otherwise); the last trip repeats the
                // range's final member, which a minimum does not notice
                const int before = minR2;
                for (int q = cellstart[cYY * numX + xl]; q < q1; q += 4) {
                    unsigned long long o[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) o[u] = cand[min(q + u, q1 - 1)];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ddx = x - (int)((unsigned)o[u] & 0xffffu), ddy = y - (int)((unsigned)o[u] >> 16);
                        const int r2 = ddx * ddx + ddy * ddy;
                        minR2 = ((float)(int)(o[u] >> 32) > s && r2 < minR2) ? r2 : minR2;
                    }
                }
                if (minR2 < before) D = min(D, ring_max(minR2));
            }
            // total order of the output: radius desc, response desc, raster position asc
            if (globalMaxR2 < (1 << 24))
                key[i] = ((unsigned long long)(unsigned)minR2 << 40) | ((unsigned long long)(unsigned)(int)(me >> 32) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)pos);
            else key[i] = (unsigned long long)(unsigned)minR2;               // compared field by field below
        }
        __threadfence_block();
        __syncthreads();
        ORB_CLK(13);
        // ---- keep the N first of the total order; rank = output position
        // (A counting sort by radius that leaves only candidates of equal radius to compare was tried: its six barriers and two atomic
        // passes cost more than this loop, 34 k cycles against 18 k per frame.)
        // (Measured and left out in round 5: the transposed form -- a wavefront holds 64 keys one per lane, key_i comes out of its lane into
        // scalar registers, one 64-bit compare tests it against 64 keys and the scalar unit counts the mask.  One vector instruction per
        // 64 pairs instead of two, but two SCALAR ones, and a SIMD issues those no faster than vector ones: 18 k -> 25 k cycles.)
        for (int i = tid; i < M; i += SEL_T) {
            int rank = 0;
            const unsigned long long ki = key[i];
            if (globalMaxR2 < (1 << 24)) {
                int j = 0;
                for (; j + 8 <= M; j += 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) rank += key[j + u] > ki;
                }
                for (; j < M; ++j) rank += key[j] > ki;
            } else {
                const unsigned long long ci = cand[i];
                const unsigned si = (unsigned)(ci >> 32), pi = (unsigned)ci;
                for (int j = 0; j < M; ++j) {
                    const unsigned long long kj = key[j], cj = cand[j];
                    const unsigned sj = (unsigned)(cj >> 32), pj = (unsigned)cj;
                    rank += (kj > ki) || (kj == ki && (sj > si || (sj == si && pj < pi)));
                }
            }
            if (rank < N && rank < a.capacity) {
                const unsigned long long c = cand[i];
                okp[rank] = make_kp(make_int2((int)(unsigned)c, (int)(c >> 32)), size_f);
            }
        }
        __syncthreads();
        ORB_CLK(14);
    };
    if (in_lds) suppress_and_rank(l_cand, l_cellstart, l_cellfill, l_key);
    else suppress_and_rank(a.cand64 + (size_t)f * a.scratch_cap, a.cell_start + (size_t)f * (a.ncells + 1), a.cell_fill + (size_t)f * (a.ncells + 1),
                           a.key64 + (size_t)f * a.scratch_cap);
    if (tid == 0) a.out_count[f] = min(N, a.capacity);
}

// ---------------------------------------------------------------------------------------------
// separable integer Gaussian, REFLECT_101, 64x16 output tile per workgroup
// ---------------------------------------------------------------------------------------------
constexpr int BT_W = 64, BT_H = 64, MAXR = 7;

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

// byte gathers straight from the blurred image: kept for patches too large for the LDS form below
__global__ __launch_bounds__(256) void k_brief_gather(const uint8_t* __restrict__ blurred, int wp, int h, const mage_keypoint* __restrict__ kps,
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

// The (2 R + 1)^2 patch of every keypoint (R = largest |coordinate| of the table) is first copied to LDS by rows -- aligned 32-bit
// loads, a row is one or two cache lines -- and the 512 samples are LDS byte reads: eight 64-lane byte GATHERS per keypoint from L2
// kept the texture addresser busy for ~11 cycles each, two row-wise loads and eight LDS reads do not (0.52 -> 0.28 ms per 2048 frames).
__global__ __launch_bounds__(256) void k_brief(const uint8_t* __restrict__ blurred, int wp, int h, const mage_keypoint* __restrict__ kps,
                                               const int* __restrict__ counts, int capacity, const signed char* __restrict__ pattern,
                                               uint8_t* __restrict__ desc, int R, int ndw, int lanes_per_row_log2)
{
    extern __shared__ uint8_t patches[];                // [wavefront][keypoint of the wavefront][row][4 ndw]
    const int f = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k0 = (blockIdx.x * 4 + wave) * BRIEF_KPW;
    const int n = counts[f];
    if (k0 >= n) return;
    const uint8_t* frame = blurred + (size_t)f * wp * h;
    const int rows = 2 * R + 1, pitch = 4 * ndw, patch_bytes = rows * pitch;
    uint8_t* mine = patches + (size_t)wave * BRIEF_KPW * patch_bytes;
    // cvRound of an integer-valued float; angleIncrement = cvRound(angle / 12) % 30 picks the pre-rotated table row
    // (OpenCVModified.cpp:523-532; angle is 0 unless UseOrientation)
    int inc[BRIEF_KPW], off[BRIEF_KPW];                 // table row; LDS offset of the patch centre
    const int lr = lane >> lanes_per_row_log2, ld = lane & ((1 << lanes_per_row_log2) - 1), rows_per_pass = 64 >> lanes_per_row_log2;
#pragma unroll
    for (int q = 0; q < BRIEF_KPW; ++q) {
        const int k = min(k0 + q, n - 1);
        const mage_keypoint kp = kps[(size_t)f * capacity + k];
        const int x = (int)rintf(kp.x), y = (int)rintf(kp.y);
        inc[q] = (int)rintf(kp.angle / 12.0f) % 30;
        const int ax = (x - R) & ~3;                    // the patch lies inside the image (RunByImageBorder), the aligned reads inside the padded rows
        off[q] = q * patch_bytes + R * pitch + (x - ax);
        const uint8_t* src = frame + (size_t)(y - R) * wp + ax;
        if (ld < ndw)
            for (int r = lr; r < rows; r += rows_per_pass)
                *reinterpret_cast<uint32_t*>(mine + q * patch_bytes + r * pitch + 4 * ld) = *reinterpret_cast<const uint32_t*>(src + (size_t)r * wp + 4 * ld);
    }
    // this lane's four pairs of the table row: 16 signed bytes, one 128-bit load (rows are 1024 bytes, lane * 16 is aligned)
    // ... decoded to LDS offsets once per table row: without orientation every keypoint uses row 0
    int o0[4], o1[4];
    auto decode = [&](int row) {
        const int4 pr = *reinterpret_cast<const int4*>(pattern + row * 1024 + lane * 16);
        const int w4[4] = { pr.x, pr.y, pr.z, pr.w };
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int x0 = (int)(signed char)(w4[b] & 0xff), y0 = (int)(signed char)((w4[b] >> 8) & 0xff);
            const int x1 = (int)(signed char)((w4[b] >> 16) & 0xff), y1 = (int)(signed char)((w4[b] >> 24) & 0xff);
            o0[b] = y0 * pitch + x0; o1[b] = y1 * pitch + x1;
        }
    };
    decode(inc[0]);
#pragma unroll
    for (int q = 0; q < BRIEF_KPW; ++q) {
        if (q > 0 && inc[q] != inc[q - 1]) decode(inc[q]);
        const uint8_t* c = mine + off[q];
        int nib = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) nib |= ((int)c[o0[b]] < (int)c[o1[b]]) << b;
        const int hi = __shfl_down(nib, 1, 64);
        if ((lane & 1) == 0 && k0 + q < n) desc[((size_t)f * capacity + k0 + q) * 32 + (lane >> 1)] = (uint8_t)(nib | (hi << 4));
    }
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace

void orb_fast_tiling(int w, int h, int* tiles_x, int* tiles_y, int* tile_cap)
{
    *tiles_x = cdiv((w + 3) & ~3, FT_W); *tiles_y = cdiv(h, FT_H); *tile_cap = F_MAXKP;
}

bool orb_blur_fuses(const OrbTaps& taps) { return taps.radius == 3 && taps.t[3] < 256; }

// The banded tap matrices of the matrix-core form of the fused blur (k_fast_keypoints<BLUR_MFMA>), as the 8-byte i8 operands of
// v_mfma_i32_16x16x32_i8: entry [lane] = B1 (pass 1: K slot 8 g + b = window column 16 J + 8 + 8 g + b, output column j = lane & 15),
// [64 + lane], [128 + lane] = B2 for output rows 0 .. 15 / 16 .. 31 (K slot b of lane group g = window row 4 g + b for b < 4,
// 16 + 4 g + b - 4 otherwise; output row 16 Q + (lane & 15) sits at window row 4 + that).  False when the taps do not fit the i8 /
// 16-bit arithmetic (a tap > 127 or a sum > 257): the caller then takes the vector-ALU form.
bool orb_blur_mfma_table(const OrbTaps& taps, unsigned long long* tab192, int* c2)
{
    if (!orb_blur_fuses(taps)) return false;
    int T = 0;
    for (int k = 0; k < 7; ++k) { if (taps.t[k] < 0 || taps.t[k] > 127) return false; T += taps.t[k]; }
    if (T > 257) return false;
    auto tap = [&](int idx) -> unsigned long long { return idx >= 0 && idx <= 6 ? (unsigned long long)taps.t[idx] : 0ull; };
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, j = lane & 15;
        unsigned long long b1 = 0, b2[2] = { 0, 0 };
        for (int b = 0; b < 8; ++b) {
            b1 |= tap(8 * g + b - j - 5) << (8 * b);
            const int row = b < 4 ? 4 * g + b : 16 + 4 * g + b - 4;
            for (int Q = 0; Q < 2; ++Q) b2[Q] |= tap(row - 1 - (16 * Q + j)) << (8 * b);
        }
        tab192[lane] = b1; tab192[64 + lane] = b2[0]; tab192[128 + lane] = b2[1];
    }
    *c2 = 128 * T * T + (1 << 15);
    return true;
}

void orb_launch_fast(const uint8_t* img, int w, int h, int stride, size_t frame_stride, int n_frames, int threshold, int border, uint8_t* raw_frame0, int wp,
                     int2* raw, int* tile_count, const OrbTaps* taps, const unsigned long long* blur_tab, int blur_c2, uint8_t* blurred, hipStream_t st)
{
    const dim3 grid(cdiv(wp, FT_W), cdiv(h, FT_H), n_frames);
    if (taps && blur_tab && !(MAGE_ORB_ABLATE & 8))
        hipLaunchKernelGGL(k_fast_keypoints<BLUR_MFMA>, grid, dim3(256), 0, st, img, w, h, stride, frame_stride, threshold, border, raw_frame0, wp, raw, tile_count,
                           *taps, blurred, blur_tab, blur_c2);
    else if (taps && !(MAGE_ORB_ABLATE & 8))
        hipLaunchKernelGGL(k_fast_keypoints<BLUR_VALU>, grid, dim3(256), 0, st, img, w, h, stride, frame_stride, threshold, border, raw_frame0, wp, raw, tile_count,
                           *taps, blurred, blur_tab, blur_c2);
    else hipLaunchKernelGGL(k_fast_keypoints<BLUR_NONE>, grid, dim3(256), 0, st, img, w, h, stride, frame_stride, threshold, border, raw_frame0, wp, raw, tile_count,
                            OrbTaps{}, blurred, blur_tab, blur_c2);
}

void orb_launch_select(const OrbSelectArgs& a, int n_frames, hipStream_t st)
{
    hipLaunchKernelGGL(k_select, dim3(n_frames), dim3(SEL_T), 0, st, a);
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
                      const signed char* pattern, int pattern_radius, uint8_t* desc, bool rotate_random, hipStream_t st)
{
    if (rotate_random) { hipLaunchKernelGGL(k_brief_rotated, dim3(cdiv(capacity, 4), n_frames), dim3(256), 0, st, blurred, wp, h, kps, counts, capacity, pattern, desc); return; }
    const int R = pattern_radius, ndw = (2 * R + 1 + 3 + 3) / 4;          // a row of 2 R + 1 bytes starting up to 3 bytes into its first dword
    int lg = 0;
    while ((1 << lg) < ndw) ++lg;
    const size_t lds = (size_t)4 * BRIEF_KPW * (2 * R + 1) * 4 * ndw;
    if (lds > 48 * 1024) { hipLaunchKernelGGL(k_brief_gather, dim3(cdiv(capacity, 4 * BRIEF_KPW), n_frames), dim3(256), 0, st, blurred, wp, h, kps, counts, capacity, pattern, desc); return; }
    hipLaunchKernelGGL(k_brief, dim3(cdiv(capacity, 4 * BRIEF_KPW), n_frames), dim3(256), lds, st, blurred, wp, h, kps, counts, capacity, pattern, desc, R, ndw, lg);
}

}  // namespace mage

#ifdef MAGE_ORB_CLOCKS
// probe build only: read (and optionally clear) the phase clocks
extern "C" __attribute__((visibility("default"))) int mage_orb_debug_clocks(unsigned long long* out32, int reset)
{
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(mage::g_orb_clk), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
    if (reset) { unsigned long long z[32] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(mage::g_orb_clk), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#endif
