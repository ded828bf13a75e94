/*
 * orb_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C, single-threaded restatement of the reference's ORB front-end:
 *
 *   Core/MAGESLAM/Source/Image/OpenCVModified.cpp
 *       :890-921   makeOffsets                    (Bresenham radius-3 ring)
 *       :926-1071  cornerScore<16>                (scalar branch :1030-1064)
 *       :1224-1512 FAST_t<16>                     (scalar branch :1415-1479, 3x3 NMS :1488-1510)
 *       :619-639   RunByImageBorder
 *       :571-617   RetainBestFeatures
 *       :144-360   AdaptiveNonMaximalSuppresion
 *       :642-761   ComputeKeyPoints               (single level; angle = 0 when !UseOrientation)
 *       :771-886   DetectAndCompute
 *       :502-549   ComputeOrbDescriptorsPrerotated + the pattern tables :74-138
 *   OpenCV 3.4.0 (external, not vendored): cv::GaussianBlur on CV_8U -- restated per SURVEY.md appendix A.7
 *
 * Where the reference leaves the result implementation-defined, this file (and the HIP path, which must
 * match it bit for bit) pins a canonical choice, documented at each site:
 *   (C1) std::nth_element in RetainBestFeatures keeps a well-defined SET; the kept points stay in raster order.
 *   (C2) std::nth_element in ANMS keeps a well-defined set given (C1)'s indices; the output is sorted by the
 *        comparator itself (suppression radius desc, strength desc, index asc).
 *   (C3) GaussianBlur: 8-bit fixed-point separable filter with taps cvRound(256 * getGaussianKernel(k, 2)),
 *        BORDER_REFLECT_101 treated as isolated.  Equality with the real OpenCV 3.4.0 binary is UNVERIFIED
 *        (OpenCV is not in the image): "parity unpinned" for the blur.
 * Out of scope here (rejected with ORBO_UNSUPPORTED): NumLevels > 1 (cv::resize
 * pyramid), UseOrientation (ICAngles/fastAtan2), patch sizes other than 15 / 31 (cv::RNG pattern).
 *
 * PARITY UNPINNED by the reference (no tests or golden vectors for ORB); pinned instead
 * by an independent numpy implementation (oracle/indep/orb_numpy.py) through tests/golden/orb_*.npz and by
 * known-answer properties (tests/test_orb_oracle.py).
 */
#define _GNU_SOURCE
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mage_brief_patterns.h"

#define ORBO_API __attribute__((visibility("default")))
#define ORBO_OK 0
#define ORBO_UNSUPPORTED (-4)

typedef struct { float x, y, size, angle, response; int octave, class_id; } orbo_keypoint;   /* cv::KeyPoint */

typedef struct {
    unsigned gaussian_kernel_size, nfeatures;
    float scale_factor;
    unsigned nlevels, patch_size, fast_threshold;
    int use_orientation;
    float feature_factor, feature_strength;
    int strong_response;
    float min_robust, max_robust;
    int cells_x, cells_y;
} orbo_params;                                   /* OrbDetector ctor arguments, OpenCVModified.h:68-82 */

/* cvRound: round half to even */
static int cv_round(double v) { return (int)nearbyint(v); }

/* ------------------------------------------------------------------------------------------ */
/* pattern tables                                                                             */
/* ------------------------------------------------------------------------------------------ */
/* MakeRandomPattern (OpenCVModified.cpp:551-560) for patch sizes without a pre-rotated table: 512 points from cv::RNG(0x34985739)
 * (OpenCV's multiply-with-carry generator, core/operations.hpp: state = (unsigned)state * 4164903690 + (state >> 32)),
 * uniform(a, b) = a + next() % (b - a).  Point i pairs as (2i, 2i+1), the same bit order as the pre-rotated tables. */
ORBO_API void orbo_random_pattern(int patch, signed char* out /* 1024: x0 y0 x1 y1 per pair */)
{
    uint64_t state = 0x34985739u;
    const int a = -patch / 2, b = patch / 2 + 1;
    for (int i = 0; i < 1024; ++i) {                       /* x then y of each of the 512 points */
        state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
        out[i] = (signed char)(a == b ? a : (int)((uint32_t)state % (uint32_t)(b - a) + (uint32_t)a));
    }
}

ORBO_API void orbo_pattern_expand(int patch, signed char* out /* 30 * 1024 */)
{
    if (patch != 15 && patch != 31) {                      /* random pattern: row 0 (the unrotated points); rotation, if any, is per keypoint */
        memset(out, 0, 30 * 1024);
        orbo_random_pattern(patch, out);
        return;
    }
    const signed char* base = patch == 31 ? MAGE_BRIEF_BASE_31 : MAGE_BRIEF_BASE_15;
    for (int k = 0; k < MAGE_BRIEF_ROTATIONS; ++k) {
        double a = k * 12.0 * M_PI / 180.0, c = cos(a), s = sin(a);
        for (int p = 0; p < 512; ++p) {
            double bx = base[p * 2], by = base[p * 2 + 1];
            double v[2] = { bx * c - by * s, bx * s + by * c };
            for (int q = 0; q < 2; ++q) {
                double h = nearbyint(v[q] * 2.0) / 2.0;
                if (fabs(v[q] - h) < 1e-9) v[q] = h;
                out[k * 1024 + p * 2 + q] = (signed char)nearbyint(v[q]);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* FAST-9/16                                                                                  */
/* ------------------------------------------------------------------------------------------ */
static void make_offsets(int pixel[25], int stride)
{
    static const int o[16][2] = { { 0, 3 }, { 1, 3 }, { 2, 2 }, { 3, 1 }, { 3, 0 }, { 3, -1 }, { 2, -2 }, { 1, -3 },
                                  { 0, -3 }, { -1, -3 }, { -2, -2 }, { -3, -1 }, { -3, 0 }, { -3, 1 }, { -2, 2 }, { -1, 3 } };
    int k = 0;
    for (; k < 16; ++k) pixel[k] = o[k][0] + o[k][1] * stride;
    for (; k < 25; ++k) pixel[k] = pixel[k - 16];
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static int corner_score16(const uint8_t* ptr, const int pixel[25], int threshold)
{
    int d[25], v = ptr[0];
    for (int k = 0; k < 25; ++k) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = imin(d[k + 1], d[k + 2]);
        a = imin(a, d[k + 3]);
        if (a <= a0) continue;
        a = imin(a, d[k + 4]); a = imin(a, d[k + 5]); a = imin(a, d[k + 6]); a = imin(a, d[k + 7]); a = imin(a, d[k + 8]);
        a0 = imax(a0, imin(a, d[k]));
        a0 = imax(a0, imin(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = imax(d[k + 1], d[k + 2]);
        b = imax(b, d[k + 3]); b = imax(b, d[k + 4]); b = imax(b, d[k + 5]);
        if (b >= b0) continue;
        b = imax(b, d[k + 6]); b = imax(b, d[k + 7]); b = imax(b, d[k + 8]);
        b0 = imin(b0, imax(b, d[k]));
        b0 = imin(b0, imax(b, d[k + 9]));
    }
    return -b0 - 1;
}

/* score[y*w + x] = (uchar)cornerScore for FAST corners, 0 elsewhere (what the rolling `curr` rows hold) */
ORBO_API void orbo_fast_score_map(const uint8_t* img, int w, int h, int stride, int threshold, uint8_t* score)
{
    memset(score, 0, (size_t)w * h);
    int pixel[25];
    make_offsets(pixel, stride);
    uint8_t tab[512];
    for (int t = -255; t <= 255; ++t) tab[t + 255] = (uint8_t)(t < -threshold ? 1 : t > threshold ? 2 : 0);
    int thr = imin(imax(threshold, 0), 255);
    const int K = 8, N = 25;
    for (int i = 3; i < h - 3; ++i) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        for (int j = 3; j < w - 3; ++j, ++ptr) {
            int v = ptr[0];
            const uint8_t* t = &tab[0] - v + 255;
            int d = t[ptr[pixel[0]]] | t[ptr[pixel[8]]];
            if (d == 0) continue;
            d &= t[ptr[pixel[2]]] | t[ptr[pixel[10]]];
            d &= t[ptr[pixel[4]]] | t[ptr[pixel[12]]];
            d &= t[ptr[pixel[6]]] | t[ptr[pixel[14]]];
            if (d == 0) continue;
            d &= t[ptr[pixel[1]]] | t[ptr[pixel[9]]];
            d &= t[ptr[pixel[3]]] | t[ptr[pixel[11]]];
            d &= t[ptr[pixel[5]]] | t[ptr[pixel[13]]];
            d &= t[ptr[pixel[7]]] | t[ptr[pixel[15]]];
            int is_corner = 0;
            if (d & 1) {
                int vt = v - thr, count = 0;
                for (int k = 0; k < N; ++k) {
                    if (ptr[pixel[k]] < vt) { if (++count > K) { is_corner = 1; break; } }
                    else count = 0;
                }
            }
            if (!is_corner && (d & 2)) {
                int vt = v + thr, count = 0;
                for (int k = 0; k < N; ++k) {
                    if (ptr[pixel[k]] > vt) { if (++count > K) { is_corner = 1; break; } }
                    else count = 0;
                }
            }
            if (is_corner) score[(size_t)i * w + j] = (uint8_t)corner_score16(ptr, pixel, thr);
        }
    }
}

typedef struct { int x, y, resp; } raw_kp;

/* 3x3 strict-greater NMS in raster order (OpenCVModified.cpp:1488-1510); returns count.
 * Note the reference tests membership in the corner list, not score != 0. */
static size_t fast_nms(const uint8_t* img, int w, int h, int stride, int threshold, const uint8_t* score, raw_kp* out, size_t cap)
{
    size_t n = 0;
    (void)img; (void)stride; (void)threshold;
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            int s = score[(size_t)y * w + x];
            /* a corner whose score is 0 (threshold 0) can never win a strict comparison against a zero neighbour,
               so testing s against its neighbours alone is equivalent to walking the corner list */
            const uint8_t* p = score + (size_t)y * w + x;
            if (s > p[1] && s > p[-1] && s > p[-w - 1] && s > p[-w] && s > p[-w + 1] && s > p[w - 1] && s > p[w] && s > p[w + 1]) {
                if (n < cap) { out[n].x = x; out[n].y = y; out[n].resp = s; }
                ++n;
            }
        }
    return n;
}

ORBO_API int orbo_fast_keypoints(const uint8_t* img, int w, int h, int stride, int threshold, int* xyr /* cap x 3 */, int cap)
{
    uint8_t* score = (uint8_t*)malloc((size_t)w * h + 1);
    raw_kp* kp = (raw_kp*)malloc(sizeof(raw_kp) * ((size_t)w * h / 4 + 16));
    orbo_fast_score_map(img, w, h, stride, threshold, score);
    size_t n = fast_nms(img, w, h, stride, threshold, score, kp, (size_t)w * h / 4 + 16);
    for (size_t i = 0; i < n && (int)i < cap; ++i) { xyr[i * 3] = kp[i].x; xyr[i * 3 + 1] = kp[i].y; xyr[i * 3 + 2] = kp[i].resp; }
    free(score); free(kp);
    return (int)n;
}

/* ------------------------------------------------------------------------------------------ */
/* RetainBestFeatures (C1) and AdaptiveNonMaximalSuppresion (C2)                              */
/* ------------------------------------------------------------------------------------------ */
static size_t retain_best(raw_kp* kp, size_t n, int min_threshold, int max_num, int min_num, float response_factor)
{
    unsigned hist[256];
    memset(hist, 0, sizeof(hist));
    for (size_t i = 0; i < n; ++i) hist[imin(imax(kp[i].resp, 0), 255)]++;
    size_t min_num_threshold = (size_t)min_threshold;
    int num = 0;
    for (int i = 255; i >= min_threshold; --i) {
        num += (int)hist[i];
        if (num >= min_num) { min_num_threshold = (size_t)i; break; }
    }
    num = 0;
    int lower = imax((int)(min_num_threshold * response_factor), min_threshold);
    int cut = lower;
    for (int i = 255; i >= lower; --i) {
        num += (int)hist[i];
        if (num >= max_num) { cut = i; break; }
    }
    /* the top `num` responses are exactly the points with response >= cut (whole histogram bins) */
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) if (kp[i].resp >= cut) kp[m++] = kp[i];
    return m;
}

typedef struct { int x, y; float strength; int r, idx, next; } anms_item;

static int anms_cmp(const void* pa, const void* pb)
{
    const anms_item* a = (const anms_item*)pa; const anms_item* b = (const anms_item*)pb;
    if (a->r != b->r) return a->r > b->r ? -1 : 1;
    if (a->strength != b->strength) return a->strength > b->strength ? -1 : 1;
    return a->idx < b->idx ? -1 : (a->idx > b->idx);
}

static size_t anms(raw_kp* kp, size_t n_before, unsigned num_to_keep, int threshold, const orbo_params* P)
{
    if (num_to_keep > n_before) return n_before;
    const int numX = P->cells_x, numY = P->cells_y;
    const float ROBUST_EPS = 0.002f;
    anms_item* it = (anms_item*)malloc(sizeof(anms_item) * n_before);
    int minX = kp[0].x, maxX = kp[0].x, minY = kp[0].y, maxY = kp[0].y;
    float minStrength = (float)kp[0].resp;
    for (size_t i = 0; i < n_before; ++i) {
        it[i].idx = (int)i; it[i].strength = (float)kp[i].resp; it[i].x = kp[i].x; it[i].y = kp[i].y; it[i].r = 0; it[i].next = -1;
        minX = imin(minX, it[i].x); minY = imin(minY, it[i].y); maxX = imax(maxX, it[i].x); maxY = imax(maxY, it[i].y);
        if (it[i].strength < minStrength) minStrength = it[i].strength;
    }
    float rf, rfInv;
    {
        float hi = (float)P->strong_response - (float)threshold;
        float val = minStrength - (float)threshold;
        val = val < 0.0f ? 0.0f : (val > hi ? hi : val);
        float range = P->max_robust - P->min_robust; if (range < 0.0f) range = 0.0f;
        rf = P->max_robust - (val / (float)(P->strong_response - threshold)) * range;
        rfInv = 1.0f / rf;
    }
    int* cells = (int*)malloc(sizeof(int) * (size_t)numX * numY);
    for (int c = 0; c < numX * numY; ++c) cells[c] = -1;
    for (size_t i = 0; i < n_before; ++i) {
        int cellX = (it[i].x - minX) * numX / (maxX + 1 - minX);
        int cellY = (it[i].y - minY) * numY / (maxY + 1 - minY);
        int b = cellY * numX + cellX;
        if (cells[b] < 0) { cells[b] = (int)i; continue; }
        if (it[cells[b]].strength < it[i].strength) { it[i].next = cells[b]; cells[b] = (int)i; continue; }
        int after = cells[b];
        while (it[after].next >= 0 && it[it[after].next].strength > it[i].strength) after = it[after].next;
        it[i].next = it[after].next;
        it[after].next = (int)i;
    }
    int globalMaxR2 = (int)(((double)(maxX - minX)) * ((double)(maxY - minY)) / (double)num_to_keep);
    int minCellDelta2;
    {
        int dx = imax((maxX - minX) / numX, 1), dy = imax((maxY - minY) / numY, 1);
        int m = imin(dx, dy);
        minCellDelta2 = m * m;
    }
    for (int cy = 0; cy < numY; ++cy)
        for (int cx = 0; cx < numX; ++cx)
            for (int item = cells[cy * numX + cx]; item >= 0; item = it[item].next) {
                int minR2 = globalMaxR2;
                float s = (it[item].strength >= 0) ? (it[item].strength * rf + ROBUST_EPS) : (it[item].strength * rfInv + ROBUST_EPS);
                for (int d = 0; imax(0, d - 1) * imax(0, d - 1) * minCellDelta2 < minR2; ++d)
                    for (int yy = -d; yy <= d; ++yy) {
                        int cYY = yy + cy;
                        if (cYY < 0 || cYY >= numY) continue;
                        for (int xx = -d; xx <= d; ++xx) {
                            int cXX = xx + cx;
                            if (cXX < 0 || cXX >= numX || imax(abs(xx), abs(yy)) != d) continue;
                            int other = cells[cYY * numX + cXX];
                            while (other >= 0 && it[other].strength > s) {
                                int ddx = it[item].x - it[other].x, ddy = it[item].y - it[other].y;
                                int r = ddx * ddx + ddy * ddy;
                                if (r < minR2) minR2 = r;
                                other = it[other].next;
                            }
                        }
                    }
                it[item].r = minR2;
            }
    qsort(it, n_before, sizeof(anms_item), anms_cmp);           /* (C2): total order, unique idx */
    raw_kp* tmp = (raw_kp*)malloc(sizeof(raw_kp) * num_to_keep);
    for (unsigned i = 0; i < num_to_keep; ++i) tmp[i] = kp[it[i].idx];
    memcpy(kp, tmp, sizeof(raw_kp) * num_to_keep);
    free(tmp); free(cells); free(it);
    return num_to_keep;
}

/* suppression radii only, in input order (for the stage tests) */
ORBO_API int orbo_select(const orbo_params* P, int* xyr, int n, int* out_xyr)
{
    raw_kp* kp = (raw_kp*)malloc(sizeof(raw_kp) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) { kp[i].x = xyr[i * 3]; kp[i].y = xyr[i * 3 + 1]; kp[i].resp = xyr[i * 3 + 2]; }
    size_t m = (size_t)n;
    if (m > P->nfeatures) {
        int max_num = (int)(P->nfeatures * P->feature_factor);
        m = retain_best(kp, m, (int)P->fast_threshold, max_num, (int)P->nfeatures, P->feature_strength);
        m = anms(kp, m, P->nfeatures, (int)P->fast_threshold, P);
    }
    for (size_t i = 0; i < m; ++i) { out_xyr[i * 3] = kp[i].x; out_xyr[i * 3 + 1] = kp[i].y; out_xyr[i * 3 + 2] = kp[i].resp; }
    free(kp);
    return (int)m;
}

/* ------------------------------------------------------------------------------------------ */
/* GaussianBlur k x k, sigma 2, 8-bit fixed point (C3)                                        */
/* ------------------------------------------------------------------------------------------ */
ORBO_API void orbo_gaussian_taps(int ksize, int* taps)
{
    /* getGaussianKernel(ksize, 2, CV_32F): float taps normalised with a double sum, then convertTo(CV_32S, 256) */
    float cf[64];
    double sum = 0, sigma = 2.0, scale2X = -0.5 / (sigma * sigma);
    for (int i = 0; i < ksize; ++i) {
        double x = i - (ksize - 1) * 0.5;
        cf[i] = (float)exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < ksize; ++i) { cf[i] = (float)(cf[i] * sum); taps[i] = cv_round((double)cf[i] * 256.0); }
}

static int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2 * (n - 1) - p; }
    return p;
}

ORBO_API void orbo_blur(const uint8_t* src, int w, int h, int stride, int ksize, uint8_t* dst /* w x h, pitch w */)
{
    int taps[64];
    orbo_gaussian_taps(ksize, taps);
    const int r = ksize / 2;
    int* rowbuf = (int*)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int t = -r; t <= r; ++t) acc += taps[t + r] * src[(size_t)y * stride + reflect101(x + t, w)];
            rowbuf[(size_t)y * w + x] = acc;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int acc = 0;
            for (int t = -r; t <= r; ++t) acc += taps[t + r] * rowbuf[(size_t)reflect101(y + t, h) * w + x];
            int v = (acc + (1 << 15)) >> 16;
            dst[(size_t)y * w + x] = (uint8_t)(v > 255 ? 255 : v);
        }
    free(rowbuf);
}

/* ------------------------------------------------------------------------------------------ */
/* ICAngles (OpenCVModified.cpp:399-437) + the u_max table (:672-688) + cv::fastAtan2 (OpenCV  */
/* 3.4.0 core, mathfuncs: 7th-order odd polynomial in float, degrees; not vendored: restated)  */
/* ------------------------------------------------------------------------------------------ */
static float fast_atan2f(float y, float x)
{
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale, p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

ORBO_API void orbo_umax(int half, int* umax /* half + 2 */)
{
    int v, v0;
    const int vmax = (int)floor(half * sqrtf(2.f) / 2 + 1);
    const int vmin = (int)ceil(half * sqrtf(2.f) / 2);
    for (v = 0; v <= half + 1; ++v) umax[v] = 0;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round(sqrt((double)half * half - (double)v * v));
    for (v = half, v0 = 0; v >= vmin; --v) {          /* make sure we are symmetric */
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

ORBO_API void orbo_ic_angles(const uint8_t* img, int stride, orbo_keypoint* kps, int n, int half)
{
    int umax[64];
    orbo_umax(half, umax);
    for (int i = 0; i < n; ++i) {
        const uint8_t* center = img + (size_t)cv_round(kps[i].y) * stride + cv_round(kps[i].x);
        int m_01 = 0, m_10 = 0;
        for (int u = -half; u <= half; ++u) m_10 += u * center[u];
        for (int v = 1; v <= half; ++v) {
            int v_sum = 0;
            const int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                const int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        kps[i].angle = fast_atan2f((float)m_01, (float)m_10);
    }
}

ORBO_API float orbo_fast_atan2(float y, float x) { return fast_atan2f(y, x); }

/* ------------------------------------------------------------------------------------------ */
/* cv::resize(src, dst, size, 0, 0, INTER_LINEAR) for CV_8UC1, OpenCV 3.4.0 (imgproc/resize.cpp: */
/* resizeGeneric_ with HResizeLinear / VResizeLinear<uchar, int, short>, fixed point with 11      */
/* coefficient bits; not vendored, restated).  Used only for the pyramid levels >= 1 (:822-841). */
/* ------------------------------------------------------------------------------------------ */
static short sat_short(float v)
{
    int r = cv_round((double)v);
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

ORBO_API void orbo_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh /* dst pitch dw */)
{
    const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int* xofs = (int*)malloc(sizeof(int) * (size_t)dw);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * (size_t)dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[dx * 2] = sat_short((1.f - fx) * 2048);
        ialpha[dx * 2 + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = (int)floorf(fy);
        fy -= sy;
        const short b0 = sat_short((1.f - fy) * 2048), b1 = sat_short(fy * 2048);
        const int r0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy), r1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        const uint8_t* S0r = src + (size_t)r0 * sstride; const uint8_t* S1r = src + (size_t)r1 * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            const int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1];
            const int S0 = S0r[sx] * a0 + S0r[sx1] * a1, S1 = S1r[sx] * a0 + S1r[sx1] * a1;
            dst[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
        }
    }
    free(xofs); free(ialpha);
}

/* ------------------------------------------------------------------------------------------ */
/* DetectAndCompute (OpenCVModified.cpp:771-886) + ComputeKeyPoints (:642-768)                 */
/* Multi-level deviation, PINNED: the reference blurs every level in place as an ROI of one    */
/* shared pyramid buffer without BORDER_ISOLATED, so border taps read the neighbouring level   */
/* or uninitialised memory; here every level is blurred as an isolated image (REFLECT_101).    */
/* ------------------------------------------------------------------------------------------ */
#define ORBO_MAX_LEVELS 16

ORBO_API int orbo_level_layout(const orbo_params* P, int w, int h, int* lw, int* lh, float* lscale, int* nfeat /* each ORBO_MAX_LEVELS */)
{
    const int L = (int)P->nlevels;
    if (L < 1 || L > ORBO_MAX_LEVELS) return ORBO_UNSUPPORTED;
    for (int l = 0; l < L; ++l) {
        lscale[l] = (float)pow((double)P->scale_factor, (double)l);               /* getScale, :564-567 */
        lw[l] = cv_round(w / lscale[l]); lh[l] = cv_round(h / lscale[l]);          /* :799 (int / float -> float) */
    }
    /* features per level, :659-669 */
    const float factor = 1.0f / P->scale_factor;
    float ndesired = (float)P->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; ++l) { nfeat[l] = cv_round(ndesired); sum += nfeat[l]; ndesired *= factor; }
    nfeat[L - 1] = (int)P->nfeatures - sum > 0 ? (int)P->nfeatures - sum : 0;
    return ORBO_OK;
}

ORBO_API int orbo_detect(const orbo_params* P, const uint8_t* img, int w, int h, int stride,
                         orbo_keypoint* kps, uint8_t* desc32, int cap, int* n_out, uint8_t* blurred_out /* optional w*h, level 0 */)
{
    *n_out = 0;
    /* other patch sizes take the random pattern (ComputeOrbDescriptors, :452-492); with angle 0 its rotation is the identity
       (cos 0 = 1, sin 0 = 0 exactly); with UseOrientation every keypoint rotates the 512 points by its own angle, see below */
    if (P->patch_size != 15 && P->patch_size != 31 && (P->patch_size < 2 || P->patch_size > 127)) return ORBO_UNSUPPORTED;
    int lw[ORBO_MAX_LEVELS], lh[ORBO_MAX_LEVELS], nfeat[ORBO_MAX_LEVELS];
    float lscale[ORBO_MAX_LEVELS];
    if (orbo_level_layout(P, w, h, lw, lh, lscale, nfeat) != ORBO_OK) return ORBO_UNSUPPORTED;
    const int L = (int)P->nlevels;
    if (L == 1) nfeat[0] = (int)P->nfeatures;
    const int half_patch = (int)P->patch_size / 2;
    /* with orientation the patch is rotated: the hypotenuse of half the patch (OpenCVModified.cpp:709-712) */
    const int half = P->use_orientation ? (int)ceil(half_patch * sqrtf(2.0f)) : half_patch;
    /* pyramid: level 0 = the image, level l = resize of level l - 1 */
    const uint8_t* lev[ORBO_MAX_LEVELS]; int lstride[ORBO_MAX_LEVELS]; uint8_t* own[ORBO_MAX_LEVELS];
    lev[0] = img; lstride[0] = stride; own[0] = NULL;
    for (int l = 1; l < L; ++l) {
        own[l] = (uint8_t*)malloc((size_t)(lw[l] > 0 ? lw[l] : 1) * (size_t)(lh[l] > 0 ? lh[l] : 1));
        if (lw[l] > 0 && lh[l] > 0 && lw[l - 1] > 0 && lh[l - 1] > 0) orbo_resize_linear(lev[l - 1], lw[l - 1], lh[l - 1], lstride[l - 1], own[l], lw[l], lh[l]);
        lev[l] = own[l]; lstride[l] = lw[l];
    }
    size_t total = 0;
    for (int l = 0; l < L; ++l) {
        const int W = lw[l], H = lh[l];
        if (W < 7 || H < 7 || nfeat[l] < 1) continue;      /* a level without quota places nothing (the reference would assert in ANMS) */
        uint8_t* score = (uint8_t*)malloc((size_t)W * H + 1);
        size_t raw_cap = (size_t)W * H / 4 + 16;
        raw_kp* kp = (raw_kp*)malloc(sizeof(raw_kp) * raw_cap);
        orbo_fast_score_map(lev[l], W, H, lstride[l], (int)P->fast_threshold, score);
        size_t n = fast_nms(lev[l], W, H, lstride[l], (int)P->fast_threshold, score, kp, raw_cap);
        free(score);
        /* RunByImageBorder */
        if (half > 0) {
            if (H <= half * 2 || W <= half * 2) n = 0;
            else {
                size_t m = 0;
                for (size_t i = 0; i < n; ++i)
                    if (kp[i].x >= half && kp[i].x < W - half && kp[i].y >= half && kp[i].y < H - half) kp[m++] = kp[i];
                n = m;
            }
        }
        if (n > (size_t)nfeat[l]) {
            int max_num = (int)(nfeat[l] * P->feature_factor);
            n = retain_best(kp, n, (int)P->fast_threshold, max_num, nfeat[l], P->feature_strength);
            n = anms(kp, n, (unsigned)nfeat[l], (int)P->fast_threshold, P);
        }
        /* ImageData::Insert: copy what still fits; a level that cannot place a single keypoint ends the loop (:731-737) */
        size_t room = (size_t)cap > total ? (size_t)cap - total : 0;
        size_t take = n < room ? n : room;
        for (size_t i = 0; i < take; ++i) {
            orbo_keypoint* k = &kps[total + i];
            k->x = (float)kp[i].x; k->y = (float)kp[i].y; k->size = (float)P->patch_size * lscale[l];
            k->angle = 0.0f; k->response = (float)kp[i].resp; k->octave = l; k->class_id = -1;
        }
        total += take;
        free(kp);
        if (n > 0 && take == 0) break;
    }
    *n_out = (int)total;
    if (total == 0) { for (int l = 1; l < L; ++l) free(own[l]); return ORBO_OK; }
    if (P->use_orientation)                                   /* ICAngles on the unblurred levels (:745-748) */
        for (size_t i = 0; i < total; ++i) orbo_ic_angles(lev[kps[i].octave], lstride[kps[i].octave], &kps[i], 1, half_patch);
    for (size_t i = 0; i < total; ++i) { const float sc = lscale[kps[i].octave]; kps[i].x *= sc; kps[i].y *= sc; }     /* :756-760 */
    /* blur every level (isolated), then the descriptors on the level of each keypoint */
    uint8_t* blur[ORBO_MAX_LEVELS];
    for (int l = 0; l < L; ++l) {
        const int W = lw[l] > 0 ? lw[l] : 1, H = lh[l] > 0 ? lh[l] : 1;
        blur[l] = (uint8_t*)malloc((size_t)W * H);
        if (lw[l] <= 0 || lh[l] <= 0) continue;
        if (P->gaussian_kernel_size > 1) orbo_blur(lev[l], lw[l], lh[l], lstride[l], (int)P->gaussian_kernel_size, blur[l]);
        else for (int y = 0; y < lh[l]; ++y) memcpy(blur[l] + (size_t)y * lw[l], lev[l] + (size_t)y * lstride[l], (size_t)lw[l]);
    }
    if (blurred_out) memcpy(blurred_out, blur[0], (size_t)w * h);
    signed char* pat = (signed char*)malloc(30 * 1024);
    orbo_pattern_expand((int)P->patch_size, pat);
    for (size_t j = 0; j < total; ++j) {
        const int l = kps[j].octave;
        const float inv = 1.f / lscale[l];                                             /* :521 */
        const uint8_t* center = blur[l] + (size_t)cv_round(kps[j].y * inv) * lw[l] + cv_round(kps[j].x * inv);
        if (P->use_orientation && P->patch_size != 15 && P->patch_size != 31) {
            /* ComputeOrbDescriptors (:452-492): angle in radians as float, a = (float)cos(angle), b = (float)sin(angle), every point
               (x, y) -> (cvRound(x a - y b), cvRound(x b + y a)) in float.  PINNED CHOICE: cos / sin are taken in double and rounded to
               float (the correctly rounded float cosine for all but ~1e-8 of the arguments); the reference's own value is whatever the
               MSVC CRT's float overload returns -- unknowable here, and one ulp of a or b only matters when x a - y b falls within
               1e-6 of a rounding boundary. */
            float ang = kps[j].angle;
            ang *= (float)(M_PI / 180.0);
            const float a = (float)cos((double)ang), b = (float)sin((double)ang);
            const signed char* p = pat;
            for (int i = 0; i < 32; ++i, p += 32) {
                int val = 0;
                for (int bit = 0; bit < 8; ++bit) {
                    const float x0 = (float)p[4 * bit], y0 = (float)p[4 * bit + 1], x1 = (float)p[4 * bit + 2], y1 = (float)p[4 * bit + 3];
                    const float rx0 = x0 * a - y0 * b, ry0 = x0 * b + y0 * a, rx1 = x1 * a - y1 * b, ry1 = x1 * b + y1 * a;
                    int t0 = center[cv_round(ry0) * lw[l] + cv_round(rx0)];
                    int t1 = center[cv_round(ry1) * lw[l] + cv_round(rx1)];
                    val |= (t0 < t1) << bit;
                }
                desc32[j * 32 + i] = (uint8_t)val;
            }
            continue;
        }
        const signed char* p = pat + (cv_round(kps[j].angle / 12.0f) % 30) * 1024;       /* angleIncrement (:523-532); 0 without orientation */
        for (int i = 0; i < 32; ++i, p += 32) {
            int val = 0;
            for (int bit = 0; bit < 8; ++bit) {
                int t0 = center[p[4 * bit + 1] * lw[l] + p[4 * bit]];
                int t1 = center[p[4 * bit + 3] * lw[l] + p[4 * bit + 2]];
                val |= (t0 < t1) << bit;
            }
            desc32[j * 32 + i] = (uint8_t)val;
        }
    }
    free(pat);
    for (int l = 0; l < L; ++l) free(blur[l]);
    for (int l = 1; l < L; ++l) free(own[l]);
    return ORBO_OK;
}


/* ------------------------------------------------------------------------------------------------
 * OrbFeatureDetector::UndistortKeypoints (Image/OrbFeatureDetector.cpp:30-62):
 *     cv::undistortPoints(points, K_distorted, distCoeffs, noArray(), K_undistorted)
 * i.e. cvUndistortPoints of OpenCV 3.4.0 (modules/imgproc/src/undistort.cpp; OpenCV is not vendored, README pins 3.4.0):
 * normalise with K, FIVE fixed-point iterations of the inverse distortion model in double, re-project with P = K_undistorted
 * (R = identity), round to float.  Coefficients in OpenCV order k1 k2 p1 p2 k3 [k4 k5 k6]; the reference passes 5
 * (Poly3k) or 8 (Rational6k) of them (Device/CameraCalibration.cpp:110-125), as float.  PARITY UNPINNED: no OpenCV here.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { float K[9]; float dist[8]; int n_dist; float P[9]; } orbo_undistort;
ORBO_API void orbo_undistort_keypoints(orbo_keypoint* kp, int n, const orbo_undistort* U)
{
    double k[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int i = 0; i < U->n_dist && i < 8; ++i) k[i] = (double)U->dist[i];
    const double fx = (double)U->K[0], fy = (double)U->K[4], cx = (double)U->K[2], cy = (double)U->K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    double RR[9];
    for (int i = 0; i < 9; ++i) RR[i] = (double)U->P[i];                   /* P * I */
    for (int i = 0; i < n; ++i) {
        double x = (double)kp[i].x, y = (double)kp[i].y, x0, y0;
        x = (x - cx) * ifx;
        y = (y - cy) * ify;
        x0 = x; y0 = y;
        for (int j = 0; j < 5; ++j) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = RR[0] * x + RR[1] * y + RR[2];
        const double yy = RR[3] * x + RR[4] * y + RR[5];
        const double ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
        kp[i].x = (float)(xx * ww);
        kp[i].y = (float)(yy * ww);
    }
}
