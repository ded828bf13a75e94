/* small_shapes_main.c -- TEST / BENCH INFRASTRUCTURE ONLY (the CPU baseline leg of bench.py; never part of the product).
 *
 * The reference's two everyday callers of BundlerLib run on the CPU oracle (oracle/ba_oracle.c, one thread like the reference's
 * g2o path), timed phase by phase from a NATIVE caller, so that the figures set beside the device path in bench.py
 * (extra.pose_only / extra.reference_window) carry no Python binding overhead -- the twin of tools/shim_small_shapes.cpp, which times
 * the device path through include/BundlerLib.h:
 *   pose-only scene.bin [frames]   TrackLocalMap::OptimizeCameraPose (Tracking/TrackLocalMap.cpp:421-501), twice per frame:
 *                                  3 iterations at Huber 4.0 / 36.0, then 4 at Huber 0.9 / 20.25 (MageSettings.h:180-195)
 *   window scene.bin [runs]        BundleAdjust::RunBundleAdjustment (BundleAdjust.cpp:281-354) with NumSteps = 1: build, ONE
 *                                  StepBundleAdjustment({1.8}, 7.25), UpdateData (GetPose per free keyframe, GetPoint per point)
 * Scene file = mageslam_amd/scene.py::save_scene ("MAGESCN1").  Prints the same lines as the device-side tool.
 */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct ba_oracle ba_oracle;
ba_oracle* bao_create(int points_fixed);
void bao_destroy(ba_oracle* b);
void bao_alloc_cameras(ba_oracle* b, size_t n);
void bao_alloc_points(ba_oracle* b, size_t n);
void bao_alloc_observations(ba_oracle* b, size_t n);
void bao_set_camera(ba_oracle* b, size_t idx, const float t[3], const float Rcm[9], const float K[4], int fixed);
void bao_set_point(ba_oracle* b, size_t idx, const float p[3]);
void bao_set_observation(ba_oracle* b, size_t idx, const float uv[2], size_t cam, size_t pt, float info);
float bao_step(ba_oracle* b, const float* huber_w, size_t n_huber, float max_err_sq, unsigned* outliers, size_t cap, size_t* n_out);
void bao_get_pose(const ba_oracle* b, size_t idx, float t[3], float Rcm[9]);
void bao_get_point(const ba_oracle* b, size_t idx, float p[3]);

typedef struct {
    uint32_t n_cams, n_pts, n_obs;
    float *cam_t, *cam_R, *cam_K, *points, *obs_uv, *obs_info;
    uint32_t *cam_fixed, *obs_cam, *obs_pt;
} scene_t;

static void* rd(FILE* f, size_t bytes)
{
    void* p = malloc(bytes ? bytes : 1);
    if (bytes && fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short read\n"); exit(1); }
    return p;
}
static scene_t read_scene(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
    char magic[8]; uint32_t hdr[4];
    if (fread(magic, 1, 8, f) != 8 || fread(hdr, 4, 4, f) != 4 || memcmp(magic, "MAGESCN1", 8) != 0) { fprintf(stderr, "not a scene file\n"); exit(1); }
    scene_t s; s.n_cams = hdr[0]; s.n_pts = hdr[1]; s.n_obs = hdr[2];
    s.cam_t = rd(f, (size_t)s.n_cams * 12); s.cam_R = rd(f, (size_t)s.n_cams * 36); s.cam_K = rd(f, (size_t)s.n_cams * 16); s.cam_fixed = rd(f, (size_t)s.n_cams * 4);
    s.points = rd(f, (size_t)s.n_pts * 12);
    s.obs_uv = rd(f, (size_t)s.n_obs * 8); s.obs_cam = rd(f, (size_t)s.n_obs * 4); s.obs_pt = rd(f, (size_t)s.n_obs * 4); s.obs_info = rd(f, (size_t)s.n_obs * 4);
    fclose(f);
    return s;
}

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }
static int cmp_d(const void* a, const void* b) { double x = *(const double*)a, y = *(const double*)b; return (x > y) - (x < y); }
static double median(double* v, int n) { if (!n) return 0; qsort(v, n, sizeof(double), cmp_d); return v[n / 2]; }

typedef struct { double *create, *set, *step, *get, *destroy, *total; int n; } phases_t;
static phases_t phases_new(int cap) { phases_t p; p.n = 0; p.create = malloc(cap * 8); p.set = malloc(cap * 8); p.step = malloc(cap * 8); p.get = malloc(cap * 8); p.destroy = malloc(cap * 8); p.total = malloc(cap * 8); return p; }
static void phases_add(phases_t* p, double t0, double t1, double t2, double t3, double t4, double t5)
{ int i = p->n++; p->create[i] = t1 - t0; p->set[i] = t2 - t1; p->step[i] = t3 - t2; p->get[i] = t4 - t3; p->destroy[i] = t5 - t4; p->total[i] = t5 - t0; }
static void phases_print(const char* tag, phases_t* p)
{
    double mn = 1e30; for (int i = 0; i < p->n; ++i) if (p->total[i] < mn) mn = p->total[i];
    printf("%s total %.5f min %.5f create %.5f set %.5f step %.5f get %.5f destroy %.5f calls %d\n", tag, median(p->total, p->n), mn,
           median(p->create, p->n), median(p->set, p->n), median(p->step, p->n), median(p->get, p->n), median(p->destroy, p->n), p->n);
}

static void optimize_camera_pose(const scene_t* s, unsigned iterations, float huber, float thr, phases_t* ph, float pos[3], float rot[9], size_t* n_out, unsigned* outl)
{
    const double t0 = now_ms();
    ba_oracle* b = bao_create(1);
    const double t1 = now_ms();
    bao_alloc_cameras(b, 1);
    bao_set_camera(b, 0, s->cam_t, s->cam_R, s->cam_K, 0);
    bao_alloc_points(b, s->n_obs);
    bao_alloc_observations(b, s->n_obs);
    for (uint32_t i = 0; i < s->n_obs; ++i) {
        bao_set_point(b, i, s->points + (size_t)s->obs_pt[i] * 3);
        bao_set_observation(b, i, s->obs_uv + (size_t)i * 2, 0, i, s->obs_info[i]);
    }
    const double t2 = now_ms();
    float widths[8];
    for (unsigned i = 0; i < iterations; ++i) widths[i] = huber;
    size_t n = 0;
    bao_step(b, widths, iterations, thr, outl, s->n_obs, &n);
    const double t3 = now_ms();
    bao_get_pose(b, 0, pos, rot);
    const double t4 = now_ms();
    bao_destroy(b);
    const double t5 = now_ms();
    if (ph) phases_add(ph, t0, t1, t2, t3, t4, t5);
    if (n_out) *n_out = n;
}

static void run_bundle_adjustment(const scene_t* s, phases_t* ph, float* mse_out, size_t* n_out, unsigned* outl, float* sink)
{
    const double t0 = now_ms();
    ba_oracle* b = bao_create(0);
    const double t1 = now_ms();
    bao_alloc_cameras(b, s->n_cams);
    for (size_t i = 0; i < s->n_cams; ++i) bao_set_camera(b, i, s->cam_t + i * 3, s->cam_R + i * 9, s->cam_K + i * 4, s->cam_fixed[i] != 0);
    bao_alloc_points(b, s->n_pts);
    for (size_t i = 0; i < s->n_pts; ++i) bao_set_point(b, i, s->points + i * 3);
    bao_alloc_observations(b, s->n_obs);
    for (size_t i = 0; i < s->n_obs; ++i) bao_set_observation(b, i, s->obs_uv + i * 2, s->obs_cam[i], s->obs_pt[i], s->obs_info[i]);
    const double t2 = now_ms();
    const float width = 1.8f;
    size_t n = 0;
    const float mse = bao_step(b, &width, 1, 7.25f, outl, s->n_obs, &n);
    const double t3 = now_ms();
    float pos[3], rot[9], p[3];
    for (size_t i = 0; i < s->n_cams; ++i) if (!s->cam_fixed[i]) { bao_get_pose(b, i, pos, rot); *sink += pos[0] + rot[0]; }
    for (size_t i = 0; i < s->n_pts; ++i) { bao_get_point(b, i, p); *sink += p[0]; }
    const double t4 = now_ms();
    bao_destroy(b);
    const double t5 = now_ms();
    if (ph) phases_add(ph, t0, t1, t2, t3, t4, t5);
    if (mse_out) *mse_out = mse;
    if (n_out) *n_out = n;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s pose-only|window scene.bin [repetitions]\n", argv[0]); return 2; }
    const scene_t s = read_scene(argv[2]);
    const int reps = argc > 3 ? atoi(argv[3]) : 200;
    unsigned* outl = malloc(((size_t)s.n_obs + 1) * sizeof(unsigned));
    if (strcmp(argv[1], "pose-only") == 0) {
        if (s.n_cams != 1) { fprintf(stderr, "pose-only wants a scene of one camera\n"); return 2; }
        phases_t p1 = phases_new(reps), p2 = phases_new(reps);
        float pos[3], rot[9];
        size_t n1 = 0, n2 = 0;
        for (int i = 0; i < 3; ++i) { optimize_camera_pose(&s, 3, 4.0f, 36.0f, NULL, pos, rot, NULL, outl); optimize_camera_pose(&s, 4, 0.9f, 20.25f, NULL, pos, rot, NULL, outl); }
        const double t0 = now_ms();
        for (int i = 0; i < reps; ++i) {
            optimize_camera_pose(&s, 3, 4.0f, 36.0f, &p1, pos, rot, &n1, outl);
            optimize_camera_pose(&s, 4, 0.9f, 20.25f, &p2, pos, rot, &n2, outl);
        }
        const double per_frame = (now_ms() - t0) / reps;
        phases_print("pose_only_pass1_ms", &p1);
        phases_print("pose_only_pass2_ms", &p2);
        printf("pose_only_frame_ms %.5f outliers %zu %zu position %.6f %.6f %.6f\n", per_frame, n1, n2, pos[0], pos[1], pos[2]);
    } else if (strcmp(argv[1], "window") == 0) {
        phases_t ph = phases_new(reps);
        float mse = 0, sink = 0;
        size_t nout = 0;
        for (int i = 0; i < 3; ++i) run_bundle_adjustment(&s, NULL, NULL, NULL, outl, &sink);
        sink = 0;
        for (int i = 0; i < reps; ++i) run_bundle_adjustment(&s, &ph, &mse, &nout, outl, &sink);
        phases_print("window_ms", &ph);
        printf("window_result mse %.6f outliers %zu checksum %.4f\n", mse, nout, sink / reps * 1.0f);
    } else { fprintf(stderr, "unknown mode %s\n", argv[1]); return 2; }
    return 0;
}
