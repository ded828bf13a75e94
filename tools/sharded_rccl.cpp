// sharded_rccl.cpp -- ONE map solved by several GPUs, sharded by landmark (include/mage_ba.h: mage_ba_set_landmark_shard; SURVEY.md 8e
// "exact algorithm"), driven from C++ with the per-trial exchange on RCCL: one process per GPU, the packed reduced camera system
// all-reduced in HBM by ncclAllReduce on the solver's own stream (xGMI between the GPUs of a node).  mageslam_amd/sharded.py is
// the Python twin.
//
//   hipcc -O2 -std=c++17 -Iinclude tools/sharded_rccl.cpp -Lmageslam_amd -lmageslam_hip -lrccl -Wl,-rpath,$PWD/mageslam_amd -o tools/_bin/sharded_rccl
//   RANK=r WORLD_SIZE=n LOCAL_RANK=r sharded_rccl scene.bin steps huber max_err_sq id_file out_prefix
//
// Every rank loads the whole scene file, keeps all cameras and the map points mage_ba_partition_landmarks gives it, and steps.
// Rank 0 publishes the ncclUniqueId through `id_file`.  Every rank writes <out_prefix>.rank<r>.bin: n_cams x 7 f64 poses, then
// its points as (global index u32 as f64, x, y, z) rows; rank 0 prints one JSON line.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mage_ba.h"
#include "scene_io.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)
#define CHECK_MAGE(x) do { if ((x) != MAGE_OK) { std::fprintf(stderr, "%s: %s\n", #x, mage_last_error()); return 1; } } while (0)

static int env_int(const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; }

struct Reduce { ncclComm_t comm; unsigned long calls = 0; size_t doubles = 0; };
static int allreduce_rccl(void* ctx, double* buf, size_t count, int op, void* stream)
{
    Reduce* r = static_cast<Reduce*>(ctx);
    r->calls++; r->doubles += count;
    return ncclAllReduce(buf, buf, count, ncclDouble, op == 0 ? ncclSum : ncclMax, r->comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}

int main(int argc, char** argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage: %s scene.bin steps huber max_err_sq id_file out_prefix\n", argv[0]); return 2; }
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", rank);
    const int steps = std::atoi(argv[2]);
    const float huber = (float)std::atof(argv[3]), max_err_sq = (float)std::atof(argv[4]);
    const std::string id_file = argv[5], out_prefix = argv[6];
    int n_dev = 0;
    CHECK_HIP(hipGetDeviceCount(&n_dev));
    if (n_dev < 1) { std::fprintf(stderr, "no HIP device\n"); return 1; }
    const int dev = local % n_dev;
    CHECK_HIP(hipSetDevice(dev));

    ncclUniqueId id;
    if (rank == 0) {
        CHECK_NCCL(ncclGetUniqueId(&id));
        const std::string tmp = id_file + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) { std::fprintf(stderr, "cannot write %s\n", tmp.c_str()); return 1; }
        std::fclose(f);
        std::rename(tmp.c_str(), id_file.c_str());
    } else {
        for (int tries = 0;; ++tries) {
            FILE* f = std::fopen(id_file.c_str(), "rb");
            if (f) { const size_t got = std::fread(&id, sizeof(id), 1, f); std::fclose(f); if (got == 1) break; }
            if (tries > 6000) { std::fprintf(stderr, "rank %d: no %s after 60 s\n", rank, id_file.c_str()); return 1; }
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    Reduce red{};
    CHECK_NCCL(ncclCommInitRank(&red.comm, world, id, rank));
    // What the communicator really spans is reported in the result line, and a run that was meant to be collective but came up with
    // ONE rank (WORLD_SIZE missing from the environment, a launcher that started the ranks separately) is refused instead of passing
    // vacuously: a single rank's all-reduce is a copy.  MAGE_ALLOW_SINGLE_RANK=1 is how the one-GPU tests ask for that on purpose.
    int comm_nranks = 0;
    CHECK_NCCL(ncclCommCount(red.comm, &comm_nranks));
    if (comm_nranks != world) { std::fprintf(stderr, "rank %d: the communicator has %d ranks, WORLD_SIZE says %d\n", rank, comm_nranks, world); return 1; }
    if (comm_nranks < 2 && !(std::getenv("MAGE_ALLOW_SINGLE_RANK") && std::atoi(std::getenv("MAGE_ALLOW_SINGLE_RANK")) != 0)) {
        std::fprintf(stderr, "the communicator has ONE rank: nothing would be exchanged (set RANK / WORLD_SIZE / LOCAL_RANK per process, or "
                             "MAGE_ALLOW_SINGLE_RANK=1 to run a single rank on purpose)\n");
        return 3;
    }

    SceneFile s;
    try { s = read_scene(argv[1]); } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
    // this rank's share: its points in map order, every observation of them in map order
    std::vector<int32_t> owner(s.n_pts);
    CHECK_MAGE(mage_ba_partition_landmarks(s.n_pts, s.n_obs, s.obs_pt.data(), world, owner.data()));
    std::vector<uint32_t> pt_global, pt_local(s.n_pts, 0xffffffffu), obs_global;
    for (uint32_t p = 0; p < s.n_pts; ++p) if (owner[p] == rank) { pt_local[p] = (uint32_t)pt_global.size(); pt_global.push_back(p); }
    std::vector<float> pts, uv, info; std::vector<uint32_t> ocam, opt;
    for (uint32_t p : pt_global) for (int k = 0; k < 3; ++k) pts.push_back(s.points[(size_t)p * 3 + k]);
    for (uint32_t e = 0; e < s.n_obs; ++e) if (owner[s.obs_pt[e]] == rank) {
        obs_global.push_back(e);
        uv.push_back(s.obs_uv[(size_t)e * 2]); uv.push_back(s.obs_uv[(size_t)e * 2 + 1]);
        ocam.push_back(s.obs_cam[e]); opt.push_back(pt_local[s.obs_pt[e]]); info.push_back(s.obs_info[e]);
    }
    std::vector<uint8_t> fixed(s.n_cams);
    for (uint32_t i = 0; i < s.n_cams; ++i) fixed[i] = s.cam_fixed[i] ? 1 : 0;

    mage_ba_params p{ 0, dev };
    mage_ba* h = nullptr;
    CHECK_MAGE(mage_ba_create(&p, &h));
    CHECK_MAGE(mage_ba_set_landmark_shard(h, rank, world, allreduce_rccl, &red));
    CHECK_MAGE(mage_ba_alloc_cameras(h, s.n_cams));
    CHECK_MAGE(mage_ba_set_cameras_bulk(h, s.n_cams, s.cam_t.data(), s.cam_R.data(), s.cam_K.data(), fixed.data()));
    CHECK_MAGE(mage_ba_alloc_points(h, pt_global.size()));
    CHECK_MAGE(mage_ba_set_points_bulk(h, pt_global.size(), pts.data()));
    CHECK_MAGE(mage_ba_alloc_observations(h, obs_global.size()));
    CHECK_MAGE(mage_ba_set_observations_bulk(h, obs_global.size(), uv.data(), ocam.data(), opt.data(), info.data()));
    CHECK_MAGE(mage_ba_alloc_fixed_distance_constraints(h, 0));
    CHECK_MAGE(mage_ba_alloc_relative_rotation_constraints(h, 0));
    CHECK_MAGE(mage_ba_alloc_relative_transform_constraints(h, 0));

    std::vector<float> mse(steps);
    std::vector<uint32_t> out_ids(obs_global.size() + 1);
    size_t n_out_total = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < steps; ++it) {
        size_t n_out = 0;
        CHECK_MAGE(mage_ba_step(h, &huber, 1, max_err_sq, out_ids.data(), out_ids.size(), &n_out, &mse[it]));
        n_out_total += n_out;
    }
    CHECK_MAGE(mage_ba_synchronize(h));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    std::vector<double> poses((size_t)s.n_cams * 7), lpts(pt_global.size() * 3 + 1);
    CHECK_MAGE(mage_ba_get_state_f64(h, poses.data(), lpts.data()));
    std::vector<double> rows(pt_global.size() * 4);
    for (size_t i = 0; i < pt_global.size(); ++i) { rows[i * 4] = (double)pt_global[i]; for (int k = 0; k < 3; ++k) rows[i * 4 + 1 + k] = lpts[i * 3 + k]; }
    const std::string out = out_prefix + ".rank" + std::to_string(rank) + ".bin";
    FILE* f = std::fopen(out.c_str(), "wb");
    if (!f || std::fwrite(poses.data(), sizeof(double), poses.size(), f) != poses.size() ||
        (rows.size() && std::fwrite(rows.data(), sizeof(double), rows.size(), f) != rows.size())) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    std::fclose(f);
    if (rank == 0) {
        std::printf("{\"world\": %d, \"comm_nranks\": %d, \"steps\": %d, \"allreduce_calls\": %lu, \"exchanged_bytes\": %zu, \"ms_total\": %.3f, \"own_points\": %zu, \"own_outliers\": %zu, \"mse\": [",
                    world, comm_nranks, steps, red.calls, red.doubles * 8, ms, pt_global.size(), n_out_total);
        for (int it = 0; it < steps; ++it) std::printf("%s%.9g", it ? ", " : "", (double)mse[it]);
        std::printf("]}\n");
    }
    mage_ba_destroy(h);
    ncclCommDestroy(red.comm);
    return 0;
}
