// windowed_rccl.cpp -- the window-sharded map (include/mage_window.h) driven from C++ with the exchange on RCCL:
// one process per GPU, the pose block all-reduced in HBM with ncclAllReduce on the rank's communicator (xGMI between the
// GPUs of a node).  This is the C++ integration INTEGRATION.md section 3b describes; mageslam_amd/windowed.py is its Python twin.
//
//   hipcc -O2 -std=c++17 -Iinclude tools/windowed_rccl.cpp -Lmageslam_amd -lmageslam_hip -lrccl -Wl,-rpath,$PWD/mageslam_amd -o tools/_bin/windowed_rccl
//   RANK=r WORLD_SIZE=n LOCAL_RANK=r windowed_rccl scene.bin n_windows overlap outer_iterations huber threads id_file out_prefix
//
// Rank 0 creates the ncclUniqueId and publishes it through `id_file` (write + rename); the others poll for it.  Every rank
// writes the final pose block (n_cams x 8 f64) to <out_prefix>.rank<r>.bin and rank 0 prints one JSON line.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mage_window.h"
#include "scene_io.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 1; } } while (0)
#define CHECK_MAGE(x) do { if ((x) != MAGE_OK) { std::fprintf(stderr, "%s: %s\n", #x, mage_last_error()); return 1; } } while (0)

static int env_int(const char* name, int dflt) { const char* e = std::getenv(name); return e ? std::atoi(e) : dflt; }

struct Reduce { ncclComm_t comm; unsigned long calls = 0; };
static int allreduce_rccl(void* ctx, double* block, size_t count, void* stream)
{
    Reduce* r = static_cast<Reduce*>(ctx);
    r->calls++;
    return ncclAllReduce(block, block, count, ncclDouble, ncclSum, r->comm, static_cast<hipStream_t>(stream)) == ncclSuccess ? 0 : 1;
}

int main(int argc, char** argv)
{
    if (argc < 9) { std::fprintf(stderr, "usage: %s scene.bin n_windows overlap outer_iterations huber threads id_file out_prefix\n", argv[0]); return 2; }
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", rank);
    const int n_windows = std::atoi(argv[2]), overlap = std::atoi(argv[3]), iters = std::atoi(argv[4]), threads = std::atoi(argv[6]);
    const float huber = (float)std::atof(argv[5]);
    const std::string id_file = argv[7], out_prefix = argv[8];
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
    std::vector<uint8_t> fixed(s.n_cams);
    for (uint32_t i = 0; i < s.n_cams; ++i) fixed[i] = s.cam_fixed[i] ? 1 : 0;
    mage_wmap_params p{ n_windows, overlap, rank, world, dev, threads };
    mage_wmap* m = nullptr;
    CHECK_MAGE(mage_wmap_create(&p, s.n_cams, s.cam_t.data(), s.cam_R.data(), s.cam_K.data(), fixed.data(), s.n_pts, s.points.data(),
                                s.n_obs, s.obs_uv.data(), s.obs_cam.data(), s.obs_pt.data(), s.obs_info.data(), &m));
    CHECK_MAGE(mage_wmap_set_allreduce(m, allreduce_rccl, &red));

    std::vector<double> mse(iters);
    const auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < iters; ++it) CHECK_MAGE(mage_wmap_outer_iteration(m, huber, 1e30f, 1, &mse[it]));
    std::vector<double> block((size_t)s.n_cams * 8);
    CHECK_MAGE(mage_wmap_get_pose_block(m, block.data()));          // synchronises the exchange stream
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

    const std::string out = out_prefix + ".rank" + std::to_string(rank) + ".bin";
    FILE* f = std::fopen(out.c_str(), "wb");
    if (!f || std::fwrite(block.data(), sizeof(double), block.size(), f) != block.size()) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    std::fclose(f);
    if (rank == 0) {
        std::printf("{\"world\": %d, \"comm_nranks\": %d, \"n_windows\": %d, \"outer_iterations\": %d, \"allreduce_calls\": %lu, \"exchange_bytes\": %zu, \"ms_total\": %.3f, \"mse_rank0\": [",
                    world, comm_nranks, n_windows, iters, red.calls, (size_t)s.n_cams * 64, ms);
        for (int it = 0; it < iters; ++it) std::printf("%s%.9g", it ? ", " : "", mse[it]);
        std::printf("]}\n");
    }
    mage_wmap_destroy(m);
    ncclCommDestroy(red.comm);
    return 0;
}
