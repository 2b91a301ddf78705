// N-GPU headless driver: the reference's main() (optixPathTracer.cpp:680-837) reduced to the hot path and sharded over the GPUs
// of one node -- load scene -> per rank: create context -> rank 0 preprocesses, tuple broadcast over RCCL -> frames: light pass of
// a rank's cores (one batch ahead), LVC all-gather, sampler build, eye megakernel on the rank's 8-row bands -> film band gather ->
// PPM.  One host thread per GPU over include/spcbpt_mgpu.h (RCCL), pure C++ above the C ABI; `--local N` runs N ranks on GPU 0
// with device copies as the transport (what a one-GPU box can do: RCCL refuses two ranks on one device).
//   spcbpt_render_mgpu scene.{scene,gltf,glb} [--data-root DIR] [--gpus N | --local N] [--dim WxH] [--frames F] [--batch B]
//                      [--light-paths M] [--out image.ppm] [--no-train]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../include/spcbpt_mgpu.h"

extern "C" int hipSetDevice(int);
extern "C" int hipGetDeviceCount(int*);

struct Job {
    spcbpt_scene_desc desc;
    float eye[3], lookat[3], up[3], fov;
    int W = 640, H = 360, frames = 16, batch = 0, M = 100000;   // batch 0 = up to 32 frames per eye / light launch, equal launches
    bool train = true, lbatch = false;
};
struct Rank {
    spcbpt_ctx* ctx = nullptr;
    spcbpt_comm* comm = nullptr;
    int id = 0, world = 1, device = 0, next_light = 1, rc = 0;
    std::vector<uint32_t> queued;
    std::string err;
};
#define RK(r, expr) do { int rc__ = (expr); if (rc__) { (r).rc = rc__; (r).err = std::string(#expr) + ": " + ((r).ctx ? spcbpt_last_error((r).ctx) : "") + ((r).comm ? std::string(" / ") + spcbpt_comm_last_error((r).comm) : ""); return; } } while (0)

static void core_range(int n, int rank, int world, int* begin, int* count) {
    const int per = n / world;
    *begin = rank * per;
    *count = rank < world - 1 ? per : n - *begin;
}

// one rank's start-up up to (not including) the communicator
static void rank_create(const Job& J, Rank& R) {
    hipSetDevice(R.device);
    RK(R, spcbpt_create(&J.desc, R.device, &R.ctx));
    RK(R, spcbpt_set_camera_lookat(R.ctx, J.eye, J.lookat, J.up, J.fov, (float)J.W / (float)J.H));
    RK(R, spcbpt_resize(R.ctx, J.W, J.H));
    spcbpt_light_trace_params lt = {J.M, 52, 1, 0, 0, 1};
    RK(R, spcbpt_set_light_trace(R.ctx, &lt));            // the whole pass: the tuple is computed with it
    if (R.id == 0) {
        if (J.train) RK(R, spcbpt_preprocess(R.ctx, 2000000, 2000000, 1));
        else RK(R, spcbpt_set_subspace(R.ctx, nullptr, 0, nullptr, 0, nullptr, nullptr));
    }
}
static void rank_shard(const Job& J, Rank& R) {
    hipSetDevice(R.device);
    int b = 0, c = 0;
    core_range(J.M, R.id, R.world, &b, &c);
    spcbpt_light_trace_params lt = {J.M, 52, 1, b, c, 1};
    RK(R, spcbpt_set_light_trace(R.ctx, &lt));
}
static void rank_prime(const Job& J, Rank& R) {           // calibrated capacity, light passes one batch ahead
    hipSetDevice(R.device);
    RK(R, spcbpt_comm_calibrate(R.comm, 2, 900000u, 1.5f));
    RK(R, spcbpt_set_light_ahead(R.ctx, 1));
    if (J.lbatch) { RK(R, spcbpt_launch_light_batch(R.ctx, (uint32_t)R.next_light, J.batch)); R.next_light += J.batch; }
    else for (int k = 0; k < J.batch; k++) RK(R, spcbpt_launch(R.ctx, "light trace", (uint32_t)R.next_light++, 0, 0, 1));
}
// the three phases of a frame; a threaded rank runs them back to back, the local driver runs each phase for every rank in turn
static void frame_light(const Job& J, Rank& R, int f) {
    hipSetDevice(R.device);
    if (!J.lbatch) RK(R, spcbpt_launch(R.ctx, "light trace", (uint32_t)R.next_light++, 0, 0, 1));
    else if (f % J.batch == 0) { RK(R, spcbpt_launch_light_batch(R.ctx, (uint32_t)R.next_light, J.batch)); R.next_light += J.batch; }   // a batch's passes as one launch
}
static void frame_exchange(const Job& J, Rank& R) {
    hipSetDevice(R.device);
    if (!J.lbatch) RK(R, spcbpt_sync_light(R.ctx));   // back-pressure only (the pass was launched a batch ago): keeps the host from queueing dozens of frames ahead
    RK(R, spcbpt_comm_exchange_lvc(R.comm));
}
static void frame_render(const Job& J, Rank& R, int f) {
    hipSetDevice(R.device);
    RK(R, spcbpt_build_sampler(R.ctx));
    R.queued.push_back((uint32_t)f);
    if ((int)R.queued.size() == J.batch || f == J.frames - 1) {
        RK(R, spcbpt_launch_eye_batch(R.ctx, (int)R.queued.size(), R.queued.data(), 8 * R.id, J.H, R.world));
        R.queued.clear();
    }
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s scene.{scene,gltf,glb} [--data-root DIR] [--gpus N | --local N] [--dim WxH] [--frames F] [--batch B] [--light-batch 0|1] [--light-paths M] [--out image.ppm] [--no-train]\n", argv[0]); return 2; }
    std::string path = argv[1], root = ".", out = "mgpu.ppm";
    int gpus = 0, local = 0, lbatch = -1;   // --light-batch 0|1: spcbpt_launch_light_batch for the passes of a batch (default: on)
    Job J;
    for (int i = 2; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--data-root") root = next();
        else if (a == "--gpus") gpus = atoi(next());
        else if (a == "--local") local = atoi(next());
        else if (a == "--dim") { if (sscanf(next(), "%dx%d", &J.W, &J.H) != 2) { fprintf(stderr, "bad --dim\n"); return 2; } }
        else if (a == "--frames") J.frames = atoi(next());
        else if (a == "--batch") J.batch = std::max(1, std::min(32, atoi(next())));
        else if (a == "--light-batch") lbatch = atoi(next());
        else if (a == "--light-paths") J.M = atoi(next());
        else if (a == "--out") out = next();
        else if (a == "--no-train") J.train = false;
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    spcbpt_scene_file* sf = nullptr;
    const bool gltf = path.size() > 5 && (path.substr(path.size() - 5) == ".gltf" || path.substr(path.size() - 4) == ".glb");
    char err[512] = {0};
    int rc = gltf ? spcbpt_gltf_load(path.c_str(), &sf, err, sizeof(err)) : spcbpt_scene_file_load(path.c_str(), root.c_str(), &sf);
    if (rc) { fprintf(stderr, "cannot load %s (%d) %s\n", path.c_str(), rc, err); return 1; }
    spcbpt_scene_file_desc(sf, &J.desc);
    spcbpt_scene_file_camera(sf, J.eye, J.lookat, J.up, &J.fov, nullptr, nullptr);
    int ndev = 0;
    hipGetDeviceCount(&ndev);
    const int world = local > 0 ? local : (gpus > 0 ? gpus : std::max(1, ndev));
    if (J.batch == 0) { const int launches = std::max(1, (J.frames + 31) / 32); J.batch = std::max(1, std::min(32, (J.frames + launches - 1) / launches)); }   // up to 32 frames per launch, equal launches
    J.lbatch = J.batch > 1 && lbatch != 0;
    if (!local && world > ndev) { fprintf(stderr, "%d GPUs asked for, %d present\n", world, ndev); return 1; }
    std::vector<Rank> R(world);
    for (int k = 0; k < world; k++) { R[k].id = k; R[k].world = world; R[k].device = local ? 0 : k; }
    auto failed = [&]() { for (auto& r : R) if (r.rc) { fprintf(stderr, "rank %d: %s (%d)\n", r.id, r.err.c_str(), r.rc); return true; } return false; };
    // spcbpt_create reads these: set ONCE, before any rank thread exists (setenv racing with another thread's getenv may move environ)
    setenv("SPCBPT_EYE_BATCH", std::to_string(J.batch).c_str(), 1);
    setenv("SPCBPT_RENDER_STREAMS", "1", 0);   // one long tile queue per launch; the light passes run beside it in a thin grid (bench.py: measured)
    auto each = [&](auto fn) {   // RCCL ranks: one host thread per GPU; local ranks: one thread, rank after rank
        if (local) { for (auto& r : R) fn(r); }
        else {
            // a rank that fails stops issuing collectives and its peers would wait in RCCL for ever: the job ends there, non-zero
            std::vector<std::thread> th;
            for (auto& r : R) th.emplace_back([&fn, &r]() { fn(r); if (r.rc) { fprintf(stderr, "rank %d: %s (%d) -- aborting the job\n", r.id, r.err.c_str(), r.rc); fflush(stderr); _exit(1); } });
            for (auto& t : th) t.join();
        }
    };
    each([&](Rank& r) { rank_create(J, r); });
    if (failed()) return 1;
    each([&](Rank& r) { rank_shard(J, r); });
    if (failed()) return 1;
    if (local) {
        std::vector<spcbpt_ctx*> ctxs; for (auto& r : R) ctxs.push_back(r.ctx);
        std::vector<spcbpt_comm*> cs(world);
        if (spcbpt_comm_create_local(ctxs.data(), world, cs.data())) { fprintf(stderr, "comm_create_local failed\n"); return 1; }
        for (int k = 0; k < world; k++) R[k].comm = cs[k];
    } else {
        char id[SPCBPT_UNIQUE_ID_BYTES];
        if (spcbpt_comm_unique_id(id)) { fprintf(stderr, "ncclGetUniqueId failed\n"); return 1; }
        each([&](Rank& r) { hipSetDevice(r.device); r.rc = spcbpt_comm_create(r.ctx, r.id, world, id, &r.comm); if (r.rc) r.err = "spcbpt_comm_create"; });
        if (failed()) return 1;
    }
    each([&](Rank& r) { hipSetDevice(r.device); RK(r, spcbpt_comm_broadcast_subspace(r.comm, 0)); });
    if (failed()) return 1;
    each([&](Rank& r) { rank_prime(J, r); });
    if (failed()) return 1;
    each([&](Rank& r) { hipSetDevice(r.device); RK(r, spcbpt_comm_barrier(r.comm)); RK(r, spcbpt_sync(r.ctx)); });
    const auto t0 = std::chrono::steady_clock::now();
    if (local) {
        for (int f = 0; f < J.frames && !failed(); f++) {
            for (auto& r : R) frame_light(J, r, f);
            for (auto& r : R) frame_exchange(J, r);       // completes when the last rank has posted
            for (auto& r : R) frame_render(J, r, f);
        }
    } else {
        each([&](Rank& r) { for (int f = 0; f < J.frames && !r.rc; f++) { frame_light(J, r, f); if (!r.rc) frame_exchange(J, r); if (!r.rc) frame_render(J, r, f); } });
    }
    if (failed()) return 1;
    each([&](Rank& r) { hipSetDevice(r.device); RK(r, spcbpt_comm_gather_film(r.comm, nullptr)); });   // read-out: ends the timed region, as bench.py's does
    if (failed()) return 1;
    each([&](Rank& r) { hipSetDevice(r.device); RK(r, spcbpt_sync(r.ctx)); });
    if (failed()) return 1;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double paths = (double)J.frames * ((double)J.W * J.H + J.M);
    printf("%d rank(s)%s, %dx%d, %d frames, %d per eye launch: %.3f s, %.3f ms/frame, %.1f Mpaths/s\n", world, local ? " on one GPU (local transport)" : " (RCCL)",
           J.W, J.H, J.frames, J.batch, dt, dt / J.frames * 1e3, paths / dt * 1e-6);
    std::vector<uint8_t> img((size_t)J.W * J.H * 4);
    std::vector<float> acc((size_t)J.W * J.H * 4);
    hipSetDevice(R[0].device);
    if (spcbpt_read_accum(R[0].ctx, acc.data())) { fprintf(stderr, "read_accum: %s\n", spcbpt_last_error(R[0].ctx)); return 1; }
    FILE* fp = fopen(out.c_str(), "wb");
    if (!fp) { fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    fprintf(fp, "P6\n%d %d\n255\n", J.W, J.H);
    for (int y = J.H - 1; y >= 0; y--)     // image row 0 is the bottom of the view (raygen.cu:338-343)
        for (int x = 0; x < J.W; x++) {
            const float* p = &acc[((size_t)y * J.W + x) * 4];
            const float lum = 0.3f * p[0] + 0.6f * p[1] + 0.1f * p[2], s = 1.0f / (1.0f + lum / 1.5f);   // ToneMap(c, 1.5), raygen.cu:50-58
            for (int k = 0; k < 3; k++) {
                float c = std::min(std::max(p[k] * s, 0.0f), 1.0f);
                c = c < 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
                fputc((int)std::min(255.0f, c * 256.0f), fp);
            }
        }
    fclose(fp);
    double sum = 0; for (size_t i = 0; i < acc.size(); i += 4) sum += acc[i] + acc[i + 1] + acc[i + 2];
    printf("mean radiance %.6f -> %s\n", sum / (3.0 * J.W * J.H), out.c_str());
    for (auto& r : R) { hipSetDevice(r.device); spcbpt_comm_destroy(r.comm); spcbpt_destroy(r.ctx); }
    spcbpt_scene_file_free(sf);
    return 0;
}
