// Microbenchmark: what does a divergent "node fetch" cost on MI355X?  Every lane walks a dependent chain of random 128-B
// nodes in a 13 MB table (the size of the bench scene's BVH).  Variants:
//   A  8 x global_load_dwordx4 per lane from the lane's own node        (what traverse<> does today)
//   B  4-lane teams: each instruction reads 64 contiguous bytes of ONE node with 4 lanes, data exchanged with DPP-free
//      __shfl (ds_bpermute) -- 16 distinct lines per instruction instead of 64
//   C  4 x dwordx4 (a 64-B node)
//   D  2 x dwordx4 (a 32-B node)
// Prints giga node-visits per second.  Build: hipcc --offload-arch=gfx950 -O3 gather_bench.hip -o gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <int QUADS>
__global__ __launch_bounds__(256) void k_own(const float4* __restrict__ nodes, uint32_t n_nodes, int steps, uint32_t* out) {
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t node = mix(tid) % n_nodes;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const float4* p = nodes + (size_t)node * 8;
        float4 q[QUADS];
#pragma unroll
        for (int i = 0; i < QUADS; i++) q[i] = p[i];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < QUADS; i++) sum += q[i].x + q[i].y + q[i].z + q[i].w;
        acc += sum;
        node = mix(__float_as_uint(sum) ^ node ^ (uint32_t)s) % n_nodes;   // dependent on the loaded data
    }
    out[tid] = __float_as_uint(acc) ^ node;
}

// 4-lane teams.  Instruction j (0..7): team lane r reads quad (r + 4 * (j & 1)) of the node of team member (j >> 1).
__global__ __launch_bounds__(256) void k_team(const float4* __restrict__ nodes, uint32_t n_nodes, int steps, uint32_t* out) {
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, r = lane & 3, team0 = lane & ~3;
    uint32_t node = mix(tid) % n_nodes;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        float4 got[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t nj = __shfl(node, team0 + (j >> 1), 64);
            got[j] = nodes[(size_t)nj * 8 + r + 4 * (j & 1)];
        }
        // member m's quad k was loaded by lane (k & 3) in instruction 2 m + (k >> 2): give every lane its own 8 quads
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            // the value I need sits in lane team0 + (k & 3), register got[2 * r + (k >> 2)] -- register index depends on MY r,
            // so each source lane selects what it hands out per round: round k serves quad k to everyone
            float4 v;
            {
                // source lane src = team0 + (k & 3) must provide got[2 * m + (k >> 2)] to member m; do 4 shuffles (one per m)
                float4 mine = make_float4(0, 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const float4 g = got[2 * m + (k >> 2)];
                    float4 t;
                    t.x = __shfl(g.x, team0 + (k & 3), 64); t.y = __shfl(g.y, team0 + (k & 3), 64);
                    t.z = __shfl(g.z, team0 + (k & 3), 64); t.w = __shfl(g.w, team0 + (k & 3), 64);
                    if (m == r) mine = t;
                }
                v = mine;
            }
            sum += v.x + v.y + v.z + v.w;
        }
        acc += sum;
        node = mix(__float_as_uint(sum) ^ node ^ (uint32_t)s) % n_nodes;
    }
    out[tid] = __float_as_uint(acc) ^ node;
}

// Same team loads, exchange through LDS (each wave owns 64 x 128 B).
__global__ __launch_bounds__(256) void k_team_lds(const float4* __restrict__ nodes, uint32_t n_nodes, int steps, uint32_t* out) {
    __shared__ float4 s_x[256 * 8];
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, r = lane & 3, team0 = lane & ~3;
    float4* wave = s_x + (threadIdx.x & ~63) * 8;
    uint32_t node = mix(tid) % n_nodes;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t nj = __shfl(node, team0 + (j >> 1), 64);
            const float4 g = nodes[(size_t)nj * 8 + r + 4 * (j & 1)];
            // quad index k = r + 4 (j & 1) of member m = j >> 1: store at [k][member lane] (conflict-free: consecutive lanes)
            wave[(r + 4 * (j & 1)) * 64 + team0 + (j >> 1)] = g;
        }
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 8; k++) { const float4 v = wave[k * 64 + lane]; sum += v.x + v.y + v.z + v.w; }
        acc += sum;
        node = mix(__float_as_uint(sum) ^ node ^ (uint32_t)s) % n_nodes;
    }
    out[tid] = __float_as_uint(acc) ^ node;
}

template <typename F>
static double time_ms(F launch) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main(int argc, char** argv) {
    const uint32_t n_nodes = argc > 1 ? (uint32_t)atoi(argv[1]) : 105000u;
    const int steps = 200, blocks = 256 * 6 * 4;
    std::vector<float> h((size_t)n_nodes * 32);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) & 0xffff) * 1e-3f;
    float4* d; uint32_t* out;
    CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const double visits = (double)blocks * 256 * steps;
    double t;
    t = time_ms([&] { hipLaunchKernelGGL(k_own<8>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, steps, out); });
    printf("A own 8 quads (128 B)  : %7.3f ms  %6.2f Gvisits/s  %7.1f GB/s\n", t, visits / t * 1e-6, visits * 128 / t * 1e-6);
    t = time_ms([&] { hipLaunchKernelGGL(k_own<4>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, steps, out); });
    printf("C own 4 quads (64 B)   : %7.3f ms  %6.2f Gvisits/s  %7.1f GB/s\n", t, visits / t * 1e-6, visits * 64 / t * 1e-6);
    t = time_ms([&] { hipLaunchKernelGGL(k_own<2>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, steps, out); });
    printf("D own 2 quads (32 B)   : %7.3f ms  %6.2f Gvisits/s  %7.1f GB/s\n", t, visits / t * 1e-6, visits * 32 / t * 1e-6);
    t = time_ms([&] { hipLaunchKernelGGL(k_own<1>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, steps, out); });
    printf("E own 1 quad  (16 B)   : %7.3f ms  %6.2f Gvisits/s  %7.1f GB/s\n", t, visits / t * 1e-6, visits * 16 / t * 1e-6);
    t = time_ms([&] { hipLaunchKernelGGL(k_team, dim3(blocks), dim3(256), 0, 0, d, n_nodes, steps, out); });
    printf("B team of 4, shfl      : %7.3f ms  %6.2f Gvisits/s\n", t, visits / t * 1e-6);
    t = time_ms([&] { hipLaunchKernelGGL(k_team_lds, dim3(blocks), dim3(256), 0, 0, d, n_nodes, steps, out); });
    printf("B' team of 4, LDS      : %7.3f ms  %6.2f Gvisits/s\n", t, visits / t * 1e-6);
    return 0;
}
