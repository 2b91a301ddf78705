// Microbenchmark 2: is the divergent 64-B node fetch of the traversal loop bound by TAG LOOKUPS in the vector L1 / texture
// addresser (one per distinct 64-B segment per wave-instruction), and do 4-lane teams that read one node with one
// instruction each (16 segments per instruction instead of 64) lift it?  Every lane walks a dependent chain of random 64-B
// nodes (4 x float4) of a table the size of the bench scene's BVH; 4 waves per SIMD like the megakernel.
//   C    own: 4 x global_load_dwordx4 from the lane's own node                                  (what SPC_NODE_STEP does)
//   TL   team of 4: instruction k reads node of member k, lane r its quad r; exchange through LDS (ds_write_b128 / ds_read_b128)
//   TD   team of 4: the same loads as LDS-DMA (global_load_lds_dwordx4: no VGPR staging), then 4 x ds_read_b128
//   TX   team of 4: the same loads into registers, 4 x 4 transpose inside the quad with DPP quad_perm
//   T2   team of 2: instruction i reads quad 2 (i & 1) + r of member i >> 1; 8 dwords swapped with the neighbour lane by DPP
// Optional VALU filler per visit (argv[2]) stands in for the slab test + sort (~120 instructions in the real loop).
// Build: hipcc --offload-arch=gfx950 -O3 gather_bench2.hip -o gather_bench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

template <int FILL>
__device__ __forceinline__ float filler(float x) {
#pragma unroll
    for (int i = 0; i < FILL; i++) x = fmaf(x, 1.0000001f, 1e-9f);
    return x;
}
__device__ __forceinline__ float sum4(float4 v) { return v.x + v.y + v.z + v.w; }

template <int FILL>
__global__ __launch_bounds__(256, 4) void k_own(const float4* __restrict__ nodes, uint32_t n_nodes, int steps, uint32_t* out) {
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t node = mix(tid) % n_nodes;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const float4* p = nodes + (size_t)node * 4;
        const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        const float sum = filler<FILL>(sum4(q0) + 2.f * sum4(q1) + 3.f * sum4(q2) + 4.f * sum4(q3));
        acc += sum;
        node = mix(__float_as_uint(sum) ^ node ^ (uint32_t)s) % n_nodes;
    }
    out[tid] = __float_as_uint(acc) ^ node;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float dppf(float v) { return __uint_as_float(dpp<CTRL>(__float_as_uint(v))); }
// quad_perm controls: broadcast member k = k * 0x55; xor 1 = [1,0,3,2] = 0xB1; xor 2 = [2,3,0,1] = 0x4E

template <int FILL>
__global__ __launch_bounds__(256, 4) void k_team_lds(const float4* __restrict__ nodes, uint32_t n_nodes, int steps, uint32_t* out) {
    __shared__ float4 s_x[256 * 4];
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, r = lane & 3, team0 = lane & ~3;
    float4* wave = s_x + (threadIdx.x & ~63) * 4;
    uint32_t node = mix(tid) % n_nodes;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const uint32_t n0 = dpp<0x00>(node), n1 = dpp<0x55>(node), n2 = dpp<0xAA>(node), n3 = dpp<0xFF>(node);
        const float4 g0 = nodes[(size_t)n0 * 4 + r], g1 = nodes[(size_t)n1 * 4 + r], g2 = nodes[(size_t)n2 * 4 + r], g3 = nodes[(size_t)n3 * 4 + r];
        // member m's node contiguous at [(team0 + m) * 4 .. + 4): lane r writes quad r of each member
        wave[(team0 + 0) * 4 + r] = g0; wave[(team0 + 1) * 4 + r] = g1; wave[(team0 + 2) * 4 + r] = g2; wave[(team0 + 3) * 4 + r] = g3;
        __builtin_amdgcn_wave_barrier();
        const float4 q0 = wave[lane * 4 + 0], q1 = wave[lane * 4 + 1], q2 = wave[lane * 4 + 2], q3 = wave[lane * 4 + 3];
        __builtin_amdgcn_wave_barrier();
        const float sum = filler<FILL>(sum4(q0) + 2.f * sum4(q1) + 3.f * sum4(q2) + 4.f * sum4(q3));
        acc += sum;
        node = mix(__float_as_uint(sum) ^ node ^ (uint32_t)s) % n_nodes;
    }
    out[tid] = __float_as_uint(acc) ^ node;
}

typedef __attribute__((address_space(3))) void lds_void;
template <int FILL>
__global__ __launch_bounds__(256, 4) void k_team_dma(const float4* __restrict__ nodes, uint32_t n_nodes, int steps, uint32_t* out) {
    __shared__ float4 s_x[256 * 4];
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, r = lane & 3, team = lane >> 2;
    // instruction k writes 64 lanes x 16 B = 1 KiB contiguously at s_k: lane (4 t + r) -> s_k + (4 t + r) * 16 = quad r of
    // the node of member k of team t; member k of team t then reads its node at s_k + t * 64
    float4* wave = s_x + (threadIdx.x & ~63) * 4;
    uint32_t node = mix(tid) % n_nodes;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const uint32_t n0 = dpp<0x00>(node), n1 = dpp<0x55>(node), n2 = dpp<0xAA>(node), n3 = dpp<0xFF>(node);
        __builtin_amdgcn_global_load_lds((const void*)(nodes + (size_t)n0 * 4 + r), (lds_void*)(wave + 0 * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(nodes + (size_t)n1 * 4 + r), (lds_void*)(wave + 1 * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(nodes + (size_t)n2 * 4 + r), (lds_void*)(wave + 2 * 64), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void*)(nodes + (size_t)n3 * 4 + r), (lds_void*)(wave + 3 * 64), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const float4* mine = wave + r * 64 + team * 4;
        const float4 q0 = mine[0], q1 = mine[1], q2 = mine[2], q3 = mine[3];
        __builtin_amdgcn_wave_barrier();
        const float sum = filler<FILL>(sum4(q0) + 2.f * sum4(q1) + 3.f * sum4(q2) + 4.f * sum4(q3));
        acc += sum;
        node = mix(__float_as_uint(sum) ^ node ^ (uint32_t)s) % n_nodes;
    }
    out[tid] = __float_as_uint(acc) ^ node;
}

// butterfly stage of the 4 x 4 transpose inside a quad: registers (a, b) with partner lane ^ X
template <int CTRL>
__device__ __forceinline__ void bfly(float& a, float& b, bool hi) {
    const float give = hi ? a : b;
    const float got = dppf<CTRL>(give);
    a = hi ? got : a;
    b = hi ? b : got;
}
template <int FILL>
__global__ __launch_bounds__(256, 4) void k_team_dpp(const float4* __restrict__ nodes, uint32_t n_nodes, int steps, uint32_t* out) {
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, r = lane & 3;
    const bool b0 = (r & 1) != 0, b1 = (r & 2) != 0;
    uint32_t node = mix(tid) % n_nodes;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const uint32_t n0 = dpp<0x00>(node), n1 = dpp<0x55>(node), n2 = dpp<0xAA>(node), n3 = dpp<0xFF>(node);
        float4 g[4] = {nodes[(size_t)n0 * 4 + r], nodes[(size_t)n1 * 4 + r], nodes[(size_t)n2 * 4 + r], nodes[(size_t)n3 * 4 + r]};
        // lane r holds g[k] = quad r of member k; wants q[j] = quad j of member r: transpose over (k, r)
#define TR(c) bfly<0xB1>(g[0].c, g[1].c, b0); bfly<0xB1>(g[2].c, g[3].c, b0); bfly<0x4E>(g[0].c, g[2].c, b1); bfly<0x4E>(g[1].c, g[3].c, b1);
        TR(x) TR(y) TR(z) TR(w)
#undef TR
        const float sum = filler<FILL>(sum4(g[0]) + 2.f * sum4(g[1]) + 3.f * sum4(g[2]) + 4.f * sum4(g[3]));
        acc += sum;
        node = mix(__float_as_uint(sum) ^ node ^ (uint32_t)s) % n_nodes;
    }
    out[tid] = __float_as_uint(acc) ^ node;
}

template <int FILL>
__global__ __launch_bounds__(256, 4) void k_team2(const float4* __restrict__ nodes, uint32_t n_nodes, int steps, uint32_t* out) {
    uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, r = lane & 1;
    uint32_t node = mix(tid) % n_nodes;
    float acc = 0.f;
    for (int s = 0; s < steps; s++) {
        const uint32_t n0 = dpp<0xA0>(node), n1 = dpp<0xF5>(node);   // [0,0,2,2] and [1,1,3,3]
        // lane r: a = quad r of n0, b = quad 2 + r of n0, c = quad r of n1, d = quad 2 + r of n1
        float4 a = nodes[(size_t)n0 * 4 + r], b = nodes[(size_t)n0 * 4 + 2 + r], c = nodes[(size_t)n1 * 4 + r], d = nodes[(size_t)n1 * 4 + 2 + r];
        // member 0 keeps a, b and needs the partner's a, b; member 1 keeps c, d and needs the partner's c, d
        float4 q0, q1, q2, q3;
#define SW(cmp) { const float g1 = r ? a.cmp : c.cmp, g2 = r ? b.cmp : d.cmp; const float t1 = dppf<0xB1>(g1), t2 = dppf<0xB1>(g2); \
                  q0.cmp = r ? t1 : a.cmp; q1.cmp = r ? c.cmp : t1; q2.cmp = r ? t2 : b.cmp; q3.cmp = r ? d.cmp : t2; }
        SW(x) SW(y) SW(z) SW(w)
#undef SW
        const float sum = filler<FILL>(sum4(q0) + 2.f * sum4(q1) + 3.f * sum4(q2) + 4.f * sum4(q3));
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

template <int FILL>
static void run(const float4* d, uint32_t n_nodes, uint32_t* out, int blocks, int steps) {
    const double visits = (double)blocks * 256 * steps;
    double t;
    printf("-- %d filler FMAs per visit, %u nodes (%.1f MB), %d blocks x 256\n", FILL, n_nodes, n_nodes * 64e-6, blocks);
    t = time_ms([&] { hipLaunchKernelGGL(k_own<FILL>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, steps, out); });
    printf("C  own 4 quads           : %7.3f ms  %6.2f Gvisits/s\n", t, visits / t * 1e-6);
    t = time_ms([&] { hipLaunchKernelGGL(k_team_lds<FILL>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, steps, out); });
    printf("TL team of 4, LDS        : %7.3f ms  %6.2f Gvisits/s\n", t, visits / t * 1e-6);
    t = time_ms([&] { hipLaunchKernelGGL(k_team_dma<FILL>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, steps, out); });
    printf("TD team of 4, LDS-DMA    : %7.3f ms  %6.2f Gvisits/s\n", t, visits / t * 1e-6);
    t = time_ms([&] { hipLaunchKernelGGL(k_team_dpp<FILL>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, steps, out); });
    printf("TX team of 4, DPP transp : %7.3f ms  %6.2f Gvisits/s\n", t, visits / t * 1e-6);
    t = time_ms([&] { hipLaunchKernelGGL(k_team2<FILL>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, steps, out); });
    printf("T2 team of 2, DPP swap   : %7.3f ms  %6.2f Gvisits/s\n", t, visits / t * 1e-6);
}

int main(int argc, char** argv) {
    const uint32_t n_nodes = argc > 1 ? (uint32_t)atoi(argv[1]) : 247000u;
    const int steps = 400, blocks = 256 * 4;
    std::vector<float> h((size_t)n_nodes * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) & 0xffff) * 1e-3f;
    float4* d; uint32_t* out;
    CHECK(hipMalloc(&d, h.size() * 4)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // every variant must compute the same chain: compare the outputs of C and the team variants
    std::vector<uint32_t> ref((size_t)blocks * 256), got(ref.size());
    hipLaunchKernelGGL(k_own<0>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, 16, out);
    CHECK(hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost));
    auto check = [&](const char* name) {
        CHECK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < ref.size(); i++) bad += ref[i] != got[i];
        printf("check %-4s: %zu mismatches\n", name, bad);
    };
    hipLaunchKernelGGL(k_team_lds<0>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, 16, out); check("TL");
    hipLaunchKernelGGL(k_team_dma<0>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, 16, out); check("TD");
    hipLaunchKernelGGL(k_team_dpp<0>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, 16, out); check("TX");
    hipLaunchKernelGGL(k_team2<0>, dim3(blocks), dim3(256), 0, 0, d, n_nodes, 16, out); check("T2");
    run<0>(d, n_nodes, out, blocks, steps);
    run<64>(d, n_nodes, out, blocks, steps);
    run<128>(d, n_nodes, out, blocks, steps);
    return 0;
}
