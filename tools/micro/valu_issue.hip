// Microbenchmark 3: what does one wave64 VALU instruction cost a SIMD on gfx950 -- 2 cycles or 4?
// MI355X_MICROARCH.md says `v_fma_f32` (wave64) runs at 2 cycles on the SIMD-32 and that ONE wave alone sustains only one per 4;
// tools/pmc_summary.py (rounds 1-3) charged 4 cycles per wave-instruction whatever the occupancy, which put k_spcbpt's VALU issue
// at 0.875 of the SIMDs' cycles.  This settles the constant: every wave runs a stream of independent `v_fma_f32` (8 accumulators,
// inline asm so that nothing is packed or folded) and stamps s_memtime around it; with w waves resident per SIMD the SIMD issues
// w x I wave-instructions in the waves' common lifetime T, so cycles per wave-instruction = T / (w x I).
//   grid = CUs x 1 block of 256 x w threads  (w = 1, 2, 4: 1024-thread blocks at most), or several 256-thread blocks per CU (w = 8)
// Also measured, because the megakernel's stream is not FMAs only: the same with every 4th instruction a `v_cndmask_b32`, a
// `v_rcp_f32` (quarter rate), an LDS read, and a scalar instruction in between (does SALU issue cost VALU slots?).
// Build: hipcc --offload-arch=gfx950 -O3 valu_issue.hip -o valu_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define FMA8                                                              \
    asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n" \
                 "v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n" \
                 "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n" \
                 "v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y))
// 6 FMAs + 2 of another instruction (MODE 1: v_cndmask_b32, 2: v_rcp_f32, 3: ds_read_b32, 4: s_mul_i32 pairs between the FMAs)
#define MIX8(OTHER0, OTHER1)                                              \
    asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n" \
                 "v_fma_f32 %2, %8, %9, %2\n" OTHER0 "\n"                 \
                 "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n" \
                 "v_fma_f32 %6, %8, %9, %6\n" OTHER1 "\n"                 \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y), "v"(lds_addr) : "vcc", "s40", "s41")

template <int MODE>
__global__ __launch_bounds__(1024) void k_issue(int iters, float x, float y, unsigned long long* stamps, float* sink, unsigned long long exec_mask) {
    __shared__ float s_buf[2048];
    s_buf[threadIdx.x & 1023] = x; s_buf[1024 + (threadIdx.x & 1023)] = y;
    __syncthreads();
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const uint32_t lds_addr = (threadIdx.x & 1023) * 4;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // MODE 5: the same FMA stream with only the lanes of exec_mask enabled (does a half-empty wave64 instruction skip its second
    // pass on the SIMD-32?  does a nearly empty one cost the same?)
    if (MODE == 5) asm volatile("s_mov_b64 exec, %0" :: "s"(exec_mask) : "memory");
    for (int i = 0; i < iters; i++) {
        if (MODE == 0 || MODE == 5) { FMA8; FMA8; FMA8; FMA8; }
        if (MODE == 7) {   // half of the stream under exec_mask, half under the full mask (what a divergent kernel's stream looks like)
            asm volatile("s_mov_b64 exec, %0" :: "s"(exec_mask) : "memory"); FMA8; FMA8;
            asm volatile("s_mov_b64 exec, -1" ::: "memory"); FMA8; FMA8;
        }
        if (MODE == 1) { MIX8("v_cndmask_b32 %3, %3, %8, vcc", "v_cndmask_b32 %7, %7, %8, vcc"); MIX8("v_cndmask_b32 %3, %3, %8, vcc", "v_cndmask_b32 %7, %7, %8, vcc");
                         MIX8("v_cndmask_b32 %3, %3, %8, vcc", "v_cndmask_b32 %7, %7, %8, vcc"); MIX8("v_cndmask_b32 %3, %3, %8, vcc", "v_cndmask_b32 %7, %7, %8, vcc"); }
        if (MODE == 2) { MIX8("v_rcp_f32 %3, %3", "v_rcp_f32 %7, %7"); MIX8("v_rcp_f32 %3, %3", "v_rcp_f32 %7, %7"); MIX8("v_rcp_f32 %3, %3", "v_rcp_f32 %7, %7"); MIX8("v_rcp_f32 %3, %3", "v_rcp_f32 %7, %7"); }
        if (MODE == 3) { MIX8("ds_read_b32 %3, %10", "ds_read_b32 %7, %10 offset:4096"); MIX8("ds_read_b32 %3, %10", "ds_read_b32 %7, %10 offset:4096");
                         MIX8("ds_read_b32 %3, %10", "ds_read_b32 %7, %10 offset:4096"); MIX8("ds_read_b32 %3, %10", "ds_read_b32 %7, %10 offset:4096");
                         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (MODE == 4) { MIX8("s_mul_i32 s40, s40, s41", "s_mul_i32 s41, s41, s40"); MIX8("s_mul_i32 s40, s40, s41", "s_mul_i32 s41, s41, s40");
                         MIX8("s_mul_i32 s40, s40, s41", "s_mul_i32 s41, s41, s40"); MIX8("s_mul_i32 s40, s40, s41", "s_mul_i32 s41, s41, s40"); }
    }
    if (MODE == 5) asm volatile("s_mov_b64 exec, -1" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { stamps[2 * wave] = t0; stamps[2 * wave + 1] = t1; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char* name, int cus, int waves_per_simd, int iters, unsigned long long exec_mask = ~0ull) {
    // w <= 4: one block of 256 w threads per CU (the dispatcher places one block per CU when the grid is the CU count; checked
    // below from the waves' overlap); w = 8: two such 1024-thread blocks per CU
    const int block = 256 * std::min(waves_per_simd, 4);
    const int grid = cus * (waves_per_simd > 4 ? waves_per_simd / 4 : 1);
    const size_t n_waves = (size_t)grid * block / 64;
    unsigned long long* d_stamps; float* d_sink;
    CHECK(hipMalloc(&d_stamps, n_waves * 16));
    CHECK(hipMalloc(&d_sink, (size_t)grid * block * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_issue<MODE>, dim3(grid), dim3(block), 0, 0, 16, 1.0000001f, 1e-9f, d_stamps, d_sink, exec_mask);   // warm-up
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_issue<MODE>, dim3(grid), dim3(block), 0, 0, iters, 1.0000001f, 1e-9f, d_stamps, d_sink, exec_mask);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st(2 * n_waves);
    CHECK(hipMemcpy(st.data(), d_stamps, n_waves * 16, hipMemcpyDeviceToHost));
    std::vector<double> dur(n_waves);
    for (size_t w = 0; w < n_waves; w++) dur[w] = (double)(st[2 * w + 1] - st[2 * w]);
    std::sort(dur.begin(), dur.end());
    const double med = dur[n_waves / 2], p10 = dur[n_waves / 10], p90 = dur[n_waves * 9 / 10];
    const double instr = (double)iters * 32.0;   // wave-instructions of the measured stream per wave (FMAs + the others)
    // both clocks are printed: the waves' own s_memtime ticks per wave-instruction per SIMD (the guide: one tick = one shader cycle) and,
    // from the event time of the whole launch (all waves resident at once), ns per wave-instruction per SIMD; their ratio is the
    // shader clock the chip really ran at under this load.
    // span of a block = first start to last end of its waves (one block = one CU = one counter): the SIMDs of that CU issued
    // (waves of the block per SIMD) x instr wave-instructions each in that span
    const int wpb = block / 64;
    std::vector<double> span(grid);
    for (int b = 0; b < grid; b++) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < wpb; w++) { lo = std::min(lo, st[2 * ((size_t)b * wpb + w)]); hi = std::max(hi, st[2 * ((size_t)b * wpb + w) + 1]); }
        span[b] = (double)(hi - lo);
    }
    std::sort(span.begin(), span.end());
    const double ticks_per_instr_simd = span[grid / 2] / (instr * std::min(waves_per_simd, 4));   // (w = 8: two blocks share a CU -- read the ns column)
    const double ns_per_instr_simd = (double)ms * 1e6 / (instr * waves_per_simd);
    printf("%-34s w=%d  waves %6zu  kernel %8.3f ms  wave ticks p10/med/p90 %9.0f %9.0f %9.0f  per wave-instruction per SIMD: %.3f ticks (block span), %.4f ns (-> %.2f GHz)\n",
           name, waves_per_simd, n_waves, ms, p10, med, p90, ticks_per_instr_simd, ns_per_instr_simd, ticks_per_instr_simd / ns_per_instr_simd);
    CHECK(hipFree(d_stamps)); CHECK(hipFree(d_sink));
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = argc > 1 ? atoi(argv[1]) : 40000;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    for (int w : {1, 2, 4, 8}) run<0>("v_fma_f32 x 32", cus, w, iters);
    for (int w : {1, 2, 4}) run<1>("24 v_fma_f32 + 8 v_cndmask_b32", cus, w, iters);
    for (int w : {1, 2, 4}) run<2>("24 v_fma_f32 + 8 v_rcp_f32", cus, w, iters);
    for (int w : {1, 2, 4}) run<3>("24 v_fma_f32 + 8 ds_read_b32", cus, w, iters);
    for (int w : {1, 2, 4}) run<4>("24 v_fma_f32 + 8 s_mul_i32", cus, w, iters);
    const struct { const char* name; unsigned long long mask; } masks[] = {
        {"EXEC all 64", ~0ull}, {"EXEC low 32", 0xffffffffull}, {"EXEC high 32", 0xffffffff00000000ull}, {"EXEC low 16", 0xffffull}, {"EXEC 8 lanes", 0xffull},
        {"EXEC 4 lanes", 0xfull}, {"EXEC 1 lane", 0x1ull}, {"EXEC every 8th", 0x0101010101010101ull}, {"EXEC every 2nd", 0x5555555555555555ull}};
    for (auto& m : masks)
        for (int w : {1, 4}) run<5>(m.name, cus, w, iters, m.mask);
    for (auto& m : masks)
        for (int w : {1, 4}) { char nm[96]; snprintf(nm, sizeof(nm), "16 FMA %s / 16 FMA all", m.name); run<7>(nm, cus, w, iters, m.mask); }
    return 0;
}
