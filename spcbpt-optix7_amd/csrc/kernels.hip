// HIP kernels of the SPCBPT hot path for gfx950 (wave64).  One lane = one pixel-sample (eye pass) or one
// light-trace core (light pass); a wave covers an 8x8 pixel tile so primary rays stay coherent.
//   k_spcbpt      <- __raygen__SPCBPT + __closesthit__eyeSubpath(+_LightSource) + __miss__BDPTVertex +
//                    __closesthit__occlusion (raygen.cu:319-443, hit_program.cu:58-147, 246-340)  [megakernel]
//   k_light_trace <- __raygen__lightTrace + __closesthit__lightSubpath (raygen.cu:620-685, hit_program.cu:341-438)
//   k_pt          <- __raygen__pinhole + __closesthit__radiance/lightsource (raygen.cu:71-170, hit_program.cu:148-180, 439-552)
//   sampler build <- MyThrustOp::LVC_Process (cuda_thrust/device_thrust.cu:241-332), on device
#include <hip/hip_runtime.h>

#include "device_lib.h"
#include "eye_walk.h"
#include "kernel_config.h"
#include "kernels.h"

namespace spc {

// The eye megakernel's block.  Its waves share nothing but the LDS copy of the hottest BVH nodes (s_hot below), and a CU holds 16 of
// them whatever the block size (4 per SIMD at 128 VGPRs; 16 x (4 KB of stack + a 5 840-B pool record) = 155 KB of its 160 KB of
// LDS): the larger the block, the fewer copies of that table share what is left -- 19 nodes in each of four 256-thread blocks, 38
// in each of two 512-thread blocks, 64 (all the builder numbers first) in ONE block of 1 024 threads per CU.
#ifndef SPC_EYE_BLOCK
#define SPC_EYE_BLOCK 256
#endif
#ifndef SPC_SECOND_STAGE_ARY
#define SPC_SECOND_STAGE_ARY 2    // 4: sampleSecondStage as a 4-ary search (below)
#endif
// Issue priority of a wave by phase (s_setprio, 0 .. 3): see the traversal pass in k_spcbpt
#ifndef SPC_PRIO_TRAV
#define SPC_PRIO_TRAV 1
#endif
#ifndef SPC_PRIO_CONNECT
#define SPC_PRIO_CONNECT 0
#endif
#ifndef SPC_PRIO_SHADE
#define SPC_PRIO_SHADE 0
#endif
#ifndef SPC_JOINT_FIRST_STAGE
#define SPC_JOINT_FIRST_STAGE 0   // 1 / 2: the first stages of a vertex's connections on one coarse fetch (below: measured, slower)
#endif
static constexpr int EYE_BLOCK = SPC_EYE_BLOCK;
#ifndef SPC_POOL_SLOTS_IN_REGS
// 1: the LVC slot and the pmf of a lane's three connections cross the traversal pass in the lane's own registers (i.e. in its scratch:
// six more dwords of parked state) and are handed to the job loop in the xy of the ray slot, whose direction the pass no longer
// needs -- instead of two 768-B arrays per wave in LDS, which become 96 more hot nodes per block.  Measured (profiles/r05_experiments.md,
// section 24): the scratch costs +1.4 %, 115 instead of 19 hot nodes give back 0.4 %.
#define SPC_POOL_SLOTS_IN_REGS 0
#endif
#if defined(SPC_EYE_HOT_OVERRIDE)
static constexpr int EYE_HOT = SPC_EYE_HOT_OVERRIDE;   // (experiments)
#elif SPC_POOL_SLOTS_IN_REGS
static constexpr int EYE_HOT = HOT_NODES < 115 || SPC_EYE_BLOCK >= 512 ? HOT_NODES : 115;   // node records [0, EYE_HOT) live in LDS (256 threads: what 40 960 B leave)
#else
static constexpr int EYE_HOT = SPC_EYE_BLOCK >= 1024 ? 64 : (SPC_EYE_BLOCK >= 512 ? 38 : 19);   // node records [0, EYE_HOT) live in LDS
#endif
static_assert(EYE_HOT <= HOT_NODES, "the builder numbers HOT_NODES nodes first (layout.h)");
static_assert(STACK_LDS >= 16, "the pooled connections publish 16 dwords per eye vertex through the traversal-stack LDS");
#ifndef SPC_EYE_WAVES
// ... and for the eye megakernel.  Measured on MI355X (bedroom 1080p, ms per frame).  With the pooled if-if traversal, one frame
// per launch: 2 (241 VGPR, no scratch) -> 12.86, 3 (168 VGPR, 252 B scratch) -> 10.72, 4 (128 VGPR, 452 B) -> 11.01,
// 5 (96 VGPR, 688 B) -> 14.16.  With pooled connections, immediate regeneration and batched launches the balance moved: the
// kernel is bound by the latency of dependent gathers (its throughput is 1 : 1.71 : 2.20 at 1, 2, 3 resident blocks per CU),
// and 4 (128 VGPR, 372 B scratch) gives 25.9 instead of 28.4 ms per 4-frame launch -- provided the block fits 4 times into
// the 160 KB of LDS, hence the 16-entry stack and the three eye-vertex dwords that travel by ds_bpermute instead (below).
#define SPC_EYE_WAVES 4
#endif

// The SPCBPT megakernel: persistent waves with per-lane path regeneration.  A wave pulls 8x8 pixel tiles from a global
// queue (one atomicAdd per tile); a lane whose eye path ends writes its pixel and immediately starts the next
// pixel-sample of the wave's pool, so the 64 lanes stay busy although path lengths differ by an order of magnitude.
// Every iteration runs the same phases for all live lanes: pooled traversal pass -> connections of the previous vertex ->
// new vertex + two-stage resampling.  The queue counter saturates, so every wave reaches the exit.
// Known cost: an eye path may live for 50 bounces (a dependent chain of milliseconds); once the queue is empty the waves drain
// their last paths with ever fewer live lanes -- measured with the wave clocks of the counting build, the average wave has
// left after 83 % of the kernel span.  Ordering the queue by the longest path each tile held in the previous frame did not
// shorten that (long paths are decided by Russian roulette, not by the pixel).
// BATCH: the tiles of p.n_frames frames (same camera and bands, each with its own sampler tables, subframe index and result
// buffer: p.frames) share one queue, so a wave keeps regenerating across frame boundaries and the drain phase is paid once per
// batch instead of once per frame -- what a rank's small share of a sharded frame needs.  Every pixel-sample is computed exactly
// as in a launch of its own frame; BATCH = false compiles to the single-frame kernel unchanged.
// CACHE: label caching (device_lib.h).  <false, *, true> are the timed kernels; <true, false, false> evaluates in the reference's
// order and charges its events (the contract's byte table, and the generic form for classifier trees with direction nodes);
// <true, false, true> counts the events the TIMED kernels execute (roofline.frac: what runs, not what the reference would run).
// ENV = false (timed forms only, chosen by the launcher for a scene with neither an environment map nor a material flagged
// `brdf`, DeviceScene::general == 0): the direction tests and the flag's divisions (device_lib.h brdf_div) are compiled out.
template <bool COUNT, bool BATCH, bool CACHE, bool ENV = true>
__global__ __launch_bounds__(EYE_BLOCK, SPC_EYE_WAVES) void k_spcbpt(const KParams p) {
    constexpr int BLOCK = EYE_BLOCK;   // (this kernel's; the other kernels of the file run 256-thread blocks)
    __shared__ uint32_t s_stack[BLOCK * STACK_LDS];
    // everything else a wave keeps in LDS sits in ONE record per wave: every field is then the wave's base (one SGPR) plus a
    // constant that folds into the ds instruction's offset.  As eight separate arrays the eight wave-uniform bases were spilled
    // SGPRs, read back with v_readlane inside the traversal loop.
    struct alignas(16) WavePool {
        float4 ray[POOL_RAYS];      // shadow ray it * 64 + lane: direction.xyz, length (< 0: none)
        float4 org[64];             // eye vertex of lane l: position.xyz (= origin of its shadow rays), lastNormalProjection
#if !SPC_POOL_SLOTS_IN_REGS
        int32_t slot[POOL_RAYS];    // LVC slot of connection it * 64 + lane
        float pmf[POOL_RAYS];       // its resampling pmf (path_count * pmf2 * pmf1)
#endif
        uint8_t job[POOL_RAYS];     // before the pass: slots that hold a ray; after it: the unoccluded connections, compacted
                                    // (the pass answers a shadow ray in the ray's own slot: an occluded pair's length becomes -1 = no ray)
        uint32_t next;              // pool cursor
        uint32_t pad[3];
    };
    __shared__ WavePool s_pool[BLOCK / 64];
    // the hottest nodes of the BVH (layout.h: HOT_NODES, numbered first by the builder), one copy per block
    __shared__ float4 s_hot[EYE_HOT * 4];
    const DeviceScene& S = p.scene;
    for (int i = (int)threadIdx.x; i < EYE_HOT * 4; i += BLOCK) s_hot[i] = i < S.tri_base * 4 ? ldq(S.nodes, (size_t)i) : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    // wave_in_block through readfirstlane: the per-wave LDS base below is then a wave-uniform value the compiler keeps in an SGPR
    const uint32_t lane = threadIdx.x & 63, wave_in_block = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    WavePool* wp = s_pool + wave_in_block;
#if SPC_POOL_SLOTS_IN_REGS
    int32_t my_slot[SPCBPT_CONNECTION_N];   // LVC slot of this lane's connection `it` ...
    float my_pmf[SPCBPT_CONNECTION_N];      // ... and its resampling pmf (path_count * pmf2 * pmf1)
#pragma unroll
    for (int it = 0; it < SPCBPT_CONNECTION_N; it++) { my_slot[it] = 0; my_pmf[it] = 1.0f; }
#else
    int32_t* w_slot = wp->slot;
    float* w_pmf = wp->pmf;
#endif
    uint8_t* w_job = wp->job;
    uint32_t* w_stack = s_stack + wave_in_block * 64;      // [entry * BLOCK + lane]: free between two traversal passes
    float4* w_ray = wp->ray;
    float4* w_org = wp->org;
    uint32_t* w_next = &wp->next;
    Counts<COUNT> cn;
    cn.clear();
    TravStack<BLOCK, STACK_LDS> st;
    st.init(s_stack, p.spill, p.spill_entries, (size_t)blockIdx.x * BLOCK + threadIdx.x, p.diag);
    const int path_count = BATCH ? 0 : p.sampler_counts[1];
    const uint32_t n_tiles = BATCH ? p.n_tiles * p.n_frames : p.n_tiles;   // queue length
    uint32_t fid = 0, pool_fid = 0, pend_fid = 0;   // frame of the lane's path / of the wave's current tile / of the parked pixel

    bool alive = false, exhausted = false;
    uint32_t pool_tile = 0;
    int pool_left = 0;
    uint32_t x = 0, y = 0;
    WalkState w;
    EyeVertex cur;
    f3 result = mk3(0.0f);
    w.done = false; w.seed = 0; w.origin = w.dir = w.next_flux = mk3(0.0f); w.next_single_pdf = 1.0f;
    cur.c.pos = cur.c.n = cur.c.color = cur.c.lastPos = mk3(0.0f); cur.c.lnp = 0.0f; cur.c.mat = 0; cur.c.lld = false;
    cur.flux = cur.R3 = mk3(0.0f); cur.pdf = cur.singlePdf = 1.0f; cur.sub = cur.lastZone = cur.depth = 0; cur.lsub = 0;

    // software pipeline: the vertex built in iteration i is connected in iteration i + 1, in the same traversal pass that
    // extends the path by its next segment (the next direction is drawn before the connections, hit_program.cu:324-337)
    bool has_vertex = false, has_ray = false;
    // A path that ended at a vertex (Russian roulette / depth) still owes that vertex's connections, which are evaluated one
    // iteration later.  Its lane does not wait for them: it parks the pixel and the radiance so far (`pend_*`), starts the next
    // pixel-sample at once (`fresh`: the camera vertex is installed after the connect phase, which still reads `cur`), and
    // writes the parked pixel when the connections have been added.
    bool pend_valid = false, fresh = false;
    uint32_t pend_xy = 0;
    f3 pend_result = mk3(0.0f);
#pragma unroll
    for (int it = 0; it < SPCBPT_CONNECTION_N; it++) w_ray[it * 64 + lane] = make_float4(0.f, 0.f, 0.f, -1.0f);
    const unsigned long long w_start = COUNT ? wall_clock64() : 0ull;
    long long t_ph = COUNT ? clock64() : 0;
#define SPC_PHASE(slot) do { if (COUNT) { const long long t1__ = clock64(); if (lane == 0) cn.add(slot, (unsigned)((t1__ - t_ph) >> 4)); t_ph = t1__; } } while (0)
    while (true) {
        // ---- regeneration: hand pixel-samples of the pool to idle lanes
        unsigned long long idle = __ballot(!alive || !has_ray);
        while (idle != 0ull && !exhausted) {
            if (pool_left == 0) {
                uint32_t t = 0;
                if (lane == (uint32_t)__ffsll((long long)idle) - 1u) t = atomicAdd(p.work_counter, 1u);
                t = __shfl(t, __ffsll((long long)idle) - 1, 64);
                if (t >= n_tiles) { exhausted = true; break; }
                pool_tile = BATCH ? t % p.n_tiles : t;
                pool_fid = BATCH ? t / p.n_tiles : 0u;
                pool_left = 64;
            }
            const int n_idle = __popcll(idle);
            const int take = n_idle < pool_left ? n_idle : pool_left;
            const int my_rank = __popcll(idle & ((1ull << lane) - 1ull));
            if ((!alive || !has_ray) && my_rank < take) {
                const uint32_t slot = (uint32_t)(64 - pool_left + my_rank);
                uint32_t nx, ny;
                if (tile_pixel(p, pool_tile, slot, nx, ny)) {
                    if (alive) { pend_valid = true; pend_xy = x | (y << 16); pend_result = result; pend_fid = fid; }
                    x = nx; y = ny;
                    fid = pool_fid;
                    alive = true;
                    has_ray = true;
                    fresh = true;
                    w.dir = camera_ray(p, x, y, w.seed, BATCH ? p.frames[fid].subframe : p.subframe);
                    w.origin = ld3(p.eye);
                    w.done = false;
                    w.next_flux = mk3(0.0f);
                    w.next_single_pdf = 1.0f;
                    result = mk3(0.0f);
                    cn.add(C_PIX); cn.add(C_EYE);
                }
            }
            pool_left -= take;
            // slots that fall outside the image (partial tiles) are consumed; their lanes stay idle for this round
            const unsigned long long still = __ballot(!alive || !has_ray);
            if (still == idle && pool_left > 0) break;  // only out-of-image slots were handed out: avoid spinning
            idle = still;
        }
        if (!__any(alive)) {
            if (exhausted) break;
            continue;
        }
        SPC_PHASE(C_T_REGEN);
        // ---- traversal pass: the next segment of every live path and the shadow rays of the vertices built last iteration
        if (lane == 0) *w_next = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the slots that hold a shadow ray, compacted (w_job is free here: the connect phase below rebuilds it after the pass)
        const uint32_t n_rays = pool_ray_list(w_ray, w_job, reinterpret_cast<float*>(w_next + 1));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the vertex's five small integers cross the pass in two registers (the pass needs every register it can get: the kernel spills)
        const uint32_t ids_a = (uint32_t)cur.sub | ((uint32_t)cur.lastZone << 10) | ((uint32_t)cur.depth << 20);
        const uint32_t ids_b = (uint32_t)cur.c.mat | ((uint32_t)cur.lsub << 16);
        HitRec h;
        // (the next segment starts at the path's last vertex -- or at the camera for a path that was started in this iteration, whose
        // `cur` still holds the parked path's vertex: w.origin would be a copy kept alive across the pass for nothing)
        // Issue priority by phase (s_setprio): the traversal pass is the phase whose instructions are the kernel's throughput (three quarters
        // of what it issues), connect and shading are chains of dependent fetches with little to issue in between -- a wave in the pass
        // goes first when both are ready.  Measured (profiles/r05_experiments.md, section 16): pass 1 / others 0: +1.1 % paths per second;
        // any phase but the pass raised: the light pass that shares the CUs (priority 0 throughout) starves and the step gets longer.
        if (SPC_PRIO_TRAV != SPC_PRIO_SHADE || SPC_PRIO_TAIL >= 0) __builtin_amdgcn_s_setprio(SPC_PRIO_TRAV);
        trace_pool(S, st, alive && has_ray, fresh ? ld3(p.eye) : cur.c.pos, w.dir, h, w_org, w_ray, w_next, w_job, n_rays, cn, s_hot, EYE_HOT);
        if (SPC_PRIO_CONNECT != SPC_PRIO_TRAV || SPC_PRIO_TAIL >= 0) __builtin_amdgcn_s_setprio(SPC_PRIO_CONNECT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cur.sub = (int)(ids_a & 1023u); cur.lastZone = (int)((ids_a >> 10) & 1023u); cur.depth = (int)(ids_a >> 20);
        cur.c.mat = (int)(ids_b & 0xffffu); cur.lsub = (int)(ids_b >> 16);
        SPC_PHASE(C_T_POOL);
        // ---- connect the unoccluded pairs of the previous vertices.  Only ~1/4 of the 192 (lane, connection) slots of a wave
        // hold an unoccluded pair, so the pairs are compacted into a job list and every lane -- whatever the state of its
        // own path -- evaluates one job per round: the eye vertices are published through the (now idle) traversal-stack
        // LDS, the results come back through the ray slots and each owner adds its own in connection order, which keeps
        // the floating-point sums identical to evaluating them in place.
        {
            uint32_t my_live = 0u, n_jobs = 0u;
#pragma unroll
            for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                const bool live = has_vertex && w_ray[it * 64 + lane].w >= 0.0f;   // a ray was shot and found nothing in the way
                const unsigned long long m = __ballot(live);
                if (live) {
                    w_job[n_jobs + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)(it * 64 + lane);
                    my_live |= 1u << it;
#if SPC_POOL_SLOTS_IN_REGS
                    w_ray[it * 64 + lane] = make_float4(__int_as_float(my_slot[it]), my_pmf[it], 0.0f, 0.0f);   // (the pass is over: the slot's direction is free)
#endif
                }
                n_jobs += (uint32_t)__popcll(m);
            }
            if (n_jobs != 0u) {
                if (my_live) {  // publish this lane's eye vertex (position and lastNormalProjection already sit in w_org)
                    uint32_t* col = w_stack + lane;
                    col[0 * BLOCK] = __float_as_uint(cur.c.n.x); col[1 * BLOCK] = __float_as_uint(cur.c.n.y); col[2 * BLOCK] = __float_as_uint(cur.c.n.z);
                    col[3 * BLOCK] = __float_as_uint(cur.c.color.x); col[4 * BLOCK] = __float_as_uint(cur.c.color.y); col[5 * BLOCK] = __float_as_uint(cur.c.color.z);
                    col[6 * BLOCK] = __float_as_uint(cur.c.lastPos.x); col[7 * BLOCK] = __float_as_uint(cur.c.lastPos.y); col[8 * BLOCK] = __float_as_uint(cur.c.lastPos.z);
                    col[9 * BLOCK] = __float_as_uint(cur.flux.x); col[10 * BLOCK] = __float_as_uint(cur.flux.y); col[11 * BLOCK] = __float_as_uint(cur.flux.z);
                    col[12 * BLOCK] = __float_as_uint(cur.pdf); col[13 * BLOCK] = __float_as_uint(cur.singlePdf);
                    // the frame of the VERTEX: a lane that parked its pixel has already taken a tile of possibly another frame
                    col[14 * BLOCK] = (uint32_t)cur.sub | ((uint32_t)cur.lastZone << 10) | ((uint32_t)cur.depth << 20) | ((pend_valid ? pend_fid : fid) << 26);   // depth <= 51 (raygen.cu:361): 6 bits; frame id: 6 bits
                    col[15 * BLOCK] = (uint32_t)cur.c.mat | ((uint32_t)cur.lsub << 16);   // material ids are < 32768 (spcbpt_create)
                    // (RMIS_pointer_3 does not fit the 16 stack entries four resident blocks leave: it travels by ds_bpermute below)
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (uint32_t base = 0; base < n_jobs; base += 64u) {   // wave-uniform: the shuffles below need every lane
                    const uint32_t j = base + lane;
                    const bool job = j < n_jobs;
                    const uint32_t slot = job ? w_job[j] : 0u, owner = slot & 63u;
                    const f3 ownerR3 = mk3(__shfl(cur.R3.x, (int)owner, 64), __shfl(cur.R3.y, (int)owner, 64), __shfl(cur.R3.z, (int)owner, 64));
                    if (COUNT) { if (job) cn.add(C_U_JOB_LANES); if (lane == 0) cn.add(C_U_JOB_SLOTS, 64); }
                    if (!job) continue;
                    const uint32_t* col = w_stack + owner;
                    const float4 po = w_org[owner];
                    EyeVertex a;
                    a.c.pos = mk3(po.x, po.y, po.z); a.c.lnp = po.w; a.c.lld = false;
                    a.c.n = mk3(__uint_as_float(col[0 * BLOCK]), __uint_as_float(col[1 * BLOCK]), __uint_as_float(col[2 * BLOCK]));
                    a.c.color = mk3(__uint_as_float(col[3 * BLOCK]), __uint_as_float(col[4 * BLOCK]), __uint_as_float(col[5 * BLOCK]));
                    a.c.lastPos = mk3(__uint_as_float(col[6 * BLOCK]), __uint_as_float(col[7 * BLOCK]), __uint_as_float(col[8 * BLOCK]));
                    a.flux = mk3(__uint_as_float(col[9 * BLOCK]), __uint_as_float(col[10 * BLOCK]), __uint_as_float(col[11 * BLOCK]));
                    a.R3 = ownerR3;
                    a.pdf = __uint_as_float(col[12 * BLOCK]); a.singlePdf = __uint_as_float(col[13 * BLOCK]);
                    const uint32_t ids = col[14 * BLOCK];
                    a.sub = (int)(ids & 1023u); a.lastZone = (int)((ids >> 10) & 1023u); a.depth = (int)((ids >> 20) & 63u);
                    const LightVertex* job_lvc = BATCH ? p.frames[ids >> 26].lvc_sorted : p.lvc_sorted;   // (w_slot holds the vertex's place in the sampler's order)
                    a.c.mat = (int)(col[15 * BLOCK] & 0xffffu); a.lsub = (int)(col[15 * BLOCK] >> 16);
                    LightVertex b;
#if SPC_POOL_SLOTS_IN_REGS
                    const float4 sp = w_ray[slot];
                    const float4* src = reinterpret_cast<const float4*>(job_lvc + __float_as_int(sp.x));
                    const float job_pmf = sp.y;
#else
                    const float4* src = reinterpret_cast<const float4*>(job_lvc + w_slot[slot]);
                    const float job_pmf = w_pmf[slot];
#endif
                    float4* dst = reinterpret_cast<float4*>(&b);
#pragma unroll
                    for (int q = 0; q < 6; q++) dst[q] = src[q];
                    f3 res = connect_vertices<COUNT, CACHE, ENV>(p, a, b, cn);
                    if (is_invalid(res)) res = mk3(0.0f);
                    res = res / job_pmf;
                    const bool ok = !is_invalid(res);
                    res = res / (float)SPCBPT_CONNECTION_N;
                    w_ray[slot] = make_float4(res.x, res.y, res.z, ok ? 1.0f : 0.0f);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                f3 sum = pend_valid ? pend_result : result;
#pragma unroll
                for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                    if (my_live & (1u << it)) {
                        const float4 r = w_ray[it * 64 + lane];
                        if (r.w != 0.0f) sum += mk3(r.x, r.y, r.z);
                    }
                }
                if (pend_valid) pend_result = sum;
                else result = sum;
            }
        }
        if (pend_valid) {
            if (BATCH) film_store(p.frames[pend_fid].result, p.width, pend_xy & 0xffffu, pend_xy >> 16, pend_result);
            else film_write(p, pend_xy & 0xffffu, pend_xy >> 16, pend_result);
            pend_valid = false;
        }
        if (fresh) {  // init_EyeSubpath (raygen.cu:216-231)
            fresh = false;
            cur.c.pos = ld3(p.eye); cur.c.n = w.dir; cur.c.color = mk3(0.0f); cur.c.lastPos = cur.c.pos; cur.c.lnp = 0.0f; cur.c.mat = 0; cur.c.lld = false;
            cur.flux = mk3(1.0f); cur.R3 = mk3(0.0f); cur.pdf = 1.0f; cur.singlePdf = 1.0f; cur.sub = 0; cur.lastZone = 0; cur.depth = 0; cur.lsub = 0;
        }
        has_vertex = false;
        bool finished = alive && !has_ray;  // the path ended at that vertex (Russian roulette / depth): nothing was traced
        SPC_PHASE(C_T_CONNECT);
        if (SPC_PRIO_SHADE != SPC_PRIO_CONNECT) __builtin_amdgcn_s_setprio(SPC_PRIO_SHADE);
        // ---- the new segment: miss, emitter, or a new vertex with its CONNECTION_N resampled light vertices
        if (alive && has_ray) {
            has_ray = false;
            if (h.tri < 0) {
                finished = true;  // __miss__BDPTVertex
            } else {
                const Geom g = local_geometry(S, h);
                const bool last_is_origin = cur.depth == 0;
                const f3 ray_dir = w.dir;
                if (g.emitter) {
                    result += eye_emitter_hit<COUNT, CACHE, ENV>(p, g, h.t, ray_dir, last_is_origin, cur, w, cn);
                    finished = true;
                } else {
                    EyeVertex mid;
                    eye_surface_hit<COUNT, CACHE, ENV>(p, g, h.t, ray_dir, last_is_origin, cur, w, mid, cn, true);
                    cur = mid;
                    has_vertex = true;
                    long long t_s0 = COUNT ? clock64() : 0;
                    // CONNECTION_N resampled connections through the subspace sampling matrix (raygen.cu:390-419).  Only the
                    // position quad of the light vertex is fetched here (visibilityTest, cuProg.h:463-487); the connection
                    // itself does not consume random numbers, so drawing all three first leaves the RNG stream unchanged.
                    // (the light vertices in the sampler's order: the vertex drawn at place k of a subspace's CMF is record jump_bias + k,
                    // next to the other vertices of its subspace -- no trip through `jump`)
                    const LightVertex* f_lvc = p.lvc_sorted; const DSubspace* f_subspace = p.subspace; const float* f_cmfs = p.cmfs; const uint32_t* f_guide = p.guide;
                    int f_path_count = path_count;
                    const int32_t* f_counts = p.sampler_counts;
                    if (BATCH) {   // the sampler tables of this path's frame
                        const FrameDesc& D = p.frames[fid];
                        f_lvc = D.lvc_sorted; f_subspace = D.subspace; f_cmfs = D.cmfs; f_guide = D.guide; f_path_count = D.sampler_counts[1];
                        f_counts = D.sampler_counts;
                    }
                    // Three stages, each over all CONNECTION_N connections, so that what does not depend on each other is in flight together:
                    // (1) per connection, in order (the random numbers are one stream, and an empty subspace draws none for its second stage):
                    //     the light subspace and its record; (2) the bisections of sampleSecondStage side by side -- one round trip per level
                    //     for the three of them instead of three; (3) the sampled slots, the light vertices' position quads and the rays.
                    float pmf1_[SPCBPT_CONNECTION_N], pmf2_[SPCBPT_CONNECTION_N], u2_[SPCBPT_CONNECTION_N];
                    int lslot_[SPCBPT_CONNECTION_N], bias_[SPCBPT_CONNECTION_N], size_[SPCBPT_CONNECTION_N];
#pragma unroll
                    for (int it = 0; it < SPCBPT_CONNECTION_N; it++) { pmf1_[it] = 1.0f; pmf2_[it] = 0.0f; u2_[it] = 0.0f; lslot_[it] = -1; bias_[it] = 0; size_[it] = 0; }
                    // The random numbers of a vertex's connections are ONE stream -- u1, [u2 unless the light subspace drawn with u1 is
                    // empty], u1, ... -- so connection k's first number is known only when connection k - 1's subspace record has arrived:
                    // four dependent round trips per connection, twelve per vertex.  An empty subspace is never drawn from a trained
                    // matrix (its Gamma column is zero) and rarely otherwise, so the numbers are drawn as if none were empty: the
                    // CONNECTION_N first stages then run side by side on one coarse fetch (sample_first_stage_n: three round trips for all
                    // of them), the subspace records follow together, and the guess is checked -- a vertex with an empty subspace in
                    // front of its last connection starts over in the reference's order (the loop below), with the seed as it was.
                    bool in_order = SPC_JOINT_FIRST_STAGE == 0 || p.uniform_lvc != 0 || p.cmf_gamma2 == nullptr;
                    if (!in_order) {
                        uint32_t sd = w.seed;
                        float u1[SPCBPT_CONNECTION_N], u2[SPCBPT_CONNECTION_N], pm[SPCBPT_CONNECTION_N];
                        uint32_t after_u1[SPCBPT_CONNECTION_N];
                        int l[SPCBPT_CONNECTION_N];
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) { u1[it] = rnd(sd); after_u1[it] = sd; u2[it] = rnd(sd); }
#if SPC_GUIDE
                        int windows[SPCBPT_CONNECTION_N];
                        sample_first_stage_guided_n<SPCBPT_CONNECTION_N>(p.cmf_gamma2, p.cmf_guide1, cur.sub, u1, l, pm, windows);
#else
                        sample_first_stage_n<SPCBPT_CONNECTION_N, SPC_JOINT_FIRST_STAGE == 2>(p.cmf_gamma2, cur.sub, u1, l, pm);
#endif
                        DSubspace ss[SPCBPT_CONNECTION_N];
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) ss[it] = f_subspace[l[it]];
#pragma unroll
                        for (int it = 0; it + 1 < SPCBPT_CONNECTION_N; it++) in_order = in_order || ss[it].size == 0;
                        if (!in_order) {
#pragma unroll
                            for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                                pmf1_[it] = pm[it];
                                if (ss[it].size != 0) { bias_[it] = ss[it].jump_bias; size_[it] = ss[it].size; u2_[it] = u2[it]; }
#if SPC_GUIDE
                                if (COUNT) cn.add(C_CMF, CACHE ? 1u + (unsigned)SPC_GUIDE_WINDOW * (unsigned)windows[it] : (unsigned)bisection_probes(l[it], SPCBPT_NUM_SUBSPACE));
#else
                                if (COUNT) cn.add(C_CMF, CACHE ? (it == 0 ? 32u : 16u) : (unsigned)bisection_probes(l[it], SPCBPT_NUM_SUBSPACE));
#endif
                            }
                            w.seed = ss[SPCBPT_CONNECTION_N - 1].size != 0 ? sd : after_u1[SPCBPT_CONNECTION_N - 1];
                        }
                    }
                    if (in_order) {
#pragma unroll
                    for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                        if (p.uniform_lvc) {   // the comparator of BASELINE config 5: uniformSample (cuProg.h:283-289), one random number
                            const int vc = f_counts[0];
                            if (vc > 0) lslot_[it] = uniform_sample_index(vc, w.seed, pmf2_[it]);   // (place in the jump buffer = record of the sorted cache)
                        } else {
                            const int l = sample_first_stage<COUNT, CACHE>(p, cur.sub, w.seed, pmf1_[it], cn);
                            const DSubspace ss = f_subspace[l];
                            if (ss.size != 0) { bias_[it] = ss.jump_bias; size_[it] = ss.size; u2_[it] = rnd(w.seed); }
                        }
                    }
                    }
#if SPC_GUIDE
                    {   // binary_sample (cuProg.h:245-264) of the three through the guide table (device_lib.h: guide_window); every sampler build
                        // writes one (capi.hip: set_guide is allocated with the CMF), so there is no bisection beside it in this build
                        GuideScan s_[SPCBPT_CONNECTION_N];
                        int pos_[SPCBPT_CONNECTION_N], first_[SPCBPT_CONNECTION_N];
                        bool open_[SPCBPT_CONNECTION_N];
                        uint32_t g_[SPCBPT_CONNECTION_N];
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++)
                            g_[it] = size_[it] > 0 ? f_guide[bias_[it] + min((int)(u2_[it] * (float)size_[it]), size_[it] - 1)] : 0u;
#ifndef SPC_GUIDE_SIDE_BY_SIDE
#define SPC_GUIDE_SIDE_BY_SIDE 0   // 1: the windows of the three connections in flight together (24 registers of CMF values: spills, measured)
#endif
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                            const int c0 = max((int)g_[it] - 1, 0);
                            s_[it].cnt = c0; s_[it].lo = -INFINITY; s_[it].hi = INFINITY;
                            first_[it] = bias_[it] + c0; pos_[it] = first_[it] & ~3;
                            open_[it] = size_[it] > 0;
                            if (COUNT && CACHE && open_[it]) cn.add(C_CMF);   // (the guide entry; the reference-order form charges the bisection's probes below)
                        }
#if SPC_GUIDE_SIDE_BY_SIDE
                        bool any_open = false;
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) any_open = any_open || open_[it];
                        while (any_open) {
                            float4 a_[SPCBPT_CONNECTION_N], b_[SPCBPT_CONNECTION_N];
#pragma unroll
                            for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                                if (open_[it]) {
                                    a_[it] = *reinterpret_cast<const float4*>(f_cmfs + pos_[it]);
                                    b_[it] = *reinterpret_cast<const float4*>(f_cmfs + pos_[it] + 4);
                                }
                            }
                            any_open = false;
#pragma unroll
                            for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                                if (open_[it]) {
                                    if (COUNT && CACHE) cn.add(C_CMF, 8);
                                    guide_window(a_[it], b_[it], pos_[it], first_[it], bias_[it] + size_[it], u2_[it], s_[it]);
                                    pos_[it] += 8;
                                    open_[it] = !(s_[it].hi < INFINITY) && pos_[it] < bias_[it] + size_[it];
                                }
                                any_open = any_open || open_[it];
                            }
                        }
#else
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                            while (open_[it]) {
                                const float4 a = *reinterpret_cast<const float4*>(f_cmfs + pos_[it]);
                                const float4 b = SPC_GUIDE_WINDOW == 8 ? *reinterpret_cast<const float4*>(f_cmfs + pos_[it] + 4) : a;
                                if (COUNT && CACHE) cn.add(C_CMF, SPC_GUIDE_WINDOW);
                                guide_window(a, b, pos_[it], first_[it], bias_[it] + size_[it], u2_[it], s_[it]);
                                pos_[it] += SPC_GUIDE_WINDOW;
                                open_[it] = !(s_[it].hi < INFINITY) && pos_[it] < bias_[it] + size_[it];
                            }
                        }
#endif
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                            if (size_[it] != 0) {
                                int k = s_[it].cnt;
                                if (k >= size_[it]) {   // no entry above u (the build ends every CMF with 1: not reached): the bisection's last bin
                                    const float* cmf = f_cmfs + bias_[it];
                                    k = size_[it] - 1;
                                    pmf2_[it] = k == 0 ? cmf[k] : cmf[k] - cmf[k - 1];
                                } else {
                                    pmf2_[it] = k == 0 ? s_[it].hi : s_[it].hi - s_[it].lo;
                                }
                                lslot_[it] = bias_[it] + k;   // its record in the sorted cache (what jump[bias + k] names in the cache's own order)
                                if (COUNT && !CACHE) cn.add(C_CMF, (unsigned)bisection_probes(k, size_[it]));
                            }
                        }
                    }
#else
                    {   // binary_sample (cuProg.h:245-264) of the three, level by level
                        int lo_[SPCBPT_CONNECTION_N], hi_[SPCBPT_CONNECTION_N], mid_[SPCBPT_CONNECTION_N];
#if SPC_SECOND_STAGE_ARY == 4
                        // ... as a 4-ary search: the sampler's CMFs are non-decreasing by construction (k_sb_cmf: a normalised prefix sum,
                        // a zero-weight subspace is uniform), so the bisection's bin is the first k with u < cmf[k], size - 1 if there is
                        // none -- three probes per level find it in half the dependent round trips (five for 263 entries instead of nine)
                        (void)mid_;
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) { lo_[it] = 0; hi_[it] = size_[it] > 0 ? size_[it] - 1 : 0; }
                        bool any_open = false;
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) any_open = any_open || hi_[it] > lo_[it];
                        while (any_open) {
                            float a_[SPCBPT_CONNECTION_N], b_[SPCBPT_CONNECTION_N], c_[SPCBPT_CONNECTION_N];
                            int m1_[SPCBPT_CONNECTION_N], m2_[SPCBPT_CONNECTION_N], m3_[SPCBPT_CONNECTION_N];
#pragma unroll
                            for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                                const int n = hi_[it] - lo_[it];
                                m1_[it] = lo_[it] + (n >> 2); m2_[it] = lo_[it] + (n >> 1); m3_[it] = lo_[it] + ((3 * n) >> 2);
                                const bool open = n > 0;
                                a_[it] = open ? f_cmfs[bias_[it] + m1_[it]] : 0.0f;
                                b_[it] = open ? f_cmfs[bias_[it] + m2_[it]] : 0.0f;
                                c_[it] = open ? f_cmfs[bias_[it] + m3_[it]] : 0.0f;
                            }
                            any_open = false;
#pragma unroll
                            for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                                if (hi_[it] > lo_[it]) {
                                    if (COUNT) cn.add(C_CMF, 3);
                                    const float u = u2_[it];
                                    if (u < a_[it]) hi_[it] = m1_[it];
                                    else if (u < b_[it]) { lo_[it] = m1_[it] + 1; hi_[it] = m2_[it]; }
                                    else if (u < c_[it]) { lo_[it] = m2_[it] + 1; hi_[it] = m3_[it]; }
                                    else lo_[it] = m3_[it] + 1;
                                }
                                any_open = any_open || hi_[it] > lo_[it];
                            }
                        }
#else
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) { lo_[it] = 0; hi_[it] = size_[it]; mid_[it] = size_[it] / 2 - 1; }
                        bool any_open = false;
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) any_open = any_open || hi_[it] - lo_[it] > 1;
                        while (any_open) {
                            float v_[SPCBPT_CONNECTION_N];
#pragma unroll
                            for (int it = 0; it < SPCBPT_CONNECTION_N; it++) v_[it] = hi_[it] - lo_[it] > 1 ? f_cmfs[bias_[it] + mid_[it]] : 0.0f;
                            any_open = false;
#pragma unroll
                            for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                                if (hi_[it] - lo_[it] > 1) {
                                    cn.add(C_CMF);
                                    if (u2_[it] < v_[it]) hi_[it] = mid_[it] + 1;
                                    else lo_[it] = mid_[it] + 1;
                                    mid_[it] = (lo_[it] + hi_[it]) / 2 - 1;
                                }
                                any_open = any_open || hi_[it] - lo_[it] > 1;
                            }
                        }
#endif
#pragma unroll
                        for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                            if (size_[it] != 0) {
                                const float* cmf = f_cmfs + bias_[it];
                                const int k = lo_[it];
                                pmf2_[it] = k == 0 ? cmf[k] : cmf[k] - cmf[k - 1];
                                lslot_[it] = bias_[it] + k;   // its record in the sorted cache (what jump[bias + k] names in the cache's own order)
                            }
                        }
                    }
#endif
#pragma unroll
                    for (int it = 0; it < SPCBPT_CONNECTION_N; it++) {
                        const float pmf1 = pmf1_[it], pmf2 = pmf2_[it];
                        const int lslot = lslot_[it];
                        float4 rq = make_float4(0.f, 0.f, 0.f, -1.0f);
                        if (lslot >= 0) {
#if SPC_POOL_SLOTS_IN_REGS
                            my_slot[it] = lslot;
#else
                            w_slot[it * 64 + lane] = lslot;
#endif
                            cn.add(C_CONN);
                            const float4 bq0 = reinterpret_cast<const float4*>(f_lvc + lslot)[0];
                            const float4 bq1 = reinterpret_cast<const float4*>(f_lvc + lslot)[1];
#if SPC_POOL_SLOTS_IN_REGS
                            my_pmf[it] = (float)f_path_count * pmf2 * pmf1;
#else
                            w_pmf[it * 64 + lane] = (float)f_path_count * pmf2 * pmf1;
#endif
                            // a light vertex that is a DIRECTION of the environment map (only scenes with one pay the flag fetch):
                            // visibilityTest shoots from the eye vertex to eye - 10 r n_b (cuProg.h:489-495)
                            const bool b_dir = ENV && (f_lvc[lslot].pad & SPCBPT_LV_DIRECTION) != 0u;
                            const f3 target = b_dir ? -10 * S.env.r * mk3(bq1.x, bq1.y, bq1.z) + cur.c.pos : mk3(bq0.x, bq0.y, bq0.z);
                            const f3 bias = target - cur.c.pos;
                            const float len = sqrtf(dot(bias, bias));
                            const f3 sdir = bias / len;
                            // a pair that faces away on either side has a BSDF factor of exactly zero (bsdf_eval / the one-sided
                            // emitter term of connect_vertices): its shadow ray cannot change the pixel and is not traced
                            if (b_dir ? !null_connection_direction(cur.c.n, mk3(bq1.x, bq1.y, bq1.z))
                                      : !null_connection(cur.c.pos, cur.c.n, mk3(bq0.x, bq0.y, bq0.z), mk3(bq1.x, bq1.y, bq1.z)))
                                rq = make_float4(sdir.x, sdir.y, sdir.z, len);
                        }
                        w_ray[it * 64 + lane] = rq;
                    }
                    if (COUNT) cn.add(C_T_SAMPLE, (unsigned)((clock64() - t_s0) >> 4));
                    w_org[lane] = make_float4(cur.c.pos.x, cur.c.pos.y, cur.c.pos.z, cur.c.lnp);
                    // the loop-top test of raygen.cu:361: a path that ends here still connects this vertex (next iteration)
                    has_ray = !(w.done || cur.depth > 50);   // (cur.depth = the number of segments traced: raygen.cu:361 counts them in payload.depth)
                }
            }
        }
        if (!has_vertex) {
#pragma unroll
            for (int it = 0; it < SPCBPT_CONNECTION_N; it++) w_ray[it * 64 + lane] = make_float4(0.f, 0.f, 0.f, -1.0f);
        }
        if (alive && finished) {
            if (BATCH) film_store(p.frames[fid].result, p.width, x, y, result);
            else film_write(p, x, y, result);
            alive = false;
        }
        SPC_PHASE(C_T_SHADE);
    }
#undef SPC_PHASE
    if (COUNT && p.counters && lane == 0) {
        const unsigned long long w_end = wall_clock64();
        atomicMin(&p.counters[C_W_START_MIN], w_start);
        atomicMax(&p.counters[C_W_END_MAX], w_end);
        atomicAdd(&p.counters[C_W_END_SUM], w_end);
        atomicAdd(&p.counters[C_W_WAVES], 1ull);
    }
    cn.flush(p.counters);
}

// ------------------------------------------------------------------------------------------------
template <bool COUNT>
__global__ __launch_bounds__(BLOCK, SPC_WAVES) void k_pt(const KParams p) {
    __shared__ uint32_t s_stack[BLOCK * STACK_LDS];
    uint32_t x, y;
    const bool active = lane_pixel(p, x, y);
    Counts<COUNT> cn;
    cn.clear();
    if (active) {
        const DeviceScene& S = p.scene;
        TravStack<BLOCK, STACK_LDS> st;
        st.init(s_stack, p.spill, p.spill_entries, (size_t)blockIdx.x * BLOCK + threadIdx.x, p.diag);
        uint32_t seed;
        f3 dir = camera_ray(p, x, y, seed);
        f3 origin = ld3(p.eye);
        f3 throughput = mk3(1.0f), result = mk3(0.0f);
        float prd_pdf = 0.0f;
        int depth = 0;
        bool done = false;
        cn.add(C_PIX); cn.add(C_EYE);
        while (true) {
            HitRec h;
            cn.add(C_CLOSEST);
            f3 current = mk3(0.0f), visA = mk3(0.0f), visB = mk3(0.0f);
            if (!traverse<false, COUNT>(S, st, origin, dir, kEps, 1e16f, h, cn)) {
                done = true;  // __miss__constant_radiance (raygen.cu:687-697): the sky is seen by primary rays only
                if (depth == 0 && S.env.valid) result = throughput * env_color(S.env, dir);
            } else {
                const Geom g = local_geometry(S, h);
                Pbr pbr = load_pbr(S, g.mat);
                if (g.emitter) {  // __closesthit__lightsource
                    const DLight& L = S.lights[pbr.light_id];
                    const LightSampleD ls = light_reverse_sample(S, L, g.u, g.v);
                    if (dot(dir, ls.normal) <= 0) {
                        float mis = 1.0f;
                        if (depth != 0) {
                            const float pdf_hit = prd_pdf * fabsf(dot(dir, ls.normal)) / (h.t * h.t);
                            mis = pdf_hit / (ls.pdf + pdf_hit);
                        }
                        result += throughput * ls.emission * mis;
                    }
                    done = true;
                } else {  // __closesthit__radiance
                    color_tex_sample(S, g, pbr, cn);
                    f3 N = g.N;
                    if (dot(N, dir) > 0.f) N = -N;
                    const f3 in_dir = -dir;
                    const float rr = clampf(max3(pbr.base), SPCBPT_MIN_RR_RATE, 1.0f);
                    const int lid = pick_light(S, seed);
                    const DLight& L = S.lights[lid];
                    if (L.type == 1) {   // next-event estimation of the environment map (hit_program.cu:502-518)
                        const f3 direction = env_sample(S.env, seed);
                        const f3 emission = env_color(S.env, direction);
                        const float lpdf = env_pdf(S.env, direction) / (float)S.n_lights;
                        const f3 V = -normalize(dir);
                        const float L_dot_N = dot(direction, N);
                        if (L_dot_N > 0.0f) {
                            visA = g.P; visB = g.P + direction + mk3(S.env.r * 2);   // float3 + float adds the scalar to every component: as written upstream
                            const f3 eval = bsdf_eval(pbr, N, V, direction);
                            current = throughput * emission / lpdf * eval * L_dot_N;
                        }
                    } else {
                        const float r1 = rnd(seed), r2 = rnd(seed);
                        const LightSampleD ls = light_reverse_sample(S, L, r1, r2);
                        const f3 dvec = ls.position - g.P;
                        const float L_dist = sqrtf(dot(dvec, dvec));
                        const f3 Ld = dvec / L_dist;
                        const f3 V = -normalize(dir);
                        const float L_dot_LN = dot(-Ld, ls.normal);
                        const float N_dot_L = dot(N, Ld), N_dot_V = dot(N, V);
                        if (N_dot_L > 0.0f && N_dot_V > 0.0f && L_dot_LN > 0.0f) {
                            visA = g.P; visB = ls.position;
                            const f3 eval = bsdf_eval(pbr, N, V, Ld);
                            const float pdf_hit = bsdf_pdf(pbr, N, V, Ld) * fabsf(L_dot_LN) / (L_dist * L_dist) * rr;
                            const float mis = ls.pdf / (pdf_hit + ls.pdf);
                            current = throughput * ls.emission * 1.0f / ls.pdf * N_dot_L * L_dot_LN / L_dist / L_dist * eval * mis;
                        }
                    }
                    origin = g.P;
                    cn.add(C_VERTEX);
                    if (rnd(seed) > rr) {
                        done = true;
                    } else {
                        dir = bsdf_sample(pbr, N, in_dir, seed);
                        const float pdf = bsdf_pdf(pbr, N, in_dir, dir);
                        if (pdf > 0.0f) {
                            throughput *= bsdf_eval(pbr, N, in_dir, dir) * fabsf(dot(dir, N)) / pdf / rr;
                            prd_pdf = pdf * rr;
                        } else {
                            done = true;
                        }
                    }
                }
            }
            if (sum3(current) > 0.0f) {  // the shadow ray is shot by raygen (raygen.cu:134-143)
                const f3 bias = visB - visA;
                const float len = sqrtf(dot(bias, bias));
                HitRec sh;
                cn.add(C_SHADOW);
                if (!traverse<true, COUNT>(S, st, visA, bias / len, kEps, len - kEps, sh, cn)) result += current;
            }
            if (done || depth > 30) break;
            depth += 1;
        }
        film_write(p, x, y, result);
    }
    cn.flush(p.counters);
}

// ---- host-callable launchers ---------------------------------------------------------------------
static inline int render_blocks(const KParams& p) {
    const int tiles_x = ((int)p.width + 7) / 8;
    const int band_begin = p.row_begin / 8;
    const int band_end = (std::min(p.row_end, (int)p.height) + 7) / 8;
    const int step = p.row_step < 1 ? 1 : p.row_step;
    const int nb = band_end > band_begin ? (band_end - band_begin + step - 1) / step : 0;
    const int waves = tiles_x * nb;
    return (waves + (BLOCK / 64) - 1) / (BLOCK / 64);   // (k_pt, the film merges: one wave per tile, 256-thread blocks)
}
// threads of the widest grid a render launch of these bands may run ("pt": one wave per tile; "SPCBPT_eye": the same waves in the eye
// kernel's blocks) -- what the traversal stack's HBM area is sized for
int render_thread_count(const KParams& p) {
    const int waves = render_blocks(p) * (BLOCK / 64);
    return (waves + (EYE_BLOCK / 64) - 1) / (EYE_BLOCK / 64) * EYE_BLOCK;
}
int spcbpt_block_threads() { return EYE_BLOCK; }

// variant: 0 = timed (label caching, no counters), 1 = reference order with counters (also the generic form), 2 = the timed
// kernel's own events, counted
void launch_spcbpt(const KParams& p, int variant, int max_blocks, hipStream_t s) {
    // persistent grid: at most `max_blocks` (resident) blocks, never more than the tile queue can feed
    const int tiles = (int)p.n_tiles;
    if (tiles <= 0) return;
    int blocks = (tiles + (EYE_BLOCK / 64) - 1) / (EYE_BLOCK / 64);
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    if (variant == 1) {   // the reference's own evaluation of Gamma / Q (three reads, one division), counted as such
        KParams q = p;
        q.gamma_q = nullptr;
        hipLaunchKernelGGL((k_spcbpt<true, false, false>), dim3(blocks), dim3(EYE_BLOCK), 0, s, q);
    }
    else if (variant == 2) hipLaunchKernelGGL((k_spcbpt<true, false, true>), dim3(blocks), dim3(EYE_BLOCK), 0, s, p);
    else if (p.scene.general) hipLaunchKernelGGL((k_spcbpt<false, false, true, true>), dim3(blocks), dim3(EYE_BLOCK), 0, s, p);
    else hipLaunchKernelGGL((k_spcbpt<false, false, true, false>), dim3(blocks), dim3(EYE_BLOCK), 0, s, p);
}
// p.frames / p.n_frames describe the batch; p.n_tiles is the tile count of ONE frame
int spcbpt_batch_blocks(const KParams& p, int max_blocks) {
    const long long tiles = (long long)p.n_tiles * p.n_frames;
    if (tiles <= 0) return 0;
    long long blocks = (tiles + (EYE_BLOCK / 64) - 1) / (EYE_BLOCK / 64);
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    return (int)blocks;
}
void launch_spcbpt_batch(const KParams& p, int max_blocks, hipStream_t s) {
    const int blocks = spcbpt_batch_blocks(p, max_blocks);
    if (blocks <= 0) return;
    if (p.scene.general) hipLaunchKernelGGL((k_spcbpt<false, true, true, true>), dim3((unsigned)blocks), dim3(EYE_BLOCK), 0, s, p);
    else hipLaunchKernelGGL((k_spcbpt<false, true, true, false>), dim3((unsigned)blocks), dim3(EYE_BLOCK), 0, s, p);
}
// resident blocks per CU of the instantiation launch_spcbpt / launch_spcbpt_batch will really launch for (variant, batch, general):
// the forms differ in registers and scratch (the ENV = false form exists because of that), so each is asked for itself
int spcbpt_blocks_per_cu(int variant, bool batch, bool general) {
    int n = 0;
    hipError_t e;
    if (variant == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spcbpt<true, false, false>, EYE_BLOCK, 0);
    else if (variant == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spcbpt<true, false, true>, EYE_BLOCK, 0);
    else if (batch) e = general ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spcbpt<false, true, true, true>, EYE_BLOCK, 0)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spcbpt<false, true, true, false>, EYE_BLOCK, 0);
    else e = general ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spcbpt<false, false, true, true>, EYE_BLOCK, 0)
                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_spcbpt<false, false, true, false>, EYE_BLOCK, 0);
    return e == hipSuccess && n > 0 ? n : 1;
}
int render_tile_count(const KParams& p) {
    const int tiles_x = ((int)p.width + 7) / 8;
    const int band_begin = p.row_begin / 8;
    const int band_end = (std::min(p.row_end, (int)p.height) + 7) / 8;
    const int step = p.row_step < 1 ? 1 : p.row_step;
    const int nb = band_end > band_begin ? (band_end - band_begin + step - 1) / step : 0;
    return tiles_x * nb;
}
// accumulate + tone-map (raygen.cu:421-442) of one subframe from the `result` buffer a render kernel filled, for the pixels of
// the selected bands.  Separate from the render kernels so that consecutive frames' render kernels may overlap.
__global__ __launch_bounds__(BLOCK) void k_film_merge(const KParams p, const float* __restrict__ result) {
    uint32_t x, y;
    if (!lane_pixel(p, x, y)) return;
    const float4 r = reinterpret_cast<const float4*>(result)[(size_t)y * p.width + x];
    film_write(p, x, y, mk3(r.x, r.y, r.z));  // p.result is null here: the direct path
}
// ... of the frames of a batched launch in one pass: per pixel the running mean takes the frames in order (the same operations as
// `frames` launches of k_film_merge), the tone map is that of the last one
__global__ __launch_bounds__(BLOCK) void k_film_merge_batch(const KParams p, const MergeBatch m, int frames) {
    uint32_t x, y;
    if (!lane_pixel(p, x, y)) return;
    const size_t idx = (size_t)y * p.width + x;
    float4* acc = reinterpret_cast<float4*>(p.accum);
    f3 c = mk3(0.0f);
    bool have = false;
    for (int k = 0; k < frames; k++) {
        const float4 r = reinterpret_cast<const float4*>(m.result[k])[idx];
        f3 v = mk3(r.x, r.y, r.z);
        if (m.subframe[k] > 0) {
            if (!have) { const float4 prev = acc[idx]; c = mk3(prev.x, prev.y, prev.z); }
            const float a = 1.0f / (float)(m.subframe[k] + 1);
            v = lerp3(c, v, a);
        }
        c = v; have = true;
    }
    acc[idx] = make_float4(c.x, c.y, c.z, 1.0f);
    if (p.frame) {
        const float lum = 0.3f * c.x + 0.6f * c.y + 0.1f * c.z;
        const float s = 1.0f / (1.0f + lum / 1.5f);
        const f3 t = c * s;
        p.frame[idx] = quant8(to_srgb(clampf(t.x, 0.f, 1.f))) | (quant8(to_srgb(clampf(t.y, 0.f, 1.f))) << 8) |
                       (quant8(to_srgb(clampf(t.z, 0.f, 1.f))) << 16) | (255u << 24);
    }
}
void launch_film_merge_batch(const KParams& p, const MergeBatch& m, int frames, hipStream_t s) {
    const int blocks = render_blocks(p);
    if (blocks <= 0 || frames <= 0) return;
    KParams q = p;
    q.result = nullptr;
    hipLaunchKernelGGL(k_film_merge_batch, dim3(blocks), dim3(BLOCK), 0, s, q, m, frames);
}
void launch_film_merge(const KParams& p, hipStream_t s) {
    const int blocks = render_blocks(p);
    if (blocks <= 0 || !p.result) return;
    KParams q = p;
    q.result = nullptr;
    hipLaunchKernelGGL(k_film_merge, dim3(blocks), dim3(BLOCK), 0, s, q, p.result);
}
void launch_pt(const KParams& p, bool count, hipStream_t s) {
    const int blocks = render_blocks(p);
    if (blocks <= 0) return;
    if (count) hipLaunchKernelGGL(k_pt<true>, dim3(blocks), dim3(BLOCK), 0, s, p);
    else hipLaunchKernelGGL(k_pt<false>, dim3(blocks), dim3(BLOCK), 0, s, p);
}

}  // namespace spc
