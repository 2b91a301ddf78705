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
#include "kernels.h"

namespace spc {

static constexpr int BLOCK = 256;
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
#ifndef SPC_PRIO_LIGHT
#define SPC_PRIO_LIGHT 0   // ... of the light pass's waves (they share CUs with the eye megakernel when passes run ahead)
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
static constexpr int STACK_LDS = kStackLds;  // LDS stack entries per lane; deeper entries spill (TravStack)
static_assert(STACK_LDS >= 16, "the pooled connections publish 16 dwords per eye vertex through the traversal-stack LDS");
#ifndef SPC_WAVES
// minimum waves per SIMD requested from the register allocator for the other kernels of this file.  4 like the eye kernel, and for
// its sake: with three 128-VGPR eye blocks resident on a CU, 128 registers per lane are what is left -- a light-pass block that
// wants 154 would only fit on CUs holding two eye blocks or fewer and starve next to a persistent eye kernel (measured: the step
// got SLOWER with the faster eye kernel until the light pass was compiled to fit)
#define SPC_WAVES 4
#endif
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

// ------------------------------------------------------------------------------------------------
// Light pass.  A core walks m_per_core light paths and fills its own padded slot range, exactly the launch geometry of
// LightTraceParams; the MI355X default is num_core = M, m_per_core = 1 (one path per core).  Persistent waves with per-lane
// regeneration, like the eye pass: a lane whose core is finished takes the next core of a global queue at once (one atomic
// per wave and refill), and every iteration of the wave advances all live paths by one segment.  Light paths end after 2.6
// vertices on average but may run for 50, so one core per lane for the whole launch kept a wave resident for its longest
// path with 1-2 live lanes; regenerating waves do the same work with a quarter of the resident blocks, which matters because
// the pass shares the GPU with persistent eye kernels that never yield a block slot.  What a core computes and where it
// stores it does not depend on the lane that runs it: seeds come from the global core index, slots from the core's range.
template <bool COUNT, bool CACHE>
__global__ __launch_bounds__(BLOCK, SPC_WAVES) void k_light_trace(const KParams p) {
    __shared__ uint32_t s_stack[BLOCK * STACK_LDS];
    if (SPC_PRIO_LIGHT) __builtin_amdgcn_s_setprio(SPC_PRIO_LIGHT);
    const DeviceScene& S = p.scene;
    const uint32_t lane = threadIdx.x & 63;
    Counts<COUNT> cn;
    cn.clear();
    TravStack<BLOCK, STACK_LDS> st;
    st.init(s_stack, p.spill, p.spill_entries, (size_t)blockIdx.x * BLOCK + threadIdx.x, p.diag);
    int paths_started = 0;
    // per-core state
    bool has_core = false, exhausted = false;
    int local_core = 0, nverts = 0, npaths = 0, origins = 0;
    uint32_t seed = 0, pseed = 0;
    LightVertex* slots = nullptr;
    const bool batched = p.n_lframes > 0;   // several frames' passes in one queue (each lane then counts its cores' paths with one atomic per core)
    const uint32_t queue_len = batched ? (uint32_t)p.core_count * (uint32_t)p.n_lframes : (uint32_t)p.core_count;
    int fk = 0;   // frame of the lane's core within a batched pass
    // per-path state
    bool in_path = false;
    f3 origin = mk3(0.0f), dir = mk3(0.0f), next_flux = mk3(0.0f);
    float next_single_pdf = 0.0f;
    int depth = 0;
    uint32_t path_id = 0;
    LightVertex last;
    uint32_t pool_base = 0;
    int pool_left = 0;
    auto store = [&](const LightVertex& v) {
        float4* dst = reinterpret_cast<float4*>(slots + nverts);
        const float4* src = reinterpret_cast<const float4*>(&v);
#pragma unroll
        for (int q = 0; q < 6; q++) dst[q] = src[q];
        nverts++;
        cn.add(C_LVCW);
    };
    auto core_done = [&]() {   // the lane's core is complete: its vertex count, and the paths it started
        if (batched) {
            p.core_counts[(size_t)fk * (p.core_count + 1) + local_core] = nverts;
            atomicAdd(p.path_counter + fk, origins);
        } else { p.core_counts[local_core] = nverts; paths_started += origins; }
    };
    while (true) {
        // ---- regeneration: cores of the queue to lanes without one
        unsigned long long idle = __ballot(!has_core);
        while (idle != 0ull && !exhausted) {
            if (pool_left == 0) {
                uint32_t t = 0;
                if (lane == (uint32_t)__ffsll((long long)idle) - 1u) t = atomicAdd(p.work_counter, 64u);
                t = __shfl(t, __ffsll((long long)idle) - 1, 64);
                if (t >= queue_len) { exhausted = true; break; }
                pool_base = t;
                pool_left = min(64, (int)(queue_len - t));
            }
            const int n_idle = __popcll(idle);
            const int take = n_idle < pool_left ? n_idle : pool_left;
            const int my_rank = __popcll(idle & ((1ull << lane) - 1ull));
            if (!has_core && my_rank < take) {
                local_core = (int)pool_base + my_rank;
                has_core = true;
                uint32_t launch_frame = p.launch_frame;
                LightVertex* scratch = p.lvc_scratch;
                if (batched) {   // the queue spans the passes of n_lframes frames (layout.h: n_lframes)
                    fk = local_core / p.core_count;
                    local_core -= fk * p.core_count;
                    launch_frame += (uint32_t)fk;
                    scratch += (size_t)fk * p.core_count * p.core_padding;
                }
                const int core = p.core_begin + local_core;
                seed = tea4((uint32_t)core, launch_frame);  // light sampling stream
                // payload.seed: BSDF stream; the reference starts it equal to `seed`
                pseed = p.lt_decorrelate ? tea4((uint32_t)core ^ 0x80000000u, launch_frame) : seed;
                slots = scratch + (size_t)local_core * p.core_padding;
                nverts = 0; npaths = 0; origins = 0;
                in_path = false;
            }
            pool_base += (uint32_t)take;
            pool_left -= take;
            idle = __ballot(!has_core);
        }
        if (!__any(has_core)) break;   // queue exhausted and every core of the wave finished
        // ---- a core without a running path starts its next one: light sample + origin vertex (raygen.cu:620-668)
        if (has_core && !in_path) {
            const int lid = pick_light(S, seed);
            const DLight& L = S.lights[lid];
            LightSampleD ls;
            float dir_pdf;
            uint32_t origin_flags = 0u;
            if (L.type == 1) {   // the environment map: a sky direction, the sub-path starts on the sky disk and runs against it
                ls = env_light_sample(S, seed, dir_pdf);
                dir = ls.normal;
                origin_flags = SPCBPT_LV_DIRECTION;
            } else {
                const float r1 = rnd(seed), r2 = rnd(seed);
                ls = light_reverse_sample(S, L, r1, r2);
                const float d1 = rnd(seed), d2 = rnd(seed);  // traceMode
                const Onb onb(ls.normal);
                dir = onb.to_world(cosine_sample_hemisphere(d1, d2));
                dir_pdf = fabsf(dot(dir, ls.normal)) * kInvPi;
            }
            origin = ls.position;
            path_id = (uint32_t)(p.core_begin + local_core) * (uint32_t)p.m_per_core + (uint32_t)npaths;
            cn.add(C_LIGHT);
            // origin vertex (init_vertex_from_lightSample raygen.cu:172-195)
            LightVertex v;
            v.position[0] = ls.position.x; v.position[1] = ls.position.y; v.position[2] = ls.position.z; v.pdf = ls.pdf;
            v.normal[0] = ls.normal.x; v.normal[1] = ls.normal.y; v.normal[2] = ls.normal.z; v.single_pdf = ls.pdf;
            v.flux[0] = ls.emission.x; v.flux[1] = ls.emission.y; v.flux[2] = ls.emission.z; v.rmis_pointer = 1.0f;
            v.color[0] = v.color[1] = v.color[2] = 0.0f; v.last_lum = 0.0f;
            v.last_position[0] = v.last_position[1] = v.last_position[2] = 0.0f; v.last_normal_projection = 0.0f;
            v.material_id = (int16_t)L.id; v.subspace_id = (int16_t)ls.subspace; v.depth = 0; v.last_zone_id = 0;
            v.path_id = path_id; v.pad = origin_flags;
            store(v);
            origins++;
            last = v;
            next_flux = mk3(0.0f);
            next_single_pdf = dir_pdf;
            depth = 0;
            in_path = nverts < p.core_padding;   // a full slot range ends the core right after the origin vertex
            if (!in_path) { core_done(); has_core = false; }
        }
        // ---- one segment of every running path (hit_program.cu:341-438)
        const bool tracing = has_core && in_path;
        HitRec h;
        h.tri = -1;
        if (tracing) {
            cn.add(C_CLOSEST);
            traverse<false, COUNT>(S, st, origin, dir, kEps, 1e16f, h, cn);
        }
        if (tracing) {
            bool done = false, full = false;
            if (h.tri < 0) { done = true; }
            else {
                const Geom g = local_geometry(S, h);
                if (g.emitter) { done = true; }  // __closesthit__lightSource_subpath
                else {
                    Pbr pbr = load_pbr(S, g.mat);
                    color_tex_sample(S, g, pbr, cn);
                    f3 N = g.N;
                    if (dot(N, dir) > 0.f) N = -N;
                    const f3 inv_dir = -dir;
                    const f3 new_dir = bsdf_sample(pbr, N, inv_dir, pseed);
                    const float pdf = bsdf_pdf(pbr, N, inv_dir, new_dir);
                    if (!(pdf > 0.0f)) done = true;
                    const f3 last_n = ld3(last.normal), last_flux = ld3(last.flux);
                    const bool last_dir = (last.pad & SPCBPT_LV_DIRECTION) != 0u;   // LastVertex.is_DIRECTION(): parallel rays from the sky, no 1 / t^2 (hit_program.cu:372-375)
                    const float pdf_G = last_dir ? fabsf(dot(N, dir) * dot(last_n, dir)) : fabsf(dot(N, dir) * dot(last_n, dir)) / (h.t * h.t);
                    const f3 flux = last.depth == 0 ? last_flux * pdf_G : next_flux * last_flux * pdf_G;
                    LightVertex m;
                    m.position[0] = g.P.x; m.position[1] = g.P.y; m.position[2] = g.P.z;
                    m.normal[0] = N.x; m.normal[1] = N.y; m.normal[2] = N.z;
                    m.flux[0] = flux.x; m.flux[1] = flux.y; m.flux[2] = flux.z;
                    m.color[0] = pbr.base.x; m.color[1] = pbr.base.y; m.color[2] = pbr.base.z;
                    m.last_position[0] = last.position[0]; m.last_position[1] = last.position[1]; m.last_position[2] = last.position[2];
                    if (last_dir) { const f3 lp = g.P - dir; m.last_position[0] = lp.x; m.last_position[1] = lp.y; m.last_position[2] = lp.z; }   // hit_program.cu:386-389
                    m.last_normal_projection = fabsf(dot(last_n, dir));
                    m.material_id = (int16_t)g.mat;
                    // light-tree label of the new vertex and eye-tree relabel of the previous one (tracing_weight_light) in lock-step
                    int new_label, eye_label;
                    const f3 last_pos = ld3(last.position);
                    uint32_t own_eye_label = 0u;   // device_lib.h: label caching -- the new vertex's own eye-tree label + 1
                    if (CACHE) {
                        int own;
                        tree_label2<COUNT, true>(p.light_tree, g.P, N, inv_dir, true, p.eye_tree, g.P, N, inv_dir, true, new_label, own, cn);
                        own_eye_label = (uint32_t)own + 1u;
                        eye_label = (int)(last.pad & 0xffffu) - 1;   // the previous vertex's, cached when it was created (unused when it is the origin)
                    } else {
                        tree_label2(p.light_tree, g.P, N, inv_dir, true, p.eye_tree, last_pos, last_n, normalize(g.P - last_pos), last.depth != 0,
                                    new_label, eye_label, cn);
                    }
                    m.subspace_id = (int16_t)new_label;
                    m.last_zone_id = last.subspace_id;
                    m.depth = (int16_t)(last.depth + 1);
                    m.single_pdf = next_single_pdf * pdf_G / fabsf(dot(last_n, dir));
                    m.pdf = last.pdf * m.single_pdf;
                    m.last_lum = sum3(last_flux / last.pdf);
                    m.path_id = path_id; m.pad = own_eye_label | (last_dir ? SPCBPT_LV_LAST_DIRECTION : 0u);   // isLastVertex_direction (hit_program.cu:412: the predecessor is the origin)
                    if (last.depth == 0) {
                        m.rmis_pointer = last.rmis_pointer / last.single_pdf;  // tracing_init_light
                    } else {  // tracing_update_light (rmis.h:80-94)
                        const VCore lc = core_of(last);
                        const Pbr mat_last = load_pbr_colored(S, lc.mat, lc.color);
                        const f3 in_dir = normalize(g.P - lc.pos);
                        const float LL_pdf = rmis_last_pdf(mat_last, lc, in_dir);
                        const float wgt = rmis_weight_light_l(p, last.last_zone_id, last.last_lum, eye_label, cn);
                        m.rmis_pointer = (last.rmis_pointer * LL_pdf + wgt) / last.single_pdf;
                    }
                    cn.add(C_VERTEX);
                    next_flux = brdf_div(pbr, bsdf_eval(pbr, N, inv_dir, new_dir), N, new_dir);   // hit_program.cu:384
                    next_single_pdf = pdf;
                    origin = g.P;
                    dir = new_dir;
                    const float r = rnd(pseed);
                    const float rr = rr_of(pbr.base);
                    if (r > rr) done = true;
                    else next_single_pdf *= rr;
                    store(m);
                    last = m;
                    if (!(nverts < p.core_padding)) full = true;
                }
            }
            // the walk loop's exit tests (raygen.cu:646-676): slot range full -> the core ends; path done or too deep -> next path
            bool path_over = full;
            if (!full) {
                if (done || depth > 50) path_over = true;
                else depth += 1;
            }
            if (path_over) {
                in_path = false;
                bool core_over = full;
                if (!full) {
                    npaths++;
                    if (npaths >= p.m_per_core || !(nverts < p.core_padding)) core_over = true;
                }
                if (core_over) { core_done(); has_core = false; }
            }
        }
    }
    // path_count of the sampler (#depth-0 vertices, device_thrust.cu:324-326): one atomic per wave
    for (int o = 32; o > 0; o >>= 1) paths_started += __shfl_down(paths_started, o, 64);
    if (lane == 0 && paths_started) atomicAdd(p.path_counter, paths_started);
    cn.flush(p.counters);
}

// ------------------------------------------------------------------------------------------------
// Sampler build on device (LVC_Process).  Input: padded scratch + per-core counts.  Steps:
//   1. exclusive scan of core_counts (hipcub)                       -> core_offsets, vertex_count
//   2. k_lvc_compact: copy to the compact LVC in (core, slot) order, emit key = subspace id, weight, path starts
//   3. stable radix sort of (subspace id -> compact index) (hipcub)  -> jump_buffer
//   4. k_subspace_ranges: first/last position of each subspace in the sorted keys -> jump_bias, size
//   5. inclusive scan (double) of the weights in sorted order (hipcub), k_cmf: per-subspace normalised CMF
__global__ void k_lvc_compact(const LightVertex* __restrict__ scratch, const int* __restrict__ core_counts,
                              const int* __restrict__ core_offsets, int core_count, int core_padding, LightVertex* __restrict__ lvc,
                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, float* __restrict__ weights,
                              int* __restrict__ sampler_counts, int capacity, uint32_t* __restrict__ overflow) {
    // one thread per padded slot; only the filled slots (slot < count of its core) copy their 96-B record.  The compact cache holds
    // `capacity` vertices (sized from a measured pass with slack, not from the padded worst case: context.h); a pass that outgrows
    // it is cut off at the capacity and reported through *overflow (SPCBPT_ERR_CAPACITY at the next sync), never written past the end.
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0 && core_offsets[core_count] > capacity) { sampler_counts[0] = capacity; *overflow = 1u; }
    if (t < (long long)core_count * core_padding) {
        const int core = (int)(t / core_padding), slot = (int)(t % core_padding);
        if (slot < core_counts[core] && core_offsets[core] + slot < capacity) {
            const int dst_i = core_offsets[core] + slot;
            const float4* src = reinterpret_cast<const float4*>(scratch + t);
            float4* dst = reinterpret_cast<float4*>(lvc + dst_i);
            float4 q[6];
#pragma unroll
            for (int k = 0; k < 6; k++) q[k] = src[k];
#pragma unroll
            for (int k = 0; k < 6; k++) dst[k] = q[k];
            const LightVertex& v = *reinterpret_cast<const LightVertex*>(q);
            float w = (v.flux[0] + v.flux[1] + v.flux[2]) / v.pdf;  // LVCSubspaceInfoCopy device_thrust.cu:191-212
            if (isinf(w) || isnan(w)) w = 0.0f;
            keys[dst_i] = (uint32_t)v.subspace_id;
            vals[dst_i] = (uint32_t)dst_i;
            weights[dst_i] = w;
        }
    }
}

// Compaction of a batched light pass: grid.y = frame of the batch.  core_offsets is ONE exclusive scan over the n * (core_count + 1)
// counts (each frame's segment ends in a zero sentinel), so frame k's offsets are relative to its first entry and its total is the
// sentinel's offset minus that.  Keys are left to the sampler build (k_fill_keys_from_lvc), which also counts the paths again.
__global__ void k_lvc_compact_batch(const LightVertex* __restrict__ scratch, const int* __restrict__ core_counts, const int* __restrict__ core_offsets,
                                    const int* __restrict__ path_counts, int core_count, int core_padding, CompactBatch dst, int capacity,
                                    uint32_t* __restrict__ overflow) {
    const int k = blockIdx.y;
    const int* counts = core_counts + (size_t)k * (core_count + 1);
    const int* offs = core_offsets + (size_t)k * (core_count + 1);
    const int base = offs[0];
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t == 0) {
        const int total = offs[core_count] - base;
        if (total > capacity) *overflow = 1u;   // k_lvc_compact: cut off at the set's capacity and reported
        dst.counts[k][0] = min(total, capacity); dst.counts[k][1] = path_counts[k];
    }
    if (t < (long long)core_count * core_padding) {
        const int core = (int)(t / core_padding), slot = (int)(t % core_padding);
        if (slot < counts[core] && offs[core] - base + slot < capacity) {
            const float4* src = reinterpret_cast<const float4*>(scratch + (size_t)k * core_count * core_padding + t);
            float4* out = reinterpret_cast<float4*>(dst.lvc[k] + (offs[core] - base + slot));
            float4 q[6];
#pragma unroll
            for (int j = 0; j < 6; j++) q[j] = src[j];
#pragma unroll
            for (int j = 0; j < 6; j++) out[j] = q[j];
        }
    }
}

__global__ void k_fill_keys_from_lvc(const LightVertex* __restrict__ lvc, int n, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                     float* __restrict__ weights, int* __restrict__ sampler_counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int start = 0;
    if (i < n) {
        const LightVertex& v = lvc[i];
        float w = (v.flux[0] + v.flux[1] + v.flux[2]) / v.pdf;
        if (isinf(w) || isnan(w)) w = 0.0f;
        keys[i] = (uint32_t)v.subspace_id;
        vals[i] = (uint32_t)i;
        weights[i] = w;
        start = v.depth == 0 ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) start += __shfl_down(start, o, 64);
    if ((threadIdx.x & 63) == 0 && start) atomicAdd(&sampler_counts[1], start);
}

// The same with the item count on the DEVICE (sampler_counts[0]) and a host-known upper bound `bound` as the grid: slots beyond the
// count get the pad key 1023 (no subspace id reaches it: ids are < 1000) and weight 0, so a 10-bit radix sort over `bound` items
// leaves the real items sorted in front.  sampler_counts[1] (path count) is left as the caller set it.
__global__ void k_fill_keys_devcount(const LightVertex* __restrict__ lvc, int bound, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                     float* __restrict__ weights, const int* __restrict__ sampler_counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bound) return;
    const int n = sampler_counts[0];
    if (i < n) {
        const LightVertex& v = lvc[i];
        float w = (v.flux[0] + v.flux[1] + v.flux[2]) / v.pdf;
        if (isinf(w) || isnan(w)) w = 0.0f;
        keys[i] = (uint32_t)v.subspace_id;
        weights[i] = w;
    } else {
        keys[i] = 1023u;
        weights[i] = 0.0f;
    }
    vals[i] = (uint32_t)i;
}

// Exchange 1 of a sharded job, receiving side: `gathered` holds `world` shards of `cap` slots each (the all-gather of every rank's
// compact shard, padded to the agreed capacity), counts_all[2 r] / [2 r + 1] the vertex / path count of rank r.  The shards are
// concatenated in rank order = global (path, depth) order into the set's LVC; the totals go to sampler_counts (device-resident:
// the sampler build sizes itself from them, no host round trip).  A shard that did not fit `cap` raises *overflow.
// Batched form (one exchange per light batch): grid.y = frame k of `nf`; rank q's block of the all-gather holds its nf shards one
// after the other, so frame k of rank q sits at (q nf + k) cap and its counts at 2 (q nf + k); every frame goes to its own set (dst).
__global__ void k_gather_compact(const LightVertex* __restrict__ gathered, const int* __restrict__ counts_all, int world, int cap, int lvc_capacity,
                                 CompactBatch dst, int nf, int* __restrict__ overflow) {
    const int chunks = (cap + 255) / 256;
    const int r = blockIdx.x / chunks, c = blockIdx.x % chunks, k = blockIdx.y;
    LightVertex* __restrict__ lvc = dst.lvc[k];
    int* __restrict__ sampler_counts = dst.counts[k];
    int base = 0, total = 0, paths = 0;
    bool over = false;
    for (int q = 0; q < world; q++) {
        const int n = counts_all[2 * (q * nf + k)];
        if (n > cap) over = true;
        if (q < r) base += min(n, cap);
        total += min(n, cap);
        paths += counts_all[2 * (q * nf + k) + 1];
    }
    if (total > lvc_capacity) over = true;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sampler_counts[0] = min(total, lvc_capacity);
        sampler_counts[1] = paths;
        if (over) *overflow = 1;
    }
    const int n_r = min(counts_all[2 * (r * nf + k)], cap);
    const int i = c * 256 + (int)threadIdx.x;
    if (i >= n_r || base + i >= lvc_capacity) return;
    const float4* src = reinterpret_cast<const float4*>(gathered + ((size_t)r * nf + k) * cap + i);
    float4* dst_q = reinterpret_cast<float4*>(lvc + base + i);
    float4 q[6];
#pragma unroll
    for (int j = 0; j < 6; j++) q[j] = src[j];
#pragma unroll
    for (int j = 0; j < 6; j++) dst_q[j] = q[j];
}
// Sending side of the batched exchange: the first min(count, cap) vertices of nf sets and their count pairs into one contiguous
// send buffer of nf x cap vertices (grid.y = frame).  The padding behind a shard is not copied (nobody reads it).
__global__ void k_pack_shards(CompactBatch src, int cap, LightVertex* __restrict__ send, int* __restrict__ send_counts) {
    const int k = blockIdx.y;
    const int n = src.counts[k][0];
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    if (i == 0) { send_counts[2 * k] = n; send_counts[2 * k + 1] = src.counts[k][1]; }
    if (i >= min(n, cap)) return;
    const float4* in = reinterpret_cast<const float4*>(src.lvc[k] + i);
    float4* out = reinterpret_cast<float4*>(send + (size_t)k * cap + i);
    float4 q[6];
#pragma unroll
    for (int j = 0; j < 6; j++) q[j] = in[j];
#pragma unroll
    for (int j = 0; j < 6; j++) out[j] = q[j];
}

// film exchange of a sharded job: the 8-row bands of rank `rank` (band b with b % world == rank) packed contiguously / unpacked
__global__ void k_pack_bands(const float4* __restrict__ accum, int width, int height, int rank, int world, float4* __restrict__ packed, int unpack_all) {
    // unpack_all == 0: accum -> packed (own bands, band-major); != 0: packed (world x bands_per_rank x 8 x width) -> accum (every band)
    const int bands = (height + 7) / 8, per_rank = (bands + world - 1) / world;
    const size_t band_px = (size_t)8 * width;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!unpack_all) {
        if (t >= (size_t)per_rank * band_px) return;
        const int k = (int)(t / band_px), b = rank + k * world;
        const size_t in_band = t % band_px;
        const int y = b * 8 + (int)(in_band / width), x = (int)(in_band % width);
        packed[t] = (b < bands && y < height) ? accum[(size_t)y * width + x] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        if (t >= (size_t)world * per_rank * band_px) return;
        const int r = (int)(t / ((size_t)per_rank * band_px));
        const size_t tr = t % ((size_t)per_rank * band_px);
        const int k = (int)(tr / band_px), b = r + k * world;
        const size_t in_band = tr % band_px;
        const int y = b * 8 + (int)(in_band / width), x = (int)(in_band % width);
        if (b < bands && y < height) reinterpret_cast<float4*>(const_cast<float4*>(accum))[(size_t)y * width + x] = packed[t];
    }
}

__global__ void k_subspace_ranges(const uint32_t* __restrict__ sorted_keys, const int* __restrict__ sampler_counts, DSubspace* __restrict__ sub) {
    const int n = sampler_counts[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = sorted_keys[i];
    if (i == 0 || sorted_keys[i - 1] != k) sub[k].jump_bias = i;
    if (i == n - 1 || sorted_keys[i + 1] != k) sub[k].size = i + 1;  // temporarily the END position; fixed in k_finish_ranges
}
__global__ void k_finish_ranges(DSubspace* __restrict__ sub) {
    // one block of 1024 threads: empty subspaces get jump_bias = end of the last non-empty one before them, like the
    // running offset of the reference's host loop (device_thrust.cu:301-309) -> inclusive max-scan of the END positions
    __shared__ int ends[1024];
    const int s = threadIdx.x;
    const int end = s < SPCBPT_NUM_SUBSPACE ? sub[s].size : 0;  // END position written by k_subspace_ranges, 0 if empty
    ends[s] = end;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = s >= off ? ends[s - off] : 0;
        __syncthreads();
        ends[s] = max(ends[s], v);
        __syncthreads();
    }
    if (s < SPCBPT_NUM_SUBSPACE) {
        if (end > 0) sub[s].size = end - sub[s].jump_bias;
        else { sub[s].jump_bias = s > 0 ? ends[s - 1] : 0; sub[s].size = 0; }
    }
}
__global__ void k_gather_weights(const float* __restrict__ weights, const uint32_t* __restrict__ sorted_vals, const int* __restrict__ sampler_counts,
                                 double* __restrict__ out) {
    const int n = sampler_counts[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)weights[sorted_vals[i]];
}
__global__ void k_cmf(const double* __restrict__ prefix, const uint32_t* __restrict__ sorted_keys, const int* __restrict__ sampler_counts,
                      DSubspace* __restrict__ sub, float* __restrict__ cmfs) {
    const int n = sampler_counts[0];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = sorted_keys[i];
    const int b = sub[k].jump_bias, e = b + sub[k].size;
    const double base = b > 0 ? prefix[b - 1] : 0.0;
    const double total = prefix[e - 1] - base;
    const bool last = i == e - 1;
    // sum_pmf == 0 gives NaN CMFs in the reference; guarded here: zero-weight subspaces sample uniformly
    float c = total > 0.0 ? (float)((prefix[i] - base) / total) : (float)(i - b + 1) / (float)(e - b);
    if (last) { c = 1.0f; sub[k].sum_pmf = (float)total; }
    cmfs[i] = c;
}

// ---- sampler build in four launches ------------------------------------------------------------------------------------------
// The build above is ~14 dependent launches (hipcub's radix sort and scan are five and two of them): 0.3 ms of launch latency
// however few vertices it sorts, and a batched eye launch waits for up to 16 of them.  Subspace ids are 10-bit keys, so one stable
// counting sort does: SB_BLOCKS single-wave blocks each own a contiguous chunk of the cache,
//   k_sb_hist     per-block histogram of the ids (LDS), keys + weights (+ the path count) on the way
//   k_sb_scan     one block: per-id running offsets over the blocks, exclusive scan over the ids -> jump_bias / size
//   k_sb_scatter  each block places its chunk in order (rank among equal ids inside a wave from ten ballots) -> jump buffer,
//                 weights in sorted order
//   k_sb_cmf      one block per subspace: double-precision scan of its weights -> CMF, sum_pmf
// Same tables as the sort: the order inside a subspace is the cache order (stable), empty subspaces carry the running offset.
// The CMF sums a subspace's weights by themselves (the scan above takes differences of a global prefix): equal to 1e-16 relative.
static constexpr int SB_BLOCKS = 512;
// (blockIdx.y = frame of a batched build: SamplerBuildBatch, kernels.h; a single build is a batch of one)
__global__ __launch_bounds__(64) void k_sb_hist(const SamplerBuildBatch B) {
    const int f = blockIdx.y;
    const LightVertex* __restrict__ lvc = B.lvc[f];
    const int n_host = B.n_host[f];
    const int* __restrict__ n_dev = B.n_dev[f];
    uint32_t* __restrict__ keys = B.keys + (size_t)f * B.item_stride;
    float* __restrict__ weights = B.weights + (size_t)f * B.item_stride;
    int* __restrict__ hist = B.hist + (size_t)f * (SB_BLOCKS + 1) * 1024;
    int* __restrict__ path_count = B.path_count[f];
    __shared__ uint32_t h[1024];
    const int lane = threadIdx.x, b = blockIdx.x;
#pragma unroll
    for (int t = 0; t < 16; t++) h[t * 64 + lane] = 0u;
    __syncthreads();
    const int n = n_dev ? n_dev[0] : n_host;
    const int chunk = (n + SB_BLOCKS - 1) / SB_BLOCKS, i0 = b * chunk, i1 = min(n, i0 + chunk);
    int starts = 0;
    for (int i = i0 + lane; i < i1; i += 64) {
        const LightVertex& v = lvc[i];
        float w = (v.flux[0] + v.flux[1] + v.flux[2]) / v.pdf;   // LVCSubspaceInfoCopy device_thrust.cu:191-212
        if (isinf(w) || isnan(w)) w = 0.0f;
        const uint32_t k = (uint32_t)v.subspace_id & 1023u;
        keys[i] = k;
        weights[i] = w;
        atomicAdd(&h[k], 1u);
        starts += v.depth == 0 ? 1 : 0;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 16; t++) hist[(size_t)b * 1024 + t * 64 + lane] = (int)h[t * 64 + lane];
    if (path_count) {
        for (int o = 32; o > 0; o >>= 1) starts += __shfl_down(starts, o, 64);
        if (lane == 0 && starts) atomicAdd(path_count, starts);
    }
}
__global__ __launch_bounds__(1024) void k_sb_scan(const SamplerBuildBatch B) {
    int* __restrict__ hist = B.hist + (size_t)blockIdx.y * (SB_BLOCKS + 1) * 1024;
    DSubspace* __restrict__ sub = B.sub[blockIdx.y];
    // thread = subspace id.  hist[b][id] becomes the number of items with that id in the blocks before b; row SB_BLOCKS receives the
    // position of the id's first item = items with smaller ids: for an empty subspace the end of the last non-empty one before it,
    // the running offset of the reference's host loop (device_thrust.cu:301-309)
    __shared__ int tot[1024];
    const int k = threadIdx.x;
    int run = 0;
#pragma unroll 8
    for (int b = 0; b < SB_BLOCKS; b++) {
        const int c = hist[(size_t)b * 1024 + k];
        hist[(size_t)b * 1024 + k] = run;
        run += c;
    }
    tot[k] = run;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = k >= off ? tot[k - off] : 0;
        __syncthreads();
        tot[k] += v;
        __syncthreads();
    }
    const int base = tot[k] - run;   // exclusive
    hist[(size_t)SB_BLOCKS * 1024 + k] = base;
    if (k < SPCBPT_NUM_SUBSPACE) { sub[k].jump_bias = base; sub[k].size = run; sub[k].sum_pmf = 0.0f; sub[k].pad = 0; }
}
__global__ __launch_bounds__(64) void k_sb_scatter(const SamplerBuildBatch B) {
    const int f = blockIdx.y;
    const uint32_t* __restrict__ keys = B.keys + (size_t)f * B.item_stride;
    const float* __restrict__ weights = B.weights + (size_t)f * B.item_stride;
    const int n_host = B.n_host[f];
    const int* __restrict__ n_dev = B.n_dev[f];
    const int* __restrict__ hist = B.hist + (size_t)f * (SB_BLOCKS + 1) * 1024;
    uint32_t* __restrict__ jump = B.jump[f];
    double* __restrict__ wsorted = B.wsorted + (size_t)f * B.item_stride;
    __shared__ uint32_t next[1024];   // where this block's next item of each id goes
    const int lane = threadIdx.x, b = blockIdx.x;
#pragma unroll
    for (int t = 0; t < 16; t++) next[t * 64 + lane] = (uint32_t)(hist[(size_t)SB_BLOCKS * 1024 + t * 64 + lane] + hist[(size_t)b * 1024 + t * 64 + lane]);
    __syncthreads();
    const int n = n_dev ? n_dev[0] : n_host;
    const int chunk = (n + SB_BLOCKS - 1) / SB_BLOCKS, i0 = b * chunk, i1 = min(n, i0 + chunk);
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = i0; base < i1; base += 64) {   // wave-uniform bounds: every lane takes part in the ballots
        const int i = base + lane;
        const bool valid = i < i1;
        const uint32_t k = valid ? keys[i] : 0u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 10; bit++) {
            const unsigned long long m = __ballot((k >> bit) & 1u);
            peers &= ((k >> bit) & 1u) ? m : ~m;
        }
        uint32_t pos = 0u;
        if (valid) pos = next[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt) == 0ull) next[k] = pos + (uint32_t)__popcll(peers);   // the first lane of each id moves the cursor on
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const uint32_t dst = pos + (uint32_t)__popcll(peers & lt);
            jump[dst] = (uint32_t)i;
            wsorted[dst] = (double)weights[i];
        }
    }
}
// the cache in the sampler's order: record i = lvc[jump[i]], one lane per QUAD (six consecutive lanes read one 96-B vertex and write
// its six quads next to each other: the stores of a wave are contiguous, the loads are whole records)
__global__ __launch_bounds__(256) void k_sb_copy(const SamplerBuildBatch B) {
    const int f = blockIdx.y;
    float4* __restrict__ dst = reinterpret_cast<float4*>(B.lvc_sorted[f]);
    if (!dst) return;
    const float4* __restrict__ src = reinterpret_cast<const float4*>(B.lvc[f]);
    const uint32_t* __restrict__ jump = B.jump[f];
    const int* __restrict__ n_dev = B.n_dev[f];
    const long long n = n_dev ? n_dev[0] : B.n_host[f];
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < n * 6; t += (long long)gridDim.x * 256) {
        const long long i = t / 6;
        const int q = (int)(t - i * 6);
        dst[t] = src[(size_t)jump[i] * 6 + q];
    }
}
// KParams::guide of one subspace: entry j = the first place k with cmf[k] > (j / n)(1 - 2^-20).  A random number u of bucket j --
// (int)(u * (float)n) == j, the product rounded to FP32 -- is at least (j / n)(1 - 2^-24), so no entry before that place is above u.
SPC_DEV void build_guide(const float* cmf, int n, uint32_t* guide, int t, int stride) {
    for (int j = t; j < n; j += stride) {
        const double tj = (double)j / (double)n * (1.0 - 1.0 / 1048576.0);
        int lo = 0, hi = n - 1;   // (the last entry is 1)
        while (lo < hi) {
            const int m = (lo + hi) >> 1;
            if ((double)cmf[m] > tj) hi = m; else lo = m + 1;
        }
        guide[j] = (uint32_t)lo;
    }
}
__global__ __launch_bounds__(256) void k_sb_cmf(const SamplerBuildBatch B) {
    DSubspace* __restrict__ sub = B.sub[blockIdx.y];
    const double* __restrict__ wsorted = B.wsorted + (size_t)blockIdx.y * B.item_stride;
    float* cmfs = B.cmfs[blockIdx.y];   // (read back for the guide table below: not __restrict__)
    __shared__ double sh[256];
    const int k = blockIdx.x, t = threadIdx.x;
    const int b = sub[k].jump_bias, sz = sub[k].size;
    if (sz <= 0) return;
    double acc = 0.0;
    for (int j = t; j < sz; j += 256) acc += wsorted[b + j];
    sh[t] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (t < off) sh[t] += sh[t + off]; __syncthreads(); }
    const double total = sh[0];
    __syncthreads();
    double carry = 0.0;
    for (int j0 = 0; j0 < sz; j0 += 256) {
        const int j = j0 + t;
        const double w = j < sz ? wsorted[b + j] : 0.0;
        sh[t] = w;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const double v = t >= off ? sh[t - off] : 0.0;
            __syncthreads();
            sh[t] += v;
            __syncthreads();
        }
        if (j < sz) {
            // sum_pmf == 0 gives NaN CMFs in the reference; guarded here: zero-weight subspaces sample uniformly
            float c = total > 0.0 ? (float)((carry + sh[t]) / total) : (float)(j + 1) / (float)sz;
            if (j == sz - 1) c = 1.0f;
            cmfs[b + j] = c;
        }
        carry += sh[255];
        __syncthreads();
    }
    if (t == 0) sub[k].sum_pmf = (float)total;
    if (B.guide[blockIdx.y]) build_guide(cmfs + b, sz, B.guide[blockIdx.y] + b, t, 256);   // (the loop above ends on a barrier: the block's CMF is written)
}
// the second-stage guide table next to a CMF that another path has written (the radix-sort build)
__global__ __launch_bounds__(256) void k_sb_guide(const DSubspace* __restrict__ sub, const float* cmfs, uint32_t* __restrict__ guide) {
    const int b = sub[blockIdx.x].jump_bias, sz = sub[blockIdx.x].size;
    if (sz > 0) build_guide(cmfs + b, sz, guide + b, threadIdx.x, 256);
}
void launch_sampler_guide(const DSubspace* sub, const float* cmfs, uint32_t* guide, hipStream_t s) {
    if (guide) hipLaunchKernelGGL(k_sb_guide, dim3(SPCBPT_NUM_SUBSPACE), dim3(256), 0, s, sub, cmfs, guide);
}
size_t sampler_build_hist_ints() { return (size_t)(SB_BLOCKS + 1) * 1024; }
void launch_sampler_build_batch(const SamplerBuildBatch& b, int frames, hipStream_t s) {
    if (frames <= 0) return;
    hipLaunchKernelGGL(k_sb_hist, dim3(SB_BLOCKS, frames), dim3(64), 0, s, b);
    hipLaunchKernelGGL(k_sb_scan, dim3(1, frames), dim3(1024), 0, s, b);
    hipLaunchKernelGGL(k_sb_scatter, dim3(SB_BLOCKS, frames), dim3(64), 0, s, b);
    hipLaunchKernelGGL(k_sb_cmf, dim3(SPCBPT_NUM_SUBSPACE, frames), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_sb_copy, dim3(256, frames), dim3(256), 0, s, b);
}
__global__ __launch_bounds__(256) void k_lvc_sorted_copy(const LightVertex* __restrict__ lvc, const uint32_t* __restrict__ jump, const int* __restrict__ counts,
                                                        LightVertex* __restrict__ out, int capacity) {
    const int n = min(counts[0], capacity);
    const float4* src = reinterpret_cast<const float4*>(lvc);
    float4* dst = reinterpret_cast<float4*>(out);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const size_t from = jump[i];
#pragma unroll
        for (int q = 0; q < 6; q++) dst[(size_t)i * 6 + q] = src[from * 6 + q];
    }
}
void launch_lvc_sorted_copy(const LightVertex* lvc, const uint32_t* jump, const int* sampler_counts, LightVertex* lvc_sorted, int capacity, hipStream_t s) {
    if (capacity <= 0 || !lvc_sorted) return;
    hipLaunchKernelGGL(k_lvc_sorted_copy, dim3(512), dim3(256), 0, s, lvc, jump, sampler_counts, lvc_sorted, capacity);
}
void launch_sampler_build(const LightVertex* lvc, int n_host, const int* n_dev, uint32_t* keys, float* weights, int* hist, int* path_count, DSubspace* sub,
                          uint32_t* jump, double* wsorted, float* cmfs, LightVertex* lvc_sorted, uint32_t* guide, hipStream_t s) {
    SamplerBuildBatch b = {};
    b.guide[0] = guide;
    b.lvc[0] = lvc; b.n_host[0] = n_host; b.n_dev[0] = n_dev; b.path_count[0] = path_count; b.sub[0] = sub; b.jump[0] = jump; b.cmfs[0] = cmfs;
    b.lvc_sorted[0] = lvc_sorted;
    b.keys = keys; b.weights = weights; b.hist = hist; b.wsorted = wsorted; b.item_stride = 0;
    launch_sampler_build_batch(b, 1, s);
}

// ------------------------------------------------------------------------------------------------
// Standalone traversal kernels (parity of the software LBVH against the oracle's BVH)
__global__ __launch_bounds__(BLOCK) void k_trace_closest(const KParams p, const float* __restrict__ rays, int n, float* __restrict__ out_t,
                                                        int* __restrict__ out_tri, float* __restrict__ out_uv) {
    __shared__ uint32_t s_stack[BLOCK * STACK_LDS];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    TravStack<BLOCK, STACK_LDS> st;
    st.init(s_stack, p.spill, p.spill_entries, (size_t)i, p.diag);
    const float* r = rays + (size_t)i * 8;
    Counts<false> cn;
    HitRec h;
    traverse<false, false>(p.scene, st, mk3(r[0], r[1], r[2]), mk3(r[4], r[5], r[6]), r[3], r[7], h, cn);
    out_t[i] = h.t;
    out_tri[i] = h.tri >= 0 ? p.scene.tri_orig[h.tri] : -1;
    out_uv[2 * i] = h.u; out_uv[2 * i + 1] = h.v;
}
__global__ __launch_bounds__(BLOCK) void k_trace_any(const KParams p, const float* __restrict__ rays, int n, int* __restrict__ out_visible) {
    __shared__ uint32_t s_stack[BLOCK * STACK_LDS];
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    TravStack<BLOCK, STACK_LDS> st;
    st.init(s_stack, p.spill, p.spill_entries, (size_t)i, p.diag);
    const float* r = rays + (size_t)i * 8;
    Counts<false> cn;
    HitRec h;
    out_visible[i] = traverse<true, false>(p.scene, st, mk3(r[0], r[1], r[2]), mk3(r[4], r[5], r[6]), r[3], r[7], h, cn) ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------
// "pretrace": one PT+NEE eye path per lane producing the training records of the sampling-matrix optimisation
// (__raygen__TrainData raygen.cu:751-868, PreTrace_buildPathInfo 708-740, TrainData::nVertex_device cuProg.h:1128-1292).
struct NVertex {  // TrainData::nVertex, the live fields
    f3 position, dir, normal, weight, color;
    float pdf;
    int materialId, label_id, depth;  // materialId < 0: area light source
};
SPC_DEV NVertex nv_from_eye(const EyeVertex& a) {  // nVertex(const BDPTVertex&, eye_side = true)
    NVertex v;
    v.position = a.c.pos; v.normal = a.c.n; v.color = a.c.color; v.materialId = a.c.mat; v.pdf = a.pdf; v.label_id = a.sub;
    v.depth = a.depth;
    v.dir = a.depth == 0 ? mk3(0.0f) : normalize(a.c.lastPos - a.c.pos);
    v.weight = mk3(a.pdf);
    return v;
}
SPC_DEV NVertex nv_from_light(const LightSampleD& ls) {  // nVertex(light BDPTVertex, eye_side = false), depth 0, QUAD
    NVertex v;
    v.position = ls.position; v.normal = ls.normal; v.color = mk3(0.0f); v.materialId = -1; v.pdf = ls.pdf; v.label_id = ls.subspace;
    v.depth = 0; v.dir = mk3(0.0f); v.weight = ls.emission;
    return v;
}
SPC_DEV Pbr nv_mat(const DeviceScene& S, const NVertex& v) { return load_pbr_colored(S, v.materialId, v.color); }
SPC_DEV f3 nv_forward_eye(const DeviceScene& S, const NVertex& self, const NVertex& b) {  // cuProg.h:1220-1243
    const f3 vec = b.position - self.position;
    const f3 c_dir = normalize(vec);
    const float g = fabsf(dot(c_dir, b.normal)) / dot(vec, vec);
    const float d_pdf = bsdf_pdf(nv_mat(S, self), self.normal, self.dir, c_dir);
    return self.weight * d_pdf * rr_of(self.color) * g;
}
SPC_DEV f3 nv_forward_light(const DeviceScene& S, const NVertex& self, const NVertex& b) {  // cuProg.h:1245-1282
    const f3 vec = b.position - self.position;
    const f3 c_dir = normalize(vec);
    const float g = fabsf(dot(c_dir, b.normal)) * fabsf(dot(c_dir, self.normal)) / dot(vec, vec);
    if (self.materialId < 0) return self.weight * g;
    return self.weight * g * bsdf_eval(nv_mat(S, self), self.normal, self.dir, c_dir);
}
SPC_DEV float nv_forward_light_pdf(const DeviceScene& S, const NVertex& self, const NVertex& b) {  // cuProg.h:1193-1218
    const f3 vec = b.position - self.position;
    const f3 c_dir = normalize(vec);
    float g = fabsf(dot(c_dir, b.normal)) / dot(vec, vec);
    if (self.materialId < 0) {
        g *= fabsf(dot(self.normal, c_dir));
        return self.pdf * g * kInvPi;
    }
    const float d_pdf = bsdf_pdf(nv_mat(S, self), self.normal, self.dir, c_dir);
    return self.pdf * d_pdf * rr_of(self.color) * g;
}
// nVertex_device(a, b, eye_side): the vertex a seen as the next vertex after b
SPC_DEV NVertex nv_extend(const DeviceScene& S, const NVertex& a, const NVertex& b, bool eye_side) {
    NVertex v;
    v.position = a.position;
    v.dir = normalize(b.position - a.position);
    v.normal = a.normal;
    v.weight = eye_side ? nv_forward_eye(S, b, a) : nv_forward_light(S, b, a);
    v.pdf = eye_side ? v.weight.x : nv_forward_light_pdf(S, b, a);
    v.color = a.color; v.materialId = a.materialId; v.label_id = a.label_id; v.depth = b.depth + 1;
    return v;
}
static constexpr int PRETRACE_MAX = 10;  // PRETRACE_CONN_PADDING

SPC_DEV void pretrace_build_path(const DeviceScene& S, const EyeVertex* buffer, int buffer_size, NVertex light,
                                 spcbpt_pretrace_path& path, spcbpt_pretrace_node* conn) {
    path.valid = 1;
    path.begin_ind = 0;
    path.end_ind = buffer_size - 1;
    int e = buffer_size - 1;
    NVertex n_eye = nv_from_eye(buffer[e]);
    const NVertex n_next_eye = nv_extend(S, light, n_eye, true);
    const f3 vec = light.position - n_eye.position;
    const f3 seg_contri = bsdf_eval(nv_mat(S, n_eye), n_eye.normal, n_eye.dir, normalize(vec));  // local_contri
    path.sample_pdf = n_next_eye.pdf + n_eye.pdf * light.pdf;
    path.fix_pdf = n_next_eye.pdf;
    f3 contri = buffer[e].flux * nv_forward_light(S, light, n_eye) * seg_contri;
    for (int i = 0; i < path.end_ind; i++) {
        spcbpt_pretrace_node& nd = conn[path.end_ind - i - 1];  // pathInfo_node(n_eye, light)
        nd.a_position[0] = n_eye.position.x; nd.a_position[1] = n_eye.position.y; nd.a_position[2] = n_eye.position.z;
        nd.b_position[0] = light.position.x; nd.b_position[1] = light.position.y; nd.b_position[2] = light.position.z;
        nd.a_dir[0] = n_eye.dir.x; nd.a_dir[1] = n_eye.dir.y; nd.a_dir[2] = n_eye.dir.z;
        nd.b_dir[0] = light.dir.x; nd.b_dir[1] = light.dir.y; nd.b_dir[2] = light.dir.z;
        nd.a_normal[0] = n_eye.normal.x; nd.a_normal[1] = n_eye.normal.y; nd.a_normal[2] = n_eye.normal.z;
        nd.b_normal[0] = light.normal.x; nd.b_normal[1] = light.normal.y; nd.b_normal[2] = light.normal.z;
        nd.peak_pdf = n_eye.weight.x * sum3(light.weight);
        nd.path_id = 0;
        nd.label_a = n_eye.depth;  // set_eye_depth
        nd.label_b = light.label_id;
        nd.valid = 1;
        nd.light_source = light.materialId < 0 ? 1 : 0;
        e--;
        light = nv_extend(S, n_eye, light, false);
        n_eye = nv_from_eye(buffer[e]);
    }
    const float wgt = sum3(contri) / path.sample_pdf;
    if (isnan(wgt) || isinf(wgt)) contri = mk3(0.0f);
    path.contri[0] = contri.x; path.contri[1] = contri.y; path.contri[2] = contri.z;
}

__global__ __launch_bounds__(BLOCK) void k_pretrace(const KParams p, uint32_t iteration, int num_core, int padding,
                                                    spcbpt_pretrace_path* __restrict__ paths, spcbpt_pretrace_node* __restrict__ nodes) {
    __shared__ uint32_t s_stack[BLOCK * STACK_LDS];
    const int launch_index = blockIdx.x * BLOCK + threadIdx.x;
    if (launch_index >= num_core) return;
    const DeviceScene& S = p.scene;
    Counts<false> cn;
    TravStack<BLOCK, STACK_LDS> st;
    st.init(s_stack, p.spill, p.spill_entries, (size_t)launch_index, p.diag);
    WalkState w;
    w.seed = tea4((uint32_t)launch_index, iteration);
    const float jx = rnd(w.seed), jy = rnd(w.seed);
    w.dir = normalize((2.0f * jx - 1.0f) * ld3(p.U) + (2.0f * jy - 1.0f) * ld3(p.V) + ld3(p.W));
    w.origin = ld3(p.eye);
    w.done = false; w.next_flux = mk3(0.0f); w.next_single_pdf = 1.0f;
    EyeVertex buffer[PRETRACE_MAX];
    EyeVertex& cam = buffer[0];
    cam.c.pos = w.origin; cam.c.n = w.dir; cam.c.color = mk3(0.0f); cam.c.lastPos = w.origin; cam.c.lnp = 0.0f; cam.c.mat = 0; cam.c.lld = false;
    cam.flux = mk3(1.0f); cam.R3 = mk3(0.0f); cam.pdf = 1.0f; cam.singlePdf = 1.0f; cam.sub = 0; cam.lastZone = 0; cam.depth = 0;
    int buffer_size = 1, resample_number = 0, depth = 0;
    spcbpt_pretrace_path path;
    path.valid = 0; path.begin_ind = path.end_ind = 0; path.choice_id = 0; path.sample_pdf = path.fix_pdf = 0.0f;
    path.contri[0] = path.contri[1] = path.contri[2] = 0.0f; path.pad = 0;
    spcbpt_pretrace_node* conn = nodes + (size_t)launch_index * padding;
    while (true) {
        HitRec h;
        if (!traverse<false, false>(S, st, w.origin, w.dir, kEps, 1e16f, h, cn)) break;
        const Geom g = local_geometry(S, h);
        const EyeVertex& last = buffer[buffer_size - 1];
        const f3 ray_dir = w.dir;
        if (g.emitter) {
            const Pbr lm = load_pbr(S, g.mat);
            const DLight& L = S.lights[lm.light_id];
            if (dot(ray_dir, ld3(L.normal)) > 0) break;                       // back of the emitter: no vertex
            if (buffer_size + 1 > 2) {                                        // payload.path.size > 2
                const float r = rnd(w.seed);                                  // rr_acc_accept
                if (1.0f / (resample_number + 1) > r) {
                    const LightSampleD ls = light_reverse_sample(S, L, g.u, g.v);
                    pretrace_build_path(S, buffer, buffer_size, nv_from_light(ls), path, conn);
                    resample_number++;
                }
            }
            break;
        }
        if (buffer_size >= PRETRACE_MAX) break;  // cannot happen: the padding check below stops the walk first
        EyeVertex mid;
        eye_surface_hit(p, g, h.t, ray_dir, last.depth == 0, last, w, mid, cn);
        buffer[buffer_size] = mid;
        buffer_size++;
        // next-event candidate
        // QUAD lights only: upstream picks among all lights here too (raygen.cu:820-823) and then reads the sample's position, which
        // the ENV branch never sets -- undefined, so the sky is left out of the training pass's next-event candidates
        const int n_quads = S.n_lights - (S.env.valid ? 1 : 0);
        const int lid = min(max((int)floorf(rnd(w.seed) * n_quads), 0), n_quads - 1);
        const float r1 = rnd(w.seed), r2 = rnd(w.seed);
        const LightSampleD ls = light_reverse_sample(S, S.lights[lid], r1, r2);
        const f3 vis_vec = ls.position - mid.c.pos;
        const float len = sqrtf(dot(vis_vec, vis_vec));
        HitRec sh;
        if (!traverse<true, false>(S, st, mid.c.pos, vis_vec / len, kEps, len - kEps, sh, cn)) {
            const float r = rnd(w.seed);
            if (1.0f / (resample_number + 1) > r) {
                if (dot(vis_vec, ls.normal) < 0) {
                    pretrace_build_path(S, buffer, buffer_size, nv_from_light(ls), path, conn);
                    resample_number++;
                }
            }
        }
        if (w.done || depth > 50) break;
        if (buffer_size >= padding) break;  // PRETRACER_PADDING_VERTICES_CHECK
        depth += 1;
    }
    int begin_index = 0;
    if (path.valid) begin_index += path.end_ind - path.begin_ind;
    for (int i = begin_index; i < padding; i++) { conn[i].valid = 0; }
    path.sample_pdf = path.sample_pdf / (float)resample_number;
    const int bias = launch_index * padding;
    path.begin_ind += bias;
    path.end_ind += bias;
    path.pixel_id[0] = (int)((float)p.width * jx);
    path.pixel_id[1] = (int)((float)p.height * jy);
    if (path.begin_ind == path.end_ind && path.valid) path.valid = 0;
    paths[launch_index] = path;
}
void launch_pretrace(const KParams& p, uint32_t iteration, int num_core, int padding, spcbpt_pretrace_path* paths,
                     spcbpt_pretrace_node* nodes, hipStream_t s) {
    if (num_core <= 0) return;
    hipLaunchKernelGGL(k_pretrace, dim3((num_core + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, p, iteration, num_core, padding, paths, nodes);
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
int light_trace_blocks(const KParams& p, int max_blocks) {
    const long long cores = (long long)p.core_count * (p.n_lframes > 0 ? p.n_lframes : 1);
    long long blocks = (cores + BLOCK - 1) / BLOCK;
    if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
    return (int)blocks;
}
void launch_light_trace(const KParams& p, int variant, int max_blocks, hipStream_t s) {   // variants as launch_spcbpt
    const int blocks = light_trace_blocks(p, max_blocks);   // p.work_counter (the core queue head) must have been zeroed on `s`
    if (blocks <= 0) return;
    if (variant == 1) {   // (the reference's own evaluation of Gamma / Q, counted as such: launch_spcbpt)
        KParams q = p;
        q.gamma_q = nullptr;
        hipLaunchKernelGGL((k_light_trace<true, false>), dim3(blocks), dim3(BLOCK), 0, s, q);
    }
    else if (variant == 2) hipLaunchKernelGGL((k_light_trace<true, true>), dim3(blocks), dim3(BLOCK), 0, s, p);
    else hipLaunchKernelGGL((k_light_trace<false, true>), dim3(blocks), dim3(BLOCK), 0, s, p);
}
void launch_lvc_compact(const LightVertex* scratch, const int* core_counts, const int* core_offsets, int core_count, int core_padding,
                        LightVertex* lvc, uint32_t* keys, uint32_t* vals, float* weights, int* sampler_counts, int capacity, uint32_t* overflow,
                        hipStream_t s) {
    const long long total = (long long)core_count * core_padding;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(k_lvc_compact, dim3(blocks), dim3(256), 0, s, scratch, core_counts, core_offsets, core_count,
                       core_padding, lvc, keys, vals, weights, sampler_counts, capacity, overflow);
}
void launch_lvc_compact_batch(const LightVertex* scratch, const int* core_counts, const int* core_offsets, const int* path_counts, int core_count,
                              int core_padding, int n, const CompactBatch& dst, int capacity, uint32_t* overflow, hipStream_t s) {
    const long long total = (long long)core_count * core_padding;
    hipLaunchKernelGGL(k_lvc_compact_batch, dim3((unsigned)((total + 255) / 256), (unsigned)n), dim3(256), 0, s, scratch, core_counts, core_offsets,
                       path_counts, core_count, core_padding, dst, capacity, overflow);
}
void launch_fill_keys(const LightVertex* lvc, int n, uint32_t* keys, uint32_t* vals, float* weights, int* sampler_counts, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fill_keys_from_lvc, dim3((n + 255) / 256), dim3(256), 0, s, lvc, n, keys, vals, weights, sampler_counts);
}
void launch_fill_keys_devcount(const LightVertex* lvc, int bound, uint32_t* keys, uint32_t* vals, float* weights, const int* sampler_counts, hipStream_t s) {
    if (bound <= 0) return;
    hipLaunchKernelGGL(k_fill_keys_devcount, dim3((bound + 255) / 256), dim3(256), 0, s, lvc, bound, keys, vals, weights, sampler_counts);
}
void launch_gather_compact(const LightVertex* gathered, const int* counts_all, int world, int cap, int lvc_capacity, const CompactBatch& dst, int nf,
                           int* overflow, hipStream_t s) {
    const int chunks = (cap + 255) / 256;
    hipLaunchKernelGGL(k_gather_compact, dim3((unsigned)(world * chunks), (unsigned)nf), dim3(256), 0, s, gathered, counts_all, world, cap, lvc_capacity, dst,
                       nf, overflow);
}
void launch_pack_shards(const CompactBatch& src, int nf, int cap, LightVertex* send, int* send_counts, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_shards, dim3((unsigned)((cap + 255) / 256), (unsigned)nf), dim3(256), 0, s, src, cap, send, send_counts);
}
void launch_pack_bands(float* accum, int width, int height, int rank, int world, float* packed, bool unpack_all, hipStream_t s) {
    const int bands = (height + 7) / 8, per_rank = (bands + world - 1) / world;
    const size_t n = (size_t)(unpack_all ? world : 1) * per_rank * 8 * width;
    if (n == 0) return;
    hipLaunchKernelGGL(k_pack_bands, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(accum), width, height, rank, world,
                       reinterpret_cast<float4*>(packed), unpack_all ? 1 : 0);
}
void launch_subspace_ranges(const uint32_t* sorted_keys, const int* sampler_counts, DSubspace* sub, int capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_subspace_ranges, dim3((capacity + 255) / 256), dim3(256), 0, s, sorted_keys, sampler_counts, sub);
    hipLaunchKernelGGL(k_finish_ranges, dim3(1), dim3(1024), 0, s, sub);
}
void launch_gather_weights(const float* weights, const uint32_t* sorted_vals, const int* sampler_counts, double* out, int capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_weights, dim3((capacity + 255) / 256), dim3(256), 0, s, weights, sorted_vals, sampler_counts, out);
}
void launch_cmf(const double* prefix, const uint32_t* sorted_keys, const int* sampler_counts, DSubspace* sub, float* cmfs, int capacity, hipStream_t s) {
    hipLaunchKernelGGL(k_cmf, dim3((capacity + 255) / 256), dim3(256), 0, s, prefix, sorted_keys, sampler_counts, sub, cmfs);
}
void launch_trace_closest(const KParams& p, const float* rays, int n, float* t, int* tri, float* uv, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_trace_closest, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, p, rays, n, t, tri, uv);
}
void launch_trace_any(const KParams& p, const float* rays, int n, int* vis, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_trace_any, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, p, rays, n, vis);
}

}  // namespace spc
