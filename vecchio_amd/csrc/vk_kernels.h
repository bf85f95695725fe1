// vk_kernels.h — the device code of libvecchio_amd.so: the persistent megakernel (render_kernel) and the small
// kernels around it (tile order, resolve, output stage, tile slabs).  Included by vk_api.hip only.
//
// Kernel structure (gfx950 / CDNA4, wave64): see the head of vk_api.hip.
#ifndef VK_KERNELS_H
#define VK_KERNELS_H

#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/vecchio_amd.h"
#include "vk_trace.h"

using namespace vkd;

namespace {

constexpr size_t LDS_PER_CU = 160 * 1024;
constexpr int TILE = 8;   // 8x8 pixels = one wave
#ifndef VK_BOX_UNROLL
#define VK_BOX_UNROLL 4
#endif
// The scheduler runs SHADE + REFILL only when its lanes outnumber the box lanes AND the primitive lanes SHADE_DEFER
// times over: a shading phase costs ~700 issue slots against ~42 of a box step, so it pays to keep traversing with
// thinning waves until nearly every lane waits for shading and then shade them all at once.  C2 / C4, Msamples/s:
// 1 (plain plurality): 4 070 / 3 435; 1.5: 4 270 / 3 645; 2: 4 430 / 3 725; 3: 4 580 / 3 815; 4: 4 630 / 3 820;
// 8: 4 600 / 3 765.  (Deferring only against BOX, not PRIM: 4 150.)  Scenes that miss L2 (C5) are bound by the
// latency of the item gathers, not by issue slots: there every parked lane is a gather less in flight, and plain
// plurality is better (C5: 69 against 58 Msamples/s), so the factor is a launch parameter.
#ifndef VK_SHADE_DEFER
#define VK_SHADE_DEFER 4
#endif
constexpr uint32_t SHADE_DEFER = VK_SHADE_DEFER;
constexpr int BOX_UNROLL = VK_BOX_UNROLL;   // box steps between two exit tests of the BOX loop

struct KArgs {
    DScene S;
    RenderConsts C;
    float *out;              // full framebuffer (width*height*3)
    float *partial;          // [n_chunks][width*height*3] when n_chunks > 1
    float4 *debug;           // optional per-sample (rgb, draws) dump
    uint32_t *counter;       // work-unit counter
    uint32_t *tile_cost;     // [tiles of the image] time spent on each tile (1.6 us ticks): written by the probe (COST) build only
    const uint32_t *tile_order;   // [n_local_tiles] local tile slots, dearest first (from the probe launch), or null = raster order
    uint32_t tiles_x, tiles_y;
    uint32_t n_local_tiles;  // tiles of this call's partition
    uint32_t tile_rank, tile_world;
    uint32_t n_chunks;
    uint32_t shade_defer;    // SHADE + REFILL runs when its lanes outnumber box and primitive lanes this many times (see SHADE_DEFER)
    uint32_t lds_items, lds_spheres, lds_boxes;   // record counts staged into LDS (LDS variant)
    unsigned long long *phase_stats;   // optional (diagnostic build of the kernel): 16 counters, see vk_debug_phase_stats
};

// LDS-resident hot records
struct LdsMem {
    const uint4 *items;      // first halves of all items (x/y bounds) ...
    const uint4 *items_hi;   // ... then the second halves (z bounds, w0, w1): 16-byte stride per array gives a
                             // ds_read_b128 16 bank slots instead of the 8 a 32-byte stride leaves it
    const float4 *spheres;
    const uint4 *boxes;      // 2 x uint4 per DBox
    const uint32_t *sphere_mat;
    __device__ __forceinline__ DBox box(uint32_t i) const {
        uint4 a = boxes[2 * i], b = boxes[2 * i + 1];
        DBox o;
        o.p0[0] = __uint_as_float(a.x); o.p0[1] = __uint_as_float(a.y); o.p0[2] = __uint_as_float(a.z);
        o.p1x = __uint_as_float(a.w); o.p1y = __uint_as_float(b.x); o.p1z = __uint_as_float(b.y);
        o.mat = b.z; o._p = 0;
        return o;
    }
    __device__ __forceinline__ DItem item(uint32_t i) const {
        uint4 a = items[i], b = items_hi[i];
        DItem n;
        n.mnx = __uint_as_float(a.x); n.mxx = __uint_as_float(a.y); n.mny = __uint_as_float(a.z); n.mxy = __uint_as_float(a.w);
        n.mnz = __uint_as_float(b.x); n.mxz = __uint_as_float(b.y);
        n.w0 = b.z; n.w1 = b.w;
        return n;
    }
    __device__ __forceinline__ DSphere sphere(uint32_t i) const {
        float4 s = spheres[i];
        DSphere o; o.cx = s.x; o.cy = s.y; o.cz = s.z; o.r = s.w;
        return o;
    }
    __device__ __forceinline__ uint32_t smat(uint32_t i) const { return sphere_mat[i]; }
};

extern __shared__ uint4 smem[];

// The persistent loop below is one big region; left alone, LLVM hoists every value that is
// invariant across it (seed hashes, camera terms, scene pointers, division magic numbers)
// into the prologue and then spills them (>1 KB of scratch per lane, reloaded inside the hot
// loop).  So nothing is read from the by-value kernel argument directly: each phase re-reads
// what it needs from the kernarg segment (scalar loads, K$-resident) through a pointer that
// is laundered by an empty asm, which pins the loads, and everything derived from them,
// inside the phase that uses them.
typedef const __attribute__((address_space(4))) KArgs *KArgsC;
__device__ __forceinline__ KArgsC kargs_fresh() {
    KArgsC p = (KArgsC)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
#define KARG(p, field) (*(const decltype(KArgs::field) *)&((p)->field))

template <uint32_t F, bool LDS_SCENE>
__device__ __forceinline__ typename std::conditional<LDS_SCENE, LdsMem, GlobalMem>::type make_mem(const DScene &S, uint32_t lds_items) {
    typename std::conditional<LDS_SCENE, LdsMem, GlobalMem>::type M;
    if constexpr (LDS_SCENE) {
        M.items = smem; M.items_hi = smem + lds_items; M.spheres = reinterpret_cast<const float4 *>(smem + 2u * lds_items);
        M.boxes = smem + 2u * lds_items + S.n_spheres; M.sphere_mat = S.sphere_mat;
    } else {
        M.items = S.items; M.spheres = S.spheres; M.sphere_mat = S.sphere_mat; M.boxes = S.boxes;
    }
    return M;
}

// Cold per-lane path state (throughput, radiance, RNG, pixel/sample ids, world ray) lives in
// LDS between SHADE phases, SoA by field (word f of lane l at cold[f*64 + l]: conflict-free),
// so that the box/primitive loops keep only the traversal state in VGPRs.
constexpr int NCOLD_BASE = 11;      // thr3 acc3 depth key2 ctr (q | sample << 6)
constexpr int NCOLD_INST = 17;      // + world-space ray (o3 d3) for scenes with instances
template <uint32_t F> constexpr int ncold() { return (F & VKF_INSTANCE) ? NCOLD_INST : NCOLD_BASE; }

template <uint32_t F>
__device__ __forceinline__ void cold_store(float *c, uint32_t lane, const Lane &L, uint32_t q) {
    c[0 * 64 + lane] = L.thr.x; c[1 * 64 + lane] = L.thr.y; c[2 * 64 + lane] = L.thr.z;
    c[3 * 64 + lane] = L.acc.x; c[4 * 64 + lane] = L.acc.y; c[5 * 64 + lane] = L.acc.z;
    c[6 * 64 + lane] = __uint_as_float(L.depth);
    c[7 * 64 + lane] = __uint_as_float((uint32_t)L.rng.key); c[8 * 64 + lane] = __uint_as_float((uint32_t)(L.rng.key >> 32));
    c[9 * 64 + lane] = __uint_as_float(L.rng.ctr);
    c[10 * 64 + lane] = __uint_as_float(q | (L.sample << 6));   // the pixel is implied by the unit's tile and q
    if (F & VKF_INSTANCE) {
        c[11 * 64 + lane] = L.wo.x; c[12 * 64 + lane] = L.wo.y; c[13 * 64 + lane] = L.wo.z;
        c[14 * 64 + lane] = L.wd.x; c[15 * 64 + lane] = L.wd.y; c[16 * 64 + lane] = L.wd.z;
    }
}
template <uint32_t F>
__device__ __forceinline__ void cold_load(const float *c, uint32_t lane, Lane &L, uint32_t &q) {
    L.thr = v3(c[0 * 64 + lane], c[1 * 64 + lane], c[2 * 64 + lane]);
    L.acc = v3(c[3 * 64 + lane], c[4 * 64 + lane], c[5 * 64 + lane]);
    L.depth = __float_as_uint(c[6 * 64 + lane]);
    L.rng.key = (uint64_t)__float_as_uint(c[7 * 64 + lane]) | ((uint64_t)__float_as_uint(c[8 * 64 + lane]) << 32);
    L.rng.ctr = __float_as_uint(c[9 * 64 + lane]);
    uint32_t qs = __float_as_uint(c[10 * 64 + lane]);
    q = qs & 63u; L.sample = qs >> 6; L.pixel = 0;
    if (F & VKF_INSTANCE) {
        L.wo = v3(c[11 * 64 + lane], c[12 * 64 + lane], c[13 * 64 + lane]);
        L.wd = v3(c[14 * 64 + lane], c[15 * 64 + lane], c[16 * 64 + lane]);
    } else {
        L.wo = L.o; L.wd = L.d;       // no instances: the current space IS world space
    }
}
template <uint32_t F>
__device__ __forceinline__ void cold_load_world_ray(const float *c, uint32_t lane, Lane &L) {
    if (F & VKF_INSTANCE) {
        L.wo = v3(c[11 * 64 + lane], c[12 * 64 + lane], c[13 * 64 + lane]);
        L.wd = v3(c[14 * 64 + lane], c[15 * 64 + lane], c[16 * 64 + lane]);
    }
}

// LDS word of a wave that holds the time its previous unit ended.  Re-derived from freshly loaded kernel arguments at
// its uses (kernel start, unit end) so that no pointer stays live across the traversal loops.
template <uint32_t F, bool LDS_SCENE>
__device__ __forceinline__ uint32_t *unit_t0_word(uint32_t wave) {
    KArgsC P = kargs_fresh();
    uint32_t scene16 = LDS_SCENE ? 2u * KARG(P, lds_items) + KARG(P, lds_spheres) + 2u * KARG(P, lds_boxes) : 0u;
    float *dyn = reinterpret_cast<float *>(smem + scene16);
    return reinterpret_cast<uint32_t *>(dyn + (blockDim.x >> 6) * (64 * 3 + 64 * ncold<F>())) + wave;
}

// number of lanes of the wave for which p holds (v_cmp -> s_bcnt1, no VGPR round trip)
__device__ __forceinline__ uint32_t lanes_with(bool p) { return (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); }

template <uint32_t F, bool LDS_SCENE, int MINW, bool STATS, bool COST = false>
__global__ __launch_bounds__((MINW <= 4 ? 1024 : (MINW == 5 ? 640 : (MINW == 6 ? 768 : 1024))), MINW) void render_kernel(KArgs A_byval) {
    (void)A_byval;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    using Mem = typename std::conditional<LDS_SCENE, LdsMem, GlobalMem>::type;
    // diagnostic counters (STATS build only): phase executions and the lanes that had work in them
    unsigned long long st_box_steps = 0, st_box_lanes = 0, st_prim_execs = 0, st_prim_lanes = 0, st_shade_execs = 0, st_shade_lanes = 0, st_sched = 0;
    unsigned long long st_t_box = 0, st_t_light = 0, st_t_heavy = 0, st_t_shade = 0, st_heavy_execs = 0, st_t0 = 0, st_t_total = 0, st_t_mat = 0, st_t_refill = 0, st_t_install = 0, st_t1 = 0;
    if (STATS) st_t_total = clock64();

    // ---- LDS layout: [items][spheres][per-wave pixel accumulators][per-wave cold lane state]
    uint32_t lds_items = 0;
    float *acc_lds, *cold;
    {
        KArgsC P = kargs_fresh();
        lds_items = LDS_SCENE ? KARG(P, lds_items) : 0u;
        uint32_t lds_spheres = LDS_SCENE ? KARG(P, lds_spheres) : 0u;
        uint32_t lds_boxes = LDS_SCENE ? KARG(P, lds_boxes) : 0u;
        float *dyn = reinterpret_cast<float *>(smem + (2u * lds_items + lds_spheres + 2u * lds_boxes));
        acc_lds = dyn + wave * (64 * 3);
        cold = dyn + (blockDim.x >> 6) * (64 * 3) + wave * (64 * ncold<F>());
        if (LDS_SCENE) {
            const uint4 *gi = reinterpret_cast<const uint4 *>(KARG(P, S.items));
            for (uint32_t k = threadIdx.x; k < 2u * lds_items; k += blockDim.x) smem[(k >> 1) + ((k & 1u) ? lds_items : 0u)] = gi[k];
            const uint4 *gs = reinterpret_cast<const uint4 *>(KARG(P, S.spheres));
            for (uint32_t k = threadIdx.x; k < lds_spheres; k += blockDim.x) smem[2u * lds_items + k] = gs[k];
            const uint4 *gb = reinterpret_cast<const uint4 *>(KARG(P, S.boxes));
            for (uint32_t k = threadIdx.x; k < 2u * lds_boxes; k += blockDim.x) smem[2u * lds_items + lds_spheres + k] = gb[k];
            __syncthreads();
        }
    }

    if (COST) { if (lane == 0) *unit_t0_word<F, LDS_SCENE>(wave) = (uint32_t)(wall_clock64() >> 4); }
    // (per-XCD work queues — contiguous image bands per XCD, stealing when empty — were tried for L2 locality on
    // C5: no gain there, -1.7 % on C2, and the two extra live scalars doubled the everything-variant's spills)
    for (;;) {
        KArgsC U = kargs_fresh();
        uint32_t unit = 0;
        if (lane == 0) unit = atomicAdd(KARG(U, counter), 1u);
        unit = __builtin_amdgcn_readfirstlane(unit);
        const uint32_t n_chunks = KARG(U, n_chunks);
        if (unit >= KARG(U, n_local_tiles) * n_chunks) break;
        const uint32_t chunk = unit % n_chunks;
        // tiles are visited dearest-first when the probe launch left an order (see enqueue_render)
        uint32_t tslot = unit / n_chunks;
        { const uint32_t *ord = KARG(U, tile_order); if (ord) tslot = ord[tslot]; }
        const uint32_t tile = KARG(U, tile_rank) + tslot * KARG(U, tile_world);
        const uint32_t tiles_x = KARG(U, tiles_x);
        const uint32_t tx = (tile % tiles_x) * TILE, ty = (tile / tiles_x) * TILE;
        const uint32_t spp = KARG(U, C.spp);
        const uint32_t s0 = (uint32_t)(((uint64_t)spp * chunk) / n_chunks);
        const uint32_t s1 = (uint32_t)(((uint64_t)spp * (chunk + 1)) / n_chunks);
        const uint32_t total = 64u * (s1 - s0);       // items: k -> (pixel slot k & 63, sample s0 + (k >> 6))

        acc_lds[lane] = 0.0f; acc_lds[64 + lane] = 0.0f; acc_lds[128 + lane] = 0.0f;

        Lane L;
        memset(&L, 0, sizeof(L));
        bool need = true;          // lane wants a new (pixel, sample)
        bool active = false;       // lane holds a live path
        uint32_t next_item = 0;    // wave-uniform
        // Wave-level phase scheduler.  Every lane is in one of four states; each round the
        // wave runs the code of the most populated state with the lanes in it (64-bit ballots
        // + s_bcnt1), so the long box loop, the primitive tests and the (expensive, rare)
        // shading / ray-generation code each execute with as many lanes as possible instead
        // of all being paid for on every iteration:
        //   BOX    pend == 0 and items (or an instance to leave) remain  -> box_step
        //   PRIM   pend != 0                                             -> prim_step
        //   SHADE  live path whose segment is fully traversed            -> shade
        //   NEED   no path; the unit still has (pixel, sample) items     -> start_sample
        // SHADE and NEED run as one phase: a path that ends hands its lane straight to the
        // next item (ballot + prefix popcount = active-ray compaction).
        const uint32_t shade_defer = KARG(U, shade_defer);
        for (;;) {
            bool is_prim = active && has_prim_work(L);
            bool is_box = active && !is_prim && traversing(L);
            bool is_shade = active && !is_prim && !is_box;
            // primitives come in two weights (sphere/rect ~50 instructions; Boxy, list, medium, instance
            // entry several times that): scheduled separately so cheap tests never pay for heavy ones
            const bool HAS_HEAVY = (F & (VKF_LIST | VKF_MEDIUM | VKF_INSTANCE | VKF_BOX)) != 0;
            bool is_heavy = HAS_HEAVY && is_prim && prim_is_heavy(L.pend);
            uint32_t n_box = lanes_with(is_box);
            uint32_t n_heavy = HAS_HEAVY ? lanes_with(is_heavy) : 0u;
            uint32_t n_light = lanes_with(is_prim && !is_heavy);
            if (n_heavy > n_light) { is_prim = is_heavy; } else { is_prim = is_prim && !is_heavy; }
            uint32_t n_prim = n_heavy > n_light ? n_heavy : n_light;
            uint32_t n_sn = lanes_with(is_shade || need);
            if ((n_box | n_prim | n_sn) == 0) break;
            if (STATS) st_sched++;
            if (n_box >= n_prim && n_box * shade_defer >= n_sn) {
                // ---- BOX: UNROLL steps under a shrinking EXEC mask per exit test, while box lanes are the plurality
                KArgsC P = kargs_fresh();
                DScene S = KARG(P, S);
                Mem M = make_mem<F, LDS_SCENE>(S, lds_items);
                if (STATS) st_t0 = clock64();
                cold_load_world_ray<F>(cold, lane, L);
                const uint32_t live = n_box + n_heavy + n_light + n_sn;   // lanes only change state here, none appear or vanish
                // steps between two exit tests (compile time: a run-time trip count costs 4-13 %).  With shading deferred the
                // optimum is 3 for the sphere-only and Cornell-type variants (against 4: C2 +0.7 %, C4 +1.3 %; 2: -3 %, 6: -1 %,
                // 8: -6 %) and 4 for the everything-variants (C3: 3 -> -2 %)
                constexpr int UNROLL = ((F & VKF_ALL_SCENE) == VKF_ALL_SCENE) ? BOX_UNROLL : BOX_UNROLL - 1;
                for (;;) {
                    if (F & VKF_INSTANCE) {     // end of an instance's item range: back to the parent space (rare)
                        if (is_box && L.i >= L.end && L.cur_inst >= 0) leave_instance<F, Mem>(L, S);
                    }
                    bool go = active && L.pend == 0u && L.i < range_end<F>(L, S);
                    if (STATS) {        // diagnostic build: same steps one at a time, counting the lanes in each
                        for (int u = 0; u < UNROLL; u++) {
                            st_box_steps += 1; st_box_lanes += lanes_with(go);
                            box_steps<F, Mem, 1>(L, S, M, go);
                            go = active && L.pend == 0u && L.i < range_end<F>(L, S);
                        }
                    } else {
                        box_steps<F, Mem, UNROLL>(L, S, M, go);
                    }
                    is_box = active && !has_prim_work(L) && traversing(L);
                    uint32_t nb = lanes_with(is_box);
                    uint32_t np = lanes_with(active && has_prim_work(L));
                    uint32_t ns = live - nb - np;
                    if (nb == 0 || nb < np || nb * shade_defer < ns) {               // another state now has more lanes parked than are stepping
                        // sphere-only variants: when that state is PRIM, test the pending spheres right here and
                        // keep stepping (saves the scheduler round trip that otherwise follows every ~10 box steps)
                        if (!HAS_HEAVY && !(F & VKF_MEDIUM) && np != 0 && np * shade_defer >= ns) {
                            if (active && has_prim_work(L)) prim_step<F, Mem>(L, S, M);
                            is_box = active && !has_prim_work(L) && traversing(L);
                            continue;
                        }
                        break;
                    }
                }
                if (STATS) st_t_box += clock64() - st_t0;
            } else if (n_prim * shade_defer >= n_sn) {
                // ---- PRIM: intersect / enter the pending object
                KArgsC P = kargs_fresh();
                DScene S = KARG(P, S);
                Mem M = make_mem<F, LDS_SCENE>(S, lds_items);
                if (STATS) { st_prim_execs++; st_prim_lanes += n_prim; st_t0 = clock64(); if (n_heavy > n_light) st_heavy_execs++; }
                if (is_prim) {
                    if (F & VKF_MEDIUM) {    // ConstantMedium::hit draws inside traversal (hittable.rs:473)
                        L.rng.key = (uint64_t)__float_as_uint(cold[7 * 64 + lane]) | ((uint64_t)__float_as_uint(cold[8 * 64 + lane]) << 32);
                        L.rng.ctr = __float_as_uint(cold[9 * 64 + lane]);
                    }
                    prim_step<F, Mem>(L, S, M);
                    if (F & VKF_MEDIUM) cold[9 * 64 + lane] = __uint_as_float(L.rng.ctr);
                }
                if (STATS) { if (n_heavy > n_light) st_t_heavy += clock64() - st_t0; else st_t_light += clock64() - st_t0; }
            } else {
                // ---- SHADE + REFILL
                KArgsC P = kargs_fresh();
                RenderConsts C = KARG(P, C);
                DScene S = KARG(P, S);
                Mem M = make_mem<F, LDS_SCENE>(S, lds_items);
                if (STATS) { st_shade_execs++; st_shade_lanes += n_sn; st_t0 = clock64(); }
                uint32_t q = 0;
                bool touched = is_shade;      // lanes whose cold state is in registers during this phase
                bool fresh = false;           // lanes that leave this phase with a new ray to install; it is parked in the
                                              // (dead) world-ray fields L.wo / L.wd / L.time, so it costs no registers
                // (the everything-variant keeps one begin_segment per call site: merging them there doubled its spills)
                constexpr bool ONE_INSTALL = (F & VKF_ALL_SCENE) != VKF_ALL_SCENE;
                if (is_shade) {
                    cold_load<F>(cold, lane, L, q);
                    if (STATS) st_t1 = clock64();
                    bool cont;
                    if (ONE_INSTALL) {
                        V3 no, nd; float nt;
                        cont = shade_core<F, Mem>(L, S, M, C, no, nd, nt);
                        if (cont) { L.wo = no; L.wd = nd; L.time = nt; fresh = true; }
                    } else {
                        cont = shade<F, Mem>(L, S, M, C);
                    }
                    if (STATS) st_t_mat += clock64() - st_t1;
                    if (!cont) {
                        float4 *dbg = KARG(P, debug);
                        if (dbg) dbg[((size_t)(ty + (q >> 3)) * C.width + (tx + (q & 7u))) * C.spp + L.sample] = make_float4(L.acc.x, L.acc.y, L.acc.z, __uint_as_float(L.rng.ctr));
                        if (isfinite(L.acc.x) && isfinite(L.acc.y) && isfinite(L.acc.z)) {   // main.rs:192-194
                            atomicAdd(&acc_lds[q * 3 + 0], L.acc.x);
                            atomicAdd(&acc_lds[q * 3 + 1], L.acc.y);
                            atomicAdd(&acc_lds[q * 3 + 2], L.acc.z);
                        }
                        active = false;
                        need = true;
                    }
                }
                if (STATS) st_t1 = clock64();
                unsigned long long need_mask = __builtin_amdgcn_ballot_w64(need);
                if (need_mask) {
                    uint32_t rank = __popcll(need_mask & ((1ull << lane) - 1ull));
                    if (need) {
                        uint32_t k = next_item + rank;
                        if (k < total) {
                            q = k & 63u;
                            uint32_t s = s0 + (k >> 6);
                            uint32_t px = tx + (q & 7u), py = ty + (q >> 3);
                            if (px < C.width && py < C.height) {   // slots outside the image (edge tiles) are skipped
                                if (ONE_INSTALL) {
                                    V3 no, nd; float nt;
                                    start_sample_core(L, C, px, py, s, no, nd, nt);
                                    L.wo = no; L.wd = nd; L.time = nt;
                                    fresh = true;
                                } else {
                                    start_sample(L, S, C, px, py, s);
                                }
                                active = true;
                                need = false;
                                touched = true;
                            }
                        } else {
                            need = false;                           // unit exhausted: this lane idles until the wave drains
                        }
                    }
                    next_item += (uint32_t)__popcll(need_mask);
                }
                if (STATS) { st_t_refill += clock64() - st_t1; st_t1 = clock64(); }
                if (ONE_INSTALL && fresh) begin_segment(L, S, L.wo, L.wd, L.time);   // one copy of the exact reciprocals for both kinds of new ray
                if (active && touched) cold_store<F>(cold, lane, L, q);
                if (STATS) { st_t_shade += clock64() - st_t0; st_t_install += clock64() - st_t1; }
            }
        }
        // ---- write the unit's pixel sums
        {
            KArgsC P = kargs_fresh();
            if (COST) {   // probe launch: time since this wave's previous unit ended = this unit's cost (units run back to back)
                uint32_t *tc = KARG(P, tile_cost);
                if (tc && lane == 0) {
                    uint32_t *w = unit_t0_word<F, LDS_SCENE>(wave);
                    uint32_t now = (uint32_t)(wall_clock64() >> 4);
                    atomicAdd(&tc[(ty / TILE) * KARG(P, tiles_x) + tx / TILE], now - *w);
                    *w = now;
                }
            }
            uint32_t width = KARG(P, C.width), height = KARG(P, C.height);
            uint32_t px = tx + (lane & 7u), py = ty + (lane >> 3);
            if (px < width && py < height) {
                size_t pix = (size_t)py * width + px;
                float r = acc_lds[lane * 3 + 0], g = acc_lds[lane * 3 + 1], b = acc_lds[lane * 3 + 2];
                if (n_chunks == 1) {
                    float n = (float)KARG(P, C.spp);                         // main.rs:196
                    float *o = KARG(P, out);
                    o[pix * 3 + 0] = r / n; o[pix * 3 + 1] = g / n; o[pix * 3 + 2] = b / n;
                } else {
                    float *p = KARG(P, partial) + ((size_t)chunk * ((size_t)width * height) + pix) * 3;
                    p[0] = r; p[1] = g; p[2] = b;
                }
            }
        }
    }
    if (STATS) {
        KArgsC P = kargs_fresh();
        unsigned long long *ps = KARG(P, phase_stats);
        if (ps && lane == 0) {
            atomicAdd(&ps[0], st_box_steps); atomicAdd(&ps[1], st_box_lanes); atomicAdd(&ps[2], st_prim_execs); atomicAdd(&ps[3], st_prim_lanes);
            atomicAdd(&ps[4], st_shade_execs); atomicAdd(&ps[5], st_shade_lanes); atomicAdd(&ps[6], st_sched); atomicAdd(&ps[7], st_heavy_execs);
            atomicAdd(&ps[8], st_t_box); atomicAdd(&ps[9], st_t_light); atomicAdd(&ps[10], st_t_heavy); atomicAdd(&ps[11], st_t_shade);
            atomicAdd(&ps[12], (unsigned long long)(clock64() - st_t_total));
            atomicAdd(&ps[13], st_t_mat); atomicAdd(&ps[14], st_t_refill); atomicAdd(&ps[15], st_t_install);
        }
    }
}

// ---- heavy-first tile order (bucket sort of the probe's per-tile times, dearest first).
// 8 buckets per octave of cost; the order inside a bucket is arbitrary, which is fine: any order renders the same image.
constexpr uint32_t ORDER_BUCKETS = 256;
__device__ __forceinline__ uint32_t cost_bucket(uint32_t c) {
    if (c < 8u) return c;
    uint32_t msb = 31u - (uint32_t)__builtin_clz(c);
    uint32_t b = (msb - 2u) * 8u + ((c >> (msb - 3u)) & 7u);
    return b < ORDER_BUCKETS ? b : ORDER_BUCKETS - 1u;
}
__global__ void order_hist_kernel(const uint32_t *cost, uint32_t n_local, uint32_t tile_rank, uint32_t tile_world, uint32_t *hist) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_local) atomicAdd(&hist[cost_bucket(cost[tile_rank + i * tile_world])], 1u);
}
__global__ void order_scan_kernel(uint32_t *hist) {      // one thread: start offset of every bucket, dearest bucket first
    uint32_t run = 0;
    for (int b = (int)ORDER_BUCKETS - 1; b >= 0; b--) { uint32_t c = hist[b]; hist[b] = run; run += c; }
}
__global__ void order_scatter_kernel(uint32_t *cost, uint32_t n_local, uint32_t tile_rank, uint32_t tile_world, uint32_t *hist, uint32_t *order) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_local) return;
    uint32_t t = tile_rank + i * tile_world;
    order[atomicAdd(&hist[cost_bucket(cost[t])], 1u)] = i;
}

// sums the sample chunks of each pixel in chunk order (deterministic) and divides by spp
__global__ void resolve_kernel(const float *partial, float *out, uint32_t width, uint32_t height, uint32_t n_chunks, uint32_t spp,
                               uint32_t tiles_x, uint32_t tile_rank, uint32_t tile_world) {
    uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_pixels = width * height;
    if (pix >= n_pixels) return;
    uint32_t x = pix % width, y = pix / width;
    uint32_t tile = (y / TILE) * tiles_x + (x / TILE);
    if (tile % tile_world != tile_rank) return;
    float r = 0.0f, g = 0.0f, b = 0.0f;
    for (uint32_t c = 0; c < n_chunks; c++) {
        const float *p = partial + ((size_t)c * n_pixels + pix) * 3;
        r += p[0]; g += p[1]; b += p[2];
    }
    float n = (float)spp;
    out[(size_t)pix * 3 + 0] = r / n; out[(size_t)pix * 3 + 1] = g / n; out[(size_t)pix * 3 + 2] = b / n;
}

// Vec3::to_color (vec3.rs:44-61): sqrt gamma, hand-written clamp (NaN falls through it), *256, `as u32`
// (saturating; NaN -> 0), for one component
__device__ __forceinline__ uint8_t to_color_u8(float x) {
    float v = sqrtf(x);
    float cl = v < 0.0f ? 0.0f : (v > 0.999f ? 0.999f : v);
    return (uint8_t)vk::sat_u32(256.0f * cl);
}
// whole image: Vec3::to_color + top-down rows (main.rs:209)
__global__ void to_color_kernel(const float *rgb, uint32_t width, uint32_t height, uint8_t *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)width * height * 3;
    if (i >= n) return;
    uint32_t c = (uint32_t)(i % 3);
    size_t pix = i / 3;
    uint32_t x = (uint32_t)(pix % width), row = (uint32_t)(pix / width);
    uint32_t y = height - 1 - row;
    out[i] = to_color_u8(rgb[((size_t)y * width + x) * 3 + c]);
}

// ---- tile slabs: the pixels of one tile partition (tiles t = rank + i*world, i = 0..n_local) packed tile by tile,
// 64 pixel slots per tile, 3 components per slot.  A multi-device scene moves one slab per device to devices[0]
// (the path's only exchange) and de-interleaves it there; RGB8 output packs bytes (to_color fused: 4x less traffic).
enum : int { TM_PACK_F32 = 0, TM_PACK_U8 = 1, TM_UNPACK_F32 = 2, TM_UNPACK_U8 = 3, TM_CONVERT_U8 = 4, TM_ZERO_F32 = 5, TM_ZERO_U8 = 6 };
//   TM_PACK_*     fb (f32, y up) -> slab          TM_UNPACK_F32  slab -> fb (f32, y up)
//   TM_UNPACK_U8  slab (u8) -> rgb8 image, top row first          TM_CONVERT_U8  fb (f32) -> rgb8 image, this partition only
//   TM_ZERO_*     this partition's pixels := 0 (max_depth 0: every sample is (0,0,0), main.rs:126-128)
template <int MODE>
__global__ void tile_move_kernel(const void *src, void *dst, uint32_t width, uint32_t height, uint32_t tiles_x,
                                 uint32_t tile_rank, uint32_t tile_world, uint32_t n_local) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // slab slot: local tile * 64 + pixel slot
    if (idx >= (size_t)n_local * 64u) return;
    uint32_t i = (uint32_t)(idx >> 6), q = (uint32_t)(idx & 63u);
    uint32_t tile = tile_rank + i * tile_world;
    uint32_t px = (tile % tiles_x) * TILE + (q & 7u), py = (tile / tiles_x) * TILE + (q >> 3);
    if (px >= width || py >= height) return;
    size_t up = ((size_t)py * width + px) * 3, down = ((size_t)(height - 1 - py) * width + px) * 3, sl = idx * 3;
    const float *sf = reinterpret_cast<const float *>(src); const uint8_t *sb = reinterpret_cast<const uint8_t *>(src);
    float *df = reinterpret_cast<float *>(dst); uint8_t *db = reinterpret_cast<uint8_t *>(dst);
    for (int c = 0; c < 3; c++) {
        if (MODE == TM_PACK_F32) df[sl + c] = sf[up + c];
        else if (MODE == TM_PACK_U8) db[sl + c] = to_color_u8(sf[up + c]);
        else if (MODE == TM_UNPACK_F32) df[up + c] = sf[sl + c];
        else if (MODE == TM_UNPACK_U8) db[down + c] = sb[sl + c];
        else if (MODE == TM_CONVERT_U8) db[down + c] = to_color_u8(sf[up + c]);
        else if (MODE == TM_ZERO_F32) df[up + c] = 0.0f;
        else db[down + c] = 0;
    }
}

// device math probe (tests: GPU transcendental/draw functions are bit-identical to the host's)
__global__ void math_probe_kernel(int op, const float *a, const float *b, float *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.0f;
    switch (op) {
        case 0: r = vk::sinf_(a[i]); break;
        case 1: r = vk::cosf_(a[i]); break;
        case 2: r = vk::logf_(a[i]); break;
        case 3: r = vk::asinf_(a[i]); break;
        case 4: r = vk::atan2f_(a[i], b[i]); break;
        case 5: r = vk::pow5f_(a[i]); break;
        case 6: r = a[i] / b[i]; break;
        case 7: r = sqrtf(a[i]); break;
        case 8: { vk::Rng g = vk::rng_for_sample(__float_as_uint(a[i]), (uint32_t)i, 0); r = vk::gen_range(g, -1.0f, 1.0f) + vk::gen_f32(g); break; }
        case 9: r = a[i] * b[i] + a[i]; break;   // must stay an unfused mul+add
    }
    out[i] = r;
}

}  // namespace
#endif
