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
// 8: 4 600 / 3 765 (round 1).  (Deferring only against BOX, not PRIM: 4 150.)  It is a launch parameter (VK_SHADE_DEFER);
// the 1 M-sphere scene prefers the same 4 once its NaN rays no longer walk the whole tree (1 / 2 / 4 -> 573 / 588 / 593).
#ifndef VK_SHADE_DEFER
#define VK_SHADE_DEFER 4
#endif
constexpr uint32_t SHADE_DEFER = VK_SHADE_DEFER;
constexpr int BOX_UNROLL = VK_BOX_UNROLL;   // box steps between two exit tests of the BOX loop

struct KArgs {
    DScene S;
    RenderConsts C;
    float *out;              // full framebuffer (width*height*3)
    long long *accum;        // [width*height*3] fixed-point pixel sums (see to_fixed); null in the probe launch
    float4 *debug;           // optional per-sample (rgb, draws) dump
    uint32_t *counter;       // work-unit counter
    unsigned long long *clamped;   // number of samples whose radiance was clamped on its way into the fixed-point sums (see to_fixed)
    float accum_clamp;       // min(1e10, 1.3e11 / spp), worked out by the host (accum_clamp_for)
    uint32_t *launch_units;  // [2] units pulled by the 1024-thread / the other launch (the dual launch's self-check, vk_api.hip)
    uint32_t *tile_cost;     // [tiles of the image] time spent on each tile (1.6 us ticks): written by the probe (COST) build only
    const uint32_t *tile_order;   // [n_local_tiles] local tile slots, dearest first (from the probe launch), or null = raster order
    uint32_t tiles_x, tiles_y;
    uint32_t n_local_tiles;  // tiles of this call's partition
    uint32_t tile_rank, tile_world;
    uint32_t n_chunks;
    uint32_t shade_defer;    // SHADE + REFILL runs when its lanes outnumber box and primitive lanes this many times (see SHADE_DEFER)
    uint32_t prim_weight;    // pending primitive tests run when prim_weight x their lanes outnumber the box lanes
    uint32_t lds_items, lds_spheres, lds_boxes;   // record counts staged into LDS (LDS variant)
    unsigned long long *phase_stats;   // optional (diagnostic build of the kernel): 24 counters, see vk_debug_phase_stats
    // Exact re-treeing (vk_trace.h): samples dropped by the first launch (the winner of one of their segments may depend on the visiting
    // order) are queued here, REDO_REGIONS queues of redo_region_cap entries {x | y << 16, sample}, one counter per region (64 bytes
    // apart); a workgroup appends to the region of its block index.  The second launch (list_mode = 1, S = the scene as handed over)
    // takes its units from these queues instead of from the tiles: unit u = entries [n k, n (k + 1)) of region u % REDO_REGIONS,
    // k = u / REDO_REGIONS, for k below the slice count redo_plan_kernel leaves in redo_plan[0] and n = redo_plan[3] entries per unit.
    // redo_list == null: nothing is dropped
    // (the probe launch, scenes without a rebuilt tree).
    // list_mode = 2: the FALLBACK launch behind the second one — an ordinary launch over the tiles on the scene as handed over, which runs
    // only if redo_plan[2] != 0, i.e. if a queue overflowed and the frame of the first two launches is incomplete (redo_reset_kernel has
    // zeroed the sums by then); otherwise every workgroup returns at once.
    uint2 *redo_list; uint32_t *redo_count; const uint32_t *redo_plan; uint32_t redo_region_cap; uint32_t list_mode;
    // diagnostic builds (-DVK_WAVE_TIMES, with the environment's VK_WAVE_TIMES=1): per wave {start, last unit pull, end}, 100 MHz ticks
    unsigned long long *wave_times;
};
constexpr uint32_t REDO_REGIONS = 512u;
constexpr uint32_t REDO_COUNT_STRIDE = 16u;        // uint32 words between two regions' counters
// most queue entries per work unit of the second launch (redo_plan_kernel picks 64..this)
constexpr uint32_t REDO_UNIT = 256u;

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
    // the traversal cursor counts BYTES of these two arrays (16 per item; see GlobalMem::ISHIFT): the skip links of the staged
    // items are scaled to match when a workgroup copies them in (render_kernel)
    static constexpr uint32_t ISHIFT = 4;
    static constexpr bool FUSED_BOX = true;      // see GlobalMem
    uint32_t items_hi_off;   // byte offset of items_hi in the workgroup's LDS
    __device__ __forceinline__ DItem item(uint32_t off) const {
        // Absolute LDS addresses: the staged scene starts at LDS address 0 (the kernels have no static LDS: pinned by
        // tests/test_kernel_resources.py), so the cursor IS the address of the first half.  Through `smem` the compiler adds
        // the array's link-time address (0) with a VALU instruction per read.
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef const u32x4 __attribute__((address_space(3))) *lds_u4;
        u32x4 a = *(lds_u4)(uintptr_t)off;
        u32x4 b = *(lds_u4)(uintptr_t)(off + items_hi_off);
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
    // the grid form (DGrid): the staged table [cells | refs] takes the items' place at LDS address 0
    __device__ __forceinline__ uint32_t grid_cell(const DScene &, uint32_t c) const {
        typedef const uint32_t __attribute__((address_space(3))) *lds_u;
        return *(lds_u)(uintptr_t)(c << 2);
    }
    __device__ __forceinline__ uint32_t grid_ref(const DScene &S, uint32_t k) const {
        typedef const uint32_t __attribute__((address_space(3))) *lds_u;
        return *(lds_u)(uintptr_t)((S.grid.nu * S.grid.nv + 1u + k) << 2);
    }
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
        M.items = smem; M.items_hi = smem + lds_items; M.items_hi_off = lds_items << 4;
        M.spheres = reinterpret_cast<const float4 *>(smem + 2u * lds_items);
        M.boxes = smem + 2u * lds_items + S.n_spheres; M.sphere_mat = S.sphere_mat;
    } else {
        M.items = S.items; M.spheres = S.spheres; M.sphere_mat = S.sphere_mat; M.boxes = S.boxes;
    }
    return M;
}

// Cold per-lane path state lives in LDS between SHADE phases, SoA by field (word f of lane l at cold[f*64 + l]:
// conflict-free), so that the box/primitive loops keep only the traversal state in VGPRs: throughput (3), depth, RNG key (2)
// + counter, the sample's pixel (x | y << 16) and sample index (+ the world-space ray (o3 d3) for scenes with instances).
// Radiance is not kept: no material both emits and scatters (material.rs:30-40,215-225), so it is 0 — or its NaN is implied
// by a non-finite throughput — until the path ends.
enum : int { CF_THR = 0, CF_DEPTH = 3, CF_KEY = 4, CF_CTR = 6, CF_XY = 7, CF_SAMPLE = 8, CF_WORLD_RAY = 9 };
constexpr int NCOLD_BASE = 9;
constexpr int NCOLD_INST = 15;
template <uint32_t F> constexpr int ncold() { return (F & VKF_INSTANCE) ? NCOLD_INST : NCOLD_BASE; }
constexpr int WAVE_STATE_WORDS = 8;   // per-wave, wave-uniform: the unit whose samples are being handed out (see render_kernel)
// one 16-byte record: tile origin (x | y << 16), first sample, items, items handed out
enum : int { WS_TXY = 0, WS_S0 = 1, WS_TOTAL = 2, WS_NEXT = 3,
              WS_KARGS = 4 };  // + the kernel-argument segment's address (2 words), for code that is called (shade_refill_call)

// per-wave LDS block, contiguous: [cold lane state][tile sums: 64 x 3 x u64][wave state]
template <uint32_t F> constexpr uint32_t wave_block_floats() { return 64u * (uint32_t)ncold<F>() + 64u * 3u * 2u +
    (uint32_t)WAVE_STATE_WORDS; }

// Pixel sums are ORDER-INDEPENDENT: every finished sample is added to its pixel's three 64-bit fixed-point accumulators
// (2^-26 units: 1.5e-8 absolute per sample, sums up to 1.4e11) with integer atomics, so the image does not depend on which
// lane, wave, work unit, tile partition or GPU traced a sample, nor on the order they finished in — without any lane ever
// waiting for another one's path.  (The reference's `c += color` in f32, main.rs:193, is re-associated anyway: it never
// feeds control flow.)  The sums SATURATE instead of wrapping: a sample's components are clamped to +-min(1e10, 1.3e11 / spp), so
// that spp of them stay below 2^63 * 2^-26 = 1.37e11, and every clamped sample is counted (vk_stats.clamped_samples): the reference
// adds such a sample in f32 (main.rs:193) and the caller can tell that this frame deviates from it.
constexpr float ACCUM_SCALE = 67108864.0f;          // 2^26
constexpr float ACCUM_CLAMP = 1.0e10f;
constexpr float ACCUM_RANGE = 1.3e11f;
inline float accum_clamp_for(uint32_t spp) { float c = ACCUM_RANGE / (float)spp; return c < ACCUM_CLAMP ? c : ACCUM_CLAMP; }     // (host)
// Components below 32 in magnitude — every sample of a scene without bright emitters, and nearly every one otherwise — fit 31 bits
// at 2^-26: ONE v_cvt_i32_f32 (truncating, like the 64-bit cast) and a sign extension, against ~17 instructions for the float ->
// 64-bit integer conversion the compiler has to expand.
constexpr float ACCUM_SMALL = 31.999f;
__device__ __forceinline__ long long to_fixed_small(float v) { return (long long)(int)(v * ACCUM_SCALE); }
__device__ __forceinline__ long long to_fixed(float v, float clampv) {
    v = fminf(fmaxf(v, -clampv), clampv);
    return (long long)(v * ACCUM_SCALE);             // scaling by a power of two is exact; the cast truncates toward zero
}

template <uint32_t F>
__device__ __forceinline__ void cold_store_path(float *c, uint32_t lane, const Lane &L) {
    c[(CF_THR + 0) * 64 + lane] = L.thr.x; c[(CF_THR + 1) * 64 + lane] = L.thr.y; c[(CF_THR + 2) * 64 + lane] = L.thr.z;
    c[CF_DEPTH * 64 + lane] = __uint_as_float(L.depth);
    c[CF_KEY * 64 + lane] = __uint_as_float((uint32_t)L.rng.key);
    c[(CF_KEY + 1) * 64 + lane] = __uint_as_float((uint32_t)(L.rng.key >> 32));
    c[CF_CTR * 64 + lane] = __uint_as_float(L.rng.ctr);
    if (F & VKF_INSTANCE) {
        c[(CF_WORLD_RAY + 0) * 64 + lane] = L.wo.x; c[(CF_WORLD_RAY + 1) * 64 + lane] = L.wo.y; c[(CF_WORLD_RAY + 2) * 64 + lane] = L.wo.z;
        c[(CF_WORLD_RAY + 3) * 64 + lane] = L.wd.x; c[(CF_WORLD_RAY + 4) * 64 + lane] = L.wd.y; c[(CF_WORLD_RAY + 5) * 64 + lane] = L.wd.z;
    }
}
template <uint32_t F>
__device__ __forceinline__ void cold_load_world_ray(const float *c, uint32_t lane, Lane &L) {
    if (F & VKF_INSTANCE) {
        L.wo = v3(c[(CF_WORLD_RAY + 0) * 64 + lane], c[(CF_WORLD_RAY + 1) * 64 + lane], c[(CF_WORLD_RAY + 2) * 64 + lane]);
        L.wd = v3(c[(CF_WORLD_RAY + 3) * 64 + lane], c[(CF_WORLD_RAY + 4) * 64 + lane], c[(CF_WORLD_RAY + 5) * 64 + lane]);
    }
}
template <uint32_t F>
__device__ __forceinline__ void cold_load_path(const float *c, uint32_t lane, Lane &L) {
    L.thr = v3(c[(CF_THR + 0) * 64 + lane], c[(CF_THR + 1) * 64 + lane], c[(CF_THR + 2) * 64 + lane]);
    L.acc = v3s(0.0f);
    L.depth = __float_as_uint(c[CF_DEPTH * 64 + lane]) & 0x7FFFFFFFu;      // (bit 31: shade_refill_body's rearm mark)
    L.rng.key = (uint64_t)__float_as_uint(c[CF_KEY * 64 + lane]) | ((uint64_t)__float_as_uint(c[(CF_KEY + 1) * 64 + lane]) << 32);
    L.rng.ctr = __float_as_uint(c[CF_CTR * 64 + lane]);
    L.pixel = 0; L.sample = 0;
    if (F & VKF_INSTANCE) cold_load_world_ray<F>(c, lane, L);
    else { L.wo = L.o; L.wd = L.d; }       // no instances: the current space IS world space
}

// adds the wave's LDS sums of tile `txy` (x | y << 16 of its origin; lane = pixel slot) to the frame's accumulators and clears them
__device__ __forceinline__ void flush_tile_sums(unsigned long long *tile_sum, long long *accum, uint32_t txy, uint32_t lane,
    uint32_t width, uint32_t height) {
    uint32_t px = (txy & 0xFFFFu) + (lane & 7u), py = (txy >> 16) + (lane >> 3);
    if (!accum || txy == 0xFFFFFFFFu || px >= width || py >= height) return;
    unsigned long long *a = reinterpret_cast<unsigned long long *>(accum) + ((size_t)py * width + px) * 3;
    for (int c = 0; c < 3; c++) {
        unsigned long long v = tile_sum[lane * 3 + c];
        if (v) { atomicAdd(a + c, v); tile_sum[lane * 3 + c] = 0ull; }
    }
}

// number of lanes of the wave for which p holds (v_cmp -> s_bcnt1, no VGPR round trip)
// lane mask of prim_is_heavy<F>(ref), straight from the compare
template <uint32_t F>
__device__ __forceinline__ unsigned long long heavy_mask(uint32_t ref) {
    if constexpr ((F & VKF_ALL_SCENE) == VKF_ALL_SCENE) return __builtin_amdgcn_uicmp(ref - ((uint32_t)DK_LIST << 28), 3u << 28,
        36 /* ult */);
    else return __builtin_amdgcn_uicmp(ref, (uint32_t)DK_LIST << 28, 35 /* uge */);
}
__device__ __forceinline__ uint32_t lanes_with(bool p) { return (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(p)); }

// ---- Perlin turbulence, the octaves of one lane spread over the wave.  perlin_turb is seven octaves of an 8-corner noise (56
// dependent gathers) and, in a scene like the final one, is needed by one or two of a shading phase's 64 lanes in three phases out of
// four: run inside the material code it keeps the whole wave for seven serial noise evaluations.  The octaves are independent until
// the final ordered sum, so here — at a point of the phase where every lane of the wave is present — the lanes that WILL evaluate a
// noise texture are found ahead of the material code (a sphere hit outside any instance: its point is o + d*T; the lineariser lists
// the few spheres with a noise material) and, nine of them per round, 63 helper lanes each evaluate one octave of one of them; points
// and values travel by ds_bpermute and each served lane sums its seven values in the reference's order (accum += weight * noise,
// material.rs:379-390: scaling the point and the weight by powers of two is exact, so octave j alone computes what iteration j of the
// loop does).  texture_value uses the value only for the texture and point it was made for and runs the loop itself otherwise (lists,
// instanced objects, SpecDiffuse picks).  C3: 763 -> 799 Msamples/s (with ONE octave instead of seven, i.e. no turbulence cost left
// to remove, it would be 809).
template <uint32_t F, class Mem>
__device__ __forceinline__ PreTurb cooperative_turb(const Lane &L, const DScene &S, const Mem &M, bool is_shade, uint32_t lane) {
    PreTurb pt = no_pre_turb();
    if constexpr ((F & VKF_TEXTURES) != 0u) {
        if (!(S.features & VKF_NOISE)) return pt;          // (wave-uniform) no noise texture in this scene: nothing to prepare
        uint32_t perlin = 0u;
        if (is_shade && L.best_prim != 0u && VKD_KIND(L.best_prim) == DK_SPHERE && (!(F & VKF_INSTANCE) || L.best_inst < 0)) {
            const uint32_t idx = VKD_INDEX(L.best_prim);
            if (S.n_noise_spheres != 0xFFFFFFFFu) {        // the scene's (few) spheres with a noise material, from the lineariser
                for (uint32_t k = 0; k < 4u; k++)
                    if (k < S.n_noise_spheres && idx == S.noise_sphere[k]) { pt.tex = S.noise_tex[k]; perlin = S.noise_perlin[k]; }
            } else {
                const DMaterial &m = S.materials[M.smat(idx)];
                if (m.tex_kind == VK_TEX_NOISE) { pt.tex = m.tex; perlin = S.textures[m.tex].a; }
            }
            if (pt.tex != 0xFFFFFFFFu) {
                V3 p = L.o + L.d * L.T;                    // simple_record's R.p for a sphere
                pt.px = p.x; pt.py = p.y; pt.pz = p.z;
            }
        }
        unsigned long long todo = __builtin_amdgcn_uicmp(pt.tex, 0xFFFFFFFFu, 33 /* ne */);
        if (todo != 0ull) {                                // wave-uniform: every lane of the wave is here
            // Up to nine lanes' turbulences per round: helper lane h < 63 evaluates octave h % 7 for the (h / 7)-th of them, so that
            // the noise evaluation — two dependent rounds of gathers — is paid once per round, not once per lane that needs one
            const uint32_t my_rank = (uint32_t)__builtin_popcountll(todo & ((1ull << lane) - 1ull));   // of a lane that needs one
            const uint32_t my_slot = lane / 7u, my_octave = lane - 7u * my_slot;
            uint32_t base = 0u;
            while (todo != 0ull) {
                uint32_t my_src = 0u; bool helper = false;
                for (uint32_t sl = 0; sl < 9u; sl++) {
                    if (todo == 0ull) break;               // (uniform)
                    const uint32_t src = (uint32_t)__builtin_ctzll(todo);
                    todo &= todo - 1ull;
                    if (my_slot == sl) { my_src = src; helper = true; }
                }
                const int sa = (int)(my_src << 2);
                const float sx = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)__float_as_uint(pt.px)));
                const float sy = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)__float_as_uint(pt.py)));
                const float sz = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)__float_as_uint(pt.pz)));
                const uint32_t sp = (uint32_t)__builtin_amdgcn_ds_bpermute(sa, (int)perlin);
                float n = 0.0f;
                if (helper) {
                    const float sc = (float)(1u << my_octave); // the point of this octave: p * 2^octave, exactly what the doublings give
                    n = perlin_noise(S.perlins[sp], v3(sx * sc, sy * sc, sz * sc));
                }
                // the lanes served this round sum their seven values in the reference's order
                const uint32_t slot = my_rank - base;      // (meaningful for lanes with a pending request of this round only)
                float accum = 0.0f, weight = 1.0f;
                for (uint32_t j = 0; j < 7u; j++) {
                    const int ga = (int)(((slot < 9u ? slot : 0u) * 7u + j) << 2);
                    accum += weight * __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(ga, (int)__float_as_uint(n)));
                    weight *= 0.5f;
                }
                // (my_rank >= base for lanes not served yet; < base wraps to huge)
                if (pt.tex != 0xFFFFFFFFu && slot < 9u) pt.val = fabsf(accum);
                base += 9u;
            }
        }
    }
    return pt;
}

// ---- random_in_unit_sphere for the lanes about to scatter off a Metal (or an Isotropic), drawn by the WHOLE wave.  The rejection loop
// (util.rs:31-40) accepts 52 % of its candidates: run lane by lane, a wave with ten such lanes iterates four times on average, and
// removing the loop altogether would make the InOneWeekend scene 7.7 % faster (profiles/r05/experiments).  The generator is
// counter-based — candidate m of a lane's loop is draws ctr + 3m + 1 .. 3 of its stream, whoever computes them — so the 64 lanes are
// dealt evenly to the K lanes that need a point: lane h draws candidate h mod G of the lane of rank h / G, G = 64 / K, and the
// owner takes the first one inside the sphere (the loop's own choice).  One round nearly always does (0.476^6 = 1 %); the rest go round
// again with the wave regrouped.
__device__ __forceinline__ PreBall cooperative_ball(bool want, const Rng &rng, uint32_t lane) {
    PreBall out = no_pre_ball();
    uint32_t m0 = 0u;                                    // candidates of this lane's loop already refused
    unsigned long long need = __builtin_amdgcn_ballot_w64(want);
    while (need != 0ull) {                               // (wave-uniform)
        const uint32_t K = (uint32_t)__popcll(need), G = 64u / K;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
        // table rank -> lane (a push: lanes that need nothing write entry 63, which is an owner's only when all 64 are)
        const int tbl = __builtin_amdgcn_ds_permute((int)((want ? rank : 63u) << 2), (int)lane);
        const uint32_t j = (lane * ((65536u + G - 1u) / G)) >> 16;      // lane / G (exact below 64)
        const uint32_t c = lane - j * G;
        const int owner = __builtin_amdgcn_ds_bpermute((int)((j < 63u ? j : 63u) << 2), tbl) << 2;
        const uint32_t klo = (uint32_t)__builtin_amdgcn_ds_bpermute(owner, (int)(uint32_t)rng.key);
        const uint32_t khi = (uint32_t)__builtin_amdgcn_ds_bpermute(owner, (int)(uint32_t)(rng.key >> 32));
        const uint32_t cb = (uint32_t)__builtin_amdgcn_ds_bpermute(owner, (int)(rng.ctr + 3u * m0));
        bool inside;
        const V3 p = ball_candidate((uint64_t)klo | ((uint64_t)khi << 32), cb + 3u * c, inside);
        const unsigned long long fl = __builtin_amdgcn_ballot_w64(inside && j < K);
        // the owner's helpers are lanes rank * G .. rank * G + G - 1
        const uint32_t lo = rank * G;
        const unsigned long long grp = want ? ((fl >> (lo & 63u)) & (G == 64u ? ~0ull : ((1ull << G) - 1ull))) : 0ull;
        const uint32_t first = grp ? (uint32_t)__builtin_ctzll(grp) : 0u;
        const int src = (int)(((lo + first) & 63u) << 2);
        const float x = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(p.x)));
        const float y = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(p.y)));
        const float z = __uint_as_float((uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)__float_as_uint(p.z)));
        if (want) {
            if (grp) { want = false; out.have = 1u; out.p = v3(x, y, z); out.skip = 3u * (m0 + first + 1u); }
            else m0 += G;
        }
        need = __builtin_amdgcn_ballot_w64(want);
    }
    return out;
}

// ---- the SHADE + REFILL phase body: shade the lanes whose segment is fully traversed, deposit finished samples, hand new
// samples to the lanes without a path.  Leaves `fresh` lanes with a new ray parked in L.wo / L.wd / L.time, which the caller
// installs with ONE begin_segment (its three exact reciprocals are ~60 instructions per call site). Used inline by the lean variants and
// through shade_refill_call (below) by the everything-variants.
struct PhaseClocks { unsigned long long mat = 0, refill = 0, t1 = 0, turb = 0, cold = 0; };
template <uint32_t F, bool LDS_SCENE, bool STATS, bool COST>
__device__ __forceinline__ void shade_refill_body(Lane &L, bool is_shade, bool early, bool &active, bool &need, bool &fresh, bool &touched,
                                                  bool &rearm,
                                                  uint32_t &cost_t0, KArgsC P, float *cold, unsigned long long *tile_sum,
                                                  uint32_t *wstate, uint32_t lane, uint32_t lds_items, PhaseClocks &clk) {
    using Mem = typename std::conditional<LDS_SCENE, LdsMem, GlobalMem>::type;
    RenderConsts C = KARG(P, C);
    DScene S = KARG(P, S);
    Mem M = make_mem<F, LDS_SCENE>(S, lds_items);
    unsigned long long &st_t_mat = clk.mat, &st_t_refill = clk.refill, &st_t1 = clk.t1;
    (void)st_t_mat; (void)st_t_refill; (void)st_t1; (void)cost_t0;
    fresh = false;                // lanes that leave this phase with a new ray to install; it is parked in the
                                  // (dead) world-ray fields L.wo / L.wd / L.time, so it costs no registers
    if (STATS) st_t1 = clock64();
    const PreTurb pre_turb = cooperative_turb<F, Mem>(L, S, M, is_shade, lane);
    if (STATS) clk.turb += clock64() - st_t1;
    rearm = false;
    // `early` (exact re-treeing): the winner of this lane's segment may depend on the visiting order (vk_trace.h segment_unsafe,
    // asked by the caller, which holds the segment's reciprocals)
    if constexpr ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u && !LDS_SCENE) {
        // scene in global memory (both trees in items[], DScene::walk_start): the segment is walked again, now on the tree as handed
        // over — the caller re-installs the same ray for that; bit 31 of the lane's depth word says so until the segment is shaded
        if (S.walk_start != 0u && is_shade && early) {
            rearm = true; is_shade = false;
            L.wo = L.o; L.wd = L.d;      // (L.time is the segment's)
            cold[CF_DEPTH * 64 + lane] = __uint_as_float(__float_as_uint(cold[CF_DEPTH * 64 + lane]) | 0x80000000u);
        }
    }
    if constexpr ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u) {
        // scene in LDS (the rebuilt tree only) — and the grid form wherever it is walked from: the sample of such a segment is dropped
        // here (a queue was handed in: redo_list) and rendered by the second launch on the
        // tree as handed over.  One counter update per wave and phase, prefix sums over the dropping lanes.
        uint2 *rl = KARG(P, redo_list);
        const unsigned long long m_drop = __builtin_amdgcn_ballot_w64(is_shade && early && rl != nullptr);
        if (m_drop != 0ull) {
            // (dual launch: 0..255 | 256..511)
            const uint32_t region = (blockIdx.x + (blockDim.x == 1024u ? 0u : gridDim.x)) & (REDO_REGIONS - 1u);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(KARG(P, redo_count) + region * REDO_COUNT_STRIDE, (uint32_t)__popcll(m_drop));
            base = __builtin_amdgcn_readfirstlane(base);
            if ((m_drop >> lane) & 1ull) {
                const uint32_t slot = base + (uint32_t)__popcll(m_drop & ((1ull << lane) - 1ull));
                const uint32_t cap = KARG(P, redo_region_cap);
                // (a full region: the counter keeps counting and the host sees the overflow, vk_api.hip)
                if (slot < cap) rl[(size_t)region * cap + slot] = make_uint2(__float_as_uint(cold[CF_XY * 64 + lane]),
                    __float_as_uint(cold[CF_SAMPLE * 64 + lane]));
                is_shade = false; active = false; need = true;
            }
        }
    }
    touched = is_shade;
    PreBall pre_ball = no_pre_ball();
    if constexpr (F == 0u && LDS_SCENE) {
        // (the sphere-only scatter variant staged in LDS — the headline's: the material is one gather away, shade_core reads the same
        // record; before the path's other cold state is loaded: the traversal state of the lanes that are not shading stays live
        // through this phase, registers are short.  Its PDF twin and the global-memory variants would spill more than they gain.)
        bool want = false;
        Rng g; g.key = 0ull; g.ctr = 0u;
        if (is_shade && L.best_prim != 0u) {
            const uint32_t kind = S.sphere_material[VKD_INDEX(L.best_prim)].kind;
            want = kind == VK_MAT_METAL || kind == VK_MAT_ISOTROPIC;
        }
        if (want) {
            g.key = (uint64_t)__float_as_uint(cold[CF_KEY * 64 + lane]) | ((uint64_t)__float_as_uint(cold[(CF_KEY + 1) * 64 + lane]) << 32);
            g.ctr = __float_as_uint(cold[CF_CTR * 64 + lane]);
        }
        pre_ball = cooperative_ball(want, g, lane);
    }
    if (is_shade) {
        if (STATS) st_t1 = clock64();
        cold_load_path<F>(cold, lane, L);
        if (STATS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); clk.cold += clock64() - st_t1; }
    }
    if (is_shade) {
        if (STATS) st_t1 = clock64();
        V3 no, nd; float nt;
        bool cont = shade_core<F, Mem>(L, S, M, C, no, nd, nt, pre_turb, pre_ball);
        if (cont) { L.wo = no; L.wd = nd; L.time = nt; fresh = true; }
        if (STATS) st_t_mat += clock64() - st_t1;
        if (!cont) {
            uint32_t xy = __float_as_uint(cold[CF_XY * 64 + lane]);
            size_t pix = (size_t)(xy >> 16) * C.width + (xy & 0xFFFFu);
            float4 *dbg = KARG(P, debug);
            if (dbg) dbg[pix * C.spp + __float_as_uint(cold[CF_SAMPLE * 64 + lane])] = make_float4(L.acc.x, L.acc.y, L.acc.z,
                __uint_as_float(L.rng.ctr));
            long long *acc = KARG(P, accum);
            if (acc && isfinite(L.acc.x) && isfinite(L.acc.y) && isfinite(L.acc.z)) {   // main.rs:192-194; c += color (main.rs:193)
                const float clampv = KARG(P, accum_clamp);
                const float big = fmaxf(fmaxf(fabsf(L.acc.x), fabsf(L.acc.y)), fabsf(L.acc.z));
                unsigned long long fx, fy, fz;
                if (big <= ACCUM_SMALL) {      // (nearly always, for the whole wave)
                    fx = (unsigned long long)to_fixed_small(L.acc.x); fy = (unsigned long long)to_fixed_small(L.acc.y);
                    fz = (unsigned long long)to_fixed_small(L.acc.z);
                } else {
                    if (big > clampv) atomicAdd(KARG(P, clamped), 1ull);     // (rare)
                    fx = (unsigned long long)to_fixed(L.acc.x, clampv); fy = (unsigned long long)to_fixed(L.acc.y, clampv);
                    fz = (unsigned long long)to_fixed(L.acc.z, clampv);
                }
                // a sample of the tile the wave is handing out (nearly all of them) lands in the wave's LDS sums, which
                // reach the frame's accumulators once per unit; a straggler of an earlier unit goes there directly
                if ((xy & 0xFFF8FFF8u) == __builtin_amdgcn_readfirstlane(wstate[WS_TXY])) {
                    unsigned long long *t = tile_sum + ((xy & 7u) | ((xy >> 13) & 0x38u)) * 3u;
                    atomicAdd(t + 0, fx); atomicAdd(t + 1, fy); atomicAdd(t + 2, fz);
                } else {
                    unsigned long long *a = reinterpret_cast<unsigned long long *>(acc) + pix * 3;
                    atomicAdd(a + 0, fx); atomicAdd(a + 1, fy); atomicAdd(a + 2, fz);
                }
            }
            if (COST) {   // probe launch: the lane-time this sample took, charged to its tile
                uint32_t *tc = KARG(P, tile_cost);
                uint32_t now = (uint32_t)(wall_clock64() >> 4);
                if (tc) atomicAdd(&tc[((xy >> 16) / TILE) * KARG(P, tiles_x) + (xy & 0xFFFFu) / TILE], now - cost_t0);
            }
            active = false;
            need = true;
        }
    }
    if (STATS) st_t1 = clock64();
    // ---- hand out samples of the wave's current unit to the lanes without a path; pull the next unit when it is used up
    for (;;) {
        unsigned long long need_mask = __builtin_amdgcn_ballot_w64(need);
        if (!need_mask) break;
        uint4 ws = *reinterpret_cast<const uint4 *>(wstate);       // one ds_read_b128, the same address in every lane
        uint32_t txy = __builtin_amdgcn_readfirstlane(ws.x), s0 = __builtin_amdgcn_readfirstlane(ws.y);
        uint32_t total = __builtin_amdgcn_readfirstlane(ws.z), next = __builtin_amdgcn_readfirstlane(ws.w);
        // the second launch of exact re-treeing (sphere-only variants): units are slices of the redo queues
        const bool list_mode = ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u) && KARG(P, list_mode) == 1u;
        if (next >= total) {
            uint32_t unit = 0;
            if (lane == 0) {
                unit = atomicAdd(KARG(P, counter), 1u);
                // (once per 2 048 samples or more; the dual launch's self-check counts the first launch's units only)
                if (KARG(P, list_mode) == 0u) atomicAdd(KARG(P, launch_units) + (blockDim.x == 1024u ? 0 : 1), 1u);
            }
            unit = __builtin_amdgcn_readfirstlane(unit);
            if (list_mode) {
                const uint32_t cap = KARG(P, redo_region_cap);
                const uint32_t *plan = KARG(P, redo_plan);
                const uint32_t usize = plan[3];                 // entries per unit (redo_plan_kernel)
                if (unit >= REDO_REGIONS * plan[0]) { need = false; break; }
                flush_tile_sums(tile_sum, KARG(P, accum), txy, lane, C.width, C.height);      // (nothing after the first unit)
                const uint32_t region = unit % REDO_REGIONS, first = (unit / REDO_REGIONS) * usize;
                uint32_t cnt = KARG(P, redo_count)[region * REDO_COUNT_STRIDE];
                cnt = __builtin_amdgcn_readfirstlane(cnt < cap ? cnt : cap);
                txy = 0xFFFFFFFEu;                              // no tile: every sample goes to the frame's sums directly
                s0 = region * cap + first;                      // (here: the unit's first queue entry)
                next = 0u; total = first < cnt ? (cnt - first < usize ? cnt - first : usize) : 0u;
                if (lane == 0) { wstate[WS_TXY] = txy; wstate[WS_S0] = s0; wstate[WS_TOTAL] = total; wstate[WS_NEXT] = 0u; }
                if (total == 0u) continue;                      // an empty slice: the next unit
            } else {
            const uint32_t n_chunks = KARG(P, n_chunks);
#ifdef VK_WAVE_TIMES
            { unsigned long long *wt = KARG(P, wave_times);
              if (wt && lane == 0) wt[3u * ((blockIdx.x + (blockDim.x == 1024u ? 0u : gridDim.x)) * 16u + (threadIdx.x >> 6)) + 1u] = wall_clock64(); }
#endif
            if (unit >= KARG(P, n_local_tiles) * n_chunks) {                      // the launch's units are all handed out:
                need = false;                                                     // these lanes idle until the wave's last path ends
                break;
            }
            flush_tile_sums(tile_sum, KARG(P, accum), txy, lane, C.width, C.height);  // the finished unit's sums so far
            const uint32_t chunk = unit % n_chunks;
            // tiles are visited dearest-first when the probe launch left an order (see enqueue_render)
            uint32_t tslot = unit / n_chunks;
            { const uint32_t *ord = KARG(P, tile_order); if (ord) tslot = ord[tslot]; }
            const uint32_t tile = KARG(P, tile_rank) + tslot * KARG(P, tile_world);
            const uint32_t tiles_x = KARG(P, tiles_x);
            s0 = (uint32_t)(((uint64_t)C.spp * chunk) / n_chunks);
            const uint32_t s1 = (uint32_t)(((uint64_t)C.spp * (chunk + 1)) / n_chunks);
            txy = ((tile % tiles_x) * TILE) | (((tile / tiles_x) * TILE) << 16);
            next = 0u; total = 64u * (s1 - s0);
            if (lane == 0) { wstate[WS_TXY] = txy; wstate[WS_S0] = s0; wstate[WS_TOTAL] = total; }
            }
        }
        uint32_t k = next + (uint32_t)__popcll(need_mask & ((1ull << lane) - 1ull));
        if (need && k < total) {
            uint32_t q = k & 63u, smp = s0 + (k >> 6);
            uint32_t px = (txy & 0xFFFFu) + (q & 7u), py = (txy >> 16) + (q >> 3);
            if (list_mode) {
                const uint2 e = KARG(P, redo_list)[(size_t)s0 + k];
                px = e.x & 0xFFFFu; py = e.x >> 16; smp = e.y;
            }
            if (px < C.width && py < C.height) {   // slots outside the image (edge tiles) are skipped: the lane asks again
                cold[CF_XY * 64 + lane] = __uint_as_float(px | (py << 16));
                cold[CF_SAMPLE * 64 + lane] = __uint_as_float(smp);
                if (COST) cost_t0 = (uint32_t)(wall_clock64() >> 4);
                V3 no, nd; float nt;               // the lane's next sample (main.rs:186-190)
                start_sample_core(L, C, px, py, smp, no, nd, nt);
                L.wo = no; L.wd = nd; L.time = nt;
                fresh = true;
                active = true;
                need = false;
                touched = true;
            }
        }
        uint32_t taken = (uint32_t)__popcll(need_mask);
        if (lane == 0) wstate[WS_NEXT] = next + taken < total ? next + taken : total;
    }
    if (STATS) { st_t_refill += clock64() - st_t1; st_t1 = clock64(); }
}

// The everything-variants (media + textures + instances + lists) call the phase out of line: its several hundred live
// values then get their own register allocation instead of squeezing the traversal loops' (left inline, the allocator spilled
// traversal state inside the box and primitive loops as soon as anything in the kernel changed: C3 moved between 330 and 520
// Msamples/s with the spill placement).  Only what shading reads of the traversal state crosses, by value.
struct ShadeIo {
    // in: 1 is_shade, 2 active, 4 need, 64 early (segment_unsafe);   out: 2 active, 4 need, 8 fresh, 16 touched, 32 rearm (same ray
    // again)
    uint32_t flags;
    float T; uint32_t best_prim; int32_t best_inst; float best_aux;
    V3 o, d; float time;       // in: the segment's ray (world ray when the scene has no instances); out: the new ray of fresh lanes
    uint32_t cost_t0;
};
template <uint32_t F, bool LDS_SCENE, bool COST>
__device__ __attribute__((noinline)) ShadeIo shade_refill_call(ShadeIo io, uint32_t lds_items, uint32_t wave_block) {
    // a called function has neither the kernel-argument pointer nor the kernel's LDS pointers: the wave's LDS block comes as its
    // offset in the workgroup's dynamic LDS, and the kernel left the argument segment's address in the wave state
    const uint32_t lane = threadIdx.x & 63u;
    float *cold = reinterpret_cast<float *>(smem) + wave_block;
    unsigned long long *tile_sum = reinterpret_cast<unsigned long long *>(cold + 64 * ncold<F>());
    uint32_t *wstate = reinterpret_cast<uint32_t *>(cold + 64 * ncold<F>() + 64 * 3 * 2);
    KArgsC P;
    {
        uint64_t a = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(wstate[WS_KARGS]) |
                     ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(wstate[WS_KARGS + 1]) << 32);
        P = (KArgsC)a;
        asm volatile("" : "+s"(P));
    }
    Lane L;
    __builtin_memset(&L, 0, sizeof(L));     // (the intrinsic: HIP's device memset() is a loop, which keeps a Lane this large in scratch)
    L.T = io.T; L.best_prim = io.best_prim; L.best_inst = io.best_inst; L.best_aux = io.best_aux;
    L.o = io.o; L.d = io.d; L.time = io.time;
    bool active = (io.flags & 2u) != 0u, need = (io.flags & 4u) != 0u, fresh = false, touched = false, rearm = false;
    uint32_t cost_t0 = io.cost_t0;
    PhaseClocks clk;
    shade_refill_body<F, LDS_SCENE, false, COST>(L, (io.flags & 1u) != 0u, (io.flags & 64u) != 0u, active, need, fresh, touched, rearm,
        cost_t0, P, cold, tile_sum,
        wstate, lane, lds_items, clk);
    // fresh lanes: L.wo / L.wd hold the NEW ray, which is what the world-ray slots want
    if (active && touched) cold_store_path<F>(cold, lane, L);
    ShadeIo out = io;
    out.flags = (active ? 2u : 0u) | (need ? 4u : 0u) | (fresh ? 8u : 0u) | (touched ? 16u : 0u) | (rearm ? 32u : 0u);
    out.o = L.wo; out.d = L.wd; out.time = L.time; out.cost_t0 = cost_t0;
    return out;
}

// Work distribution.  A work UNIT is (8x8 tile, sample chunk), pulled by a WAVE from a global atomic counter and handed
// out to its lanes sample by sample (item k -> pixel slot k & 63, sample s0 + (k >> 6): the 64 primary rays of one sample
// index start together, which keeps the first segments coherent).  A lane whose path ended takes the next item by ballot +
// prefix popcount (active-ray compaction), and the wave pulls the NEXT unit the moment the current one is handed out:
// nobody waits for the slowest path of a unit (the "drain" cost 5 % on C2 and most of the lanes on C5). That is possible because pixel sums
// are order independent (to_fixed above).
// GRID: the sphere-only variants' walk on the grid form of exact re-treeing (DGrid; vk_trace.h grid_step) instead of a tree.  Staged in
// LDS, the table [cells | refs] takes the items' place (lds_items = its size in 32-byte units).  A failed segment requeues its sample for
// the second launch — also from global memory (walking the tree as handed over in place was tried: with a lane or two per wave on that
// tree nearly every step of the wave pays for both walks, the 1 M-sphere scene ran at half the tree forms' rate).
template <uint32_t F, bool LDS_SCENE, int MINW, bool STATS, bool COST = false, bool GRID = false>
__global__ __launch_bounds__((MINW <= 4 ? 1024 : (MINW == 5 ? 640 : (MINW == 6 ? 768 : 1024))), MINW) void render_kernel(KArgs A_byval) {
    // (MINW == 7: the sphere-only LDS variants' dual launch, 1024- and 768-thread workgroups of the same build: vk_api.hip launch_dual)
    (void)A_byval;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    using Mem = typename std::conditional<LDS_SCENE, LdsMem, GlobalMem>::type;
    // diagnostic counters (STATS build only): phase executions and the lanes that had work in them
    unsigned long long st_box_steps = 0, st_box_lanes = 0, st_prim_execs = 0, st_prim_lanes = 0, st_shade_execs = 0, st_shade_lanes = 0,
        st_sched = 0;
    unsigned long long st_t_box = 0, st_t_light = 0, st_t_heavy = 0, st_t_shade = 0, st_heavy_execs = 0, st_t0 = 0, st_t_total = 0,
        st_t_mat = 0, st_t_refill = 0, st_t_install = 0, st_t1 = 0, st_t_turb = 0, st_t_cold = 0;
    if (STATS) st_t_total = clock64();

    if constexpr ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u) {      // the fallback launch of exact re-treeing: nothing to do, nearly always
        KArgsC P = kargs_fresh();
        if (KARG(P, list_mode) == 2u && KARG(P, redo_plan)[2] == 0u) return;
    }
    // ---- LDS layout: [items][spheres][boxes][per wave: cold lane state | tile sums | wave state]
    uint32_t lds_items = 0;
    float *cold;
    unsigned long long *tile_sum;   // the wave's current tile: 64 pixels x 3 fixed-point sums
    uint32_t *wstate;
    uint32_t wave_block;            // offset of this wave's LDS block in the dynamic LDS, in floats
    {
        KArgsC P = kargs_fresh();
        lds_items = LDS_SCENE ? KARG(P, lds_items) : 0u;
        uint32_t lds_spheres = LDS_SCENE ? KARG(P, lds_spheres) : 0u;
        uint32_t lds_boxes = LDS_SCENE ? KARG(P, lds_boxes) : 0u;
        wave_block = 4u * (2u * lds_items + lds_spheres + 2u * lds_boxes) + wave * wave_block_floats<F>();     // in floats from smem
        cold = reinterpret_cast<float *>(smem) + wave_block;
        tile_sum = reinterpret_cast<unsigned long long *>(cold + 64 * ncold<F>());
        wstate = reinterpret_cast<uint32_t *>(cold + 64 * ncold<F>() + 64 * 3 * 2);
        if (lane == 0) {      // for called code: the address of the kernel-argument segment
            uint64_t a = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
            wstate[WS_KARGS] = (uint32_t)a; wstate[WS_KARGS + 1] = (uint32_t)(a >> 32);
        }
        if (LDS_SCENE) {
            if constexpr (GRID) {
                const uint4 *gt = reinterpret_cast<const uint4 *>(KARG(P, S.grid_cells));      // [cells | refs], padded to 32 bytes
                for (uint32_t k = threadIdx.x; k < 2u * lds_items; k += blockDim.x) smem[k] = gt[k];
            } else {
            const uint4 *gi = reinterpret_cast<const uint4 *>(KARG(P, S.items));
            for (uint32_t k = threadIdx.x; k < 2u * lds_items; k += blockDim.x) {
                uint4 h = gi[k];
                if ((k & 1u) && (h.z >> 28) == 0u) h.z <<= LdsMem::ISHIFT;      // an inner item's skip link, in cursor units
                smem[(k >> 1) + ((k & 1u) ? lds_items : 0u)] = h;
            }
            }
            const uint4 *gs = reinterpret_cast<const uint4 *>(KARG(P, S.spheres));
            for (uint32_t k = threadIdx.x; k < lds_spheres; k += blockDim.x) smem[2u * lds_items + k] = gs[k];
            const uint4 *gb = reinterpret_cast<const uint4 *>(KARG(P, S.boxes));
            for (uint32_t k = threadIdx.x; k < 2u * lds_boxes; k += blockDim.x) smem[2u * lds_items + lds_spheres + k] = gb[k];
            __syncthreads();
        }
    }
#ifdef VK_WAVE_TIMES          // (diagnostic builds only, tools/experiments/build_exp.sh -DVK_WAVE_TIMES: the stores cost the production kernels 1 %)
    { KArgsC P = kargs_fresh(); unsigned long long *wt = KARG(P, wave_times);
      if (wt && lane == 0) wt[3u * ((blockIdx.x + (blockDim.x == 1024u ? 0u : gridDim.x)) * 16u + wave)] = wall_clock64(); }
#endif
    // the wave has no unit yet: the first SHADE + REFILL phase pulls one
    tile_sum[lane * 3 + 0] = 0ull; tile_sum[lane * 3 + 1] = 0ull; tile_sum[lane * 3 + 2] = 0ull;
    if (lane == 0) { wstate[WS_TXY] = 0xFFFFFFFFu; wstate[WS_NEXT] = 0u; wstate[WS_TOTAL] = 0u; }

    Lane L;
    __builtin_memset(&L, 0, sizeof(L));     // (the intrinsic: HIP's device memset() is a loop, which keeps a Lane this large in scratch)
    uint32_t cost_t0 = 0;      // probe (COST) build only: when this lane's current sample started
    (void)cost_t0;
    bool need = true;          // lane has no path and wants a sample (false once the launch's units are all handed out)
    bool active = false;       // lane holds a live path
    // Wave-level phase scheduler.  Every lane is in one of four states; each round the
    // wave runs the code of the most populated state with the lanes in it (64-bit ballots
    // + s_bcnt1), so the long box loop, the primitive tests and the (expensive, rare)
    // shading / ray-generation code each execute with as many lanes as possible instead
    // of all being paid for on every iteration:
    //   BOX    pend == 0 and items (or an instance to leave) remain  -> box_step
    //   PRIM   pend != 0                                             -> prim_step
    //   SHADE  live path whose segment is fully traversed            -> shade
    //   NEED   no path; samples remain                               -> start_sample
    // SHADE and NEED run as one phase: a path that ends hands its lane straight to the
    // next sample (ballot + prefix popcount = active-ray compaction).
    // (per-XCD work queues — contiguous image bands per XCD, stealing when empty — were tried for L2 locality on
    // C5: no gain there, -1.7 % on C2, and the two extra live scalars doubled the everything-variant's spills)
    uint32_t shade_defer, prim_weight;
    { KArgsC U = kargs_fresh(); shade_defer = KARG(U, shade_defer); prim_weight = KARG(U, prim_weight); }
    // lanes holding a path / wanting a sample, as wave masks: both only change in the SHADE + REFILL phase
    unsigned long long m_act = 0ull, m_need = ~0ull;
    for (;;) {
        // the lane masks of the four states from compares, combined and counted with scalar instructions (see the BOX loop)
        // primitives come in two weights (sphere/rect ~50 instructions; Boxy, list, medium, instance
        // entry several times that): scheduled separately so cheap tests never pay for heavy ones
        const bool HAS_HEAVY = (F & (VKF_LIST | VKF_MEDIUM | VKF_INSTANCE | VKF_BOX)) != 0;
        const unsigned long long m_pend0 = __builtin_amdgcn_uicmp(L.pend, 0u, 33 /* ne */) & m_act;
        unsigned long long m_trav = __builtin_amdgcn_uicmp(L.i, L.end, 36 /* ult */);
        if (F & VKF_INSTANCE) m_trav |= __builtin_amdgcn_sicmp(L.cur_inst, 0, 39 /* sge */);
        if (GRID) m_trav |= __builtin_amdgcn_uicmp(L.cell, GRID_LAST, 36 /* ult */);
        const unsigned long long m_heavy = HAS_HEAVY ? (heavy_mask<F>(L.pend) & m_pend0) : 0ull;
        const unsigned long long m_light = m_pend0 & ~m_heavy;
        const unsigned long long m_shade = m_act & ~m_pend0 & ~m_trav;
        const uint32_t n_box = (uint32_t)__builtin_popcountll(m_act & ~m_pend0 & m_trav);
        const uint32_t n_heavy = (uint32_t)__builtin_popcountll(m_heavy), n_light = (uint32_t)__builtin_popcountll(m_light);
        const uint32_t n_prim = n_heavy > n_light ? n_heavy : n_light;
        const uint32_t n_sn = (uint32_t)__builtin_popcountll(m_shade | m_need);
        if ((n_box | n_prim | n_sn) == 0) break;
        if (STATS) st_sched++;
        if (n_box >= n_prim * prim_weight && n_box * shade_defer >= n_sn) {
            // ---- BOX: UNROLL steps under a shrinking EXEC mask per exit test, while box lanes are the plurality
            KArgsC P = kargs_fresh();
            DScene S = KARG(P, S);
            Mem M = make_mem<F, LDS_SCENE>(S, lds_items);
            if (STATS) st_t0 = clock64();
            const uint32_t live = n_box + n_heavy + n_light + n_sn;   // lanes only change state here, none appear or vanish
            // steps between two exit tests (compile time: a run-time trip count costs 4-13 %).  With shading deferred the optimum is 3
            // for scenes traversed from LDS (C2: 2 -> -4 %, 4 -> -0.5 %; C4 the same), 2 for sphere-only scenes traversed from global
            // memory (C5 on the tree handed over: 3 / 4 / 5 / 7 -> 706 / 724 / 722 / 703 Msamples/s; on the shorter walks of the rebuilt
            // tree: VK_GLOBAL_SPHERE_UNROLL below) and 5 for the everything-variants (C3: 4 / 5 / 6 / 8 ->
            // 617 / 627 / 622 / 616): the longer a step waits for its item, the less an exit test per step group is worth
            constexpr bool SPHERES = (F & ~(uint32_t)VKF_INTEG_PDF) == 0u;
#ifndef VK_CORNELL_UNROLL
#define VK_CORNELL_UNROLL 3
#endif
#ifndef VK_GLOBAL_SPHERE_UNROLL
// (on the rebuilt trees of exact re-treeing, 1 M spheres, 1 / 2 / 3 / 4 steps per exit test: 1 046 / 1 080 / 1 066 / 1 038 Msamples/s in the
// empirical unit form, round 3; in the near form, round 5, whose walks alternate between the rebuilt and the handed-over tree:
// 2 / 3 / 4 -> 1 263 / 1 276 / 1 254)
#define VK_GLOBAL_SPHERE_UNROLL 3
#endif
            constexpr int UNROLL = ((F & VKF_ALL_SCENE) == VKF_ALL_SCENE)
                ? BOX_UNROLL + 1 : (SPHERES ? (LDS_SCENE ? BOX_UNROLL - 1 : VK_GLOBAL_SPHERE_UNROLL) : VK_CORNELL_UNROLL);
            // The lane masks of the states come straight out of compares (uicmp = v_cmp into an SGPR pair) and are combined and
            // counted with scalar instructions; a ballot of a compound lane boolean goes through a VGPR (v_cndmask 0/1 + v_cmp_ne)
            // for every term.  The lanes that step are the box lanes of the last exit test: their mask is at hand in SGPRs and
            // the inverse ballot — the lane predicate of a mask — is free.
            unsigned long long m_pend, m_lt, m_inst = 0ull;
            auto masks = [&]() {
                m_pend = __builtin_amdgcn_uicmp(L.pend, 0u, 33 /* ne */);
                if constexpr (GRID) m_lt = __builtin_amdgcn_uicmp(L.i, L.end, 36 /* ult */) | __builtin_amdgcn_uicmp(L.cell, GRID_LAST, 36 /* ult */);
                else m_lt = __builtin_amdgcn_uicmp(L.i, range_end<F, Mem>(L, S), 36 /* ult */);
                if (F & VKF_INSTANCE) m_inst = __builtin_amdgcn_sicmp(L.cur_inst, 0, 39 /* sge */);
            };
            masks();
            for (;;) {
                if (F & VKF_INSTANCE) {     // end of an instance's item range: back to the parent space (rare)
                    const unsigned long long m_leave = m_act & ~m_pend & ~m_lt & m_inst;
                    if (m_leave != 0ull) {
                        if (__builtin_amdgcn_inverse_ballot_w64(m_leave)) { cold_load_world_ray<F>(cold, lane, L);
                            leave_instance<F, Mem>(L, S); }
                        masks();
                    }
                }
                bool go = __builtin_amdgcn_inverse_ballot_w64(m_act & ~m_pend & m_lt);
                if (STATS) {        // diagnostic build: same steps one at a time, counting the lanes in each
                    for (int u = 0; u < UNROLL; u++) {
                        st_box_steps += 1; st_box_lanes += lanes_with(go);
                        box_steps<F, Mem, 1>(L, S, M, go);
                        go = active && L.pend == 0u && L.i < range_end<F, Mem>(L, S);
                    }
                } else if constexpr (GRID) {
                    // one grid step per exit test: the next two references of the lane's cell, or the next cell
                    if (go) (void)grid_step(L, S, M);
                } else {
                    box_steps<F, Mem, UNROLL>(L, S, M, go);
                }
                // exit test on wave-uniform counts
                masks();
                const unsigned long long m_prim = m_pend & m_act, m_box = (m_lt | m_inst) & ~m_pend & m_act;
                const uint32_t nb = (uint32_t)__builtin_popcountll(m_box), np = (uint32_t)__builtin_popcountll(m_prim);
                const uint32_t ns = live - nb - np;
                // sphere-only diagnostic: lanes per exit test
                if (STATS && !HAS_HEAVY) { st_heavy_execs += live; st_t_light += np; st_t_heavy += ns; st_prim_execs += 1; }
                // keep stepping while nb != 0, nb >= np * prim_weight and nb * shade_defer >= ns: as sign tests of differences (the
                // counts are < 2^7), which is a third of the scalar instructions of three compares or-ed together
                const int keep1 = (int)nb - (int)(np * prim_weight > 1u ? np * prim_weight : 1u), keep2 = (int)(nb * shade_defer) - (int)ns;
                if ((keep1 | keep2) < 0) {   // another state now has more lanes parked than are stepping
                    // when that state is a LIGHT primitive test (Sphere / MovingSphere / Rect: never draws, never changes the
                    // space), run it right here and keep stepping: saves the scheduler round trip that otherwise follows every
                    // ~10 box steps
                    const unsigned long long m_light = HAS_HEAVY ? (m_prim & ~heavy_mask<F>(L.pend)) : m_prim;
                    const uint32_t nl = (uint32_t)__builtin_popcountll(m_light);
                    // nl != 0, 2 nl >= np, nl * shade_defer >= ns
                    if ((((int)nl - 1) | ((int)(2u * nl) - (int)np) | ((int)(nl * shade_defer) - (int)ns)) >= 0) {
                        if (__builtin_amdgcn_inverse_ballot_w64(m_light)) prim_step<F, Mem>(L, S, M);
                        masks();
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
            if (__builtin_amdgcn_inverse_ballot_w64(n_heavy > n_light ? m_heavy : m_light)) {
                if (F & VKF_MEDIUM) {    // ConstantMedium::hit draws inside traversal (hittable.rs:473)
                    L.rng.key = (uint64_t)__float_as_uint(cold[CF_KEY * 64 + lane]) |
                                ((uint64_t)__float_as_uint(cold[(CF_KEY + 1) * 64 + lane]) << 32);
                    L.rng.ctr = __float_as_uint(cold[CF_CTR * 64 + lane]);
                }
                prim_step<F, Mem>(L, S, M);
                if (F & VKF_MEDIUM) cold[CF_CTR * 64 + lane] = __uint_as_float(L.rng.ctr);
            }
            if (STATS) { if (n_heavy > n_light) st_t_heavy += clock64() - st_t0; else st_t_light += clock64() - st_t0; }
        } else {
            // ---- SHADE + REFILL
            if (STATS) { st_shade_execs++; st_shade_lanes += n_sn; st_t0 = clock64(); }
            // out of line for the everything-variants (see shade_refill_call); one begin_segment for both kinds of new ray
            // ... and for an 8-waves-per-SIMD build of the sphere-only variants (64 VGPRs hold the traversal loops but not shading), which
            // global-memory scenes ran until exact re-treeing halved their walks: now seven waves with the phase inline (vk_api.hip
            // launch_variant), and no instance of the kernel has MINW == 8
            constexpr bool SPLIT = ((F & VKF_ALL_SCENE) == VKF_ALL_SCENE || MINW == 8) && !STATS;
            const bool is_shade = __builtin_amdgcn_inverse_ballot_w64(m_shade);
            bool touched = false, fresh = false;
            bool early = false;       // exact re-treeing: this segment's winner may depend on the visiting order
            if constexpr ((F & ~(uint32_t)VKF_INTEG_PDF) == 0u) {
                KArgsC P = kargs_fresh();
                DScene S = KARG(P, S);
                if (S.t_pad > 0.0f) {      // (wave-uniform)
                    Mem M = make_mem<F, LDS_SCENE>(S, lds_items);
                    bool on_ref = false;   // the lane has just walked the tree as handed over: its answer stands
                    if constexpr (!LDS_SCENE) {
                        // (bit 31 of the depth word: walked again after an early winner; depth 1 under DScene::primary_ref: a primary ray
                        // that begin_segment started on the tree as handed over)
                        const uint32_t dw = __float_as_uint(cold[CF_DEPTH * 64 + lane]);
                        on_ref = (dw >> 31) != 0u || (S.primary_ref != 0u && S.walk_start != 0u && dw == 1u);
                    }
                    if (is_shade && !on_ref) early = segment_unsafe<F, Mem>(L, S, M);
                }
            }
            if constexpr (SPLIT) {
                ShadeIo io;
                io.flags = (is_shade ? 1u : 0u) | (active ? 2u : 0u) | (need ? 4u : 0u) | (early ? 64u : 0u);
                io.T = L.T; io.best_prim = L.best_prim; io.best_inst = L.best_inst; io.best_aux = L.best_aux;
                io.o = L.o; io.d = L.d; io.time = L.time; io.cost_t0 = cost_t0;
                io = shade_refill_call<F, LDS_SCENE, COST>(io, lds_items, wave_block);
                active = (io.flags & 2u) != 0u; need = (io.flags & 4u) != 0u; fresh = (io.flags & 8u) != 0u;
                const bool rearm = (io.flags & 32u) != 0u;      // exact re-treeing: the same ray again, on the tree as handed over
                cost_t0 = io.cost_t0;
                if (fresh | rearm) {
                    KArgsC P = kargs_fresh();
                    DScene S = KARG(P, S);
                    // (the callee stored the path state, new world ray included)
                    begin_segment<Mem::ISHIFT, fused_box<F, Mem>(), spheres_only<F>()>(L, S, io.o, io.d, io.time, rearm);
                }
            } else {
                PhaseClocks clk;
                bool rearm = false;
                shade_refill_body<F, LDS_SCENE, STATS, COST>(L, is_shade, early, active, need, fresh, touched, rearm, cost_t0,
                    kargs_fresh(), cold,
                    tile_sum, wstate, lane, lds_items, clk);
                if (STATS) { st_t_mat += clk.mat; st_t_refill += clk.refill; st_t_turb += clk.turb; st_t_cold += clk.cold; st_t1 = clock64(); }
                if (fresh | rearm) {
                    KArgsC P = kargs_fresh();
                    DScene S = KARG(P, S);
                    // one copy of the exact reciprocals for both kinds of new ray
                    begin_segment<Mem::ISHIFT, fused_box<F, Mem>(), spheres_only<F>()>(L, S, L.wo, L.wd, L.time, rearm);
                }
                if (active && touched) cold_store_path<F>(cold, lane, L);
                if (STATS) st_t_install += clock64() - st_t1;
            }
            m_act = __builtin_amdgcn_ballot_w64(active); m_need = __builtin_amdgcn_ballot_w64(need);
            if (STATS) st_t_shade += clock64() - st_t0;
        }
    }
    {   // the last unit's sums
        KArgsC P = kargs_fresh();
#ifdef VK_WAVE_TIMES
        unsigned long long *wt = KARG(P, wave_times);
        if (wt && lane == 0) wt[3u * ((blockIdx.x + (blockDim.x == 1024u ? 0u : gridDim.x)) * 16u + wave) + 2u] = wall_clock64();
#endif
        flush_tile_sums(tile_sum, KARG(P, accum), __builtin_amdgcn_readfirstlane(wstate[WS_TXY]), lane, KARG(P, C.width),
            KARG(P, C.height));
    }
    if (STATS) {
        KArgsC P = kargs_fresh();
        unsigned long long *ps = KARG(P, phase_stats);
        if (ps && lane == 0) {
            atomicAdd(&ps[0], st_box_steps); atomicAdd(&ps[1], st_box_lanes); atomicAdd(&ps[2], st_prim_execs);
            atomicAdd(&ps[3], st_prim_lanes);
            atomicAdd(&ps[4], st_shade_execs); atomicAdd(&ps[5], st_shade_lanes); atomicAdd(&ps[6], st_sched);
            atomicAdd(&ps[7], st_heavy_execs);
            atomicAdd(&ps[8], st_t_box); atomicAdd(&ps[9], st_t_light); atomicAdd(&ps[10], st_t_heavy); atomicAdd(&ps[11], st_t_shade);
            atomicAdd(&ps[12], (unsigned long long)(clock64() - st_t_total));
            atomicAdd(&ps[13], st_t_mat); atomicAdd(&ps[14], st_t_refill); atomicAdd(&ps[15], st_t_install);
            atomicAdd(&ps[16], st_t_turb); atomicAdd(&ps[17], st_t_cold);
        }
    }
}

// Between the two launches of exact re-treeing (ONE block of REDO_REGIONS threads): plan[1] = samples queued (vk_stats), plan[2] = entries
// that did not fit their queue (must be 0), plan[3] = entries per work unit of the second launch — 64 (one sample per lane: a short list
// then reaches every wave and the launch lasts about as long as its longest path) up to REDO_UNIT for long lists — and plan[0] = units of
// that size in the fullest queue (the second launch's units are REDO_REGIONS x that).
__global__ void redo_plan_kernel(const uint32_t *count, uint32_t cap, uint32_t *plan, uint32_t n_waves) {
    __shared__ uint32_t s_total, s_max, s_lost;
    const uint32_t r = threadIdx.x;
    if (r == 0) { s_total = 0; s_max = 0; s_lost = 0; }
    __syncthreads();
    const uint32_t c = r < REDO_REGIONS ? count[r * REDO_COUNT_STRIDE] : 0u;
    const uint32_t kept = c < cap ? c : cap;
    atomicAdd(&s_total, kept); atomicMax(&s_max, kept);
    if (c > cap) atomicAdd(&s_lost, c - cap);
    __syncthreads();
    if (r == 0) {
        uint32_t unit = 64u;
        while (unit < REDO_UNIT && (uint64_t)s_total > (uint64_t)unit * n_waves * 2u) unit *= 2u;
        plan[0] = (s_max + unit - 1u) / unit; plan[1] = s_total; plan[2] = s_lost; plan[3] = unit;
    }
}

// Behind the second launch: if a queue overflowed (plan[2] != 0) the frame is incomplete — the sums and the unit counter are cleared and the
// fallback launch (list_mode = 2) renders the partition again on the tree as handed over.  Nearly always: nothing.
__global__ void redo_reset_kernel(const uint32_t *plan, unsigned long long *accum, size_t n_words, uint32_t *counter) {
    if (plan[2] == 0u) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { counter[0] = 0u; counter[2] = 0u; counter[3] = 0u; }      // the unit counter and the clamped-sample count
    for (size_t k = i; k < n_words; k += (size_t)gridDim.x * blockDim.x) accum[k] = 0ull;
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
__global__ void order_scatter_kernel(uint32_t *cost, uint32_t n_local, uint32_t tile_rank, uint32_t tile_world, uint32_t *hist,
    uint32_t *order) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_local) return;
    uint32_t t = tile_rank + i * tile_world;
    order[atomicAdd(&hist[cost_bucket(cost[t])], 1u)] = i;
}

// pixel mean = fixed-point sum / spp (main.rs:196), for the pixels of this call's tile partition
__global__ void resolve_kernel(const long long *accum, float *out, uint32_t width, uint32_t height, uint32_t spp,
                               uint32_t tiles_x, uint32_t tile_rank, uint32_t tile_world) {
    uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_pixels = width * height;
    if (pix >= n_pixels) return;
    uint32_t x = pix % width, y = pix / width;
    uint32_t tile = (y / TILE) * tiles_x + (x / TILE);
    if (tile % tile_world != tile_rank) return;
    float n = (float)spp;
    const long long *a = accum + (size_t)pix * 3;
    const float inv_scale = 1.0f / ACCUM_SCALE;
    out[(size_t)pix * 3 + 0] = ((float)a[0] * inv_scale) / n;
    out[(size_t)pix * 3 + 1] = ((float)a[1] * inv_scale) / n;
    out[(size_t)pix * 3 + 2] = ((float)a[2] * inv_scale) / n;
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

#ifdef VK_DEBUG_LIB
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
        case 8: { vk::Rng g = vk::rng_for_sample(__float_as_uint(a[i]), (uint32_t)i, 0);
            r = vk::gen_range(g, -1.0f, 1.0f) + vk::gen_f32(g); break; }
        case 9: r = a[i] * b[i] + a[i]; break;   // must stay an unfused mul+add
        case 10: r = div_by_a(a[i], b[i], refined_rcp(b[i]), true); break;      // the sphere test's quotient by a shared reciprocal
    }
    out[i] = r;
}

#endif

}  // namespace
#endif
