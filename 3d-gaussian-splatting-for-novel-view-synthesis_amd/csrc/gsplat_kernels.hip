// gsplat_kernels.hip -- MI355X (gfx950, wave64) kernels and the C ABI of include/gsplat_mi355x.h.
//
// Pipeline (stage ids of SURVEY.md §8a in brackets):
//   K0 camera_kernel            c2w (device) -> Camera block                         [F5 setup]
//   K1 project_kernel           per Gaussian: culls, EWA, eigen clamp, conic, rect   [F1-F8, F10, F13]
//   K2 rocprim inclusive_scan   tiles-per-Gaussian -> pair offsets                   [F11]
//   K3 emit_pairs_kernel        key = tile id, payload = depth bits << 32 | id        [F11]
//   K4 rocprim radix_sort_pairs radix sort on the tile-id bits only (2 passes)       [F12]
//   K4b tile_sort_kernel        per-tile bitonic sort of the payloads in LDS          [F9, F12]
//   K5 tile_ranges_kernel       per-tile [start, end)                                [F12]
//   K6 raster_forward_kernel    one wave64 per 16x8 half tile, 2 pixels per lane     [F14, F15]
//   K7 raster_backward_kernel   same traversal, analytic gradients, wave reduction   [B1]
//   K8 project_backward_kernel  chain rule to the reference's input tensors          [B2, B3]
//
// No MFMA: there is no dense contraction on this path.  No CPU fallback: without a GPU every entry
// point returns GSPLAT_ERR_HIP.
#include <cstdio>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "gs_body.h"

using namespace gsm;

thread_local char gsplat_err_buf[512] = "";      // shared with gsplat_loss.hip; read through gsplat_last_error()

namespace {

char (&g_err)[512] = gsplat_err_buf;

int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
    snprintf(g_err, sizeof(g_err), fmt, a, b);
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) return fail(GSPLAT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define LAUNCH_CHECK(name)                                                              \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) return fail(GSPLAT_ERR_HIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

constexpr int64_t ALIGN = 256;
inline int64_t up(int64_t x) { return (x + ALIGN - 1) / ALIGN * ALIGN; }

// ---- layout of the caller-owned buffers (private to the library) ---------------------------------
struct DevCounts {           // device-side counters; copied into gsplat_counts
    int32_t n_survivors, n_visible;
    int64_t n_pairs;
    int32_t max_tiles, reserved;
};
static_assert(sizeof(DevCounts) == sizeof(gsplat_counts), "counts layout");

constexpr int COUNT_SHARDS = 256;     // per-wave counters are spread over 256 cache lines (same-address atomics serialise)
struct alignas(64) CountShard { int32_t survivors, visible, max_tiles, pad[13]; };

struct ProjectState {
    Camera* cam;
    DevCounts* counts;
    CountShard* shards;
    Rec64* rec;
    u2* rect;
    float* depth;
    uint32_t* tiles;
    uint32_t* offsets;       // inclusive prefix sum of tiles
    int64_t bytes;
};

ProjectState carve_project(void* base, int64_t n) {
    ProjectState s;
    char* p = (char*)base;
    int64_t o = 0;
    s.cam = (Camera*)(p + o); o += up(sizeof(Camera));
    s.counts = (DevCounts*)(p + o); o += up(sizeof(DevCounts));
    s.shards = (CountShard*)(p + o); o += up(sizeof(CountShard) * COUNT_SHARDS);
    s.rec = (Rec64*)(p + o); o += up(n * 64);
    s.rect = (u2*)(p + o); o += up(n * 8);
    s.depth = (float*)(p + o); o += up(n * 4);
    s.tiles = (uint32_t*)(p + o); o += up(n * 4);
    s.offsets = (uint32_t*)(p + o); o += up(n * 4);
    s.bytes = o;
    return s;
}

struct BinState {
    uint32_t* sorted_ids;    // [P] Gaussian ids in (tile, depth, id) order
    uint2* ranges;           // [tiles] start, end
    uint32_t* order_fwd;     // [tiles] launch order of the forward raster (longest list first)
    uint32_t* order_bwd;     // [tiles] launch order of the backward raster (most visited first)
    uint32_t* visited;       // [2 * tiles] Gaussians the forward visited per half tile
    int64_t bytes;
};

BinState carve_bin(void* base, int64_t n_pairs, int64_t n_tiles) {
    BinState s;
    char* p = (char*)base;
    int64_t o = 0;
    s.sorted_ids = (uint32_t*)(p + o); o += up((n_pairs > 0 ? n_pairs : 1) * 4);
    s.ranges = (uint2*)(p + o); o += up(n_tiles * 8);
    s.order_fwd = (uint32_t*)(p + o); o += up(n_tiles * 4);
    s.order_bwd = (uint32_t*)(p + o); o += up(n_tiles * 4);
    s.visited = (uint32_t*)(p + o); o += up(n_tiles * 8);
    s.bytes = o;
    return s;
}

// Binning scratch.  Pairs are radix-sorted by tile id only (ceil(log2 tiles) key bits: 2 onesweep passes at 1080p) with a
// 64-bit payload (depth bits << 32 | Gaussian id); the depth order inside every tile is then produced by a per-tile
// bitonic sort in LDS (tile_sort_kernel).  Sorting 64-bit (tile, depth) keys globally took 6 passes.
struct BinScratch {
    uint32_t *keys_in, *keys_out;     // [P] tile id
    uint64_t *vals_in, *vals_out;     // [P] depth bits << 32 | id
    void* temp;
    size_t temp_bytes;
    int64_t bytes;
};

size_t bin_temp_bytes(int64_t n_pairs) {
    size_t c = 0;
    (void)rocprim::radix_sort_pairs(nullptr, c, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint64_t*)nullptr, (uint64_t*)nullptr,
                                    (size_t)(n_pairs > 0 ? n_pairs : 1), 0u, 32u, (hipStream_t)0);
    return c;
}

BinScratch carve_bin_scratch(void* base, int64_t n_pairs) {
    BinScratch s;
    char* p = (char*)base;
    int64_t o = 0;
    const int64_t np = n_pairs > 0 ? n_pairs : 1;
    s.keys_in = (uint32_t*)(p + o); o += up(np * 4);
    s.keys_out = (uint32_t*)(p + o); o += up(np * 4);
    s.vals_in = (uint64_t*)(p + o); o += up(np * 8);
    s.vals_out = (uint64_t*)(p + o); o += up(np * 8);
    s.temp_bytes = bin_temp_bytes(np);
    s.temp = (void*)(p + o); o += up((int64_t)s.temp_bytes);
    s.bytes = o;
    return s;
}

size_t scan_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::inclusive_scan(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)(n > 0 ? n : 1),
                            rocprim::plus<uint32_t>(), (hipStream_t)0);
    return bytes;
}

int check_view(const gsplat_view* v) {
    if (!v) return fail(GSPLAT_ERR_BAD_ARG, "view is NULL");
    if (v->H <= 0 || v->W <= 0) return fail(GSPLAT_ERR_BAD_ARG, "image size must be positive");
    if (v->tile != 16) return fail(GSPLAT_ERR_BAD_ARG, "only tile size T=16 is built (the image does not depend on T)");
    if ((v->W + 15) / 16 > 65535 || (v->H + 15) / 16 > 65535) return fail(GSPLAT_ERR_BAD_ARG, "image too large");
    return GSPLAT_OK;
}

int check_gaussians(const gsplat_gaussians* g, bool* fused) {
    if (!g) return fail(GSPLAT_ERR_BAD_ARG, "gaussians is NULL");
    if (g->n < 0 || g->n > 0x7fffffffLL) return fail(GSPLAT_ERR_BAD_ARG, "n out of range");
    const bool f = g->scale_raw || g->q_raw || g->f_dc || g->f_rest;
    const bool u = g->color || g->sigma;
    if (f == u) return fail(GSPLAT_ERR_BAD_ARG, "give either (color, sigma) or (scale_raw, q_raw, f_dc, f_rest)");
    if (g->n > 0) {
        if (!g->pos || !g->opacity_raw) return fail(GSPLAT_ERR_BAD_ARG, "pos / opacity_raw is NULL");
        if (f && !(g->scale_raw && g->q_raw && g->f_dc && g->f_rest)) return fail(GSPLAT_ERR_BAD_ARG, "fused inputs incomplete");
        if (u && !(g->color && g->sigma)) return fail(GSPLAT_ERR_BAD_ARG, "color / sigma is NULL");
    }
    const void* ptrs[] = {g->pos, g->opacity_raw, g->color, g->sigma, g->scale_raw, g->q_raw, g->f_dc, g->f_rest};
    for (const void* q : ptrs)
        if (q && (reinterpret_cast<uintptr_t>(q) & 15u)) return fail(GSPLAT_ERR_BAD_ARG, "Gaussian arrays must be 16-byte aligned");
    *fused = f;
    return GSPLAT_OK;
}

// ---- device helpers ------------------------------------------------------------------------------

__device__ __forceinline__ float dpp_row_sum(float v) {
    // sum across the 64 lanes with DPP; the total ends up in lane 63 (row 3)
    int x;
#define DPP_ADD(ctrl, rmask)                                                                       \
    x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, false);               \
    v += __int_as_float(x);
    DPP_ADD(0xB1, 0xF)    // quad_perm [1,0,3,2]
    DPP_ADD(0x4E, 0xF)    // quad_perm [2,3,0,1]
    DPP_ADD(0x141, 0xF)   // row_half_mirror
    DPP_ADD(0x140, 0xF)   // row_mirror
    DPP_ADD(0x142, 0xA)   // row_bcast15 -> rows 1, 3
    DPP_ADD(0x143, 0xC)   // row_bcast31 -> rows 2, 3
#undef DPP_ADD
    return v;
}

// Nine wave totals at once, step-interleaved so that no DPP instruction reads a register written by the previous
// instruction (no s_nop padding between dependent VALU -> DPP pairs).
__device__ __forceinline__ void wave_total9(float (&v)[9]) {
#define DPP_STEP(ctrl, rmask)                                                                              \
    _Pragma("unroll") for (int k = 0; k < 9; ++k) {                                                         \
        const int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v[k]), ctrl, rmask, 0xF, false);        \
        v[k] += __int_as_float(x);                                                                          \
    }
    DPP_STEP(0xB1, 0xF)
    DPP_STEP(0x4E, 0xF)
    DPP_STEP(0x141, 0xF)
    DPP_STEP(0x140, 0xF)
    DPP_STEP(0x142, 0xA)
    DPP_STEP(0x143, 0xC)
#undef DPP_STEP
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k]), 63));
}

__device__ __forceinline__ float wave_total(float v) {
    v = dpp_row_sum(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// XCD-aware block -> tile map: blocks b, b+8, b+16, ... share an XCD (round-robin dispatch), so give each
// XCD group one contiguous range of tile ids (contiguous tile rows => neighbouring tiles => shared L2 lines).
// Bijective for any tile count (cdna_hip_programming.md, "XCD swizzle must be bijective").
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t n) {
    const uint32_t xcd = b & 7u, q = n >> 3, r = n & 7u;
    const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + (b >> 3);
}

// ---- K0 ------------------------------------------------------------------------------------------
__global__ void camera_kernel(const float* __restrict__ c2w, Camera* cam, DevCounts* counts, CountShard* shards) {
    if (threadIdx.x < COUNT_SHARDS) { shards[threadIdx.x].survivors = 0; shards[threadIdx.x].visible = 0; shards[threadIdx.x].max_tiles = 0; }
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float m[16];
        for (int i = 0; i < 16; ++i) m[i] = c2w[i];
        Camera c;
        build_camera(m, c);
        *cam = c;
        counts->n_survivors = 0; counts->n_visible = 0; counts->n_pairs = 0; counts->max_tiles = 0; counts->reserved = 0;
    }
}

// ---- K1 ------------------------------------------------------------------------------------------
// One wave64 per 64 Gaussians.  The reference layout is array-of-structures (pos[N,3], f_rest[N,45] ...): a lane
// reading its own row directly issues 45 loads that each touch 64 different cache lines.  Instead the wave copies its
// 64 contiguous rows into LDS with fully coalesced 16-byte accesses and every lane then reads its row from LDS
// (row strides 3, 4, 9, 45 words are conflict-free or 2-way at worst).  The SH block (f_dc + f_rest, 192 of the 236
// input bytes) is only fetched when at least one Gaussian of the wave survived the culls.
template <int R>
__device__ __forceinline__ void stage_rows(float* __restrict__ lds, const float* __restrict__ g, int64_t row0, int64_t n, int lane) {
    const int64_t left = n - row0;
    const int total = (int)(left < 64 ? left : 64) * R;       // floats to copy
    const float* __restrict__ src = g + row0 * R;             // 16-B aligned: row0 % 64 == 0, base 16-B aligned (host checks)
    constexpr int PIECES = 64 * R / 4;
#pragma unroll
    for (int it = 0; it < (PIECES + 63) / 64; ++it) {
        const int piece = it * 64 + lane;
        if (piece * 4 + 3 < total) {
            *reinterpret_cast<f4*>(lds + piece * 4) = *reinterpret_cast<const f4*>(src + piece * 4);
        } else if (piece * 4 < total) {
            for (int k = piece * 4; k < total; ++k) lds[k] = src[k];
        }
    }
}

template <int R>
__device__ __forceinline__ void unstage_rows(float* __restrict__ g, const float* __restrict__ lds, int64_t row0, int64_t n, int lane) {
    const int64_t left = n - row0;
    const int total = (int)(left < 64 ? left : 64) * R;
    float* __restrict__ dst = g + row0 * R;
    constexpr int PIECES = 64 * R / 4;
#pragma unroll
    for (int it = 0; it < (PIECES + 63) / 64; ++it) {
        const int piece = it * 64 + lane;
        if (piece * 4 + 3 < total) {
            *reinterpret_cast<f4*>(dst + piece * 4) = *reinterpret_cast<const f4*>(lds + piece * 4);
        } else if (piece * 4 < total) {
            for (int k = piece * 4; k < total; ++k) dst[k] = lds[k];
        }
    }
}

struct ProjectLds {
    float pos[64 * 3];
    float opa[64];
    float a[64 * 9];        // fused: q_raw [64][4]       un-fused: sigma [64][9]
    float b[64 * 3];        // fused: scale_raw [64][3]   un-fused: colour [64][3]
};

struct ShCoefLds {          // same access as ShCoefGlobal, on the staged copy
    const float* dc;
    const float* rest;
    __device__ __forceinline__ float operator()(int k, int ch) const { return k == 0 ? dc[ch] : rest[ch * 15 + (k - 1)]; }
};

template <bool FUSED>
__device__ __forceinline__ GaussIn gauss_from_lds(const ProjectLds& s, int lane) {
    GaussIn in;
#pragma unroll
    for (int k = 0; k < 3; ++k) in.p[k] = s.pos[lane * 3 + k];
    in.o_raw = s.opa[lane];
    if (FUSED) {
#pragma unroll
        for (int k = 0; k < 4; ++k) in.qr[k] = s.a[lane * 4 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) in.sr[k] = s.b[lane * 3 + k];
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) in.S9[k] = s.a[lane * 9 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k) in.col[k] = s.b[lane * 3 + k];
    }
    return in;
}

template <bool FUSED>
__device__ __forceinline__ void stage_geometry(ProjectLds& s, const gsplat_gaussians& g, int64_t row0, int lane) {
    stage_rows<3>(s.pos, g.pos, row0, g.n, lane);
    stage_rows<1>(s.opa, g.opacity_raw, row0, g.n, lane);
    if (FUSED) {
        stage_rows<4>(s.a, g.q_raw, row0, g.n, lane);
        stage_rows<3>(s.b, g.scale_raw, row0, g.n, lane);
    } else {
        stage_rows<9>(s.a, g.sigma, row0, g.n, lane);
        stage_rows<3>(s.b, g.color, row0, g.n, lane);
    }
}

template <bool FUSED>
__global__ __launch_bounds__(64) void project_kernel(gsplat_gaussians g, const Camera* __restrict__ camp, ViewK vk, Records out,
                                                     CountShard* shards) {
    __shared__ ProjectLds s;
    __shared__ float s_dc[FUSED ? 64 * 3 : 4];
    __shared__ float s_rest[FUSED ? 64 * 45 : 4];
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + lane;
    const Camera cam = *camp;
    stage_geometry<FUSED>(s, g, row0, lane);
    __syncthreads();
    GaussIn in;
    Proj o;
    o.vis = VIS_CULLED;
    if (i < g.n) {
        in = gauss_from_lds<FUSED>(s, lane);
        o = project_geometry(in, FUSED, cam, vk);
    }
    RecOut r;
    r.vis = o.vis; r.tiles = 0;
    if (FUSED) {
        if (__any(o.vis == VIS_OK)) {                        // wave-uniform: skip 192 B / Gaussian when all are culled
            stage_rows<3>(s_dc, g.f_dc, row0, g.n, lane);
            stage_rows<45>(s_rest, g.f_rest, row0, g.n, lane);
            __syncthreads();
            if (o.vis == VIS_OK) r = project_finish(in, o, true, ShCoefLds{s_dc + lane * 3, s_rest + lane * 45}, cam);
        }
    } else if (o.vis == VIS_OK) {
        r = project_finish(in, o, false, ShCoefLds{nullptr, nullptr}, cam);
    }
    if (i < g.n) {
        if (r.vis == VIS_OK) {
            Rec64 line;
            line.r0 = r.r0; line.r1 = r.r1; line.r2 = r.r2; line.pad = f4{0.f, 0.f, 0.f, 0.f};
            out.rec[i] = line;                   // 64 contiguous bytes per lane, 4 KB per wave
            out.rect[i] = r.rect;
            out.depth[i] = r.r2.w;
        }
        out.tiles[i] = r.tiles;
    }
    const unsigned long long surv = __ballot(o.vis != VIS_CULLED);
    const unsigned long long seen = __ballot(o.vis == VIS_OK);
    uint32_t mx = r.tiles;
    for (int sft = 32; sft > 0; sft >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, sft));
    if (lane == 0) {
        CountShard* sh = shards + (blockIdx.x % COUNT_SHARDS);
        if (surv) atomicAdd(&sh->survivors, (int)__popcll(surv));
        if (seen) atomicAdd(&sh->visible, (int)__popcll(seen));
        if (mx) atomicMax(&sh->max_tiles, (int)mx);
    }
}

__global__ __launch_bounds__(COUNT_SHARDS) void finish_counts_kernel(const uint32_t* __restrict__ offsets, int64_t n,
                                                                     const CountShard* __restrict__ shards, DevCounts* counts) {
    __shared__ int part[3][COUNT_SHARDS / 64];
    int a = shards[threadIdx.x].survivors, b = shards[threadIdx.x].visible, c = shards[threadIdx.x].max_tiles;
    for (int sft = 32; sft > 0; sft >>= 1) { a += __shfl_xor(a, sft); b += __shfl_xor(b, sft); c = max(c, __shfl_xor(c, sft)); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = a; part[1][threadIdx.x >> 6] = b; part[2][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int sa = 0, sb = 0, sc = 0;
        for (int k = 0; k < COUNT_SHARDS / 64; ++k) { sa += part[0][k]; sb += part[1][k]; sc = max(sc, part[2][k]); }
        counts->n_survivors = sa; counts->n_visible = sb; counts->max_tiles = sc;
        counts->n_pairs = n > 0 ? (int64_t)offsets[n - 1] : 0;
    }
}

// ---- K3 ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emit_pairs_kernel(int64_t n, const float* __restrict__ depth, const u2* __restrict__ rect,
                                                         const uint32_t* __restrict__ tiles, const uint32_t* __restrict__ offsets,
                                                         int tiles_x, int64_t n_pairs, uint32_t* __restrict__ keys,
                                                         uint64_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t nt = tiles[i];
    if (nt == 0) return;
    const u2 r = rect[i];
    const int tx0 = r.x & 0xFFFF, ty0 = r.x >> 16, tx1 = r.y & 0xFFFF, ty1 = r.y >> 16;
    // z > 0: the float's bit pattern orders like its value; equal depths fall back to the Gaussian index (low word)
    const uint64_t payload = ((uint64_t)f2u(depth[i]) << 32) | (uint64_t)(uint32_t)i;
    int64_t o = (int64_t)offsets[i] - nt;
    for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) {
            if (o < n_pairs) {                     // defensive: never write past the caller's buffer
                keys[o] = (uint32_t)(ty * tiles_x + tx);
                vals[o] = payload;
            }
            ++o;
        }
}

// ---- K5 ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_ranges_kernel(int64_t n_pairs, const uint32_t* __restrict__ keys, uint2* ranges) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
    const uint32_t t = keys[i];
    if (i == 0 || keys[i - 1] != t) ranges[t].x = (uint32_t)i;
    if (i == n_pairs - 1 || keys[i + 1] != t) ranges[t].y = (uint32_t)(i + 1);
}

// ---- K4b: per-tile depth sort ----------------------------------------------------------------------
// One workgroup per tile sorts the tile's (depth bits << 32 | id) payloads ascending with a bitonic network and writes
// the ids.  Lists of up to TILE_SORT_LDS entries are sorted in LDS; longer ones in place in global memory by the same
// workgroup (slow path, exact).  Keys are unique (the id is part of the key), so the result is deterministic.
// Two size classes share the code: lists of up to 512 entries (almost all tiles) use 128 threads and 4 KB of LDS, so
// 16 workgroups fit a CU; longer lists use 256 threads and 32 KB.  Each launch covers all tiles and a workgroup
// returns at once when its tile belongs to the other class.
constexpr int TILE_SORT_LDS = 4096;            // largest list sorted in LDS (32 KB of 64-bit keys)
constexpr int TILE_SORT_SMALL = 512;

// Direction-free bitonic network: every merge of size k starts with a mirror step (i <-> block_end - i), followed by
// the half-cleaner steps j = k/4 .. 1; every compare-exchange puts the smaller key at the lower index.  With virtual
// +inf padding above n no real element is ever exchanged with the padding, so the network also runs in place.
template <int THREADS, class Swap>
__device__ __forceinline__ void bitonic_network(uint32_t n, uint32_t m, int tid, Swap swap_if_greater) {
    uint32_t lk = 1;                                              // log2(k)
    for (uint32_t k = 2; k <= m; k <<= 1, ++lk) {
        const uint32_t half = k >> 1, lh = lk - 1;
        for (uint32_t t = tid; t < (m >> 1); t += THREADS) {
            const uint32_t r = t & (half - 1), i = ((t >> lh) << lk) + r, l = i + (k - 1 - 2 * r);
            if (l < n) swap_if_greater(i, l);
        }
        __syncthreads();
        uint32_t lj = lh;                                         // log2(j) + 1
        for (uint32_t j = half >> 1; j > 0; j >>= 1) {
            --lj;
            for (uint32_t t = tid; t < (m >> 1); t += THREADS) {
                const uint32_t i = ((t >> lj) << (lj + 1)) + (t & (j - 1)), l = i + j;
                if (l < n) swap_if_greater(i, l);
            }
            __syncthreads();
        }
    }
}

template <int THREADS, int CAP, bool LARGE>
__global__ __launch_bounds__(THREADS) void tile_sort_kernel(const uint2* __restrict__ ranges, uint64_t* __restrict__ vals,
                                                            uint32_t* __restrict__ sorted_ids) {
    __shared__ uint64_t sk[CAP];
    const uint2 rg = ranges[blockIdx.x];
    const uint32_t n = rg.y - rg.x;
    if (n == 0) return;
    if (LARGE ? (n <= (uint32_t)TILE_SORT_SMALL) : (n > (uint32_t)TILE_SORT_SMALL)) return;     // the other launch's tile
    const int tid = threadIdx.x;
    uint64_t* g = vals + rg.x;
    uint32_t* __restrict__ out = sorted_ids + rg.x;
    uint32_t m = 2;
    while (m < n) m <<= 1;                     // padded size (power of two); entries >= n are virtual +inf
    if (n <= (uint32_t)CAP) {
        for (uint32_t i = tid; i < n; i += THREADS) sk[i] = g[i];
        __syncthreads();
        bitonic_network<THREADS>(n, m, tid, [&](uint32_t i, uint32_t l) {
            const uint64_t a = sk[i], b = sk[l];
            if (a > b) { sk[i] = b; sk[l] = a; }
        });
        for (uint32_t i = tid; i < n; i += THREADS) out[i] = (uint32_t)sk[i];
    } else {
        bitonic_network<THREADS>(n, m, tid, [&](uint32_t i, uint32_t l) {
            const uint64_t a = g[i], b = g[l];
            if (a > b) { g[i] = b; g[l] = a; }
        });
        for (uint32_t i = tid; i < n; i += THREADS) out[i] = (uint32_t)g[i];
    }
}

// ---- K5b: launch order ------------------------------------------------------------------------------
// Longest-processing-time-first order of the tiles (descending work, 1/8-octave buckets).  The raster kernels are
// tail-bound: a few dense tiles take 5x the mean, so they must start first (and get issue priority, see s_setprio).
// work(t) = list length (forward) or the number of Gaussians the forward actually visited in the tile (backward).
__device__ __forceinline__ uint32_t work_bucket(uint32_t w) {
    if (w < 8u) return w;
    const uint32_t e = 31u - (uint32_t)__clz((int)w);
    return (e - 2u) * 8u + ((w >> (e - 3u)) & 7u);          // <= 239
}

__global__ __launch_bounds__(1024) void order_tiles_kernel(int n_tiles, const uint2* __restrict__ ranges,
                                                           const uint32_t* __restrict__ visited, uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[256], cursor[256];
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int t = tid; t < n_tiles; t += 1024) {
        const uint32_t w = visited ? visited[2 * t] + visited[2 * t + 1] : ranges[t].y - ranges[t].x;
        atomicAdd(&hist[work_bucket(w)], 1u);
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t above = 0;
        for (int k = 255; k > tid; --k) above += hist[k];
        cursor[tid] = above;
    }
    __syncthreads();
    for (int t = tid; t < n_tiles; t += 1024) {
        const uint32_t w = visited ? visited[2 * t] + visited[2 * t + 1] : ranges[t].y - ranges[t].x;
        order[atomicAdd(&cursor[work_bucket(w)], 1u)] = (uint32_t)t;
    }
}

// ---- K6 / K7: rasterizer -----------------------------------------------------------------------------
// One wave64 per HALF tile (16 x 8 pixels): lane l owns column (l & 15) and rows (l >> 4) and (l >> 4) + 4 of the
// half, so the two pixels of a lane form a float2 and the arithmetic runs on packed fp32 (v_pk_fma_f32 ...).
//
// Staging = binning at wave granularity: the wave walks its tile's depth-sorted list 64 entries at a time; lane l
// fetches entry l's record and tests the Gaussian's tight box {q <= chi} (ex, ey of the record) against the wave's
// 16 x 8 pixel rectangle.  __ballot + mbcnt give every survivor its slot, in list order, and the survivors are
// compacted into LDS.  The wave-wide inner loop then only visits Gaussians that can touch its pixels; a rejected
// Gaussian costs one lane a few instructions instead of costing the whole wave an inner-loop iteration.  Skipping is
// exact: a Gaussian whose box misses the rectangle has q > chi at every pixel there, i.e. alpha = 0 and T unchanged.
// The next chunk's records are fetched while the current chunk's survivors are composited.
//
// Launch order: block b takes tile order[b >> 1], half b & 1, with `order` from order_tiles_kernel (heaviest first),
// and the first blocks raise their wave priority so that a dense tile is not slowed down by light co-resident waves.
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int BATCH = 64;

// Diagnostics (tools/raster_stats.py): when a buffer is registered with gsplat_debug_set_stats(), every raster wave
// writes {list length, chunks staged, survivors visited, shader cycles} for its region.  Never set in normal use.
struct WaveStats { uint32_t list_len, chunks, visited, cycles; };
WaveStats* g_stats_fwd = nullptr;
WaveStats* g_stats_bwd = nullptr;
int g_ablate = 0;      // diagnostics: bit 0 = no atomics, bit 1 = no wave reduction (results are WRONG when set)

constexpr float QK = -0.72134752044448170368f;      // -0.5 * log2(e)

struct RasterStage {
    f4 r0[BATCH + 2];      // u, v, k A11, 2 k A12    (+2: null records that pad an odd survivor count)
    f4 r1[BATCH + 2];      // k A22, opacity, r, g
    float bl[BATCH + 2];   // b
    uint32_t id[BATCH + 2];
};

struct Candidate {         // one list entry held by one lane between fetch and test
    f4 q0, q1, q2;
    uint32_t id;
    bool valid;
};

__device__ __forceinline__ Candidate fetch_candidate(int lane, uint32_t base, uint32_t end, const uint32_t* __restrict__ ids,
                                                     const Rec64* __restrict__ rec) {
    Candidate c;
    const uint32_t idx = base + lane;
    c.valid = idx < end;
    c.id = 0;
    c.q0 = c.q1 = c.q2 = f4{0.f, 0.f, 0.f, 0.f};
    if (c.valid) {
        c.id = ids[idx];
        const Rec64* __restrict__ r = rec + c.id;        // one 64-byte line
        c.q0 = r->r0;
        c.q1 = r->r1;
        c.q2 = r->r2;
    }
    return c;
}

// Cull + compact the fetched candidates into LDS, padded to an even count.  Returns the survivor count (uniform).
template <bool WITH_ID>
__device__ __forceinline__ int compact_candidates(RasterStage& s, const Candidate& c, float x0, float x1, float y0, float y1) {
    const bool pass = c.valid && (c.q0.x + c.q1.z >= x0) && (c.q0.x - c.q1.z <= x1) && (c.q0.y + c.q1.w >= y0) &&
                      (c.q0.y - c.q1.w <= y1);
    const unsigned long long mask = __ballot(pass);
    if (mask == 0ull) return 0;
    const int n = (int)__popcll(mask);
    __syncthreads();       // previous chunk's LDS reads are done (single-wave block: orders LDS traffic only)
    if (pass) {
        const int slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        // conic pre-scaled by k = -0.5 log2(e): the loop evaluates q' = k q and alpha = o * exp2(q') (v_exp_f32 directly)
        s.r0[slot] = f4{c.q0.x, c.q0.y, QK * c.q0.z, (2.0f * QK) * c.q0.w};
        s.r1[slot] = f4{QK * c.q1.x, c.q1.y, c.q2.x, c.q2.y};
        s.bl[slot] = c.q2.z;
        if (WITH_ID) s.id[slot] = c.id;
    }
    if (threadIdx.x == 0) {                                   // null record: opacity 0 -> alpha 0, T unchanged
        s.r0[n] = f4{0.f, 0.f, 0.f, 0.f};
        s.r1[n] = f4{0.f, 0.f, 0.f, 0.f};
        s.bl[n] = 0.f;
    }
    __syncthreads();
    return n;
}

__device__ __forceinline__ int launch_priority(uint32_t b, uint32_t grid) {
    return b * 64u < grid ? 3 : (b * 16u < grid ? 2 : (b * 4u < grid ? 1 : 0));
}

__global__ __launch_bounds__(64) void raster_forward_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ ids,
                                                            const Rec64* __restrict__ rec, const uint32_t* __restrict__ order,
                                                            int tiles_x, int H, int W, float chi, float alpha_max,
                                                            float alpha_cutoff, float* __restrict__ image,
                                                            float* __restrict__ accum, uint32_t* __restrict__ visited,
                                                            WaveStats* __restrict__ stats) {
    __shared__ RasterStage s;
    const int lane = threadIdx.x;
    const uint32_t tile = order[blockIdx.x >> 1];
    const int half = blockIdx.x & 1;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int prio = launch_priority(blockIdx.x, gridDim.x);
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    const unsigned long long t_begin = stats ? __builtin_amdgcn_s_memtime() : 0ull;
    uint32_t st_chunks = 0, st_visited = 0;
    const int px = tx * 16 + (lane & 15);
    const int pya = ty * 16 + half * 8 + (lane >> 4), pyb = pya + 4;
    const bool va = (px < W) && (pya < H), vb = (px < W) && (pyb < H);
    const float fpx = (float)px;
    const v2f fpy = {(float)pya, (float)pyb};
    v2f T = {va ? 1.0f : 0.0f, vb ? 1.0f : 0.0f};
    v2f Cr = {0.f, 0.f}, Cg = {0.f, 0.f}, Cb = {0.f, 0.f};
    const float x0 = (float)(tx * 16), x1 = x0 + 15.0f, y0 = (float)(ty * 16 + half * 8), y1 = y0 + 7.0f;
    const uint2 rg = ranges[tile];
    const float chik = chi * QK;
    bool alive_any = __any(va || vb);
    uint32_t base = rg.x;
    Candidate cand;
    if (alive_any && base < rg.y) cand = fetch_candidate(lane, base, rg.y, ids, rec);
    while (alive_any && base < rg.y) {
        const int n = compact_candidates<false>(s, cand, x0, x1, y0, y1);
        base += BATCH;
        if (base < rg.y) cand = fetch_candidate(lane, base, rg.y, ids, rec);   // in flight during the loop below
        ++st_chunks;
        st_visited += (uint32_t)n;
        for (int j = 0; j < n; j += 2) {
            const f4 a0 = s.r0[j], b0 = s.r1[j], a1 = s.r0[j + 1], b1 = s.r1[j + 1];
            const float du0 = fpx - a0.x, du1 = fpx - a1.x;
            const v2f dv0 = fpy - a0.y, dv1 = fpy - a1.y;
            const v2f q0 = (a0.z * du0 * du0) + dv0 * ((a0.w * du0) + b0.x * dv0);        // k q  (k < 0)
            const v2f q1 = (a1.z * du1 * du1) + dv1 * ((a1.w * du1) + b1.x * dv1);
            const bool i00 = q0.x >= chik, i01 = q0.y >= chik, i10 = q1.x >= chik, i11 = q1.y >= chik;   // q <= chi
            if (__any(i00 || i01 || i10 || i11)) {
                const float cb0 = s.bl[j], cb1 = s.bl[j + 1];
                v2f g0, g1;
                g0.x = __builtin_amdgcn_exp2f(q0.x); g0.y = __builtin_amdgcn_exp2f(q0.y);
                g1.x = __builtin_amdgcn_exp2f(q1.x); g1.y = __builtin_amdgcn_exp2f(q1.y);
                v2f al0 = b0.y * g0, al1 = b1.y * g1;
                al0.x = fminf(al0.x, alpha_max); al0.y = fminf(al0.y, alpha_max);
                al1.x = fminf(al1.x, alpha_max); al1.y = fminf(al1.y, alpha_max);
                // alpha = 0 outside the chi-square clip and below the cutoff (one select for both)
                al0.x = (i00 && al0.x >= alpha_cutoff) ? al0.x : 0.0f; al0.y = (i01 && al0.y >= alpha_cutoff) ? al0.y : 0.0f;
                al1.x = (i10 && al1.x >= alpha_cutoff) ? al1.x : 0.0f; al1.y = (i11 && al1.y >= alpha_cutoff) ? al1.y : 0.0f;
                v2f w0 = al0 * T;
                w0.x = (T.x > 5e-5f) ? w0.x : 0.0f; w0.y = (T.y > 5e-5f) ? w0.y : 0.0f;
                T = T - al0 * T;
                v2f w1 = al1 * T;
                w1.x = (T.x > 5e-5f) ? w1.x : 0.0f; w1.y = (T.y > 5e-5f) ? w1.y : 0.0f;
                T = T - al1 * T;
                Cr += w0 * b0.z; Cg += w0 * b0.w; Cb += w0 * cb0;
                Cr += w1 * b1.z; Cg += w1 * b1.w; Cb += w1 * cb1;
            }
        }
        alive_any = __any(T.x > 5e-5f || T.y > 5e-5f);        // once per chunk: dead pixels stay dead
    }
    if (visited && lane == 0) visited[tile * 2 + half] = st_visited;
    if (stats && lane == 0)
        stats[tile * 2 + half] = WaveStats{rg.y - rg.x, st_chunks, st_visited, (uint32_t)(__builtin_amdgcn_s_memtime() - t_begin)};
    if (va) {
        const int64_t o = ((int64_t)pya * W + px) * 3;
        image[o + 0] = fminf(fmaxf(Cr.x, 0.0f), 1.0f); image[o + 1] = fminf(fmaxf(Cg.x, 0.0f), 1.0f);
        image[o + 2] = fminf(fmaxf(Cb.x, 0.0f), 1.0f);
        if (accum) { accum[o + 0] = Cr.x; accum[o + 1] = Cg.x; accum[o + 2] = Cb.x; }
    }
    if (vb) {
        const int64_t o = ((int64_t)pyb * W + px) * 3;
        image[o + 0] = fminf(fmaxf(Cr.y, 0.0f), 1.0f); image[o + 1] = fminf(fmaxf(Cg.y, 0.0f), 1.0f);
        image[o + 2] = fminf(fmaxf(Cb.y, 0.0f), 1.0f);
        if (accum) { accum[o + 0] = Cr.y; accum[o + 1] = Cg.y; accum[o + 2] = Cb.y; }
    }
}

// Nine per-lane partial sums -> their wave totals, total k delivered in lane k (k = 0..8) of row 0.
// Reduce-scatter: two quad steps halve the number of live values (9 -> 5 -> 3) while summing over the quad (select +
// DPP quad_perm add); then each lane's 3 values are summed over the 4 quads of its row (row_ror 4, 8), over the rows
// (permlane16/32 swaps), and two DPP row shifts move values 1 and 2 next to value 0.  ~44 instructions against ~94
// for nine independent butterfly reductions + readlane + select.
__device__ __forceinline__ float reduce9_to_lanes(const float (&v)[9], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float u[5], t[3];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float lo = v[2 * i], hi = (2 * i + 1 < 9) ? v[2 * i + 1] : 0.0f;
        const float keep = b0 ? hi : lo, send = b0 ? lo : hi;
        u[i] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, false));
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float lo = u[2 * j], hi = (2 * j + 1 < 5) ? u[2 * j + 1] : 0.0f;
        const float keep = b1 ? hi : lo, send = b1 ? lo : hi;
        t[j] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xF, 0xF, false));
    }
    // lane (b1, b0) now holds, in t[j], the quad sum of component 4 j + 2 b1 + b0
#pragma unroll
    for (int j = 0; j < 3; ++j) t[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t[j]), 0x124, 0xF, 0xF, false));
#pragma unroll
    for (int j = 0; j < 3; ++j) t[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t[j]), 0x128, 0xF, 0xF, false));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(t[j]), __float_as_uint(t[j]), false, false);
        t[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[j]), __float_as_uint(t[j]), false, false);
        t[j] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    // lanes 4-7 take t[1] from lanes 0-3 (row_shr 4, bank 1), lanes 8-11 take t[2] from lanes 0-3 (row_shr 8, bank 2)
    int out = __builtin_amdgcn_update_dpp(__float_as_int(t[0]), __float_as_int(t[1]), 0x114, 0xF, 0x2, false);
    out = __builtin_amdgcn_update_dpp(out, __float_as_int(t[2]), 0x118, 0xF, 0x4, false);
    return __int_as_float(out);
}

// K7: same traversal as K6 (identical T_i and alive decisions).  For pixel p and Gaussian i:
//   d alpha_i = alive_i T_i (c_i . Gc) - (sum_{k>i} w_k (c_k . Gc)) / (1 - alpha_i),
// the suffix sum being (total - running prefix), total = Gc . C_unclamped, Gc = dL/dO masked by the output clamp.
// Nine per-Gaussian sums are reduced over the wave and added to grad2d[id][0..8] by lanes 0..8 (one 36-byte atomic
// request per (half tile, Gaussian) pair that actually touched a pixel).
__global__ __launch_bounds__(64) void raster_backward_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ ids,
                                                             const Rec64* __restrict__ rec, const uint32_t* __restrict__ order,
                                                             int tiles_x, int H, int W, float chi, float alpha_max,
                                                             float alpha_cutoff, const float* __restrict__ accum,
                                                             const float* __restrict__ gimg, float* __restrict__ grad2d,
                                                             WaveStats* __restrict__ stats, int ablate) {
    __shared__ RasterStage s;
    const int lane = threadIdx.x;
    const uint32_t tile = order[blockIdx.x >> 1];
    const int half = blockIdx.x & 1;
    const uint2 rg = ranges[tile];
    if (rg.x >= rg.y) return;
    const int prio = launch_priority(blockIdx.x, gridDim.x);
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    const unsigned long long t_begin = stats ? __builtin_amdgcn_s_memtime() : 0ull;
    uint32_t st_chunks = 0, st_visited = 0;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int px = tx * 16 + (lane & 15);
    const int pya = ty * 16 + half * 8 + (lane >> 4), pyb = pya + 4;
    const bool va = (px < W) && (pya < H), vb = (px < W) && (pyb < H);
    const float fpx = (float)px;
    const v2f fpy = {(float)pya, (float)pyb};
    v2f T = {va ? 1.0f : 0.0f, vb ? 1.0f : 0.0f};
    v2f Gr = {0.f, 0.f}, Gg = {0.f, 0.f}, Gb = {0.f, 0.f}, suffix = {0.f, 0.f};
    {
        float g[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, sfx[2] = {0.f, 0.f};
        const bool vv[2] = {va, vb};
        const int py[2] = {pya, pyb};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (vv[k]) {
                const int64_t o = ((int64_t)py[k] * W + px) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float cu = accum[o + c];
                    // clamp(C, 0, 1) passes the gradient where 0 <= C <= 1 (render.py:410)
                    const float gv = (cu >= 0.0f && cu <= 1.0f) ? gimg[o + c] : 0.0f;
                    g[k][c] = gv;
                    sfx[k] += gv * cu;
                }
            }
        }
        Gr = v2f{g[0][0], g[1][0]}; Gg = v2f{g[0][1], g[1][1]}; Gb = v2f{g[0][2], g[1][2]};
        suffix = v2f{sfx[0], sfx[1]};
    }
    const float x0 = (float)(tx * 16), x1 = x0 + 15.0f, y0 = (float)(ty * 16 + half * 8), y1 = y0 + 7.0f;
    const float chik = chi * QK;
    bool alive_any = __any(va || vb);
    uint32_t base = rg.x;
    Candidate cand;
    if (alive_any) cand = fetch_candidate(lane, base, rg.y, ids, rec);
    while (alive_any && base < rg.y) {
        const int n = compact_candidates<true>(s, cand, x0, x1, y0, y1);
        base += BATCH;
        if (base < rg.y) cand = fetch_candidate(lane, base, rg.y, ids, rec);   // in flight during the loop below
        ++st_chunks;
        st_visited += (uint32_t)n;
        for (int j = 0; j < n; ++j) {
            const f4 a = s.r0[j], b = s.r1[j];
            const float du = fpx - a.x;
            const v2f dv = fpy - a.y;
            const float c0 = a.z * du * du, c1 = a.w * du;
            const v2f q = c0 + dv * (c1 + b.x * dv);                                   // k q  (k < 0)
            const bool i0 = q.x >= chik, i1 = q.y >= chik;                              // q <= chi
            if (!__any(i0 || i1)) continue;
            const float cbl = s.bl[j], go = b.y;
            v2f g;
            g.x = __builtin_amdgcn_exp2f(q.x);
            g.y = __builtin_amdgcn_exp2f(q.y);
            const v2f og = go * g;
            v2f al;
            al.x = fminf(og.x, alpha_max); al.y = fminf(og.y, alpha_max);
            al.x = (i0 && al.x >= alpha_cutoff) ? al.x : 0.0f; al.y = (i1 && al.y >= alpha_cutoff) ? al.y : 0.0f;
            const bool act0 = (T.x > 5e-5f) && (al.x > 0.0f), act1 = (T.y > 5e-5f) && (al.y > 0.0f);
            if (__any(act0 || act1)) {
                v2f w = al * T;
                w.x = act0 ? w.x : 0.0f; w.y = act1 ? w.y : 0.0f;
                const v2f sdot = b.z * Gr + b.w * Gg + cbl * Gb;
                const v2f ar = w * Gr, ag = w * Gg, ab = w * Gb;
                suffix -= w * sdot;                                            // now the sum over k > i
                v2f om = 1.0f - al;
                om.x = __builtin_amdgcn_rcpf(om.x); om.y = __builtin_amdgcn_rcpf(om.y);   // 1 - alpha >= 0.01
                v2f dal = T * sdot - suffix * om;
                // clamp_max passes the gradient where o g <= alpha_max (render.py:372); alpha > 0 implies q <= chi
                dal.x = (act0 && og.x <= alpha_max) ? dal.x : 0.0f;
                dal.y = (act1 && og.y <= alpha_max) ? dal.y : 0.0f;
                const v2f ao = dal * g;
                const v2f dq = (-0.5f * go) * (g * dal);                       // dL/dq (q un-scaled)
                const v2f dvq = dv * dq;
                const float dqs = dq.x + dq.y, dvqs = dvq.x + dvq.y;
                const v2f aA22 = dv * dvq;
                // the staged conic is k * (A11, 2 A12, A22): undo the scale with 1/k for the two position gradients
                const float r[9] = {-(2.0f * a.z * du * dqs + a.w * dvqs) * (1.0f / QK),        // d u
                                    -(a.w * du * dqs + 2.0f * b.x * dvqs) * (1.0f / QK),        // d v
                                    du * du * dqs,                                // d A11
                                    2.0f * du * dvqs,                             // d A12
                                    aA22.x + aA22.y,                              // d A22
                                    ao.x + ao.y,                                  // d opacity
                                    ar.x + ar.y, ag.x + ag.y, ab.x + ab.y};       // d rgb
                float mine;
                if (ablate & 2) {
                    mine = r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6] + r[7] + r[8];
                } else {
                    mine = reduce9_to_lanes(r, lane);
                }
                if (ablate & 1) {
                    asm volatile("" ::"v"(mine));
                } else if (lane < 9) {
                    atomicAdd(&grad2d[(int64_t)s.id[j] * 16 + lane], mine);
                }
            }
            T = T - al * T;
        }
        alive_any = __any(T.x > 5e-5f || T.y > 5e-5f);        // once per chunk: dead pixels stay dead
    }
    if (stats && lane == 0)
        stats[tile * 2 + half] = WaveStats{rg.y - rg.x, st_chunks, st_visited, (uint32_t)(__builtin_amdgcn_s_memtime() - t_begin)};
}

// ---- K8 ------------------------------------------------------------------------------------------
// Same per-wave LDS staging as K1 for the inputs; the gradients go the other way: every lane writes its rows into LDS
// (f_rest gradient over the staged f_rest: each coefficient is read before its gradient is written) and the wave
// stores the 64 rows with coalesced 16-byte accesses.  Direct per-lane stores of a [N,45] gradient wrote 3.8x the
// algorithmic bytes (partial lines evicted before they filled).
struct ShEmitLds {
    float* dc;
    float* rest;
    __device__ __forceinline__ void operator()(int k, int ch, float v) const {
        if (k == 0) dc[ch] = v; else rest[ch * 15 + (k - 1)] = v;
    }
};

template <bool FUSED>
__global__ __launch_bounds__(64) void project_backward_kernel(gsplat_gaussians g, const Camera* __restrict__ camp, ViewK vk,
                                                              const uint32_t* __restrict__ tiles, const float* __restrict__ grad2d,
                                                              gsplat_gaussian_grads out) {
    __shared__ ProjectLds s;
    __shared__ float s_dc[FUSED ? 64 * 3 : 4];
    __shared__ float s_rest[FUSED ? 64 * 45 : 4];
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + lane;
    const Camera cam = *camp;
    const bool vis = (i < g.n) && tiles[i] != 0;
    const bool any_vis = __any(vis);
    float r9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (any_vis) {
        stage_geometry<FUSED>(s, g, row0, lane);
        if (FUSED) {
            stage_rows<3>(s_dc, g.f_dc, row0, g.n, lane);
            stage_rows<45>(s_rest, g.f_rest, row0, g.n, lane);
        }
        if (vis) {
            const f4 g0 = *reinterpret_cast<const f4*>(grad2d + i * 16), g1 = *reinterpret_cast<const f4*>(grad2d + i * 16 + 4);
            r9[0] = g0.x; r9[1] = g0.y; r9[2] = g0.z; r9[3] = g0.w; r9[4] = g1.x; r9[5] = g1.y; r9[6] = g1.z; r9[7] = g1.w;
            r9[8] = grad2d[i * 16 + 8];
        }
    }
    __syncthreads();
    GradOut go;
    if (vis) {
        const GaussIn in = gauss_from_lds<FUSED>(s, lane);
        go = project_backward_core(in, FUSED, ShCoefLds{s_dc + lane * 3, s_rest + lane * 45},
                                   ShEmitLds{s_dc + lane * 3, s_rest + lane * 45}, cam, vk, true, r9);
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) { go.p[k] = 0.f; go.sr[k] = 0.f; go.col[k] = 0.f; }
#pragma unroll
        for (int k = 0; k < 4; ++k) go.qr[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) go.S9[k] = 0.f;
        go.o_raw = 0.f;
        if (FUSED) {
#pragma unroll
            for (int k = 0; k < 3; ++k) s_dc[lane * 3 + k] = 0.f;
            for (int k = 0; k < 45; ++k) s_rest[lane * 45 + k] = 0.f;
        }
    }
    __syncthreads();      // every lane has read its inputs: the geometry buffers can take the gradients
#pragma unroll
    for (int k = 0; k < 3; ++k) s.pos[lane * 3 + k] = go.p[k];
    s.opa[lane] = go.o_raw;
    if (FUSED) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s.a[lane * 4 + k] = go.qr[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) s.b[lane * 3 + k] = go.sr[k];
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) s.a[lane * 9 + k] = go.S9[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) s.b[lane * 3 + k] = go.col[k];
    }
    __syncthreads();
    unstage_rows<3>(out.pos, s.pos, row0, g.n, lane);
    unstage_rows<1>(out.opacity_raw, s.opa, row0, g.n, lane);
    if (FUSED) {
        unstage_rows<4>(out.q_raw, s.a, row0, g.n, lane);
        unstage_rows<3>(out.scale_raw, s.b, row0, g.n, lane);
        unstage_rows<3>(out.f_dc, s_dc, row0, g.n, lane);
        unstage_rows<45>(out.f_rest, s_rest, row0, g.n, lane);
    } else {
        unstage_rows<9>(out.sigma, s.a, row0, g.n, lane);
        unstage_rows<3>(out.color, s.b, row0, g.n, lane);
    }
}

// ---- stand-alone ops -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_sigma_kernel(int64_t n, const float* __restrict__ sr, const float* __restrict__ qr,
                                                          float* __restrict__ sigma) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) build_sigma_one(i, sr, qr, sigma);
}
__global__ __launch_bounds__(256) void build_sigma_backward_kernel(int64_t n, const float* __restrict__ sr, const float* __restrict__ qr,
                                                                   const float* __restrict__ gs, float* __restrict__ gsr,
                                                                   float* __restrict__ gq) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) build_sigma_backward_one(i, sr, qr, gs, gsr, gq);
}
__global__ __launch_bounds__(256) void evaluate_sh_kernel(int64_t n, const float* __restrict__ dc, const float* __restrict__ rest,
                                                          const float* __restrict__ pts, const float* __restrict__ c2w,
                                                          float* __restrict__ color) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float m[16];
    for (int k = 0; k < 16; ++k) m[k] = c2w[k];
    Camera cam;
    build_camera(m, cam);
    evaluate_sh_one(i, dc, rest, pts, cam, color);
}
__global__ __launch_bounds__(256) void evaluate_sh_backward_kernel(int64_t n, const float* __restrict__ dc, const float* __restrict__ rest,
                                                                   const float* __restrict__ pts, const float* __restrict__ c2w,
                                                                   const float* __restrict__ gcol, float* __restrict__ gdc,
                                                                   float* __restrict__ grest, float* __restrict__ gpts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float m[16];
    for (int k = 0; k < 16; ++k) m[k] = c2w[k];
    Camera cam;
    build_camera(m, cam);
    evaluate_sh_backward_one(i, dc, rest, pts, cam, gcol, gdc, grest, gpts);
}

inline unsigned blocks256(int64_t n) { return (unsigned)((n + 255) / 256); }
inline unsigned blocks64(int64_t n) { return (unsigned)((n + 63) / 64); }

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// ======================================================================================================
// C ABI
// ======================================================================================================
extern "C" {

int gsplat_abi_version(void) { return GSPLAT_ABI_VERSION; }

// Diagnostics only (not declared in include/gsplat_mi355x.h): register device buffers of 2 * tiles * 16 bytes each that
// the raster kernels fill with per-wave statistics; pass NULL to switch the statistics off again.
void gsplat_debug_set_stats(void* fwd, void* bwd) { g_stats_fwd = (WaveStats*)fwd; g_stats_bwd = (WaveStats*)bwd; }
void gsplat_debug_set_ablation(int bits) { g_ablate = bits; }

const char* gsplat_last_error(void) { return g_err; }

int gsplat_classify_counts(const gsplat_counts* c) {
    if (!c) return GSPLAT_SCENE_ALL_CULLED;
    if (c->n_survivors == 0) return GSPLAT_SCENE_ALL_CULLED;       // render.py:109-112,139-142,190-193
    if (c->n_visible == 0) return GSPLAT_SCENE_ALL_OFFSCREEN;      // render.py:235-236
    return GSPLAT_SCENE_OK;
}

int64_t gsplat_project_state_bytes(int64_t n) { return carve_project(nullptr, n > 0 ? n : 1).bytes; }

int64_t gsplat_project_scratch_bytes(int64_t n) { return up((int64_t)scan_temp_bytes(n)) + ALIGN; }

int64_t gsplat_bin_state_bytes(int64_t n_pairs, const gsplat_view* v) {
    if (!v) return -1;
    const int64_t nt = (int64_t)((v->W + 15) / 16) * ((v->H + 15) / 16);
    return carve_bin(nullptr, n_pairs, nt).bytes;
}

int64_t gsplat_bin_scratch_bytes(int64_t n, int64_t n_pairs) {
    (void)n;
    return carve_bin_scratch(nullptr, n_pairs).bytes;
}

int gsplat_project(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, void* project_state, void* scratch,
                   int64_t scratch_bytes, gsplat_counts* counts_host, void* stream_) {
    bool fused = false;
    int rc = check_gaussians(g, &fused);
    if (rc) return rc;
    if ((rc = check_view(v))) return rc;
    if (!c2w || !project_state) return fail(GSPLAT_ERR_BAD_ARG, "c2w / project_state is NULL");
    hipStream_t st = (hipStream_t)stream_;
    const int64_t n = g->n;
    ProjectState ps = carve_project(project_state, n > 0 ? n : 1);
    const ViewK vk = make_viewk(*v);
    hipLaunchKernelGGL(camera_kernel, dim3(1), dim3(COUNT_SHARDS), 0, st, c2w, ps.cam, ps.counts, ps.shards);
    LAUNCH_CHECK("camera_kernel");
    if (n > 0) {
        Records out{ps.rec, ps.rect, ps.depth, ps.tiles};
        if (fused)
            hipLaunchKernelGGL(project_kernel<true>, dim3(blocks64(n)), dim3(64), 0, st, *g, ps.cam, vk, out, ps.shards);
        else
            hipLaunchKernelGGL(project_kernel<false>, dim3(blocks64(n)), dim3(64), 0, st, *g, ps.cam, vk, out, ps.shards);
        LAUNCH_CHECK("project_kernel");
        size_t need = scan_temp_bytes(n);
        if (!scratch || (int64_t)need > scratch_bytes) return fail(GSPLAT_ERR_WORKSPACE, "project scratch too small");
        HIP_TRY(rocprim::inclusive_scan(scratch, need, ps.tiles, ps.offsets, (size_t)n, rocprim::plus<uint32_t>(), st));
        hipLaunchKernelGGL(finish_counts_kernel, dim3(1), dim3(COUNT_SHARDS), 0, st, ps.offsets, n, ps.shards, ps.counts);
        LAUNCH_CHECK("finish_counts_kernel");
    }
    if (counts_host) HIP_TRY(hipMemcpyAsync(counts_host, ps.counts, sizeof(gsplat_counts), hipMemcpyDeviceToHost, st));
    return GSPLAT_OK;
}

int gsplat_bin(int64_t n, int64_t n_pairs, const gsplat_view* v, const void* project_state, void* bin_state, void* scratch,
               int64_t scratch_bytes, void* stream_) {
    int rc = check_view(v);
    if (rc) return rc;
    if (n < 0 || n_pairs < 0 || n_pairs > 0xFFFFFFFFLL) return fail(GSPLAT_ERR_BAD_ARG, "n / n_pairs out of range");
    if (!project_state || !bin_state) return fail(GSPLAT_ERR_BAD_ARG, "state is NULL");
    hipStream_t st = (hipStream_t)stream_;
    const ViewK vk = make_viewk(*v);
    const int64_t nt = (int64_t)vk.tiles_x * vk.tiles_y;
    ProjectState ps = carve_project((void*)project_state, n > 0 ? n : 1);
    BinState bs = carve_bin(bin_state, n_pairs, nt);
    HIP_TRY(hipMemsetAsync(bs.ranges, 0, nt * sizeof(uint2), st));
    if (n_pairs == 0 || n == 0) {
        hipLaunchKernelGGL(order_tiles_kernel, dim3(1), dim3(1024), 0, st, (int)nt, bs.ranges, (const uint32_t*)nullptr, bs.order_fwd);
        LAUNCH_CHECK("order_tiles_kernel");
        return GSPLAT_OK;
    }
    BinScratch sc = carve_bin_scratch(scratch, n_pairs);
    if (!scratch || sc.bytes > scratch_bytes) return fail(GSPLAT_ERR_WORKSPACE, "bin scratch too small");
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(blocks256(n)), dim3(256), 0, st, n, ps.depth, ps.rect, ps.tiles, ps.offsets, vk.tiles_x,
                       n_pairs, sc.keys_in, sc.vals_in);
    LAUNCH_CHECK("emit_pairs_kernel");
    // F12 (tile part): radix sort on the tile-id bits only
    unsigned tile_bits = 1;
    while ((1LL << tile_bits) < nt) ++tile_bits;
    size_t tb = sc.temp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(sc.temp, tb, sc.keys_in, sc.keys_out, sc.vals_in, sc.vals_out, (size_t)n_pairs, 0u, tile_bits,
                                      st));
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(blocks256(n_pairs)), dim3(256), 0, st, n_pairs, sc.keys_out, bs.ranges);
    LAUNCH_CHECK("tile_ranges_kernel");
    // F9 + F12 (depth part): per-tile sort by (depth, index)
    hipLaunchKernelGGL((tile_sort_kernel<128, TILE_SORT_SMALL, false>), dim3((unsigned)nt), dim3(128), 0, st, bs.ranges, sc.vals_out,
                       bs.sorted_ids);
    LAUNCH_CHECK("tile_sort_kernel<small>");
    hipLaunchKernelGGL((tile_sort_kernel<256, TILE_SORT_LDS, true>), dim3((unsigned)nt), dim3(256), 0, st, bs.ranges, sc.vals_out,
                       bs.sorted_ids);
    LAUNCH_CHECK("tile_sort_kernel<large>");
    hipLaunchKernelGGL(order_tiles_kernel, dim3(1), dim3(1024), 0, st, (int)nt, bs.ranges, (const uint32_t*)nullptr, bs.order_fwd);
    LAUNCH_CHECK("order_tiles_kernel");
    return GSPLAT_OK;
}

int gsplat_rasterize_forward(int64_t n, int64_t n_pairs, const gsplat_view* v, const void* project_state, const void* bin_state,
                             float* image, float* accum, void* stream_) {
    int rc = check_view(v);
    if (rc) return rc;
    if (!project_state || !bin_state || !image) return fail(GSPLAT_ERR_BAD_ARG, "state / image is NULL");
    hipStream_t st = (hipStream_t)stream_;
    const ViewK vk = make_viewk(*v);
    const int64_t nt = (int64_t)vk.tiles_x * vk.tiles_y;
    ProjectState ps = carve_project((void*)project_state, n > 0 ? n : 1);
    BinState bs = carve_bin((void*)bin_state, n_pairs, nt);
    hipLaunchKernelGGL(raster_forward_kernel, dim3((unsigned)(2 * nt)), dim3(64), 0, st, bs.ranges, bs.sorted_ids, ps.rec,
                       bs.order_fwd, vk.tiles_x, vk.H, vk.W, vk.chi_clip, vk.alpha_max, vk.alpha_cutoff, image, accum,
                       accum ? bs.visited : (uint32_t*)nullptr, g_stats_fwd);
    LAUNCH_CHECK("raster_forward_kernel");
    if (accum) {   // a backward pass will follow: order it by the work the forward actually did
        hipLaunchKernelGGL(order_tiles_kernel, dim3(1), dim3(1024), 0, st, (int)nt, bs.ranges, bs.visited, bs.order_bwd);
        LAUNCH_CHECK("order_tiles_kernel");
    }
    return GSPLAT_OK;
}

int gsplat_rasterize_backward(int64_t n, int64_t n_pairs, const gsplat_view* v, const void* project_state, const void* bin_state,
                              const float* accum, const float* grad_image, float* grad2d, void* stream_) {
    int rc = check_view(v);
    if (rc) return rc;
    if (!project_state || !bin_state || !accum || !grad_image || !grad2d) return fail(GSPLAT_ERR_BAD_ARG, "NULL argument");
    hipStream_t st = (hipStream_t)stream_;
    const ViewK vk = make_viewk(*v);
    const int64_t nt = (int64_t)vk.tiles_x * vk.tiles_y;
    ProjectState ps = carve_project((void*)project_state, n > 0 ? n : 1);
    BinState bs = carve_bin((void*)bin_state, n_pairs, nt);
    HIP_TRY(hipMemsetAsync(grad2d, 0, (size_t)(n > 0 ? n : 0) * 16 * sizeof(float), st));
    if (n == 0 || n_pairs == 0) return GSPLAT_OK;
    hipLaunchKernelGGL(raster_backward_kernel, dim3((unsigned)(2 * nt)), dim3(64), 0, st, bs.ranges, bs.sorted_ids, ps.rec,
                       bs.order_bwd, vk.tiles_x, vk.H, vk.W, vk.chi_clip, vk.alpha_max, vk.alpha_cutoff, accum, grad_image,
                       grad2d, g_stats_bwd, g_ablate);
    LAUNCH_CHECK("raster_backward_kernel");
    return GSPLAT_OK;
}

int gsplat_project_backward(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, const void* project_state,
                            const float* grad2d, const gsplat_gaussian_grads* out, void* stream_) {
    bool fused = false;
    int rc = check_gaussians(g, &fused);
    if (rc) return rc;
    if ((rc = check_view(v))) return rc;
    if (!c2w || !project_state || !grad2d || !out) return fail(GSPLAT_ERR_BAD_ARG, "NULL argument");
    if (g->n == 0) return GSPLAT_OK;
    if (!out->pos || !out->opacity_raw) return fail(GSPLAT_ERR_BAD_ARG, "grad pos / opacity_raw is NULL");
    if (fused && !(out->scale_raw && out->q_raw && out->f_dc && out->f_rest)) return fail(GSPLAT_ERR_BAD_ARG, "fused grads incomplete");
    if (!fused && !(out->color && out->sigma)) return fail(GSPLAT_ERR_BAD_ARG, "grad color / sigma is NULL");
    hipStream_t st = (hipStream_t)stream_;
    ProjectState ps = carve_project((void*)project_state, g->n);
    const ViewK vk = make_viewk(*v);
    if (fused)
        hipLaunchKernelGGL(project_backward_kernel<true>, dim3(blocks64(g->n)), dim3(64), 0, st, *g, ps.cam, vk, ps.tiles, grad2d, *out);
    else
        hipLaunchKernelGGL(project_backward_kernel<false>, dim3(blocks64(g->n)), dim3(64), 0, st, *g, ps.cam, vk, ps.tiles, grad2d, *out);
    LAUNCH_CHECK("project_backward_kernel");
    return GSPLAT_OK;
}

int gsplat_build_sigma(int64_t n, const float* scale_raw, const float* q_raw, float* sigma, void* stream_) {
    if (n < 0) return fail(GSPLAT_ERR_BAD_ARG, "n < 0");
    if (n == 0) return GSPLAT_OK;
    if (!scale_raw || !q_raw || !sigma) return fail(GSPLAT_ERR_BAD_ARG, "NULL argument");
    hipLaunchKernelGGL(build_sigma_kernel, dim3(blocks256(n)), dim3(256), 0, (hipStream_t)stream_, n, scale_raw, q_raw, sigma);
    LAUNCH_CHECK("build_sigma_kernel");
    return GSPLAT_OK;
}

int gsplat_build_sigma_backward(int64_t n, const float* scale_raw, const float* q_raw, const float* grad_sigma, float* grad_scale_raw,
                                float* grad_q_raw, void* stream_) {
    if (n < 0) return fail(GSPLAT_ERR_BAD_ARG, "n < 0");
    if (n == 0) return GSPLAT_OK;
    if (!scale_raw || !q_raw || !grad_sigma || !grad_scale_raw || !grad_q_raw) return fail(GSPLAT_ERR_BAD_ARG, "NULL argument");
    hipLaunchKernelGGL(build_sigma_backward_kernel, dim3(blocks256(n)), dim3(256), 0, (hipStream_t)stream_, n, scale_raw, q_raw,
                       grad_sigma, grad_scale_raw, grad_q_raw);
    LAUNCH_CHECK("build_sigma_backward_kernel");
    return GSPLAT_OK;
}

int gsplat_evaluate_sh(int64_t n, const float* f_dc, const float* f_rest, const float* points, const float* c2w, float* color,
                       void* stream_) {
    if (n < 0) return fail(GSPLAT_ERR_BAD_ARG, "n < 0");
    if (n == 0) return GSPLAT_OK;
    if (!f_dc || !f_rest || !points || !c2w || !color) return fail(GSPLAT_ERR_BAD_ARG, "NULL argument");
    hipLaunchKernelGGL(evaluate_sh_kernel, dim3(blocks256(n)), dim3(256), 0, (hipStream_t)stream_, n, f_dc, f_rest, points, c2w, color);
    LAUNCH_CHECK("evaluate_sh_kernel");
    return GSPLAT_OK;
}

int gsplat_evaluate_sh_backward(int64_t n, const float* f_dc, const float* f_rest, const float* points, const float* c2w,
                                const float* grad_color, float* grad_f_dc, float* grad_f_rest, float* grad_points, void* stream_) {
    if (n < 0) return fail(GSPLAT_ERR_BAD_ARG, "n < 0");
    if (n == 0) return GSPLAT_OK;
    if (!f_dc || !f_rest || !points || !c2w || !grad_color || !grad_f_dc || !grad_f_rest || !grad_points)
        return fail(GSPLAT_ERR_BAD_ARG, "NULL argument");
    hipLaunchKernelGGL(evaluate_sh_backward_kernel, dim3(blocks256(n)), dim3(256), 0, (hipStream_t)stream_, n, f_dc, f_rest, points, c2w,
                       grad_color, grad_f_dc, grad_f_rest, grad_points);
    LAUNCH_CHECK("evaluate_sh_backward_kernel");
    return GSPLAT_OK;
}

}  // extern "C"
