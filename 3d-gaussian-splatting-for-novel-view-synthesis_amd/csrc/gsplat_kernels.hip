// gsplat_kernels.hip -- MI355X (gfx950, wave64) kernels and the C ABI of include/gsplat_mi355x.h.
//
// Pipeline (stage ids of SURVEY.md §8a in brackets):
//   K1  project_kernel         per Gaussian: camera block, culls, EWA, eigen clamp, conic, rectangles; optionally the SH colour;
//                              counters totalled by its last wave                              [F1-F8, F10, F13 (+F3)]
//   K1b colour_kernel          SH colour of the binned Gaussians as a pass of its own (host waits for the counters)   [F3]
//   K3  bin_count / bin_scatter / split_count / split_scatter_kernel   two-level counting sort of the pairs by list  [F11, F12]
//   K4  list_sort_kernel       per-list depth sort in LDS (lists of 8192+: in global memory)    [F9, F12]
//   K5  plan_kernel            longest-first launch order of the lists, sort size classes
//   K6  raster_forward_kernel  one wave64 per 16 x 8-pixel list = eight 8-lane groups, one 4 x 4 sub-tile each   [F14, F15]
//   K7  raster_backward_kernel same traversal, analytic gradients, sums per group -> LDS slots -> one atomic per pair   [B1]
//       (+ tile_block_sum / pair_base / pair_reduce_kernel: deterministic mode)
//   K8  project_backward_kernel chain rule to the reference's input tensors                     [B2, B3]
//
// Everything is hand-written HIP for gfx950; no library kernels.  No MFMA: there is no dense contraction on this path.
// No CPU fallback: without a GPU every entry point returns GSPLAT_ERR_HIP.
#include <cstdio>
#include <cstring>

#include <hip/hip_runtime.h>
#include <algorithm>

#include "gs_body.h"
#include "gs_adam.h"

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
    int64_t n_pairs;         // the reference's (tile, Gaussian) pairs (F11)
    int32_t max_tiles, reserved;
    int64_t n_binned;        // (list, Gaussian) pairs actually binned
};
static_assert(sizeof(DevCounts) == sizeof(gsplat_counts), "counts layout");

constexpr int COUNT_SHARDS = 256;     // per-wave counters are spread over 256 cache lines (same-address atomics serialise)
struct alignas(64) CountShard { int32_t survivors, visible, max_tiles; uint32_t ref_pairs, bin_pairs, arrived; int32_t pad[10]; };
// The caller's `scratch` of gsplat_project: counters that must be ZERO when a call starts.  The caller zeroes the block once;
// the wave of the projection kernel that finishes last adds the shards up and clears them again (no clearing kernel, no
// totals kernel: the front of the pipeline is latency-bound and every dependent launch costs ~5 us).
struct CounterBlock {
    CountShard shards[COUNT_SHARDS];
    uint32_t done;            // shards whose waves have all added their counts (arrivals are counted per shard first: ONE word takes
                              // only ~88 returning atomics per microsecond, 15 625 waves on it cost 0.18 ms)
    uint32_t pad[15];
};

// A "list" is the depth-ordered set of Gaussians of one 16 x 8-pixel region: the unit one wave64 rasterises.
// Binning is a two-level counting sort: (list, Gaussian) pairs go to coarse bins of 64 consecutive lists first
// (bin_count_kernel / bin_scatter_kernel, blocks of 2048 Gaussians with an LDS histogram, one global atomic per block and
// bin), then every bin is split into its 64 lists (split_count_kernel / split_scatter_kernel), then every list is sorted by depth.
constexpr int BIN_SHIFT = 6;                 // 64 lists per coarse bin
constexpr int BIN_GAUSS = 2048;              // Gaussians per block of bin_count_kernel / bin_scatter_kernel
constexpr int MAX_BINS = 8192;               // LDS histogram of the two kernels (32 KB): images up to 8192 x 8192 / 128 lists
constexpr int SPLIT_CHUNK = 4096;            // pairs per block of split_count_kernel / split_scatter_kernel
// Pair payload (64 bit): float_bits(z) << 32 | list index inside its bin << ID_BITS | Gaussian index.  z > 0, so the
// bit pattern orders like the value; inside one list the middle field is constant: sorting payloads = (depth, index) order.
constexpr int ID_BITS = 32 - BIN_SHIFT;      // 26: up to 67 M Gaussians per call
constexpr uint32_t ID_MASK = (1u << ID_BITS) - 1u;

struct ProjectState {
    Camera* cam;
    DevCounts* counts;
    Rec64* rec;
    u2* rect;                // per Gaussian: inclusive rectangle of lists
    float* depth;
    uint32_t* tiles;         // per Gaussian: number of lists (0 = contributes nowhere)
    uint32_t* mask;          // per Gaussian: which lists of the rectangle (ellipse / list test; all ones above 32 lists)
    uint32_t* bin_total;     // [3 x bins] pairs per coarse bin of the small Gaussians | of the large ones (rectangles of more than 32
                             //            lists: bin_total + bins) | the large ones' scatter cursor (bin_total + 2 bins); zero per frame
    uint32_t* bin_start;     // [bins + 1] exclusive prefix of bin_total
    uint32_t* block_off;     // [blocks x bins] where a block's pairs start inside a bin
    uint32_t* list_count;    // [bins x 64] pairs per list (split_count_kernel)
    uint2* ranges;           // [lists] start, end in the pair arrays
    uint32_t* order;         // [lists] launch order: longest list first
    uint32_t* class_bounds;  // [8] boundaries of the sort size classes inside `order`
    float* kj;               // [n][12] fused inputs, GSPLAT_PROJECT_SAVE_SH_JACOBIAN: d rgb / d logit (3), d logit / d position (9)
    uint32_t* big_flag;      // [ceil(n / 64)] does this wave of the projection kernel hold a large Gaussian (rectangle of more than 32
                             // lists)?  written by EVERY wave in every frame: the binning kernels' big blocks look here before anything else
    int64_t bytes;
};

inline int64_t n_lists(const gsplat_view* v) { return (int64_t)((v->W + LIST_W - 1) / LIST_W) * ((v->H + LIST_H - 1) / LIST_H); }
inline int64_t n_bins(int64_t nl) { return (nl + (1 << BIN_SHIFT) - 1) >> BIN_SHIFT; }
// A block of the two binning kernels takes `bin_batches(n)` batches of 2048 Gaussians, one after the other, into the same LDS
// histogram / cursors: for a very large scene this keeps the number of blocks near a thousand -- each block pays one returning global
// atomic per bin (config 5: 4883 blocks x 1012 bins = 4.9 M of them) and keeps one half-written line per bin open in the scatter.
inline int bin_batches(int64_t n) { const int64_t b = n / ((int64_t)BIN_GAUSS * 1024); return (int)(b < 1 ? 1 : (b > 4 ? 4 : b)); }
inline int64_t n_bin_blocks(int64_t n) { const int64_t per = (int64_t)BIN_GAUSS * bin_batches(n); return (n + per - 1) / per; }

ProjectState carve_project(void* base, int64_t n, int64_t nl) {
    ProjectState s;
    char* p = (char*)base;
    int64_t o = 0;
    const int64_t nb = n_bins(nl);
    s.cam = (Camera*)(p + o); o += up(sizeof(Camera));
    s.counts = (DevCounts*)(p + o); o += up(sizeof(DevCounts));
    s.rec = (Rec64*)(p + o); o += up(n * 64);
    s.rect = (u2*)(p + o); o += up(n * 8);
    s.depth = (float*)(p + o); o += up(n * 4);
    s.tiles = (uint32_t*)(p + o); o += up(n * 4);
    s.mask = (uint32_t*)(p + o); o += up(n * 4);
    s.bin_total = (uint32_t*)(p + o); o += up(3 * nb * 4);
    s.bin_start = (uint32_t*)(p + o); o += up((nb + 1) * 4);
    s.block_off = (uint32_t*)(p + o); o += up(n_bin_blocks(n) * nb * 4);
    s.list_count = (uint32_t*)(p + o); o += up((nb << BIN_SHIFT) * 4);
    s.ranges = (uint2*)(p + o); o += up(nl * 8);
    s.order = (uint32_t*)(p + o); o += up(nl * 4);
    s.class_bounds = (uint32_t*)(p + o); o += up(8 * 4);
    s.kj = (float*)(p + o); o += up(n * 48);
    s.big_flag = (uint32_t*)(p + o); o += up(((n + 255) / 256 * 4) * 4);      // (one word per projection wave = range of 64 Gaussians)
    s.bytes = o;
    return s;
}

// Binning scratch: payloads in coarse-bin order, then in list order (unsorted inside a list), and the offsets the
// chunks of the split kernels drew inside their lists.
struct BinScratch {
    uint64_t* bvals;         // [P] bin order
    uint64_t* vals;          // [P] list order
    uint32_t* seg_off;       // [(chunks + bins) x 64]
    int64_t bytes;
};

inline int64_t n_chunks(int64_t n_pairs) { return (n_pairs + SPLIT_CHUNK - 1) / SPLIT_CHUNK; }

BinScratch carve_bin_scratch(void* base, int64_t n_pairs, int64_t nb) {
    BinScratch s;
    char* p = (char*)base;
    int64_t o = 0;
    const int64_t np = n_pairs > 0 ? n_pairs : 1;
    s.bvals = (uint64_t*)(p + o); o += up(np * 8);
    s.vals = (uint64_t*)(p + o); o += up(np * 8);
    s.seg_off = (uint32_t*)(p + o); o += up((n_chunks(np) + nb) * 64 * 4);
    s.bytes = o;
    return s;
}

int check_view(const gsplat_view* v) {
    if (!v) return fail(GSPLAT_ERR_BAD_ARG, "view is NULL");
    if (v->H <= 0 || v->W <= 0) return fail(GSPLAT_ERR_BAD_ARG, "image size must be positive");
    if (v->tile < 1) return fail(GSPLAT_ERR_BAD_ARG, "tile size T must be >= 1");
    if ((v->W + 15) / 16 > 65535 || (v->H + 7) / 8 > 65535 || n_bins(n_lists(v)) > MAX_BINS) return fail(GSPLAT_ERR_BAD_ARG, "image too large");
    return GSPLAT_OK;
}

int check_gaussians(const gsplat_gaussians* g, bool* fused) {
    if (!g) return fail(GSPLAT_ERR_BAD_ARG, "gaussians is NULL");
    if (g->n < 0 || g->n > (int64_t)ID_MASK + 1) return fail(GSPLAT_ERR_BAD_ARG, "n out of range (at most 2^26 Gaussians per call)");
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

// ---- K1 ------------------------------------------------------------------------------------------
// One wave64 per 64 Gaussians.  The reference layout is array-of-structures (pos[N,3], f_rest[N,45] ...): a lane
// reading its own row directly issues 45 loads that each touch 64 different cache lines.  Instead the wave copies its
// 64 contiguous rows into LDS with fully coalesced 16-byte accesses and every lane then reads its row from LDS
// (row strides 3, 4, 9, 45 words are conflict-free or 2-way at worst).  The SH block (f_dc + f_rest, 192 of the 236
// input bytes) is only fetched when at least one Gaussian of the wave survived the culls.
// Full 64-row blocks go global -> LDS directly (global_load_lds_dwordx4, gfx950): no VGPR round trip, and a wave can put
// all of its ~15 KB of inputs in flight at once and wait for them once (the kernels run at 8-10 waves per CU, so bytes
// in flight per wave are what buys bandwidth).  One such instruction writes 64 lanes x 16 B contiguously at a
// wave-uniform LDS base: exactly the row-block image.  The caller's __syncthreads() (vmcnt(0) + barrier) retires them.
// The last, partial block of an array takes the register path.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int R>
__device__ __forceinline__ void stage_rows(float* __restrict__ lds, const float* __restrict__ g, int64_t row0, int64_t n, int lane) {
    const int64_t left = n - row0;
    const float* __restrict__ src = g + row0 * R;             // 16-B aligned: row0 % 64 == 0, base 16-B aligned (host checks)
    constexpr int PIECES = 64 * R / 4;
    if (left >= 64) {
#pragma unroll
        for (int it = 0; it < (PIECES + 63) / 64; ++it) {
            const int piece = it * 64 + lane;
            if (piece < PIECES)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(src + piece * 4), (lds_ptr_t)(lds + it * 256), 16, 0, 0);
        }
        return;
    }
    const int total = (int)left * R;       // floats to copy
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

// ACC: the rows are ADDED to what g holds (the gradient of several views summed in place, GSPLAT_BACKWARD_ACCUMULATE)
template <int R, bool ACC = false>
__device__ __forceinline__ void unstage_rows(float* __restrict__ g, const float* __restrict__ lds, int64_t row0, int64_t n, int lane) {
    const int64_t left = n - row0;
    const int total = (int)(left < 64 ? left : 64) * R;
    float* __restrict__ dst = g + row0 * R;
    constexpr int PIECES = 64 * R / 4;
#pragma unroll
    for (int it = 0; it < (PIECES + 63) / 64; ++it) {
        const int piece = it * 64 + lane;
        if (piece * 4 + 3 < total) {
            f4 v = *reinterpret_cast<const f4*>(lds + piece * 4);
            if (ACC) { const f4 o = *reinterpret_cast<const f4*>(dst + piece * 4); v = f4{o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w}; }
            *reinterpret_cast<f4*>(dst + piece * 4) = v;
        } else if (piece * 4 < total) {
            for (int k = piece * 4; k < total; ++k) dst[k] = ACC ? dst[k] + lds[k] : lds[k];
        }
    }
}

// The rows of a wave's R-float gradients (in LDS, as unstage_rows would write them) applied to the parameter instead: one Adam step
// of rows [row0, row0 + 64) of p with the moments m, v -- 16-byte pieces, the same lanes reading and writing them.
// counts / capacity: the frame's device counters and the pair capacity it was queued with -- a frame that outgrew its buffers (its
// gradients are garbage and the host will render it again) or that has nothing on screen (the host will raise the reference's
// exception) must not step anything: the guard is on the device because the host has not looked at the counters yet.
struct AdamRest { float* p; float* m; float* v; AdamStep k; const void* counts; long long capacity; };
typedef float fv4 __attribute__((ext_vector_type(4)));
template <int R>
__device__ __forceinline__ void adam_rows(const AdamRest& a, const float* __restrict__ lds, int64_t row0, int64_t n, int lane) {
    const int64_t left = n - row0;
    const int total = (int)(left < 64 ? left : 64) * R;
    float* __restrict__ P = a.p + row0 * R; float* __restrict__ M = a.m + row0 * R; float* __restrict__ V = a.v + row0 * R;
    constexpr int PIECES = 64 * R / 4;
#pragma unroll
    for (int it = 0; it < (PIECES + 63) / 64; ++it) {
        const int piece = it * 64 + lane;
        if (piece * 4 + 3 < total) {
            fv4 p = *reinterpret_cast<const fv4*>(P + piece * 4), g = *reinterpret_cast<const fv4*>(lds + piece * 4);
            fv4 m = __builtin_nontemporal_load(reinterpret_cast<const fv4*>(M + piece * 4)), v = __builtin_nontemporal_load(reinterpret_cast<const fv4*>(V + piece * 4));
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float pc = p[c], gc = g[c], mc = m[c], vc = v[c];
                adam_one(pc, gc, mc, vc, 1.0f, false, a.k.step_size, a.k.b1, a.k.b2, a.k.inv_sqrt_bc2, a.k.eps);
                p[c] = pc; m[c] = mc; v[c] = vc;
            }
            __builtin_nontemporal_store(m, reinterpret_cast<fv4*>(M + piece * 4));
            __builtin_nontemporal_store(v, reinterpret_cast<fv4*>(V + piece * 4));
            *reinterpret_cast<fv4*>(P + piece * 4) = p;
        } else if (piece * 4 < total) {
            for (int k = piece * 4; k < total; ++k) {
                float gk = lds[k];
                adam_one(P[k], gk, M[k], V[k], 1.0f, false, a.k.step_size, a.k.b1, a.k.b2, a.k.inv_sqrt_bc2, a.k.eps);
            }
        }
    }
}

// (LDS is handed out in pieces of 1280 B: with fused inputs the projection kernels use 15 104 B -> 10 waves per CU; the 9-float
// rows of the un-fused layout would cost the fused kernels a piece, and a wave per CU, for nothing)
template <bool FUSED>
struct ProjectLds {
    float pos[64 * 3];
    float opa[64];
    float a[64 * (FUSED ? 4 : 9)];   // fused: q_raw [64][4]       un-fused: sigma [64][9]
    float b[64 * 3];                 // fused: scale_raw [64][3]   un-fused: colour [64][3]
};

struct ShCoefLds {          // same access as ShCoefGlobal, on the staged copy
    const float* dc;
    const float* rest;
    __device__ __forceinline__ float operator()(int k, int ch) const { return k == 0 ? dc[ch] : rest[ch * 15 + (k - 1)]; }
};

template <bool FUSED>
__device__ __forceinline__ GaussIn gauss_from_lds(const ProjectLds<FUSED>& s, int lane) {
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
__device__ __forceinline__ void stage_geometry(ProjectLds<FUSED>& s, const gsplat_gaussians& g, int64_t row0, int lane) {
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

__device__ __forceinline__ bool rect_is_big(u2 rect) {
    const int w = (int)(rect.y & 0xFFFF) - (int)(rect.x & 0xFFFF) + 1, h = (int)(rect.y >> 16) - (int)(rect.x >> 16) + 1;
    return w * h > 32;
}

// Calls f(list, ordinal, a, b) for every list of a Gaussian's rectangle whose mask bit is set (row-major; ordinal 0 .. nt - 1
// counts the calls; a, b = the owning lane's values).  Rectangles of up to 32 lists only: each lane walks its own.  Larger ones
// (large Gaussians) are walked row by row by whole waves (for_each_big_row): the caller passes nt = 0 for them.
template <class F>
__device__ __forceinline__ void for_each_list(u2 rect, uint32_t nt, uint32_t mask, int lists_x, int lane, uint64_t a, uint32_t b, F f) {
    const int x0 = rect.x & 0xFFFF, y0 = rect.x >> 16, x1 = rect.y & 0xFFFF, y1 = rect.y >> 16;
    if (nt) {                                  // (load_block_pairs leaves nt = 0 for a large Gaussian)
        uint32_t k = 0, m = mask;
        // (left to itself the compiler, knowing the rectangle has at most 32 lists here, unrolls the walk into chains of predicated
        //  LDS atomics: 2 us slower in bin_count_kernel at config 3 than the plain loops)
#pragma clang loop unroll(disable)
        for (int y = y0; y <= y1; ++y)
#pragma clang loop unroll(disable)
            for (int x = x0; x <= x1; ++x, m >>= 1)
                if (m & 1u) f((uint32_t)(y * lists_x + x), k++, a, b);
    }
    (void)lane;
}

// The lists of the LARGE Gaussians held by the lanes of one wave (`big`: rectangle of more than 32 lists and binned at all), one
// Gaussian after the other, the lanes taking the ROWS of its rectangle: f(first list of the row's span, lists in the span, payload)
// per non-empty row (gs_math.h big_row_span: the same spans the projection kernel counted into tiles[]).  Call with all 64 lanes.
// `mine`: which lanes' Gaussians this wave takes (the four waves of a big block hold the same 64 and take every fourth each).
template <class F>
__device__ __forceinline__ void for_each_big_row(bool big, u2 rect, f4 uvexy, f4 k4, uint64_t payload, int lists_x, int lane, F f,
                                                 unsigned long long mine = ~0ull) {
    unsigned long long m = __ballot(big) & mine;
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
#define RL_F(x) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src))
        const uint32_t rx = (uint32_t)__builtin_amdgcn_readlane((int)rect.x, src), ry = (uint32_t)__builtin_amdgcn_readlane((int)rect.y, src);
        const float kk[4] = {RL_F(k4.x), RL_F(k4.y), RL_F(k4.z), RL_F(k4.w)};
        const uint64_t pl = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(payload >> 32), src) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)payload, src);
        const int x0 = rx & 0xFFFF, y0 = rx >> 16, x1 = ry & 0xFFFF, y1 = ry >> 16;
        const BigSpanK bk = big_span_setup(RL_F(uvexy.x), RL_F(uvexy.y), RL_F(uvexy.z), RL_F(uvexy.w), kk, x0, x1);
#undef RL_F
        for (int y = y0 + lane; y <= y1; y += 64) {
            const RowSpan sp = big_row_span(bk, y);
            if (sp.xb >= sp.xa) f((uint32_t)(y * lists_x + sp.xa), (uint32_t)(sp.xb - sp.xa + 1), pl);
        }
    }
}

// Inclusive prefix sum over the 64 lanes in the VALU (DPP row shifts inside the rows of 16, row broadcasts across them): six adds,
// where six __shfl_up are six round trips through the LDS crossbar.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
#define DPP_U32(v, ctrl, rmask) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xF, false)
    x += DPP_U32(x, 0x111, 0xF);            // row_shr:1
    x += DPP_U32(x, 0x112, 0xF);            // row_shr:2
    x += DPP_U32(x, 0x114, 0xF);            // row_shr:4
    x += DPP_U32(x, 0x118, 0xF);            // row_shr:8
    x += DPP_U32(x, 0x142, 0xA);            // row_bcast:15 -> rows 1, 3
    x += DPP_U32(x, 0x143, 0xC);            // row_bcast:31 -> rows 2, 3
#undef DPP_U32
    return x;
}

// Sum / maximum / minimum over the 64 lanes the same way (the result in every lane, through lane 63 and an SGPR): six DPP steps
// instead of six ds_bpermute butterflies.
#define GS_DPP_U32(v, idn, ctrl, rmask) (uint32_t)__builtin_amdgcn_update_dpp((int)(idn), (int)(v), ctrl, rmask, 0xF, false)
__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(x), 63);
}
__device__ __forceinline__ uint32_t wave_max(uint32_t x) {
    x = max(x, GS_DPP_U32(x, 0u, 0x111, 0xF)); x = max(x, GS_DPP_U32(x, 0u, 0x112, 0xF));
    x = max(x, GS_DPP_U32(x, 0u, 0x114, 0xF)); x = max(x, GS_DPP_U32(x, 0u, 0x118, 0xF));
    x = max(x, GS_DPP_U32(x, 0u, 0x142, 0xA)); x = max(x, GS_DPP_U32(x, 0u, 0x143, 0xC));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ uint32_t wave_min(uint32_t x) {
    x = min(x, GS_DPP_U32(x, 0xFFFFFFFFu, 0x111, 0xF)); x = min(x, GS_DPP_U32(x, 0xFFFFFFFFu, 0x112, 0xF));
    x = min(x, GS_DPP_U32(x, 0xFFFFFFFFu, 0x114, 0xF)); x = min(x, GS_DPP_U32(x, 0xFFFFFFFFu, 0x118, 0xF));
    x = min(x, GS_DPP_U32(x, 0xFFFFFFFFu, 0x142, 0xA)); x = min(x, GS_DPP_U32(x, 0xFFFFFFFFu, 0x143, 0xC));
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
#undef GS_DPP_U32

// COLOUR = false (fused inputs): geometry only, 44 of the 236 input bytes; colour_kernel evaluates the SH colour later,
// queued behind the copy of the counters so that it runs while the host reads them and sizes the binning buffers.
// The camera block (w2c, eye) is derived from c2w by every wave itself (16 uniform loads + 30 flops: cheaper than the launch of
// a 1-thread kernel in front); wave 0 stores it for the later kernels.  The first waves clear the coarse-bin totals
// bin_count_kernel accumulates into.  Epilogue: per-wave counts -> sharded counters -> the LAST wave to arrive (agent-scope
// acq_rel counter) adds the shards up, writes the totals (device, and the caller's mapped host block if given) and leaves
// the counter block zeroed for the next call.
// JAC (FUSED && COLOUR only): also store, per visible Gaussian, the 12 values that spare the backward the SH coefficients.
// TOTALS = false (GSPLAT_PROJECT_COUNTS_LATE): the waves only add to the sharded counters and leave; bin_count_kernel, queued
// right behind, totals and clears them.  (With the totals in here every wave waits for ALL its stores and atomics and then for
// a returning arrival atomic before it can retire: a quarter of a wave's life.)
// (Workgroups of 2 / 4 waves instead of one: 89 / 91 us against 90 -- the kernel is not held by the rate at which one-wave
// workgroups can be dispatched.  As a STREAM -- 6 persistent waves per CU, two sets of LDS rows, block k + 1 requested before
// block k is computed -- 158 us against 96: with 1.5 waves per SIMD the long dependent chains of the geometry math issue at a
// fraction of the VALU rate; this kernel lives on wave-level parallelism.)
template <bool FUSED, bool COLOUR, bool JAC = false, bool TOTALS = true>
__global__ __launch_bounds__(64) void project_kernel(gsplat_gaussians g, const float* __restrict__ c2w, Camera* __restrict__ cam_out, ViewK vk,
                                                     Records out, CounterBlock* cb, DevCounts* counts, DevCounts* counts_mapped,
                                                     uint32_t* __restrict__ bin_total, int nb, float* __restrict__ kj_out,
                                                     uint32_t* __restrict__ big_flag) {
    // DIRECT (fused inputs with the colour inside): the 44 bytes of geometry per Gaussian are loaded by the lanes themselves (rows
    // of 3 / 4 floats coalesce well enough) and only the 180 bytes of f_rest go through LDS: 11 520 B per wave instead of
    // 15 104 -> 12 waves per CU instead of 10, and the geometry math starts while the coefficients are still arriving.
    constexpr bool DIRECT = FUSED && COLOUR;
    __shared__ float s_geo[DIRECT ? 4 : sizeof(ProjectLds<FUSED>) / 4];
    ProjectLds<FUSED>& s = *reinterpret_cast<ProjectLds<FUSED>*>(s_geo);
    __shared__ float s_rest[FUSED && COLOUR ? 64 * 45 : 4];
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + lane;
    GaussIn in;
    float dc[3] = {0.f, 0.f, 0.f};                           // (DIRECT: the 3 f_dc values with the geometry; 11 520 B of LDS would allow 14
                                                             //  waves per CU, but the Jacobian variant needs 132 VGPRs: forced to 128 it spills, 94 us against 90)
    if (DIRECT) {                                            // (issued BEFORE the LDS-DMA: vmcnt counts in order)
        if (i < g.n) {
#pragma unroll
            for (int k = 0; k < 3; ++k) dc[k] = g.f_dc[i * 3 + k];
#pragma unroll
            for (int k = 0; k < 3; ++k) in.p[k] = g.pos[i * 3 + k];
            in.o_raw = g.opacity_raw[i];
            const f4 q = *reinterpret_cast<const f4*>(g.q_raw + i * 4);
            in.qr[0] = q.x; in.qr[1] = q.y; in.qr[2] = q.z; in.qr[3] = q.w;
#pragma unroll
            for (int k = 0; k < 3; ++k) in.sr[k] = g.scale_raw[i * 3 + k];
        }
    } else {
        stage_geometry<FUSED>(s, g, row0, lane);
    }
    if (FUSED && COLOUR) stage_rows<45>(s_rest, g.f_rest, row0, g.n, lane);          // all inputs of the wave in flight at once
    Camera cam;                                              // (derived while the inputs are in flight)
    {
        float m[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) m[k] = c2w[k];
        build_camera(m, cam);
        if (blockIdx.x == 0 && lane == 0) *cam_out = cam;
    }
    for (int b = blockIdx.x * 64 + lane; b < 3 * nb; b += gridDim.x * 64) bin_total[b] = 0u;     // (+ the large Gaussians' totals and cursor)
    if (!DIRECT) __syncthreads();
    Proj o;
    o.vis = VIS_CULLED;
    if (i < g.n) {
        if (!DIRECT) in = gauss_from_lds<FUSED>(s, lane);
        o = project_geometry(in, FUSED, cam, vk);
    }
    if (DIRECT) __syncthreads();                             // the SH coefficients have arrived
    RecOut r;
    r.vis = o.vis; r.tiles = 0; r.mask = 0u; r.ref_tiles = 0; r.rect = u2{0u, 0u}; r.ref_rect = u2{0u, 0u};
    float kj[12];
    if (FUSED) {
        if (o.vis == VIS_OK) r = project_finish(in, o, true, ShCoefLds{dc, s_rest + lane * 45}, cam, COLOUR, JAC ? kj : nullptr);
    } else if (o.vis == VIS_OK) {
        r = project_finish(in, o, false, ShCoefLds{nullptr, nullptr}, cam);
    }
    if (i < g.n) {
        if (r.vis == VIS_OK) {
            Rec64 line;
            line.r0 = r.r0; line.r1 = r.r1; line.r2 = r.r2; line.pad = r.r3;
            out.rec[i] = line;                   // 64 contiguous bytes per lane, 4 KB per wave
            out.rect[i] = r.rect;
            out.depth[i] = r.r2.w;
            out.mask[i] = r.mask;
            if (JAC) {                           // 48 contiguous bytes per lane
                f4* dst = reinterpret_cast<f4*>(kj_out + i * 12);
                dst[0] = f4{kj[0], kj[1], kj[2], kj[3]};
                dst[1] = f4{kj[4], kj[5], kj[6], kj[7]};
                dst[2] = f4{kj[8], kj[9], kj[10], kj[11]};
            }
        }
        out.tiles[i] = r.tiles;
#ifdef GSPLAT_DIAGNOSTICS
        if (out.ref_rect) out.ref_rect[i] = r.ref_rect;
        if (out.ref_tiles) out.ref_tiles[i] = r.vis == VIS_OK ? r.ref_tiles : 0u;
#endif
    }
    {
        const bool any_large = __any(r.tiles != 0u && rect_is_big(r.rect));
        if (lane == 0) big_flag[blockIdx.x] = any_large ? 1u : 0u;
    }
    const unsigned long long surv = __ballot(o.vis != VIS_CULLED);
    const unsigned long long seen = __ballot(o.vis == VIS_OK);
    const uint32_t mx = wave_max(r.tiles), refp = wave_sum(r.ref_tiles), binp = wave_sum(r.tiles);
    uint32_t arrived = 0u;
    if (lane == 0) {
        CountShard* sh = cb->shards + (blockIdx.x % COUNT_SHARDS);
        if (surv) atomicAdd(&sh->survivors, (int)__popcll(surv));
        if (seen) atomicAdd(&sh->visible, (int)__popcll(seen));
        if (mx) atomicMax(&sh->max_tiles, (int)mx);
        if (refp) atomicAdd(&sh->ref_pairs, refp);
        if (binp) atomicAdd(&sh->bin_pairs, binp);
        // The adds above are agent-scope atomics (performed at the memory side, coherent without any cache maintenance); they
        // only have to be COMPLETE before this wave reports in: s_waitcnt vmcnt(0) (atomics stay counted until performed).  (An
        // agent-scope release fence here costs an L2 write-back per wave: 15 625 of them took 0.8 ms.)
        if (TOTALS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // arrival, two levels: last wave of its shard -> last shard of the grid
            const uint32_t shard = blockIdx.x % COUNT_SHARDS, shards_used = min(gridDim.x, (uint32_t)COUNT_SHARDS);
            const uint32_t waves_of_shard = (gridDim.x - shard + COUNT_SHARDS - 1u) / COUNT_SHARDS;
            if (__hip_atomic_fetch_add(&sh->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == waves_of_shard - 1u)
                arrived = (__hip_atomic_fetch_add(&cb->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards_used - 1u) ? 1u : 0u;
        }
    }
    if (!TOTALS) return;
    arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
    if (!arrived) return;
    // ---- last wave: totals of the 256 shards (4 per lane; agent-scope atomic loads: the adds were made at that scope)
    unsigned long long t4[4] = {0ull, 0ull, 0ull, 0ull};
    uint32_t mxt = 0u;
#pragma unroll
    for (int k = 0; k < COUNT_SHARDS / 64; ++k) {
        CountShard* sh = cb->shards + k * 64 + lane;
        t4[0] += (unsigned long long)(uint32_t)__hip_atomic_load(&sh->survivors, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t4[1] += (unsigned long long)(uint32_t)__hip_atomic_load(&sh->visible, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t4[2] += (unsigned long long)__hip_atomic_load(&sh->ref_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t4[3] += (unsigned long long)__hip_atomic_load(&sh->bin_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        mxt = max(mxt, (uint32_t)__hip_atomic_load(&sh->max_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        // leave the block zeroed for the next call (agent-scope stores: not parked in this XCD's L2 behind the atomics)
        __hip_atomic_store(&sh->survivors, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sh->visible, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sh->ref_pairs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sh->bin_pairs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sh->max_tiles, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sh->arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int sft = 32; sft > 0; sft >>= 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) t4[k] += (unsigned long long)__shfl_xor((long long)t4[k], sft);
        mxt = max(mxt, (uint32_t)__shfl_xor((int)mxt, sft));
    }
    if (lane == 0) {
        DevCounts c;
        c.n_survivors = (int32_t)t4[0]; c.n_visible = (int32_t)t4[1]; c.n_pairs = (int64_t)t4[2]; c.max_tiles = (int32_t)mxt;
        c.reserved = 0; c.n_binned = (int64_t)t4[3];
        *counts = c;
        if (counts_mapped) *counts_mapped = c;               // pinned host memory: visible to the host once the event behind us fires
        __hip_atomic_store(&cb->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- K1b: SH colour (fused inputs) -------------------------------------------------------------------
// F3 for the Gaussians that were binned: 192 of the 236 input bytes per Gaussian are SH coefficients.  Writes r, g, b into
// the record line the geometry pass left (z stays).
template <bool JAC>
__global__ __launch_bounds__(64) void colour_kernel(gsplat_gaussians g, const Camera* __restrict__ camp, const uint32_t* __restrict__ tiles,
                                                    Rec64* __restrict__ rec, float* __restrict__ kj_out) {
    __shared__ float s_pos[64 * 3], s_dc[64 * 3], s_rest[64 * 45];
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + lane;
    const bool need = i < g.n && tiles[i] != 0u;
    if (!__any(need)) return;                                // wave-uniform: skip 204 B / Gaussian when none is binned
    stage_rows<3>(s_pos, g.pos, row0, g.n, lane);
    stage_rows<3>(s_dc, g.f_dc, row0, g.n, lane);
    stage_rows<45>(s_rest, g.f_rest, row0, g.n, lane);
    const Camera cam = *camp;
    __syncthreads();
    if (need) {
        const float p[3] = {s_pos[lane * 3], s_pos[lane * 3 + 1], s_pos[lane * 3 + 2]};
        ShMid sm;
        sh_basis(p, cam.eye, sm);
        float rgb[3];
        if (JAC) {
            float kj[12];
            sh_colour_jac(sm, ShCoefLds{s_dc + lane * 3, s_rest + lane * 45}, rgb, kj);
            f4* dst = reinterpret_cast<f4*>(kj_out + i * 12);
            dst[0] = f4{kj[0], kj[1], kj[2], kj[3]};
            dst[1] = f4{kj[4], kj[5], kj[6], kj[7]};
            dst[2] = f4{kj[8], kj[9], kj[10], kj[11]};
        } else {
            sh_colour(sm, ShCoefLds{s_dc + lane * 3, s_rest + lane * 45}, rgb);
        }
        float* r2 = reinterpret_cast<float*>(&rec[i].r2);
        r2[0] = rgb[0]; r2[1] = rgb[1]; r2[2] = rgb[2];
    }
}

// ---- K3: coarse bins (F11) -----------------------------------------------------------------------------
// bin_count_kernel: a block of 2048 Gaussians histograms its (list, Gaussian) pairs over the coarse bins in LDS and takes
// its share of every bin it touches with ONE returning global atomic per bin (device-scope atomics run at ~20 G/s and
// serialise per address: one per pair was 10x slower than the radix sort this replaces; one per block and bin is noise).
// Needs no pair buffer, so it is queued with the colour pass behind the counters and runs during the host round trip.
struct BlockPairs {                    // the 8 Gaussians of one thread of a binning workgroup
    static constexpr int K = BIN_GAUSS / 256;
    uint32_t nt[K], mk[K];
    u2 r[K];
    uint64_t payload[K];
};

// ALL loads of the thread's Gaussians in flight together: one round trip, and the caller can put its own set-up (prefix sums,
// clearing LDS, barriers) between this and for_block_pairs.  The rectangle, mask and depth of a Gaussian that is not binned are
// stale values: read and ignored.
__device__ __forceinline__ BlockPairs load_block_pairs(int64_t n, const u2* __restrict__ rect, const uint32_t* __restrict__ tiles,
                                                       const uint32_t* __restrict__ mask, const float* __restrict__ depth, int64_t batch) {
    constexpr int K = BlockPairs::K;
    BlockPairs bp;
    float dz[K];
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int64_t i = batch * BIN_GAUSS + k * 256 + tid;
        const bool in = i < n;
        bp.nt[k] = in ? tiles[i] : 0u;
        bp.r[k] = in ? rect[i] : u2{0u, 0u};
        if (rect_is_big(bp.r[k])) bp.nt[k] = 0u;           // a large Gaussian: binned by the big blocks (the rectangle of a Gaussian that is
                                                           // not binned at all is stale, its nt is 0 anyway)
        bp.mk[k] = in ? mask[i] : 0u;
        dz[k] = (in && depth) ? depth[i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int64_t i = batch * BIN_GAUSS + k * 256 + tid;
        bp.payload[k] = ((uint64_t)f2u(dz[k]) << 32) | (uint64_t)(uint32_t)i;
    }
    return bp;
}

template <class F>
__device__ __forceinline__ void for_block_pairs(const BlockPairs& bp, int lists_x, F f) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < BlockPairs::K; ++k) for_each_list(bp.r[k], bp.nt[k], bp.mk[k], lists_x, lane, bp.payload[k], 0u, f);
}

// What a big block (a range of 64 Gaussians held by each of its four waves, see bin_count_kernel) holds per thread.  Returns false (uniformly) when the block's range has no
// large Gaussian: the block then leaves -- at config 3 (none at all) that is all these blocks ever do.
struct BigLane { bool big; u2 rect; f4 uvexy, k4; uint64_t payload; };      // centre + extents, row-span constants (record)
__device__ __forceinline__ bool load_big_lane(int64_t i, int64_t n, const u2* __restrict__ rect, const uint32_t* __restrict__ tiles,
                                              const Rec64* __restrict__ rec, const float* __restrict__ depth, BigLane& b,
                                              const uint32_t* __restrict__ big_flag) {
    // the projection wave of this range (one uniform load): nothing large -> nothing else is even loaded
    if (!big_flag[(i - (threadIdx.x & 63)) / 64]) return false;
    b.rect = u2{0u, 0u};
    b.big = false;
    if (i < n) {
        const uint32_t nt = tiles[i];
        b.rect = rect[i];                                  // (stale for a Gaussian that is not binned: nt = 0)
        b.big = nt != 0u && rect_is_big(b.rect);
    }
    if (!__syncthreads_or(b.big)) return false;
    b.uvexy = b.k4 = f4{0.f, 0.f, 0.f, 0.f};
    b.payload = 0ull;
    if (b.big) {
        const Rec64* r = rec + i;
        const f4 q0 = r->r0, q1 = r->r1;
        b.uvexy = f4{q0.x, q0.y, q1.z, q1.w};
        b.k4 = r->pad;
        b.payload = ((uint64_t)f2u(depth ? depth[i] : 0.f) << 32) | (uint64_t)(uint32_t)i;
    }
    return true;
}

// body(range) for every range of this big block (first, first + stride, ...) whose projection wave flagged a large Gaussian.  The
// flags are read in rounds of 256 candidate ranges, one per thread (one round trip per round), and the flagged ones listed in LDS:
// a block with nothing to do -- every one of them at config 3 -- leaves after one round trip, and a scene with FEW large Gaussians
// (config 5: 156 K ranges, 200 per block) does not probe its ranges one dependent load after the other.  body may synchronise the
// workgroup (it is called uniformly).
struct FlaggedLds { uint32_t list[256]; uint32_t count; };
template <class Body>
__device__ __forceinline__ void for_flagged_ranges(const uint32_t* __restrict__ big_flag, int64_t first, int64_t stride, int64_t ranges,
                                                   FlaggedLds& fl, Body body) {
    for (int64_t base = first; base < ranges; base += 256 * stride) {
        if (threadIdx.x == 0) fl.count = 0u;
        __syncthreads();
        const int64_t r = base + (int64_t)threadIdx.x * stride;
        if (r < ranges && big_flag[r] != 0u) fl.list[atomicAdd(&fl.count, 1u)] = (uint32_t)r;
        __syncthreads();
        const uint32_t cnt = fl.count;
        for (uint32_t i = 0; i < cnt; ++i) body((int64_t)fl.list[i]);
        __syncthreads();
    }
}

// pieces of a run of consecutive lists [l0, l0 + cnt) by coarse bin: g(bin, first list of the piece, lists in the piece)
template <class G>
__device__ __forceinline__ void for_bin_pieces(uint32_t l0, uint32_t cnt, G g) {
    const uint32_t l1 = l0 + cnt - 1u;
    for (uint32_t b = l0 >> BIN_SHIFT; b <= (l1 >> BIN_SHIFT); ++b) {
        const uint32_t a = max(l0, b << BIN_SHIFT), e = min(l1, (b << BIN_SHIFT) + (1u << BIN_SHIFT) - 1u);
        g(b, a, e - a + 1u);
    }
}

__device__ __forceinline__ void bin_count_big(int64_t n, const u2* __restrict__ rect, const uint32_t* __restrict__ tiles, int lists_x, int nb,
                                                        uint32_t* __restrict__ bin_total, const Rec64* __restrict__ rec,
                                                        const uint32_t* __restrict__ big_flag, uint32_t small_blocks, uint32_t* hist) {
    const int tid = threadIdx.x;
    const int64_t ranges = (n + 63) / 64;
    const unsigned long long mine = 0x1111111111111111ull << (tid >> 6);          // this wave's quarter of the range's Gaussians
    // the block's ranges are counted into ONE histogram, flushed once (a flush per range of 64: 123 K atomics on the 79 totals of
    // config 6, +14 us)
    __shared__ FlaggedLds fl;
    bool any = false;
    for (int b = tid; b < nb; b += 256) hist[b] = 0u;
    for_flagged_ranges(big_flag, blockIdx.x - small_blocks, gridDim.x - small_blocks, ranges, fl, [&](int64_t range) {
        BigLane bl;
        if (!load_big_lane(range * 64 + (tid & 63), n, rect, tiles, rec, nullptr, bl, big_flag)) return;
        any = true;
        for_each_big_row(bl.big, bl.rect, bl.uvexy, bl.k4, 0ull, lists_x, tid & 63, [&](uint32_t l0, uint32_t cnt, uint64_t) {
            for_bin_pieces(l0, cnt, [&](uint32_t b, uint32_t, uint32_t c) { atomicAdd(&hist[b], c); });
        }, mine);
    });
    if (!any) return;                                      // (uniform; for_flagged_ranges ends with a barrier)
    for (int b = tid; b < nb; b += 256) {
        const uint32_t c = hist[b];
        if (c) atomicAdd(&bin_total[nb + b], c);           // (no offset is drawn here: bin_scatter_kernel's big blocks draw theirs)
    }
}

// Grid = the blocks of 2048 Gaussians, which bin the SMALL Gaussians (rectangles of up to 32 lists, each lane walking its own), then
// `big_blocks` blocks, which bin the LARGE ones of ranges of 64 Gaussians wave-cooperatively (for_each_big_row; every wave of the block
// holds the range's 64 Gaussians and takes every fourth): a Gaussian of a trained scene covers hundreds of lists, and 2048 of them per
// block left the chip with 49 workgroups walking half a million lists each.  (Ranges of 256 with 64 Gaussians per wave, one after the
// other, the first version: 1.5 waves per SIMD in a chain of dependent LDS atomics and shuffles -- 161 us for the scatter at config 6.)
__global__ __launch_bounds__(256) void bin_count_kernel(int64_t n, const u2* __restrict__ rect, const uint32_t* __restrict__ tiles,
                                                        const uint32_t* __restrict__ mask, int lists_x, int nb, uint32_t* __restrict__ bin_total,
                                                        uint32_t* __restrict__ block_off, uint32_t* __restrict__ list_count,
                                                        uint2* __restrict__ ranges, int nl, CounterBlock* cb, DevCounts* counts,
                                                        DevCounts* counts_mapped, const Rec64* __restrict__ rec, const uint32_t* __restrict__ big_flag, uint32_t small_blocks,
                                                        int batches) {
    // hist[nb]: dynamic LDS, sized by the launch (a static array for the largest image, 32 KB, held the two binning kernels at 3-4
    // workgroups per CU whatever the image: the blocks of the large Gaussians ran in three rounds)
    extern __shared__ uint32_t bin_lds[];
    uint32_t* const hist = bin_lds;
    const int tid = threadIdx.x;
    if (blockIdx.x >= small_blocks) {            // ---- ranges of 64 Gaussians (grid-stride): the large ones of each range
        bin_count_big(n, rect, tiles, lists_x, nb, bin_total, rec, big_flag, small_blocks, hist);
        return;
    }
    BlockPairs bp = load_block_pairs(n, rect, tiles, mask, nullptr, (int64_t)blockIdx.x * batches);
    if (cb && blockIdx.x == 0) {                 // GSPLAT_PROJECT_COUNTS_LATE: totals of the projection's sharded counters; shards cleared
        static_assert(COUNT_SHARDS == 256, "one shard per thread");
        __shared__ unsigned long long tsum[4][4];
        __shared__ uint32_t tmax[4];
        CountShard* sh = cb->shards + tid;
        unsigned long long t4[4];
        t4[0] = (unsigned long long)(uint32_t)__hip_atomic_load(&sh->survivors, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t4[1] = (unsigned long long)(uint32_t)__hip_atomic_load(&sh->visible, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t4[2] = (unsigned long long)__hip_atomic_load(&sh->ref_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t4[3] = (unsigned long long)__hip_atomic_load(&sh->bin_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t mxt = (uint32_t)__hip_atomic_load(&sh->max_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sh->survivors, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sh->visible, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sh->ref_pairs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sh->bin_pairs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sh->max_tiles, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int sft = 32; sft > 0; sft >>= 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) t4[k] += (unsigned long long)__shfl_xor((long long)t4[k], sft);
            mxt = max(mxt, (uint32_t)__shfl_xor((int)mxt, sft));
        }
        if ((tid & 63) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) tsum[tid >> 6][k] = t4[k];
            tmax[tid >> 6] = mxt;
        }
        __syncthreads();
        if (tid == 0) {
            DevCounts c;
            c.n_survivors = (int32_t)(tsum[0][0] + tsum[1][0] + tsum[2][0] + tsum[3][0]);
            c.n_visible = (int32_t)(tsum[0][1] + tsum[1][1] + tsum[2][1] + tsum[3][1]);
            c.n_pairs = (int64_t)(tsum[0][2] + tsum[1][2] + tsum[2][2] + tsum[3][2]);
            c.max_tiles = (int32_t)max(max(tmax[0], tmax[1]), max(tmax[2], tmax[3]));
            c.reserved = 0;
            c.n_binned = (int64_t)(tsum[0][3] + tsum[1][3] + tsum[2][3] + tsum[3][3]);
            *counts = c;
            if (counts_mapped) *counts_mapped = c;
        }
    }
    for (int l = blockIdx.x * 256 + tid; l < (nb << BIN_SHIFT); l += (int)small_blocks * 256) {     // for the split kernels
        list_count[l] = 0u;
        if (l < nl) ranges[l] = uint2{0u, 0u};
    }
    for (int b = tid; b < nb; b += 256) hist[b] = 0u;
    __syncthreads();
    for (int bt = 0;;) {
        for_block_pairs(bp, lists_x, [&](uint32_t l, uint32_t, uint64_t, uint32_t) { atomicAdd(&hist[l >> BIN_SHIFT], 1u); });
        if (++bt >= batches) break;
        bp = load_block_pairs(n, rect, tiles, mask, nullptr, (int64_t)blockIdx.x * batches + bt);
    }
    __syncthreads();
    for (int b = tid; b < nb; b += 256) {
        const uint32_t c = hist[b];
        if (c) block_off[(int64_t)blockIdx.x * nb + b] = atomicAdd(&bin_total[b], c);
    }
}

// bin_scatter_kernel: the same enumeration; a pair goes to bin_start[bin] + the block's offset in the bin + its arrival
// rank inside the block (LDS atomic).  The order inside a bin is arbitrary; the per-list sort by the unique payload makes
// the final order deterministic.
// The scatter of a wave's large Gaussians, one after the other: the lanes take the rows of the rectangle, every row's span is cut at the
// coarse-bin boundaries (a span of up to 33 lists crosses at most one: two rounds), each piece draws a run of slots from its bin's
// cursor in LDS -- and then the PAIRS, not the rows, are dealt to the lanes (`owner`: which lane's piece pair k belongs to), so that a
// store instruction writes up to 64 consecutive payloads instead of one 8-byte word into each of ~20 different runs.
// GSPLAT_BIG_DIRECT=1: every lane writes its own row's run instead (no owner array, no shuffles; 8-byte stores into ~20 runs per
// instruction).  Config 6: 105 against 131 us while the kernel sat at 3 workgroups per CU, 112 against 108 at 8 (the dynamic histogram
// below); config 5, whose scatter is bound by half-written lines: +15 us.  Off.
#ifndef GSPLAT_BIG_DIRECT
#define GSPLAT_BIG_DIRECT 0
#endif
constexpr int BIG_ROUND_PAIRS = 64 * 34;             // 64 rows x the widest span a rectangle can have (radius <= 250 px: 33 lists)
__device__ __forceinline__ void scatter_big_rows(const BigLane& bl, int lists_x, int lane, uint32_t* cur, uint8_t* owner, uint32_t n_binned,
                                                 uint64_t* __restrict__ bvals, unsigned long long mine) {
    unsigned long long m = __ballot(bl.big) & mine;
    while (m) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1;
#define RL_F(x) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src))
        const uint32_t rx = (uint32_t)__builtin_amdgcn_readlane((int)bl.rect.x, src), ry = (uint32_t)__builtin_amdgcn_readlane((int)bl.rect.y, src);
        const float kk[4] = {RL_F(bl.k4.x), RL_F(bl.k4.y), RL_F(bl.k4.z), RL_F(bl.k4.w)};
        const uint64_t pl = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(bl.payload >> 32), src) << 32) |
                            (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bl.payload, src);
        const int x0 = rx & 0xFFFF, y0 = rx >> 16, x1 = ry & 0xFFFF, y1 = ry >> 16;
        const BigSpanK bk = big_span_setup(RL_F(bl.uvexy.x), RL_F(bl.uvexy.y), RL_F(bl.uvexy.z), RL_F(bl.uvexy.w), kk, x0, x1);
#undef RL_F
        for (int yb = y0; yb <= y1; yb += 64) {                         // 64 rows per pass (one pass up to 512-pixel-high rectangles)
            const int y = yb + lane;
            RowSpan sp = RowSpan{1, 0};
            if (y <= y1) sp = big_row_span(bk, y);
            const bool has = sp.xb >= sp.xa;
            const uint32_t l0 = has ? (uint32_t)(y * lists_x + sp.xa) : 0u, l1 = has ? (uint32_t)(y * lists_x + sp.xb) : 0u;
            const uint32_t cut = ((l0 >> BIN_SHIFT) + 1u) << BIN_SHIFT;  // first list of the next bin
            for (int round = 0; round < 2; ++round) {
                // piece of this round: [a, a + c)
                const uint32_t a = round == 0 ? l0 : cut;
                const uint32_t c = !has ? 0u : (round == 0 ? min(l1 + 1u, cut) - l0 : (l1 >= cut ? l1 + 1u - cut : 0u));
                if (!__any(c != 0u)) continue;
                const uint32_t pos = c ? atomicAdd(&cur[a >> BIN_SHIFT], c) : 0u;
#if GSPLAT_BIG_DIRECT
                // every lane writes its own row's run: 8-byte stores into ~20 different runs per instruction, consecutive instructions
                // filling the same lines (the L2 merges them)
                for (uint32_t j = 0; j < c; ++j) {
                    const uint32_t dst = pos + j, l = a + j;
                    if (dst < n_binned) bvals[dst] = pl | ((uint64_t)(l & ((1u << BIN_SHIFT) - 1u)) << ID_BITS);
                }
                (void)owner;
                continue;
#endif
                const uint32_t incl = wave_inclusive_scan(c);
                const uint32_t pre = incl - c, total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                for (uint32_t j = 0; j < c; ++j) owner[pre + j] = (uint8_t)lane;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
                for (uint32_t k0 = 0; k0 < total; k0 += 64) {                   // (uniform trip count: a shuffle reads nothing from a lane that
                    const uint32_t k = k0 + (uint32_t)lane;                     //  has left the loop)
                    const int o = k < total ? owner[k] : 0;
                    const uint32_t j = k - (uint32_t)__shfl((int)pre, o);
                    const uint32_t dst = (uint32_t)__shfl((int)pos, o) + j, l = (uint32_t)__shfl((int)a, o) + j;
                    if (k < total && dst < n_binned) bvals[dst] = pl | ((uint64_t)(l & ((1u << BIN_SHIFT) - 1u)) << ID_BITS);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
                __builtin_amdgcn_wave_barrier();                                // `owner` is rewritten by the next round
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            }
        }
    }
}

// A bin's region of the bin-ordered array: [ pairs of the small Gaussians | pairs of the large ones ]; the small blocks place theirs
// with the offsets bin_count_kernel drew (block_off), the big blocks (same split of the grid as there) count their range again,
// draw ONE offset per touched bin from the bin's cursor (bin_total + 2 nb) and scatter.
//
// Exclusive prefix of the bin totals (small + large) for both kinds of block: thread t owns a contiguous run of ceil(nb / 256) bins and
// calls own(b, start of bin b, total of bin b, k) for each of them (k = index inside the run; the first four totals are in bt[]).
// The first four of a thread's totals are loaded up front (all of them up to 1024 bins = 4 M pixels): one round trip, not three.
struct BinPrefix { int per, first; uint32_t bt[4], small[4], run; };      // bt = small + large pairs of the bin, small = the small Gaussians' part
__device__ __forceinline__ BinPrefix bin_prefix_load(const uint32_t* __restrict__ bin_total, int nb) {
    BinPrefix p;
    p.per = (nb + 255) / 256;
    p.first = (int)threadIdx.x * p.per;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool in = k < p.per && p.first + k < nb;
        p.small[k] = in ? bin_total[p.first + k] : 0u;
        p.bt[k] = p.small[k] + (in ? bin_total[nb + p.first + k] : 0u);
    }
    p.run = p.bt[0] + p.bt[1] + p.bt[2] + p.bt[3];
    for (int k = 4; k < p.per; ++k) p.run += p.first + k < nb ? bin_total[p.first + k] + bin_total[nb + p.first + k] : 0u;
    return p;
}
// start of the thread's first bin (one workgroup barrier inside)
__device__ __forceinline__ uint32_t bin_prefix_scan(const BinPrefix& p, uint32_t* wsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_inclusive_scan(p.run);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t st = incl - p.run;
    for (int k = 0; k < wave; ++k) st += wsum[k];
    return st;
}

__device__ __forceinline__ void bin_scatter_big(int64_t n, const u2* __restrict__ rect, const uint32_t* __restrict__ tiles,
                                                          const float* __restrict__ depth, int lists_x, int nb, uint32_t* __restrict__ bin_total,
                                                          uint32_t n_binned, uint64_t* __restrict__ bvals, const Rec64* __restrict__ rec,
                                                          const uint32_t* __restrict__ big_flag, uint32_t small_blocks, uint32_t* cur, uint32_t* wsum,
                                                          uint8_t* owner) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t ranges = (n + 63) / 64;
    const unsigned long long mine = 0x1111111111111111ull << (tid >> 6);          // this wave's quarter of the range's Gaussians
    __shared__ FlaggedLds fl;
    for_flagged_ranges(big_flag, blockIdx.x - small_blocks, gridDim.x - small_blocks, ranges, fl, [&](int64_t range) {
        BigLane bl;
        if (!load_big_lane(range * 64 + lane, n, rect, tiles, rec, depth, bl, big_flag)) return;
        const BinPrefix bpf = bin_prefix_load(bin_total, nb);
        for (int b = tid; b < nb; b += 256) cur[b] = 0u;
        __syncthreads();
        for_each_big_row(bl.big, bl.rect, bl.uvexy, bl.k4, 0ull, lists_x, lane, [&](uint32_t l0, uint32_t cnt, uint64_t) {
            for_bin_pieces(l0, cnt, [&](uint32_t b, uint32_t, uint32_t c) { atomicAdd(&cur[b], c); });      // this range's pairs per bin
        }, mine);
        uint32_t st = bin_prefix_scan(bpf, wsum);              // (its barrier also closes the counting)
        // start of the bin + its small part + what this block draws from the large part's cursor (ONE returning atomic per touched
        // bin and block; the atomics of a thread's first four bins are in flight together: a range pays one round trip for them,
        // not one per bin)
        uint32_t mine4[4], got4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) mine4[k] = (k < bpf.per && bpf.first + k < nb) ? cur[bpf.first + k] : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) got4[k] = mine4[k] ? atomicAdd(&bin_total[2 * nb + bpf.first + k], mine4[k]) : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < bpf.per && bpf.first + k < nb) {
                if (mine4[k]) cur[bpf.first + k] = st + bpf.small[k] + got4[k];
                st += bpf.bt[k];
            }
        }
        for (int k = 4; k < bpf.per; ++k) {                    // (more than 1024 bins: images beyond 4 M pixels)
            const int b = bpf.first + k;
            if (b < nb) {
                const uint32_t mine = cur[b];
                if (mine) cur[b] = st + bin_total[b] + atomicAdd(&bin_total[2 * nb + b], mine);
                st += bin_total[b] + bin_total[nb + b];
            }
        }
        __syncthreads();
        scatter_big_rows(bl, lists_x, lane, cur, owner, n_binned, bvals, mine);
        __syncthreads();                                        // cur is cleared again by the next range
    });
}

__global__ __launch_bounds__(256) void bin_scatter_kernel(int64_t n, const u2* __restrict__ rect, const uint32_t* __restrict__ tiles,
                                                          const uint32_t* __restrict__ mask, const float* __restrict__ depth, int lists_x, int nb,
                                                          uint32_t* __restrict__ bin_total, const uint32_t* __restrict__ block_off,
                                                          uint32_t* __restrict__ bin_start, uint32_t n_binned,
                                                          uint64_t* __restrict__ bvals, const Rec64* __restrict__ rec,
                                                          const uint32_t* __restrict__ big_flag, uint32_t small_blocks, int batches) {
    extern __shared__ uint32_t bin_lds[];                  // cur[nb] (see bin_count_kernel)
    uint32_t* const cur = bin_lds;
    __shared__ uint32_t wsum[4];
    __shared__ uint8_t owner[4][BIG_ROUND_PAIRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x >= small_blocks) {            // ---- ranges of 64 Gaussians (grid-stride): the large ones of each range
        bin_scatter_big(n, rect, tiles, depth, lists_x, nb, bin_total, n_binned, bvals, rec, big_flag, small_blocks, cur, wsum, owner[wave]);
        return;
    }
    BlockPairs bp = load_block_pairs(n, rect, tiles, mask, depth, (int64_t)blockIdx.x * batches);
    const BinPrefix bpf = bin_prefix_load(bin_total, nb);
    uint32_t bo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bo[k] = (k < bpf.per && bpf.first + k < nb) ? block_off[(int64_t)blockIdx.x * nb + bpf.first + k] : 0u;
    uint32_t st = bin_prefix_scan(bpf, wsum);
    for (int k = 0; k < bpf.per; ++k) {
        const int b = bpf.first + k;
        if (b < nb) {
            const uint32_t c = k < 4 ? bpf.bt[k & 3] : bin_total[b] + bin_total[nb + b];
            cur[b] = st + (k < 4 ? bo[k & 3] : block_off[(int64_t)blockIdx.x * nb + b]);   // garbage for bins this block never touches: unused
            if (blockIdx.x == 0) {
                bin_start[b] = st;
                if (b == nb - 1) bin_start[nb] = st + c;
            }
            st += c;
        }
    }
    __syncthreads();
    for (int bt = 0;;) {
        for_block_pairs(bp, lists_x, [&](uint32_t l, uint32_t, uint64_t pl, uint32_t) {
            const uint32_t pos = atomicAdd(&cur[l >> BIN_SHIFT], 1u);
            if (pos < n_binned)                             // defensive: never write past the caller's buffer
                bvals[pos] = pl | ((uint64_t)(l & ((1u << BIN_SHIFT) - 1u)) << ID_BITS);
        });
        if (++bt >= batches) break;
        bp = load_block_pairs(n, rect, tiles, mask, depth, (int64_t)blockIdx.x * batches + bt);
    }
}

// split_count_kernel / split_scatter_kernel: every bin is split into its 64 lists.  Work is cut into chunks of 4096 pairs
// of the bin-ordered array (a dense bin of 60 K pairs is shared by 15 workgroups; one workgroup per bin was tail-bound);
// a chunk that crosses bin boundaries handles one segment per bin.  Count: LDS histogram of the segment over the bin's 64
// lists, one returning global atomic per list -> the segment's offset inside each list.  Scatter: list start = bin
// start + prefix of the bin's final list counts; a pair goes to list start + segment offset + arrival rank (LDS atomic).
// The segment that begins a bin also writes the [start, end) of the bin's lists.  Segment id = chunk + bin (unique: from
// one segment to the next at least one of the two grows).
//
// Which bin holds pair p (the b with bin_start[b] <= p < bin_start[b + 1]): every thread looks at its bins, the one that finds it
// reports it -- ONE round trip (a binary search is 8 dependent loads at config 3: 4-5 us at the start of every workgroup).
// Ends with a barrier.
template <int THREADS>
__device__ __forceinline__ int bin_of_pair_parallel(const uint32_t* __restrict__ bin_start, int nb, uint32_t p, int* slot) {
    if (threadIdx.x == 0) *slot = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < nb; t += THREADS)
        if (bin_start[t] <= p && p < bin_start[t + 1]) *slot = t;
    __syncthreads();
    return *slot;
}

__device__ __forceinline__ uint32_t local_list(uint64_t v) { return (uint32_t)(v >> ID_BITS) & ((1u << BIN_SHIFT) - 1u); }

// The grids of the two kernels come from the CAPACITY of the pair buffers (the host need not know the count); the pairs
// really binned are counts->n_binned.  More pairs than the buffers hold: only the first `capacity` are processed and the
// ranges are clipped to the buffers -- memory-safe garbage; the caller sees n_binned > capacity in the counters and renders
// the frame again with larger buffers.
__device__ __forceinline__ uint32_t pairs_to_process(const DevCounts* counts, uint32_t capacity) {
    const long long nb_ = counts->n_binned;
    return nb_ < (long long)capacity ? (uint32_t)nb_ : capacity;
}

__global__ __launch_bounds__(256) void split_count_kernel(int nb, const uint32_t* __restrict__ bin_start, const uint64_t* __restrict__ bvals,
                                                          uint32_t capacity, const DevCounts* __restrict__ counts,
                                                          uint32_t* __restrict__ list_count, uint32_t* __restrict__ seg_off) {
    constexpr int L = 1 << BIN_SHIFT, U = SPLIT_CHUNK / 256;
    __shared__ uint32_t cnt[L];
    __shared__ int s_b0;
    const int tid = threadIdx.x;
    const uint32_t n_binned = pairs_to_process(counts, capacity);
    const uint32_t c0 = blockIdx.x * (uint32_t)SPLIT_CHUNK, c1 = min(c0 + (uint32_t)SPLIT_CHUNK, n_binned);
    if (c0 >= n_binned) return;
    for (int b = bin_of_pair_parallel<256>(bin_start, nb, c0, &s_b0); b < nb && bin_start[b] < c1; ++b) {
        const uint32_t s = max(c0, bin_start[b]), e = min(c1, bin_start[b + 1]);
        if (s >= e) continue;                                  // empty bin (uniform)
        uint32_t k[U];                                          // (loads issued before the barrier: one round trip less)
#pragma unroll
        for (int u = 0; u < U; ++u) k[u] = s + u * 256 + tid < e ? local_list(bvals[s + u * 256 + tid]) : 0xFFFFFFFFu;
        if (tid < L) cnt[tid] = 0u;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (k[u] != 0xFFFFFFFFu) atomicAdd(&cnt[k[u]], 1u);
        __syncthreads();
        if (tid < L) {
            const uint32_t c = cnt[tid];
            if (c) seg_off[((int64_t)blockIdx.x + b) * L + tid] = atomicAdd(&list_count[b * L + tid], c);
        }
        __syncthreads();
    }
}

// ---- K5: plan ----------------------------------------------------------------------------------------
// Longest-processing-time-first launch order of the lists (1/8-octave buckets of the list length: the raster kernels
// are tail-bound, a few dense lists take 5x the mean, so they must start first), and the boundaries of the sort size
// classes inside that order.
__device__ __forceinline__ uint32_t work_bucket(uint32_t w) {
    if (w < 8u) return w;
    const uint32_t e = 31u - (uint32_t)__clz((int)w);
    return (e - 2u) * 8u + ((w >> (e - 3u)) & 7u);          // <= 239; 256 -> 48, 1024 -> 64, 4096 -> 80, 8192 -> 88
}
constexpr int SORT_CLASSES = 4;                               // list length >= 4096 | >= 1024 | >= 256 | >= 1
__device__ __forceinline__ uint32_t class_first_bucket(int c) { return c == 0 ? 80u : (c == 1 ? 64u : (c == 2 ? 48u : 1u)); }

// Counting sort of the lists by work bucket, descending.  Same-address LDS atomics serialise and neighbouring lists
// often share a bucket, so every bucket has 16 sub-counters selected by the lane (flat index = (255 - bucket) * 16 + sub:
// ascending flat index = descending bucket).
template <int THREADS, class Len>
__device__ __forceinline__ void plan_body(int nl, Len len, uint32_t* __restrict__ order, uint32_t* __restrict__ class_bounds,
                                          uint32_t* cnt, uint32_t* wsum) {
    constexpr int SUB = 16, NF = 256 * SUB, K = 16384 / THREADS, CPT = NF / THREADS, WAVES = THREADS / 64;
    // K lists per thread and round, held in registers; CPT counters per thread in the scan
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = tid & (SUB - 1);
    for (int f = tid; f < NF; f += THREADS) cnt[f] = 0u;
    __syncthreads();
    const bool one_round = nl <= THREADS * K;
    uint32_t flat[K];                                          // counter index of list (round base + k * THREADS + tid), or ~0
    for (int base = 0; base < nl; base += THREADS * K) {
        uint32_t w[K];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = base + k * THREADS + tid < nl ? len(base + k * THREADS + tid) : 0u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            flat[k] = base + k * THREADS + tid < nl ? (255u - work_bucket(w[k])) * SUB + sub : 0xFFFFFFFFu;
            if (flat[k] != 0xFFFFFFFFu) atomicAdd(&cnt[flat[k]], 1u);
        }
    }
    __syncthreads();
    // exclusive prefix over the NF counters: thread t owns CPT consecutive ones
    uint32_t c[CPT], run = 0u;
#pragma unroll
    for (int k = 0; k < CPT; ++k) { c[k] = cnt[tid * CPT + k]; run += c[k]; }
    const uint32_t incl = wave_inclusive_scan(run);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t st = incl - run;
    for (int k = 0; k < wave; ++k) st += wsum[k];
#pragma unroll
    for (int k = 0; k < CPT; ++k) { cnt[tid * CPT + k] = st; st += c[k]; }
    __syncthreads();
    // lists in buckets >= first bucket of a class = prefix at the first sub-counter of the bucket below it
    if (tid < SORT_CLASSES) class_bounds[tid] = cnt[(256u - class_first_bucket(tid)) * SUB];
    __syncthreads();
    for (int base = 0; base < nl; base += THREADS * K) {
        if (!one_round) {                                      // more than 16384 lists: recompute the counter indices
            uint32_t w[K];
#pragma unroll
            for (int k = 0; k < K; ++k) w[k] = base + k * THREADS + tid < nl ? len(base + k * THREADS + tid) : 0u;
#pragma unroll
            for (int k = 0; k < K; ++k)
                flat[k] = base + k * THREADS + tid < nl ? (255u - work_bucket(w[k])) * SUB + sub : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (flat[k] != 0xFFFFFFFFu) order[atomicAdd(&cnt[flat[k]], 1u)] = (uint32_t)(base + k * THREADS + tid);
    }
    (void)WAVES;
}
constexpr int PLAN_LDS_WORDS = 256 * 16 + 16;

// the plan by itself: only when there is nothing to bin (all ranges empty)
__global__ __launch_bounds__(1024) void plan_kernel(int nl, const uint2* __restrict__ ranges, uint32_t* __restrict__ order,
                                                    uint32_t* __restrict__ class_bounds) {
    __shared__ uint32_t lds[PLAN_LDS_WORDS];
    plan_body<1024>(nl, [&](int l) { const uint2 r = ranges[l]; return r.y - r.x; }, order, class_bounds, lds, lds + 256 * 16);
}

// ---- K3c, second half (after the plan it carries) ----------------------------------------------------
// Block 0 does not scatter: it is the PLAN (the list lengths are final after split_count_kernel, and a one-workgroup kernel of
// its own was 13 us of latency at config 3; here it runs beside the scatter).
#ifndef SS_THREADS
#define SS_THREADS 1024
#endif
__global__ __launch_bounds__(SS_THREADS) void split_scatter_kernel(int nl, int nb, const uint32_t* __restrict__ bin_start,
                                                            const uint64_t* __restrict__ bvals, uint32_t capacity,
                                                            const DevCounts* __restrict__ counts,
                                                            const uint32_t* __restrict__ list_count, const uint32_t* __restrict__ seg_off,
                                                            uint2* __restrict__ ranges, uint64_t* __restrict__ vals,
                                                            uint32_t* __restrict__ order, uint32_t* __restrict__ class_bounds) {
    constexpr int L = 1 << BIN_SHIFT, U = SPLIT_CHUNK / SS_THREADS;
    __shared__ uint32_t lds[PLAN_LDS_WORDS];
    if (blockIdx.x == 0) {
        plan_body<SS_THREADS>(nl, [&](int l) { return list_count[l]; }, order, class_bounds, lds, lds + 256 * 16);
        return;
    }
    uint32_t* const cur = lds;                               // [L]
    const int tid = threadIdx.x;
    const uint32_t n_binned = pairs_to_process(counts, capacity);
    const uint32_t chunk = blockIdx.x - 1u;
    const uint32_t c0 = chunk * (uint32_t)SPLIT_CHUNK, c1 = min(c0 + (uint32_t)SPLIT_CHUNK, n_binned);
    if (c0 >= n_binned) return;
    int* const s_b0 = reinterpret_cast<int*>(lds + L);
    for (int b = bin_of_pair_parallel<SS_THREADS>(bin_start, nb, c0, s_b0); b < nb && bin_start[b] < c1; ++b) {
        const uint32_t bs = bin_start[b], s = max(c0, bs), e = min(c1, bin_start[b + 1]);
        if (s >= e) continue;
        uint64_t v[U];                                          // (loads issued before the scan and the barrier: one round trip less)
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s + u * SS_THREADS + tid < e ? bvals[s + u * SS_THREADS + tid] : ~0ull;
        if (tid < L) {                                          // one wave: exclusive scan of the bin's 64 list sizes
            const uint32_t c = list_count[b * L + tid];
            const uint32_t incl = wave_inclusive_scan(c);
            const uint32_t st = bs + incl - c;
            cur[tid] = st + seg_off[((int64_t)chunk + b) * L + tid];           // garbage where the segment has no pair: unused
            const int list = b * L + tid;
            if (s == bs && list < nl) ranges[list] = uint2{min(st, capacity), min(st + c, capacity)};    // (clipped: overflow only)
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (s + u * SS_THREADS + tid < e) {
                const uint32_t pos = atomicAdd(&cur[local_list(v[u])], 1u);
                if (pos < n_binned) vals[pos] = v[u];
            }
        __syncthreads();
    }
}

// ---- K4: per-list depth sort ------------------------------------------------------------------------
// One workgroup per list sorts the list's payloads ascending = (depth, Gaussian index) order and writes the ids.
// Keys are unique, so the result does not depend on the arrival order of the scatter.
//
// list_sort_kernel (lists shorter than T * E): one-pass distribution sort in LDS.  The depth bits (monotone in z) are
// mapped to B >= 2 n buckets by subtracting the list minimum and shifting; count (LDS atomics) -> exclusive scan ->
// scatter gives bucket order; inside a bucket (expected occupancy <= 0.5) every element counts the smaller keys to find
// its rank.  ~8 barriers instead of the ~70 compare-exchange rounds of a bitonic network.  A list whose depths are so
// clustered that a bucket holds more than DENSE_BUCKET entries takes the bitonic network instead (exact, slower).
// Lists of 8192 and more (longer than the largest LDS class holds): the same bitonic network in place in global memory.
//
// Direction-free bitonic network: every merge of size k starts with a mirror step (i <-> block_end - i), followed by
// the half-cleaner steps j = k/4 .. 1; every compare-exchange puts the smaller key at the lower index.  With virtual
// +inf padding above n no real element is ever exchanged with the padding, so the network also runs in place.
// Synchronisation of the threads that sort one list: the workgroup, or -- when a wave sorts a list by itself inside a larger
// workgroup -- nothing but the order of the wave's own LDS instructions (the LDS executes one wave's instructions in issue
// order; the fences keep the compiler from moving accesses across).  The fences name the LDS ("local"): a plain wavefront-scope
// release also waits for the wave's outstanding GLOBAL stores (s_waitcnt vmcnt(0)) -- the large Gaussians' scatter stood 2 us per
// round on that, the wave-per-list sort once per list.
template <bool WAVE>
__device__ __forceinline__ void group_sync() {
    if (WAVE) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    } else {
        __syncthreads();
    }
}

template <int THREADS, bool WAVE, class Swap>
__device__ __forceinline__ void bitonic_network(uint32_t n, uint32_t m, int tid, Swap swap_if_greater) {
    uint32_t lk = 1;                                              // log2(k)
    for (uint32_t k = 2; k <= m; k <<= 1, ++lk) {
        const uint32_t half = k >> 1, lh = lk - 1;
        for (uint32_t t = tid; t < (m >> 1); t += THREADS) {
            const uint32_t r = t & (half - 1), i = ((t >> lh) << lk) + r, l = i + (k - 1 - 2 * r);
            if (l < n) swap_if_greater(i, l);
        }
        group_sync<WAVE>();
        uint32_t lj = lh;                                         // log2(j) + 1
        for (uint32_t j = half >> 1; j > 0; j >>= 1) {
            --lj;
            for (uint32_t t = tid; t < (m >> 1); t += THREADS) {
                const uint32_t i = ((t >> lj) << (lj + 1)) + (t & (j - 1)), l = i + j;
                if (l < n) swap_if_greater(i, l);
            }
            group_sync<WAVE>();
        }
    }
}

constexpr uint32_t DENSE_BUCKET = 48;

// LDS of one sorting group: T threads, lists shorter than T * E, 2^LOG2B depth buckets
template <int T, int E, int LOG2B>
struct SortLds {
    uint64_t sk[T * E];
    uint32_t cnt[(1 << LOG2B) + T];                               // padded: counter c lives at c + c / CPT (conflict-free scan)
    uint32_t red[4 + T / 64];
};

// Sort one list: T threads (`tid` = index inside the group).  WAVE: the group is one wave of a larger workgroup (T = 64).
// GLOBAL: lists of T * E entries and more are sorted in place in global memory (only the class of the longest lists has them).
template <int T, int E, int LOG2B, bool WAVE, bool GLOBAL>
__device__ __forceinline__ void sort_list(SortLds<T, E, LOG2B>& s, int tid, uint2 rg, uint64_t* __restrict__ vals,
                                          uint32_t* __restrict__ sorted_ids) {
    static_assert(!WAVE || T == 64, "a wave-synchronised group is one wave");
    constexpr int CAP = T * E, B = 1 << LOG2B, CPT = B / T;       // CPT counters per thread in the scan
    static_assert((CPT & (CPT - 1)) == 0 && CPT >= 2, "B / T must be a power of two");
    constexpr int LOG2CPT = __builtin_ctz(CPT);
    uint64_t* const sk = s.sk;
    uint32_t* const cnt = s.cnt;
    uint32_t* const red = s.red;
    group_sync<WAVE>();                                           // the LDS arrays are reused from list to list
    uint32_t n = rg.y - rg.x;                                     // 1 <= n; n < CAP by the class bounds, except in the GLOBAL class
    uint64_t* __restrict__ g = vals + rg.x;
    uint32_t* __restrict__ out = sorted_ids + rg.x;
    if (n >= (uint32_t)CAP) {
        if (!GLOBAL) {
            n = CAP - 1;                                          // cannot happen (class bounds); memory-safe if it ever did
        } else {                                                  // longer than the LDS holds -> in place in global memory
            uint32_t m = 2;
            while (m < n) m <<= 1;
            bitonic_network<T, false>(n, m, tid, [&](uint32_t i, uint32_t l) {
                const uint64_t a = g[i], b = g[l];
                if (a > b) { g[i] = b; g[l] = a; }
            });
            for (uint32_t i = tid; i < n; i += T) out[i] = (uint32_t)g[i] & ID_MASK;
            return;
        }
    }
#define PADC(c) ((c) + ((c) >> LOG2CPT))
    uint64_t key[E];
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)(e * T + tid);
        key[e] = i < n ? g[i] : ~0ull;
        if (i < n) { mn = min(mn, (uint32_t)(key[e] >> 32)); mx = max(mx, (uint32_t)(key[e] >> 32)); }
    }
    for (int c = tid; c < B + T; c += T) cnt[c] = 0u;
    if (tid == 0) { red[0] = 0xFFFFFFFFu; red[1] = 0u; red[2] = 0u; }
    mn = wave_min(mn);
    mx = wave_max(mx);
    group_sync<WAVE>();
    if ((tid & 63) == 0) { atomicMin(&red[0], mn); atomicMax(&red[1], mx); }
    group_sync<WAVE>();
    mn = red[0];
    const uint32_t range = red[1] - mn;
    const int bl = range ? 32 - __clz((int)range) : 0;
    const int shift = bl > LOG2B ? bl - LOG2B : 0;              // (range >> shift) < B
#pragma unroll
    for (int e = 0; e < E; ++e)
        if ((uint32_t)(e * T + tid) < n) {
            const uint32_t b = ((uint32_t)(key[e] >> 32) - mn) >> shift;
            atomicAdd(&cnt[PADC(b)], 1u);
        }
    group_sync<WAVE>();
    // exclusive scan of the B counters: thread t owns counters [t CPT, (t + 1) CPT)
    uint32_t loc[CPT], run = 0u, big = 0u;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const uint32_t v = cnt[tid * (CPT + 1) + k];
        loc[k] = run;
        run += v;
        big = max(big, v);
    }
    const uint32_t incl = wave_inclusive_scan(run);
    big = wave_max(big);
    if ((tid & 63) == 63) red[4 + (tid >> 6)] = incl;
    if ((tid & 63) == 0) atomicMax(&red[2], big);
    group_sync<WAVE>();
    uint32_t toff = incl - run;
    for (int k = 0; k < (tid >> 6); ++k) toff += red[4 + k];
    if (red[2] > DENSE_BUCKET) {                                  // clustered depths: exact fallback (uniform branch)
#pragma unroll
        for (int e = 0; e < E; ++e)
            if ((uint32_t)(e * T + tid) < n) sk[e * T + tid] = key[e];
        uint32_t m = 2;
        while (m < n) m <<= 1;
        group_sync<WAVE>();
        bitonic_network<T, WAVE>(n, m, tid, [&](uint32_t i, uint32_t l) {
            const uint64_t a = sk[i], b = sk[l];
            if (a > b) { sk[i] = b; sk[l] = a; }
        });
        for (uint32_t i = tid; i < n; i += T) out[i] = (uint32_t)sk[i] & ID_MASK;
        return;
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k) cnt[tid * (CPT + 1) + k] = toff + loc[k];
    group_sync<WAVE>();
#pragma unroll
    for (int e = 0; e < E; ++e)
        if ((uint32_t)(e * T + tid) < n) {
            const uint32_t b = ((uint32_t)(key[e] >> 32) - mn) >> shift;
            sk[atomicAdd(&cnt[PADC(b)], 1u)] = key[e];
        }
    group_sync<WAVE>();
    // cnt[b] is now the END of bucket b; rank inside the bucket by counting smaller keys
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t p = (uint32_t)(e * T + tid);
        if (p < n) {
            const uint64_t k = sk[p];
            const uint32_t b = ((uint32_t)(k >> 32) - mn) >> shift;
            const uint32_t st = b ? cnt[PADC(b - 1u)] : 0u, en = cnt[PADC(b)];
            uint32_t r = st;
            for (uint32_t q = st; q < en; ++q) r += sk[q] < k ? 1u : 0u;
            out[r] = (uint32_t)k & ID_MASK;
        }
    }
#undef PADC
}

// The two classes of long lists, one workgroup per list, grid-stride over the class's lists (the grids are sized for the chip, not
// for the worst-case number of lists):
//   class 0 (4096 entries and more)  <1024, 8, 13>: 100 KB of LDS, one workgroup per CU; lists of 8192+ sort in global memory
//   class 1 (1024 .. 4095)           <512, 8, 12>:   50 KB, three per CU.  (One class for everything from 1024 up kept a whole CU busy
//                                    with every 1100-entry list: 178 us at config 5, 350 us with the footprints of a trained scene.)
// (class 0 is only launched where lists of 4096 entries are plausible -- see gsplat_bin --; otherwise class 1's launch covers it
//  (`with_class0`), sorting the odd list that long in global memory: exact, slow, rare)
template <int T, int E, int LOG2B, int CLASS>
__global__ __launch_bounds__(T) void list_sort_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ class_bounds,
                                                      const uint2* __restrict__ ranges, uint64_t* __restrict__ vals,
                                                      uint32_t* __restrict__ sorted_ids, int with_class0) {
    __shared__ SortLds<T, E, LOG2B> s;
    const uint32_t lo = (CLASS == 0 || with_class0) ? 0u : class_bounds[CLASS - 1], hi = class_bounds[CLASS];
    for (uint32_t b = lo + blockIdx.x; b < hi; b += gridDim.x)
        sort_list<T, E, LOG2B, false, true>(s, threadIdx.x, ranges[order[b]], vals, sorted_ids);
}

// classes 1 and 2 in ONE launch (each was a latency-bound kernel of its own: 16 + 12 us at config 3, the chip half empty):
// workgroups [0, mid_blocks) sort the lists of 256..1023 entries, one per workgroup; the others sort the short lists, one per
// WAVE (four per workgroup, no workgroup barrier on that path).  Same LDS footprint either way (17.6 KB).
union SortSmallLds {
    SortLds<256, 4, 11> mid;
    SortLds<64, 4, 9> small[4];
};
__global__ __launch_bounds__(256) void list_sort_small_kernel(const uint32_t* __restrict__ order, const uint32_t* __restrict__ class_bounds,
                                                              uint32_t mid_blocks, const uint2* __restrict__ ranges,
                                                              uint64_t* __restrict__ vals, uint32_t* __restrict__ sorted_ids) {
    __shared__ SortSmallLds s;
    if (blockIdx.x < mid_blocks) {
        const uint32_t lo = class_bounds[1], hi = class_bounds[2];
        for (uint32_t b = blockIdx.x; lo + b < hi; b += mid_blocks)
            sort_list<256, 4, 11, false, false>(s.mid, threadIdx.x, ranges[order[lo + b]], vals, sorted_ids);
    } else {
        const uint32_t lo = class_bounds[2], hi = class_bounds[3];
        const uint32_t wave = threadIdx.x >> 6, stride = (gridDim.x - mid_blocks) * 4u;
        for (uint32_t b = (blockIdx.x - mid_blocks) * 4u + wave; lo + b < hi; b += stride)
            sort_list<64, 4, 9, true, false>(s.small[wave], (int)(threadIdx.x & 63), ranges[order[lo + b]], vals, sorted_ids);
    }
}

// ---- K6 / K7: rasterizer -----------------------------------------------------------------------------
// One wave64 per list (16 x 8 pixels).  The wave is EIGHT groups of 8 lanes: group g owns the 4 x 4-pixel sub-tile
// (g & 3, g >> 2) of the list; lane j of a group owns pixels (j & 3, j >> 2) and (j & 3, (j >> 2) + 2) of the sub-tile, so a
// lane's two pixels form a float2 and the arithmetic runs on packed fp32 (v_pk_fma_f32 ...).
//
// Why groups: a projected Gaussian covers ~57 pixels on the benchmark scene, a list 128: with the whole wave evaluating
// every list entry only 13 % of the lane evaluations were inside the ellipse, and both raster kernels are VALU-bound.
// So the wave walks its depth-sorted list 64 entries at a time; lane l fetches entry l's 64-byte record, stages it in LDS
// (conic pre-scaled for exp2) and tests the entry's bounding box {|du| <= ex, |dv| <= ey} against the 8 sub-tiles; one
// ballot per sub-tile compacts the touching entries, in depth order, into that sub-tile's queue (LDS, 2-byte record
// offsets).  In the inner loop every group pops ITS OWN queue: one iteration composites eight different (sub-tile,
// Gaussian) pairs, 2.7x fewer pixel evaluations than list-wide evaluation (tools/subtile_stats.py); the loop runs to the
// longest of the 8 queues (queues padded with a null record: opacity 0 -> alpha 0).  A Gaussian missing from a sub-tile's
// queue has q > chi, i.e. alpha = 0, on all of its pixels: the composite is unchanged term by term.
// The next chunk's records are fetched while the current chunk is composited.
//
// Launch order: block b takes list order[b] (longest first, from plan_kernel), and the first blocks raise their wave
// priority so that a dense list is not slowed down by light co-resident waves.
typedef float v2f __attribute__((ext_vector_type(2)));
// Exact ellipse / sub-tile test at staging time (subtile_mask_exact), measured on one box: the backward, whose iterations cost
// 2.5x the forward's, gains (217 -> 207 us); the forward loses (85.6 -> 91.5 us) and keeps the box test.
// GSPLAT_TWO_LEVEL=1: colour (forward) and suffix sums (backward) are formed per chunk and joined once per chunk -- fewer roundings
// at full magnitude under hundreds of layers.  Costs 1.6 us in each raster kernel at config 3 and changed none of the measured
// parity figures (the outliers it was written for turned out to be chi-square flips on the far end of needle-shaped Gaussians,
// tools/moments_check.py): off.
#ifndef GSPLAT_TWO_LEVEL
#define GSPLAT_TWO_LEVEL 0
#endif
constexpr int CHUNK = 64;                            // list entries staged per round (one per lane)
constexpr int QCAP = CHUNK + 8;                      // queue capacity: the inner loops read entries in pairs
constexpr uint32_t NULL_OFF = CHUNK * 16;            // byte offset of the null record
constexpr int N_SUB = 8;                             // 4 x 2 sub-tiles of 4 x 4 pixels

// Per-wave statistics for tools/raster_stats.py: only a diagnostics build (-DGSPLAT_DIAGNOSTICS, libgsplat_mi355x_diag.so)
// can register a buffer; the product library always passes NULL.
struct WaveStats { uint32_t list_len, chunks, visited, cycles, begin_lo, launch_index; };   // chunks | duration in 100 MHz ticks << 12; begin: 100 MHz ticks; launch_index | XCD << 24
__device__ __forceinline__ uint32_t xcc_id() {                // (every XCD has its own s_memtime counter)
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xFu;
}
#ifdef GSPLAT_DIAGNOSTICS
WaveStats* g_stats_fwd = nullptr;
WaveStats* g_stats_bwd = nullptr;
u2* g_ref_rect = nullptr;            // tools/ref_pairs_diff.py: the reference's own tile rectangle (F10) and tile count per Gaussian
uint32_t* g_ref_tiles = nullptr;
#define STATS_FWD g_stats_fwd
#define STATS_BWD g_stats_bwd
#else
#define STATS_FWD ((WaveStats*)nullptr)
#define STATS_BWD ((WaveStats*)nullptr)
#endif

constexpr float QK = -0.72134752044448170368f;      // -0.5 * log2(e)

template <int Q>               // Q = slots per queue (the forward kernel: QCAP; the backward kernel, whose queues are capped: fewer)
struct RasterLdsT {
    static constexpr int QSLOTS = Q;
    f4 r0[CHUNK + 1];          // u, v, k A11, 2 k A12                    [CHUNK] = the null record
    f4 r1[CHUNK + 1];          // k A22, opacity, r, g
    f4 r2[CHUNK + 1];          // b, Gaussian id (bits), 0, 0
    uint16_t q[N_SUB][Q];      // per sub-tile: record offsets (16 * entry) of the entries that touch it, depth order
};
using RasterLds = RasterLdsT<QCAP>;
static_assert(sizeof(uint16_t) * N_SUB * QCAP == 16 * QCAP, "queue block = QCAP 16-byte pieces");

struct Candidate {         // one list entry held by one lane between fetch and staging
    f4 q0, q1, q2;
    uint32_t id;
    uint32_t saved_mask;   // (backward) the sub-tile mask the forward left for this pair
};

__device__ __forceinline__ Candidate fetch_candidate(int lane, uint32_t base, uint32_t end, const uint32_t* __restrict__ ids,
                                                     const Rec64* __restrict__ rec, uint32_t id_max,
                                                     const uint8_t* __restrict__ pair_mask = nullptr) {
    Candidate c;
    const uint32_t idx = base + lane;
    c.id = 0;
    c.saved_mask = 0u;
    c.q0 = c.q1 = c.q2 = f4{0.f, 0.f, 0.f, 0.f};
    if (idx < end) {
        if (pair_mask) c.saved_mask = pair_mask[idx];
        c.id = min(ids[idx], id_max);                     // never gather outside the record array
        const Rec64* __restrict__ r = rec + c.id;        // one 64-byte line
        c.q0 = r->r0;
        c.q1 = r->r1;
        c.q2 = r->r2;
    }
    return c;
}

// Which of the list's 8 sub-tiles can the Gaussian touch?  Bit s = its box [u - ex, u + ex] x [v - ey, v + ey] (the padded
// half-extents of {q <= chi} from the projection, gs_math.h) meets the pixel centres of sub-tile s.  (ox, oy) = list origin.
__device__ __forceinline__ uint32_t subtile_mask(const Candidate& c, float ox, float oy) {
    const float x0 = c.q0.x - c.q1.z - ox, x1 = c.q0.x + c.q1.z - ox;
    const float y0 = c.q0.y - c.q1.w - oy, y1 = c.q0.y + c.q1.w - oy;
    uint32_t cm = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) cm |= (x1 >= (float)(4 * k) && x0 <= (float)(4 * k + 3)) ? (1u << k) : 0u;
    uint32_t m = 0u;
    if (y1 >= 0.f && y0 <= 3.f) m |= cm;
    if (y1 >= 4.f && y0 <= 7.f) m |= cm << 4;
    return m;
}

// Stage one chunk: records into LDS, sub-tile queues built.  n = entries offered (uniform, <= CHUNK).  A queue holds at most
// MAXQ entries: when a sub-tile would get more, the chunk is cut to the longest prefix of the list that fits (the rest
// comes back in the next chunk).  Returns {entries taken, length of the longest queue} (uniform); m8 = the lane's sub-tile
// mask (0 beyond the entries taken), ranks = the lane's position in each of its queues (8 bits per sub-tile).
struct Staged { int n, maxc; };
// The same question answered exactly: does {q <= chi} (padded by 1e-3 like the list test of the projection, gs_math.h) reach the
// pixel centres of sub-tile s?  q is convex: its minimum over the sub-tile's rectangle is 0 if the centre is inside, else it lies
// on an edge that FACES the centre.  With X = the centre's x clamped to the rectangle (0 in centre-relative coordinates if it
// is inside the x-range, else the nearer vertical edge) the line x = X is that vertical edge -- or, when there is none, a line
// through the rectangle, whose points are harmless extra candidates -- and the minimum of q along it is a clamped 1-D quadratic
// (v_med3); the same with Y.  min(qx, qy) is then the exact minimum in every case, the centre-inside case (X = Y = 0 -> 0)
// included.  ~135 instructions per entry for the 8 sub-tiles; removes ~14 % of the (sub-tile, Gaussian) pairs the box test lets
// through.  Non-PD conics: every sub-tile.
__device__ __forceinline__ uint32_t subtile_mask_exact(const Candidate& c, float ox, float oy, float chi_pad) {
    const float u = c.q0.x - ox, v = c.q0.y - oy, A = c.q0.z, B = c.q0.w, C = c.q1.x;
    const float tB_C = -B * __builtin_amdgcn_rcpf(C), tB_A = -B * __builtin_amdgcn_rcpf(A), B2 = 2.0f * B;
    float dx0[4], dx1[4], ax[4], bx[4], tx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        dx0[k] = (float)(4 * k) - u; dx1[k] = (float)(4 * k + 3) - u;
        const float X = __builtin_amdgcn_fmed3f(0.0f, dx0[k], dx1[k]);
        ax[k] = A * X * X; bx[k] = B2 * X; tx[k] = tB_C * X;
    }
    uint32_t m = 0u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float dy0 = (float)(4 * r) - v, dy1 = (float)(4 * r + 3) - v;
        const float Y = __builtin_amdgcn_fmed3f(0.0f, dy0, dy1);
        const float cy = C * Y * Y, by = B2 * Y, sy = tB_A * Y;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = __builtin_amdgcn_fmed3f(tx[k], dy0, dy1);          // minimiser of q on the line x = X, clamped to the rectangle
            const float qx = ax[k] + (bx[k] + C * t) * t;
            const float sc = __builtin_amdgcn_fmed3f(sy, dx0[k], dx1[k]);
            const float qy = cy + (by + A * sc) * sc;
            if (!(fminf(qx, qy) > chi_pad)) m |= 1u << (4 * r + k);             // NaN -> touched
        }
    }
    return (A > 0.f && C > 0.f && A * C - B * B > 0.f) ? m : 0xFFu;
}

// MASK: 0 = box test, 1 = box and exact test, 2 = the mask the forward pass saved for this pair (c.saved_mask)
template <int MAXQ, int MASK = 0, class Lds = RasterLds>
__device__ __forceinline__ Staged stage_chunk(Lds& s, const Candidate& c, int n, int lane, float ox, float oy, uint32_t& m8,
                                              uint64_t& ranks, float chi_pad = 0.f) {
    constexpr int QS = Lds::QSLOTS;                         // 16-byte pieces of the queue block
    static_assert(MAXQ >= CHUNK || MAXQ + 4 <= QS, "a capped queue is read up to MAXQ + 3");
    __syncthreads();       // previous chunk's LDS reads are done (single-wave block: orders LDS traffic only)
    m8 = 0u;
    if (lane == 63) { s.r0[CHUNK] = f4{0.f, 0.f, 0.f, 0.f}; s.r1[CHUNK] = f4{0.f, 0.f, 0.f, 0.f}; s.r2[CHUNK] = f4{0.f, 0.f, 0.f, 0.f}; }
    if (lane < n) {
        // conic pre-scaled by k = -0.5 log2(e): the loop evaluates q' = k q and alpha = o * exp2(q') (v_exp_f32 directly)
        s.r0[lane] = f4{c.q0.x, c.q0.y, QK * c.q0.z, (2.0f * QK) * c.q0.w};
        s.r1[lane] = f4{QK * c.q1.x, c.q1.y, c.q2.x, c.q2.y};
        s.r2[lane] = f4{c.q2.z, __uint_as_float(c.id), 0.f, 0.f};
        if (MASK == 2) m8 = c.saved_mask;
        else if (MASK == 1) m8 = subtile_mask_exact(c, ox, oy, chi_pad);      // (conservative by itself: the box test adds nothing)
        else m8 = subtile_mask(c, ox, oy);
    }
    {   // every queue slot -> the null record (QCAP 16-byte pieces)
        const uint32_t nn = NULL_OFF | (NULL_OFF << 16);
        uint4* qv = reinterpret_cast<uint4*>(&s.q[0][0]);
        if (QS >= 64 || lane < QS) qv[lane] = uint4{nn, nn, nn, nn};
        if (QS > 64 && lane < QS - 64) qv[64 + lane] = uint4{nn, nn, nn, nn};
    }
    ranks = 0ull;
    if (MAXQ >= CHUNK) {                      // no cap (forward): queue entries written as the ballots come
        int maxc = 0;
#pragma unroll
        for (int t = 0; t < N_SUB; ++t) {
            const bool hit = (m8 >> t) & 1u;
            const unsigned long long b = __ballot(hit);
            if (hit) s.q[t][__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u))] = (uint16_t)(lane * 16);
            maxc = max(maxc, (int)__popcll(b));
        }
        __syncthreads();
        return Staged{n, maxc};
    }
    unsigned long long bal[N_SUB];
    int maxc = 0;
#pragma unroll
    for (int t = 0; t < N_SUB; ++t) {
        bal[t] = __ballot((m8 >> t) & 1u);
        maxc = max(maxc, (int)__popcll(bal[t]));
    }
    {
        while (maxc > MAXQ) {                 // rare (dense lists of large Gaussians): scalar work only
            n = max(n - 4, MAXQ);             // n = MAXQ always fits
            const unsigned long long keep = (1ull << n) - 1ull;
            maxc = 0;
#pragma unroll
            for (int t = 0; t < N_SUB; ++t) maxc = max(maxc, (int)__popcll(bal[t] & keep));
        }
        if (lane >= n) m8 = 0u;
    }
#pragma unroll
    for (int t = 0; t < N_SUB; ++t) {
        if ((m8 >> t) & 1u) {
            const uint32_t r = __builtin_amdgcn_mbcnt_hi((uint32_t)(bal[t] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal[t], 0u));
            s.q[t][r] = (uint16_t)(lane * 16);
            ranks |= (uint64_t)r << (8 * t);
        }
    }
    __syncthreads();
    return Staged{n, maxc};
}

template <class T>
__device__ __forceinline__ T lds_at(const T* base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}

// min as ONE v_min_f32 (fminf() first canonicalises a scalar operand with a v_max_f32 every time it is used; no NaNs here)
// b is wave-uniform (a kernel argument): taken from its SGPR as src0, not copied to a VGPR first
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(b));
    return r;
}

__device__ __forceinline__ int launch_priority(uint32_t b, uint32_t grid) {
    return b * 64u < grid ? 3 : (b * 16u < grid ? 2 : (b * 4u < grid ? 1 : 0));
}

// (6 waves per SIMD as the compiler leaves it: 76 VGPRs.  Forced to 7 -- 69 VGPRs, no spill -- 90 us against 87.5; to 8: spills, 99 us)
// SAVE (a backward pass will follow: accum is given): the queues come from the exact ellipse / sub-tile test, which costs this
// kernel 6 us more than it saves it, and the resulting mask is left per pair (pair_mask, one byte) for the backward, which then
// needs no test of its own.
template <bool SAVE>
__global__ __launch_bounds__(64) void raster_forward_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ ids,
                                                            const Rec64* __restrict__ rec, const uint32_t* __restrict__ order,
                                                            int lists_x, int H, int W, float chi, float alpha_max,
                                                            float alpha_cutoff, float* __restrict__ image,
                                                            float* __restrict__ accum, WaveStats* __restrict__ stats, uint32_t id_max,
                                                            float* __restrict__ zero_rows, int64_t n_zero_rows, uint8_t* __restrict__ pair_mask) {
    __shared__ RasterLds s;
    const int lane = threadIdx.x;
    if (zero_rows) {        // the coming backward accumulates into grad2d: clear this wave's share now (the kernel is VALU-bound,
                            // the stores ride along; a separate 64 MB fill cost 10 us + a dependent launch)
        const int64_t per = (n_zero_rows + gridDim.x - 1) / gridDim.x, r0 = (int64_t)blockIdx.x * per;
        const int64_t r1 = r0 + per < n_zero_rows ? r0 + per : n_zero_rows;
        const f4 z = f4{0.f, 0.f, 0.f, 0.f};
        for (int64_t r = r0 + lane; r < r1; r += 64) {
            f4* row = reinterpret_cast<f4*>(zero_rows + r * 16);
            row[0] = z; row[1] = z; row[2] = z; row[3] = z;
        }
    }
    const uint32_t list = order[blockIdx.x];
    const int tx = list % lists_x, hy = list / lists_x;
    const int prio = launch_priority(blockIdx.x, gridDim.x);
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    const unsigned long long t_begin = stats ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long t_real = stats ? __builtin_amdgcn_s_memrealtime() : 0ull;      // 100 MHz, one clock for the whole chip
    uint32_t st_chunks = 0, st_visited = 0;
    const int grp = lane >> 3, j = lane & 7;
    const int px = tx * LIST_W + (grp & 3) * 4 + (j & 3);
    const int pya = hy * LIST_H + (grp >> 2) * 4 + (j >> 2), pyb = pya + 2;
    const bool va = (px < W) && (pya < H), vb = (px < W) && (pyb < H);
    const float fpx = (float)px;
    const v2f fpy = {(float)pya, (float)pyb};
    const float ox = (float)(tx * LIST_W), oy = (float)(hy * LIST_H);
    v2f T = {va ? 1.0f : 0.0f, vb ? 1.0f : 0.0f};
    v2f Cr = {0.f, 0.f}, Cg = {0.f, 0.f}, Cb = {0.f, 0.f};
    const uint2 rg = ranges[list];
    const float chik = chi * QK;
    bool alive_any = __any(va || vb);
    uint32_t base = rg.x;
    Candidate cand;
    if (alive_any && base < rg.y) cand = fetch_candidate(lane, base, rg.y, ids, rec, id_max);
    const uint16_t* myq = &s.q[grp][0];
    while (alive_any && base < rg.y) {
        uint32_t m8;
        uint64_t ranks;
        const int maxc = stage_chunk<CHUNK, SAVE ? 1 : 0>(s, cand, (int)min(rg.y - base, (uint32_t)CHUNK), lane, ox, oy, m8, ranks, chi * 1.001f + 1e-4f).maxc;
        if (SAVE && base + (uint32_t)lane < rg.y) pair_mask[base + lane] = (uint8_t)m8;     // 64 contiguous bytes per chunk
        base += CHUNK;
        if (base < rg.y) cand = fetch_candidate(lane, base, rg.y, ids, rec, id_max);   // in flight during the loop below
        ++st_chunks;
        st_visited += (uint32_t)maxc;
        // Two-level sum of the colour: the chunk's terms are added up from zero and join the running colour ONCE per chunk.  A
        // pixel under several hundred layers otherwise rounds its running sum at full magnitude in every step; this way the
        // roundings at full magnitude are one per chunk (the backward pass forms its suffix sums the same way: the two must
        // agree to well below T_i c_i for the deep layers, whose gradients are a difference against them).
#if GSPLAT_TWO_LEVEL
        v2f Lr = {0.f, 0.f}, Lg = {0.f, 0.f}, Lb = {0.f, 0.f};
#else
        v2f& Lr = Cr; v2f& Lg = Cg; v2f& Lb = Cb;
#endif
        for (int k0 = 0; k0 < maxc; k0 += 16) {
          const int k1 = min(k0 + 16, maxc);
          for (int k = k0; k < k1; k += 2) {
            const uint32_t offs = *reinterpret_cast<const uint32_t*>(myq + k);       // two queue entries
            const uint32_t o0 = offs & 0xFFFFu, o1 = offs >> 16;
            const f4 a0 = lds_at(s.r0, o0), b0 = lds_at(s.r1, o0), a1 = lds_at(s.r0, o1), b1 = lds_at(s.r1, o1);
            const float cb0 = lds_at(reinterpret_cast<const float*>(s.r2), o0), cb1 = lds_at(reinterpret_cast<const float*>(s.r2), o1);
            const float du0 = fpx - a0.x, du1 = fpx - a1.x;
            const v2f dv0 = fpy - a0.y, dv1 = fpy - a1.y;
            const v2f q0 = (a0.z * du0 * du0) + dv0 * ((a0.w * du0) + b0.x * dv0);        // k q  (k < 0)
            const v2f q1 = (a1.z * du1 * du1) + dv1 * ((a1.w * du1) + b1.x * dv1);
            const bool i00 = q0.x >= chik, i01 = q0.y >= chik, i10 = q1.x >= chik, i11 = q1.y >= chik;   // q <= chi
            v2f g0, g1;
            g0.x = __builtin_amdgcn_exp2f(q0.x); g0.y = __builtin_amdgcn_exp2f(q0.y);
            g1.x = __builtin_amdgcn_exp2f(q1.x); g1.y = __builtin_amdgcn_exp2f(q1.y);
            v2f al0 = b0.y * g0, al1 = b1.y * g1;
            al0.x = vmin(al0.x, alpha_max); al0.y = vmin(al0.y, alpha_max);
            al1.x = vmin(al1.x, alpha_max); al1.y = vmin(al1.y, alpha_max);
            // alpha = 0 outside the chi-square clip, below the cutoff, and on a dead pixel (T <= 5e-5: the term is masked, and a dead
            // pixel stays dead whether or not its T keeps shrinking) -- ONE select for the three, and w = alpha T needs none
            al0.x = (i00 && al0.x >= alpha_cutoff && T.x > 5e-5f) ? al0.x : 0.0f;
            al0.y = (i01 && al0.y >= alpha_cutoff && T.y > 5e-5f) ? al0.y : 0.0f;
            const v2f w0 = al0 * T;
            T = T - al0 * T;
            al1.x = (i10 && al1.x >= alpha_cutoff && T.x > 5e-5f) ? al1.x : 0.0f;
            al1.y = (i11 && al1.y >= alpha_cutoff && T.y > 5e-5f) ? al1.y : 0.0f;
            const v2f w1 = al1 * T;
            T = T - al1 * T;
            Lr += w0 * b0.z; Lg += w0 * b0.w; Lb += w0 * cb0;
            Lr += w1 * b1.z; Lg += w1 * b1.w; Lb += w1 * cb1;
          }
          if (!__any(T.x > 5e-5f || T.y > 5e-5f)) break;       // every 16 entries: all pixels dead
        }
#if GSPLAT_TWO_LEVEL
        Cr += Lr; Cg += Lg; Cb += Lb;
#endif
        alive_any = __any(T.x > 5e-5f || T.y > 5e-5f);        // dead pixels stay dead
    }
    if (stats && lane == 0)
        stats[list] = WaveStats{rg.y - rg.x, (st_chunks & 0xFFFu) | ((uint32_t)(__builtin_amdgcn_s_memrealtime() - t_real) << 12), st_visited, (uint32_t)(__builtin_amdgcn_s_memtime() - t_begin), (uint32_t)t_real, blockIdx.x | (xcc_id() << 24)};
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

// x + y of a two-pixel value as ONE v_add_f32 the SLP vectoriser cannot see: left to itself it pairs these horizontal adds
// into v_pk_add_f32 and pays three v_mov shuffles per pair (-2.5 % on the backward kernel).
__device__ __forceinline__ float hadd(v2f a) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a.x), "v"(a.y));
    return r;
}

// Eight per-lane partial sums v[0..7] -> their totals over the lane's GROUP of 8 lanes, total i delivered in lane i of the
// group: a reduce-scatter inside every group at once (the 8 groups of the wave reduce 8 different Gaussians' sums in the
// same instructions).  Every level halves the number of live values while it sums over one more lane pairing:
//   row_half_mirror (l <-> 7 - l), quad_perm [2,3,0,1] (l <-> l ^ 2), quad_perm [1,0,3,2] (l <-> l ^ 1):
// levels 2 and 3: two selects and a DPP add per pair of values; level 1: two bank-masked DPP adds (17 instructions for 8 sums).
#define DPP_MOV_F32(x, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xF, 0xF, false))
#define DPP_ADD_F32(x, ctrl) ((x) + DPP_MOV_F32(x, ctrl))
__device__ __forceinline__ float reduce_scatter8(float (&v)[8], int lane) {
    const bool b1 = lane & 2, b0 = lane & 1;
    // level 1 without selects: lane bit 2 is the parity of the lane's DPP bank (4 lanes), so two bank-masked DPP adds write the
    // two halves of the result: banks 0, 2 (lanes 0-3 of every group) get v[i] + mirror(v[i]), banks 1, 3 get v[i+4] + mirror(v[i+4])
    {
        float t0, t1, t2, t3;
        asm("s_nop 1\n"
            "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0x5\n"
            "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0x5\n"
            "v_add_f32_dpp %2, %6, %6 row_half_mirror row_mask:0xf bank_mask:0x5\n"
            "v_add_f32_dpp %3, %7, %7 row_half_mirror row_mask:0xf bank_mask:0x5\n"
            "v_add_f32_dpp %0, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xa\n"
            "v_add_f32_dpp %1, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n"
            "v_add_f32_dpp %2, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xa\n"
            "v_add_f32_dpp %3, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xa"
            : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
            : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        v[0] = t0; v[1] = t1; v[2] = t2; v[3] = t3;
    }
    // levels 2, 3: per pair of values (a, b) and partner lane p, this lane keeps one of the two sums and gives the other to its
    // partner: keep = mine(kept) + partner's(given) -- two selects (1.3 ns each) and ONE DPP add (1.8 ns)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b1 ? v[i + 2] : v[i], give = b1 ? v[i] : v[i + 2];
        v[i] = keep + DPP_MOV_F32(give, 0x4E);                                              // quad_perm [2,3,0,1]
    }
    const float keep = b0 ? v[1] : v[0], give = b0 ? v[0] : v[1];
    return keep + DPP_MOV_F32(give, 0xB1);                                                  // quad_perm [1,0,3,2]
}
// total of x over the lane's group of 8, in every lane of the group
__device__ __forceinline__ float all_reduce8(float x) {
    x = DPP_ADD_F32(x, 0x141);
    x = DPP_ADD_F32(x, 0x4E);
    return DPP_ADD_F32(x, 0xB1);
}
#undef DPP_ADD_F32
#undef DPP_MOV_F32

// backward: longest queue per chunk (sizes the slot block below; see the occupancy note at RasterLdsBwd).
#ifndef GSPLAT_MAXQ_BWD
#define GSPLAT_MAXQ_BWD 24
#endif
constexpr int MAXQ_BWD = GSPLAT_MAXQ_BWD;                // (even: the loop evaluates entries in pairs)
constexpr int QSLOTS_BWD = (MAXQ_BWD + 4 + 7) / 8 * 8;   // the loop reads entries k + 2, k + 3 ahead; rows of 16 bytes

// LDS of the backward kernel.  LDS float atomics are slow on this hardware (a ds_add_f32 wave-instruction with 64 lanes cost
// ~100 LDS cycles here: 230 us of a 450 us kernel), so nothing is accumulated with them: every group writes the nine sums of
// iteration k to its own slot (plain stores), and after the chunk each entry's lane adds up the slots of the sub-tiles it
// was queued in (it knows its rank in every queue) and leaves the row in `acc` for the flush.
// LDS per wave decides the occupancy here (12.8 KB -> 12 waves per CU): the chunk's rows `acc` [entry][9] reuse the record
// arrays, which are dead once the chunk's loop is over (the null record is rewritten by every stage_chunk).
using RasterLdsB = RasterLdsT<QSLOTS_BWD>;
template <bool DET>
struct RasterLdsBwd {
    RasterLdsB f;
    float slots[N_SUB * MAXQ_BWD * 9];   // [sub-tile][queue position][9 sums]
    uint32_t eid[CHUNK];                 // Gaussian id of every entry of the chunk
    uint32_t eslot[DET ? CHUNK : 1];     // (deterministic mode) the row's slot
};
// LDS is handed out in coarse pieces (1280 B by the look of it): at 12 848 B per wave 11 waves were resident per CU (measured with
// tools/raster_stats.py: 2816 waves), at 12 608 B twelve (231 -> 221 us), at 11 456 B and 128 VGPRs fourteen (214 us); sixteen
// (queue cap 18: 10 KB) lose more to chunks cut short than they gain (228 us).
#ifndef GSPLAT_BWD_LDS_MAX
#define GSPLAT_BWD_LDS_MAX 11520
#endif
static_assert(sizeof(RasterLdsBwd<false>) <= GSPLAT_BWD_LDS_MAX, "the backward kernel's LDS per wave decides its occupancy");
static_assert(sizeof(f4) * 3 * (CHUNK + 1) >= sizeof(float) * CHUNK * 9, "acc must fit into the record arrays");

// K7: same traversal as K6 (identical T_i and alive decisions).  For pixel p and Gaussian i:
//   d alpha_i = alive_i T_i (c_i . Gc) - (sum_{k>i} w_k (c_k . Gc)) / (1 - alpha_i),
// the suffix sum being (total - running prefix), total = Gc . C_unclamped, Gc = dL/dO masked by the output clamp.
// Every group reduces its Gaussian's nine sums over its 8 lanes (reduce-scatter: 8 Gaussians at once in the same
// instructions); the rows of a chunk leave with ONE 36-byte global atomic request per (list, Gaussian) pair, 7 rows per
// instruction (the memory-side atomic units take ~20 G requests/s: per sub-tile requests would cost 3x the time).
//
// DET (deterministic gradients): float atomics add in arrival order, so gradients differ from run to run at the 1e-6 level.
// With DET the rows are STORED instead, one row per (list, Gaussian) pair at slot pair_base[Gaussian] + (ordinal of the list
// in the Gaussian's own rectangle), and pair_reduce_kernel adds each Gaussian's rows in that fixed order: bitwise
// reproducible (the sums inside a wave are already in a fixed order).
struct DetArgs {
    const u2* rect; const uint32_t* mask; const uint32_t* tiles; const uint32_t* pair_base;
    float* part;             // [pair capacity][9]
    uint32_t capacity;
};

// (4 waves per SIMD: the kernel needs 131 VGPRs left alone, 128 -- no spill -- when asked; with 11.4 KB of LDS 14 waves fit a CU)
template <bool DET>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void raster_backward_kernel(const uint2* __restrict__ ranges, const uint32_t* __restrict__ ids,
                                                             const Rec64* __restrict__ rec, const uint32_t* __restrict__ order,
                                                             int lists_x, int H, int W, float chi, float alpha_max,
                                                             float alpha_cutoff, const float* __restrict__ accum,
                                                             const float* __restrict__ gimg, float* __restrict__ grad2d,
                                                             WaveStats* __restrict__ stats, uint32_t id_max, DetArgs det,
                                                             const uint8_t* __restrict__ pair_mask) {
    __shared__ RasterLdsBwd<DET> sb;
    RasterLdsB& s = sb.f;
    const int lane = threadIdx.x;
    const uint32_t list = order[blockIdx.x];
    const uint2 rg = ranges[list];
    if (rg.x >= rg.y) return;
    const int prio = launch_priority(blockIdx.x, gridDim.x);
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
    const unsigned long long t_begin = stats ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long t_real = stats ? __builtin_amdgcn_s_memrealtime() : 0ull;      // 100 MHz, one clock for the whole chip
    uint32_t st_chunks = 0, st_visited = 0;
    const int tx = list % lists_x, hy = list / lists_x;
    const int grp = lane >> 3, j = lane & 7;
    const int px = tx * LIST_W + (grp & 3) * 4 + (j & 3);
    const int pya = hy * LIST_H + (grp >> 2) * 4 + (j >> 2), pyb = pya + 2;
    const bool va = (px < W) && (pya < H), vb = (px < W) && (pyb < H);
    const float fpx = (float)px;
    const v2f fpy = {(float)pya, (float)pyb};
    const float ox = (float)(tx * LIST_W), oy = (float)(hy * LIST_H);
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
    const float chik = chi * QK, amax = alpha_max;
    bool alive_any = __any(va || vb);
    uint32_t base = rg.x;
    Candidate cand;
    if (alive_any) cand = fetch_candidate(lane, base, rg.y, ids, rec, id_max, pair_mask);
    const int my_g = lane / 9, my_k = lane - 9 * my_g;             // flush: lane i carries sum my_k of the round's row my_g
    const uint16_t* myq = &s.q[grp][0];
    float* const myslot = &sb.slots[grp * MAXQ_BWD * 9 + j];       // + 9 k: where lane j of the group puts sum j of iteration k
    float* const acc = reinterpret_cast<float*>(&sb.f);            // [entry][9]: over the record arrays, between a chunk's loop and the next stage
    while (alive_any && base < rg.y) {
        uint32_t m8;
        uint64_t ranks;
        const Staged sg = stage_chunk<MAXQ_BWD, 2, RasterLdsB>(s, cand, (int)min(rg.y - base, (uint32_t)CHUNK), lane, ox, oy, m8, ranks);
        const int n = sg.n, maxc = sg.maxc;
        sb.eid[lane] = cand.id;
        base += (uint32_t)n;
        if (base < rg.y) cand = fetch_candidate(lane, base, rg.y, ids, rec, id_max, pair_mask);   // in flight during the loop below
        ++st_chunks;
        st_visited += (uint32_t)maxc;
        int kdone = 0;                       // iterations executed (uniform): slots [0, kdone) of every queue are valid
        v2f lpre = {0.f, 0.f};               // sum of w (c . G) over the chunk's entries so far (two-level, as the forward's colour)
        // One queue entry: the group's 16 pixels against one Gaussian; the nine sums go to slot k of the group's queue.
        auto entry = [&](const f4& a, const f4& b, const float cbl, const int k) {
            const float go = b.y;
            const float du = fpx - a.x;
            const v2f dv = fpy - a.y;
            const float c0 = a.z * du * du, c1 = a.w * du;
            const v2f q = c0 + dv * (c1 + b.x * dv);                                   // k q  (k < 0)
            const bool i0 = q.x >= chik, i1 = q.y >= chik;                              // q <= chi
            v2f g;
            g.x = __builtin_amdgcn_exp2f(q.x);
            g.y = __builtin_amdgcn_exp2f(q.y);
            const v2f og = go * g;
            // alpha = min(o g, alpha_max) where q <= chi and that is >= alpha_cutoff (<=> o g >= alpha_cutoff: cutoff <= alpha_max), else 0
            // (NOT folded with the alive test as in the forward kernel: alpha would then wait for the previous entry's T, and the
            //  chain alpha -> 1 / (1 - alpha) -> d alpha of two consecutive entries could no longer overlap: +7 us, measured)
            const bool p0 = i0 && og.x >= alpha_cutoff, p1 = i1 && og.y >= alpha_cutoff;
            const float cl0 = vmin(og.x, amax), cl1 = vmin(og.y, amax);                 // (unconditional: a select, not a branch)
            v2f al;
            al.x = p0 ? cl0 : 0.0f; al.y = p1 ? cl1 : 0.0f;
            const bool alive0 = T.x > 5e-5f, alive1 = T.y > 5e-5f;
            v2f w = al * T;
            w.x = alive0 ? w.x : 0.0f; w.y = alive1 ? w.y : 0.0f;
            const v2f sdot = b.z * Gr + b.w * Gg + cbl * Gb;
            const v2f ar = w * Gr, ag = w * Gg, ab = w * Gb;
#if GSPLAT_TWO_LEVEL
            lpre += w * sdot;
            const v2f sfx = suffix - lpre;                                 // the sum over k > i
#else
            suffix -= w * sdot;
            const v2f sfx = suffix;
#endif
            v2f om = 1.0f - al;
            om.x = __builtin_amdgcn_rcpf(om.x); om.y = __builtin_amdgcn_rcpf(om.y);   // 1 - alpha >= 0.01
            v2f dal = T * sdot - sfx * om;
            // the pixel is alive, alpha passed its two tests, and clamp_max passes the gradient where o g <= alpha_max (render.py:372)
            dal.x = (alive0 && p0 && og.x <= alpha_max) ? dal.x : 0.0f;
            dal.y = (alive1 && p1 && og.y <= alpha_max) ? dal.y : 0.0f;
            // a = dL/d alpha * g.  dL/d opacity = sum a, and dL/dq = -0.5 o a: the factor -0.5 o is the same for all pixels of a
            // Gaussian, so the moments are taken of `a` and project_backward_kernel multiplies once per Gaussian:
            //   d u = o (A11 Mx + A12 My), d v = o (A12 Mx + A22 My), d A11 = -0.5 o Mxx, d A12 = -o Mxy, d A22 = -0.5 o Myy
            const v2f ao = dal * g;
            const v2f dva = dv * ao;
            const float m0 = hadd(ao), my = hadd(dva);
            float r[8];
            r[0] = du * m0;                                                // Mx  = sum du a
            r[1] = my;                                                     // My  = sum dv a
            r[2] = du * r[0];                                              // Mxx = sum du^2 a
            r[3] = du * my;                                                // Mxy = sum du dv a
            r[4] = hadd(dv * dva);                                         // Myy = sum dv^2 a
            r[5] = m0;                                                     // M0  = sum a = dL/d opacity
            r[6] = hadd(ar); r[7] = hadd(ag);                              // d r, d g
            const float tot_b = all_reduce8(hadd(ab));                     // d b
            myslot[k * 9] = reduce_scatter8(r, lane);
            if (j == 0) myslot[k * 9 + 8] = tot_b;         // (two unconditional stores instead -- 3 instructions fewer -- measured no gain)
            T = T - al * T;
        };
        // Software pipeline over the queue (LDS latency is not covered by occupancy here: 3 waves per SIMD), two entries per step
        // in two register sets: entry k + 2 is requested into set 0 as soon as entry k has been evaluated from it, while entry
        // k + 1 is evaluated from set 1, and so on -- no register copies.  An odd queue ends on a null record (its slot gets zeros).
        const float* r2f = reinterpret_cast<const float*>(s.r2);
        uint32_t oo = *reinterpret_cast<const uint32_t*>(myq);
        f4 a0 = lds_at(s.r0, oo & 0xFFFFu), b0 = lds_at(s.r1, oo & 0xFFFFu), a1 = lds_at(s.r0, oo >> 16), b1 = lds_at(s.r1, oo >> 16);
        float cb0 = lds_at(r2f, oo & 0xFFFFu), cb1 = lds_at(r2f, oo >> 16);
        for (int k0 = 0; k0 < maxc; k0 += 8) {
          const int k1 = min(k0 + 8, maxc);
          for (int k = k0; k < k1; k += 2) {
            oo = *reinterpret_cast<const uint32_t*>(myq + k + 2);                       // entries k + 2, k + 3 (null past the end; k + 3 < QCAP)
            entry(a0, b0, cb0, k);
            a0 = lds_at(s.r0, oo & 0xFFFFu); b0 = lds_at(s.r1, oo & 0xFFFFu); cb0 = lds_at(r2f, oo & 0xFFFFu);
            entry(a1, b1, cb1, k + 1);
            a1 = lds_at(s.r0, oo >> 16); b1 = lds_at(s.r1, oo >> 16); cb1 = lds_at(r2f, oo >> 16);
          }
          kdone = k1;
          if (!__any(T.x > 5e-5f || T.y > 5e-5f)) break;          // every 8 entries: all pixels dead
        }
#if GSPLAT_TWO_LEVEL
        suffix -= lpre;                                       // (once per chunk)
#endif
        alive_any = __any(T.x > 5e-5f || T.y > 5e-5f);        // dead pixels stay dead
        __syncthreads();
        {   // entry `lane`: add up the slots of the sub-tiles it was queued in
            float tot[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < N_SUB; ++t) {
                const int r = (int)((ranks >> (8 * t)) & 0xFFu);
                if (((m8 >> t) & 1u) && r < kdone) {
                    const float* p = &sb.slots[(t * MAXQ_BWD + r) * 9];
#pragma unroll
                    for (int v = 0; v < 9; ++v) tot[v] += p[v];
                }
            }
#pragma unroll
            for (int v = 0; v < 9; ++v) acc[lane * 9 + v] = tot[v];
        }
        if (DET && lane < n) {       // entry `lane`: its row's slot = first slot of its Gaussian + ordinal of this list in its rectangle
            const uint32_t id = sb.eid[lane];
            const u2 rc = det.rect[id];
            const uint32_t mk = det.mask[id], nt = det.tiles[id];
            const int x0 = (int)(rc.x & 0xFFFFu), y0 = (int)(rc.x >> 16), x1 = (int)(rc.y & 0xFFFFu);
            const uint32_t bit = (uint32_t)((hy - y0) * (x1 - x0 + 1) + (tx - x0));          // row-major, like for_each_list
            uint32_t ord;
            if (rect_is_big(rc)) {               // a large Gaussian: its lists are row spans (for_each_big_row): lists in the rows above + offset in this row
                const Rec64* r = rec + id;
                const f4 q0 = r->r0, q1 = r->r1, q3 = r->pad;
                const float kk[4] = {q3.x, q3.y, q3.z, q3.w};
                const BigSpanK bk = big_span_setup(q0.x, q0.y, q1.z, q1.w, kk, x0, x1);
                ord = 0u;
                for (int y = y0; y < hy; ++y) {
                    const RowSpan sp = big_row_span(bk, y);
                    if (sp.xb >= sp.xa) ord += (uint32_t)(sp.xb - sp.xa + 1);
                }
                ord += (uint32_t)(tx - big_row_span(bk, hy).xa);
            } else {
                ord = (uint32_t)__popc(mk & ((1u << (bit & 31u)) - 1u));
            }
            (void)nt;
            sb.eslot[lane] = det.pair_base[id] + ord;
        }
        __syncthreads();
        // the chunk's rows -> grad2d: 7 rows x 9 sums per atomic instruction, one 36-byte request per row
        for (int t0 = 0; t0 < n; t0 += 7) {
            const int c = t0 + my_g;
            if (lane < 63 && c < n) {
                const float val = acc[c * 9 + my_k];
                if (DET) {
                    const uint32_t slot = sb.eslot[c];
                    if (slot < det.capacity) det.part[(int64_t)slot * 9 + my_k] = val;
                } else if (val != 0.0f) {
                    atomicAdd(&grad2d[(int64_t)sb.eid[c] * 16 + my_k], val);
                }
            }
        }
    }
    if (stats && lane == 0)
        stats[list] = WaveStats{rg.y - rg.x, (st_chunks & 0xFFFu) | ((uint32_t)(__builtin_amdgcn_s_memrealtime() - t_real) << 12), st_visited, (uint32_t)(__builtin_amdgcn_s_memtime() - t_begin), (uint32_t)t_real, blockIdx.x | (xcc_id() << 24)};
}

// ---- deterministic mode: slots of the (list, Gaussian) rows and their fixed-order sum ------------------------------------
constexpr int PB_BLOCK = 2048;                       // Gaussians per block of the two kernels below
__global__ __launch_bounds__(256) void tile_block_sum_kernel(int64_t n, const uint32_t* __restrict__ tiles, uint32_t* __restrict__ block_sum) {
    __shared__ uint32_t ws[4];
    uint32_t t = 0u;
    for (int k = 0; k < PB_BLOCK / 256; ++k) {
        const int64_t i = (int64_t)blockIdx.x * PB_BLOCK + k * 256 + threadIdx.x;
        t += i < n ? tiles[i] : 0u;
    }
    for (int sft = 32; sft > 0; sft >>= 1) t += (uint32_t)__shfl_xor((int)t, sft);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// pair_base[i] = number of rows of the Gaussians before i (exclusive scan of tiles[]): thread t of a block owns 8 CONSECUTIVE
// Gaussians, so the scan order is the index order.
__global__ __launch_bounds__(256) void pair_base_kernel(int64_t n, const uint32_t* __restrict__ tiles, const uint32_t* __restrict__ block_sum,
                                                        uint32_t* __restrict__ pair_base) {
    __shared__ uint32_t ws[4], s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t before = 0u;
    for (int b = tid; b < (int)blockIdx.x; b += 256) before += block_sum[b];
    for (int sft = 32; sft > 0; sft >>= 1) before += (uint32_t)__shfl_xor((int)before, sft);
    if (lane == 0) ws[wave] = before;
    __syncthreads();
    if (tid == 0) s_base = ws[0] + ws[1] + ws[2] + ws[3];
    __syncthreads();
    constexpr int K = PB_BLOCK / 256;
    uint32_t v[K], run = 0u;
    const int64_t i0 = (int64_t)blockIdx.x * PB_BLOCK + (int64_t)tid * K;
#pragma unroll
    for (int k = 0; k < K; ++k) { v[k] = i0 + k < n ? tiles[i0 + k] : 0u; run += v[k]; }
    const uint32_t incl = wave_inclusive_scan(run);
    __syncthreads();
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    uint32_t st = s_base + incl - run;
    for (int k = 0; k < wave; ++k) st += ws[k];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (i0 + k < n) pair_base[i0 + k] = st;
        st += v[k];
    }
}

// grad2d[i][0..8] = sum of Gaussian i's rows, in the order of its lists (row-major in its rectangle): the same order every run.
__global__ __launch_bounds__(256) void pair_reduce_kernel(int64_t n, const uint32_t* __restrict__ tiles, const uint32_t* __restrict__ pair_base,
                                                          const float* __restrict__ part, uint32_t capacity, float* __restrict__ grad2d) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float t[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint32_t nt = tiles[i], pb = pair_base[i];
    for (uint32_t k = 0; k < nt && pb + k < capacity; ++k) {
        const float* row = part + (int64_t)(pb + k) * 9;
#pragma unroll
        for (int v = 0; v < 9; ++v) t[v] += row[v];
    }
    float* o = grad2d + i * 16;
#pragma unroll
    for (int v = 0; v < 9; ++v) o[v] = t[v];
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

// JAC (fused inputs): the forward left d rgb / d logit and d logit / d position in project_state (GSPLAT_PROJECT_SAVE_SH_JACOBIAN),
// so the 192 bytes of SH coefficients are not read again: 48 instead of 192 bytes per visible Gaussian, and no dY accumulators.
// ADAM (fused inputs, saved Jacobian, not factored): the 45 f_rest gradients of a Gaussian are not written: the rows are stepped in
// place (adam_rows) -- the 192 of the 236 gradient bytes per Gaussian neither leave this kernel nor come back into the optimiser's.
// ACC (fused inputs, saved Jacobian, not factored): every gradient is ADDED to what `out` holds -- the gradients of the views of one
// iteration summed by the kernel that forms them, instead of a pass of the host's autograd per view (read two, write one).
template <bool FUSED, bool JAC = false, bool ADAM = false, bool ACC = false>
__global__ __launch_bounds__(64) void project_backward_kernel(gsplat_gaussians g, const Camera* __restrict__ camp, ViewK vk,
                                                              const uint32_t* __restrict__ tiles, const float* __restrict__ grad2d,
                                                              gsplat_gaussian_grads out, bool factored, const float* __restrict__ kj_in,
                                                              AdamRest ar) {
    static_assert(!ADAM || (FUSED && JAC), "the in-place step needs the direct path");
    static_assert(!ACC || (FUSED && JAC && !ADAM), "accumulation is built for the direct path");
    // DIRECT (fused inputs, saved Jacobian): nothing is staged IN (the 44 bytes of geometry are loaded by the lanes), and of the
    // gradients only the 45 f_rest rows go OUT through LDS (the rows of 1 / 3 / 4 floats are stored by the lanes): 11 520 B per
    // wave instead of 15 104 -> 14 waves per CU instead of 10.
    constexpr bool DIRECT = FUSED && JAC;
    __shared__ float s_geo[DIRECT ? 4 : sizeof(ProjectLds<FUSED>) / 4];
    ProjectLds<FUSED>& s = *reinterpret_cast<ProjectLds<FUSED>*>(s_geo);
    __shared__ float s_dc[FUSED && !DIRECT ? 64 * 3 : 4];
    __shared__ float s_rest[FUSED ? 64 * 45 : 4];
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + lane;
    const Camera cam = *camp;
    const bool vis = (i < g.n) && tiles[i] != 0;
    const bool any_vis = __any(vis);
    float r9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float kj[12];
    GaussIn in;
    if (any_vis) {
        if (DIRECT) {
            if (vis) {
#pragma unroll
                for (int k = 0; k < 3; ++k) in.p[k] = g.pos[i * 3 + k];
                in.o_raw = g.opacity_raw[i];
                const f4 q = *reinterpret_cast<const f4*>(g.q_raw + i * 4);
                in.qr[0] = q.x; in.qr[1] = q.y; in.qr[2] = q.z; in.qr[3] = q.w;
#pragma unroll
                for (int k = 0; k < 3; ++k) in.sr[k] = g.scale_raw[i * 3 + k];
            }
        } else {
            stage_geometry<FUSED>(s, g, row0, lane);
        }
        if (FUSED && !JAC) {
            stage_rows<3>(s_dc, g.f_dc, row0, g.n, lane);
            stage_rows<45>(s_rest, g.f_rest, row0, g.n, lane);
        }
        if (JAC && vis) {
            const f4* src = reinterpret_cast<const f4*>(kj_in + i * 12);
            const f4 k0 = src[0], k1 = src[1], k2 = src[2];
            kj[0] = k0.x; kj[1] = k0.y; kj[2] = k0.z; kj[3] = k0.w; kj[4] = k1.x; kj[5] = k1.y; kj[6] = k1.z; kj[7] = k1.w;
            kj[8] = k2.x; kj[9] = k2.y; kj[10] = k2.z; kj[11] = k2.w;
        }
        if (vis) {
            const f4 g0 = *reinterpret_cast<const f4*>(grad2d + i * 16), g1 = *reinterpret_cast<const f4*>(grad2d + i * 16 + 4);
            r9[0] = g0.x; r9[1] = g0.y; r9[2] = g0.z; r9[3] = g0.w; r9[4] = g1.x; r9[5] = g1.y; r9[6] = g1.z; r9[7] = g1.w;
            r9[8] = grad2d[i * 16 + 8];
        }
    }
    if (!DIRECT) __syncthreads();
    GradOut go;
    float gdc[3] = {0.f, 0.f, 0.f};                       // (DIRECT) d L / d f_dc of this lane's Gaussian
    float* const dc_rows = DIRECT ? gdc : s_dc + lane * 3;
    if (vis) {
        if (!DIRECT) in = gauss_from_lds<FUSED>(s, lane);
        go = project_backward_core(in, FUSED, ShCoefLds{dc_rows, s_rest + lane * 45},
                                   ShEmitLds{dc_rows, s_rest + lane * 45}, cam, vk, true, r9, true, JAC ? kj : nullptr);
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
            for (int k = 0; k < 3; ++k) dc_rows[k] = 0.f;
            for (int k = 0; k < 45; ++k) s_rest[lane * 45 + k] = 0.f;
        }
    }
    if (DIRECT) {
        if (ACC) {
            if (vis) {                                      // (a Gaussian that is not visible adds nothing)
#pragma unroll
                for (int k = 0; k < 3; ++k) out.pos[i * 3 + k] += go.p[k];
                out.opacity_raw[i] += go.o_raw;
                const f4 q0 = *reinterpret_cast<const f4*>(out.q_raw + i * 4);
                *reinterpret_cast<f4*>(out.q_raw + i * 4) = f4{q0.x + go.qr[0], q0.y + go.qr[1], q0.z + go.qr[2], q0.w + go.qr[3]};
#pragma unroll
                for (int k = 0; k < 3; ++k) out.scale_raw[i * 3 + k] += go.sr[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) out.f_dc[i * 3 + k] += gdc[k];
            }
            if (any_vis) {
                __syncthreads();
                unstage_rows<45, true>(out.f_rest, s_rest, row0, g.n, lane);
            }
            return;
        }
        if (i < g.n) {                                      // every row is written (zeros for a Gaussian that is not visible)
#pragma unroll
            for (int k = 0; k < 3; ++k) out.pos[i * 3 + k] = go.p[k];
            out.opacity_raw[i] = go.o_raw;
            *reinterpret_cast<f4*>(out.q_raw + i * 4) = f4{go.qr[0], go.qr[1], go.qr[2], go.qr[3]};
#pragma unroll
            for (int k = 0; k < 3; ++k) out.scale_raw[i * 3 + k] = go.sr[k];
            if (factored) {
                // d L / d f_dc = (d L / d colour logit) * Y0: hand out the 3 logit gradients instead of the 48 SH gradients
                if (out.color) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) out.color[i * 3 + k] = gdc[k] * (1.0f / GS_K0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) out.f_dc[i * 3 + k] = gdc[k];
            }
        }
        if (!factored) {
            __syncthreads();
            if (ADAM) {
                const DevCounts* cnt = reinterpret_cast<const DevCounts*>(ar.counts);
                if (cnt->n_visible > 0 && cnt->n_binned <= ar.capacity) adam_rows<45>(ar, s_rest, row0, g.n, lane);      // (uniform)
            } else {
                unstage_rows<45>(out.f_rest, s_rest, row0, g.n, lane);
            }
        }
        return;
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
        if (factored) {
            // d L / d f_dc = (d L / d colour logit) * Y0: hand out the 3 logit gradients instead of the 48 SH gradients
            // (gsplat_sh_accumulate rebuilds those, for any number of views, from logit gradients and view directions)
#pragma unroll
            for (int k = 0; k < 3; ++k) s_dc[lane * 3 + k] *= 1.0f / GS_K0;      // each lane its own slots: no barrier needed
            __syncthreads();
            if (out.color) unstage_rows<3>(out.color, s_dc, row0, g.n, lane);
        } else {
            unstage_rows<3>(out.f_dc, s_dc, row0, g.n, lane);
            unstage_rows<45>(out.f_rest, s_rest, row0, g.n, lane);
        }
    } else {
        unstage_rows<9>(out.sigma, s.a, row0, g.n, lane);
        unstage_rows<3>(out.color, s.b, row0, g.n, lane);
    }
}

// ---- colour-logit gradients straight from the raster backward's sums (data-parallel exchange) -----------
// d L / d logit[ch] = (d L / d colour[ch]) * c (1 - c): only needs the raster backward's colour sums and the colour in the
// record, so the all-gather of the logit gradients can start BEFORE gsplat_project_backward and overlap it.
__global__ __launch_bounds__(256) void logit_grad_kernel(int64_t n, const uint32_t* __restrict__ tiles, const Rec64* __restrict__ rec,
                                                         const float* __restrict__ grad2d, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g[3] = {0.f, 0.f, 0.f};
    if (tiles[i] != 0u) {
        const f4 c = rec[i].r2;
        const float* r = grad2d + i * 16;
        g[0] = r[6] * c.x * (1.f - c.x); g[1] = r[7] * c.y * (1.f - c.y); g[2] = r[8] * c.z * (1.f - c.z);
    }
    out[i * 3] = g[0]; out[i * 3 + 1] = g[1]; out[i * 3 + 2] = g[2];
}

// ---- SH gradients from logit gradients (data-parallel exchange) ---------------------------------------
// grad f_dc[i, ch] = scale * sum_v glogit[v, i, ch] * Y0,  grad f_rest[i, ch * 15 + k - 1] = scale * sum_v glogit[v, i, ch] * Y_k(d_v(i)),
// d_v(i) = unit vector from camera v's position to Gaussian i (spherical_harmonics.py:132-133).
__global__ __launch_bounds__(64) void sh_accumulate_kernel(int64_t n, int n_views, const float* __restrict__ pos, const float* __restrict__ eyes,
                                                           const float* __restrict__ glogit, float scale, float* __restrict__ grad_f_dc,
                                                           float* __restrict__ grad_f_rest) {
    __shared__ float s_pos[64 * 3], s_dc[64 * 3], s_rest[64 * 45];
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + lane;
    stage_rows<3>(s_pos, pos, row0, n, lane);
    __syncthreads();
    float acc[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) acc[k] = 0.f;
    if (i < n) {
        const float p[3] = {s_pos[lane * 3], s_pos[lane * 3 + 1], s_pos[lane * 3 + 2]};
        for (int v = 0; v < n_views; ++v) {
            const float* gl = glogit + ((int64_t)v * n + i) * 3;
            const float g0 = gl[0], g1 = gl[1], g2 = gl[2];
            if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;       // not binned in this view
            const float eye[3] = {eyes[v * 3], eyes[v * 3 + 1], eyes[v * 3 + 2]};
            ShMid sm;
            sh_basis(p, eye, sm);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc[k] += g0 * sm.Y[k]; acc[16 + k] += g1 * sm.Y[k]; acc[32 + k] += g2 * sm.Y[k];
            }
        }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        s_dc[lane * 3 + ch] = scale * acc[ch * 16];
#pragma unroll
        for (int k = 1; k < 16; ++k) s_rest[lane * 45 + ch * 15 + (k - 1)] = scale * acc[ch * 16 + k];
    }
    __syncthreads();
    unstage_rows<3>(grad_f_dc, s_dc, row0, n, lane);
    unstage_rows<45>(grad_f_rest, s_rest, row0, n, lane);
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
// Stand-alone evaluate_sh backward, row blocks staged through LDS like K8 (a lane reading and writing its own 180-byte rows
// directly touched every cache line 45 times: 120 us against 60 us for 1 M Gaussians).
__global__ __launch_bounds__(64) void evaluate_sh_backward_kernel(int64_t n, const float* __restrict__ dc, const float* __restrict__ rest,
                                                                  const float* __restrict__ pts, const float* __restrict__ c2w,
                                                                  const float* __restrict__ gcol, float* __restrict__ gdc,
                                                                  float* __restrict__ grest, float* __restrict__ gpts) {
    __shared__ float s_pos[64 * 3], s_gc[64 * 3], s_dc[64 * 3], s_rest[64 * 45];
    const int lane = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * 64, i = row0 + lane;
    stage_rows<3>(s_pos, pts, row0, n, lane);
    stage_rows<3>(s_gc, gcol, row0, n, lane);
    stage_rows<3>(s_dc, dc, row0, n, lane);
    stage_rows<45>(s_rest, rest, row0, n, lane);
    float m[16];
    for (int k = 0; k < 16; ++k) m[k] = c2w[k];
    Camera cam;
    build_camera(m, cam);
    __syncthreads();
    float gp[3] = {0.f, 0.f, 0.f};
    if (i < n) {
        const float p[3] = {s_pos[lane * 3], s_pos[lane * 3 + 1], s_pos[lane * 3 + 2]};
        const float gc[3] = {s_gc[lane * 3], s_gc[lane * 3 + 1], s_gc[lane * 3 + 2]};
        ShMid sm;
        sh_basis(p, cam.eye, sm);
        ShCoefLds coef{s_dc + lane * 3, s_rest + lane * 45};
        float rgb[3];
        sh_colour(sm, coef, rgb);
        sh_colour_backward(sm, coef, rgb, gc, ShEmitLds{s_dc + lane * 3, s_rest + lane * 45}, gp);     // gradients over the coefficients
    }
    s_pos[lane * 3] = gp[0]; s_pos[lane * 3 + 1] = gp[1]; s_pos[lane * 3 + 2] = gp[2];                  // own slots: read above
    __syncthreads();
    unstage_rows<3>(gdc, s_dc, row0, n, lane);
    unstage_rows<45>(grest, s_rest, row0, n, lane);
    unstage_rows<3>(gpts, s_pos, row0, n, lane);
}

inline unsigned blocks256(int64_t n) { return (unsigned)((n + 255) / 256); }
// Dynamic LDS of the two binning kernels: nb counters -- padded so that only `per_cu` of their workgroups fit a CU when the image has
// many bins.  Every resident block of bin_scatter_kernel keeps one open line per bin it writes to; with 8 blocks per CU and the 1012
// bins of a 4K image that is 2 M lines under 32 MB of L2: lines leave half written and come back (config 5: 202 us at 3 blocks per
// CU, 285 at 8); a small image has few bins and wants the occupancy (config 6: 131 us at 3, 108 at 8).
inline size_t bin_lds_bytes(size_t nb, size_t static_bytes, size_t least = 2) {
    size_t per_cu = 2048 / (nb ? nb : 1);
#ifdef GSPLAT_BIN_PER_CU
    per_cu = GSPLAT_BIN_PER_CU;
#endif
    if (per_cu < least) per_cu = least;
    if (per_cu >= 8) return nb * 4;
    const size_t want = (160 * 1024) / (per_cu + 1) + 1024;          // one more block must not fit
    const size_t dyn = want > static_bytes ? want - static_bytes : 0;
    return std::min(std::max(nb * 4, dyn), (size_t)(64 * 1024 - 256));
}
// workgroups of the binning kernels that look for LARGE Gaussians (ranges of 64, grid-stride): enough to fill the chip when every
// Gaussian is large, few enough to cost a scene without any (config 3: 15 625 ranges, one flag word each) almost nothing
#ifndef GSPLAT_BIG_BLOCKS_MAX
#define GSPLAT_BIG_BLOCKS_MAX 2048u
#endif
inline unsigned big_bin_blocks(int64_t n) { return std::min((unsigned)((n + 63) / 64), GSPLAT_BIG_BLOCKS_MAX); }
// (bin_count_kernel: fewer, each adding its ranges up before it touches the global totals)
inline unsigned big_count_blocks(int64_t n) { return std::min((unsigned)((n + 63) / 64), 768u); }
inline unsigned blocks64(int64_t n) { return (unsigned)((n + 63) / 64); }

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// ======================================================================================================
// C ABI
// ======================================================================================================
extern "C" {

int gsplat_abi_version(void) { return GSPLAT_ABI_VERSION; }

#ifdef GSPLAT_DIAGNOSTICS
// Diagnostics build only (libgsplat_mi355x_diag.so, loaded by tools/ alone; not declared in include/gsplat_mi355x.h and not
// in the product library): register device buffers of lists * 16 bytes each that the raster kernels fill with per-wave
// statistics; pass NULL to switch the statistics off again.
void gsplat_debug_set_stats(void* fwd, void* bwd) { g_stats_fwd = (WaveStats*)fwd; g_stats_bwd = (WaveStats*)bwd; }
// device buffers of n x 8 and n x 4 bytes that the projection kernel fills with the reference's tile rectangle / tile count
void gsplat_debug_set_ref_rect(void* rect, void* tiles) { g_ref_rect = (u2*)rect; g_ref_tiles = (uint32_t*)tiles; }
#endif

const char* gsplat_last_error(void) { return g_err; }

int gsplat_classify_counts(const gsplat_counts* c) {
    if (!c) return GSPLAT_SCENE_ALL_CULLED;
    if (c->n_survivors == 0) return GSPLAT_SCENE_ALL_CULLED;       // render.py:109-112,139-142,190-193
    if (c->n_visible == 0) return GSPLAT_SCENE_ALL_OFFSCREEN;      // render.py:235-236
    return GSPLAT_SCENE_OK;
}

int64_t gsplat_project_state_bytes(int64_t n, const gsplat_view* v) {
    if (!v || v->H <= 0 || v->W <= 0) return -1;
    return carve_project(nullptr, n > 0 ? n : 1, n_lists(v)).bytes;
}

int64_t gsplat_project_scratch_bytes(int64_t n) { (void)n; return up(sizeof(CounterBlock)); }   // the persistent counter block

// bin_state = sorted ids [capacity] (4 B each), then one byte per pair: the sub-tile mask gsplat_rasterize_forward leaves for
// gsplat_rasterize_backward when it is given `accum`
inline uint8_t* pair_mask_of(const void* bin_state, int64_t pair_capacity) {
    return (uint8_t*)bin_state + up((pair_capacity > 0 ? pair_capacity : 1) * 4);
}

int64_t gsplat_bin_state_bytes(int64_t pair_capacity, const gsplat_view* v) {
    if (!v) return -1;
    const int64_t cap = pair_capacity > 0 ? pair_capacity : 1;
    return up(cap * 4) + up(cap);
}

int64_t gsplat_bin_scratch_bytes(int64_t pair_capacity, const gsplat_view* v) {
    if (!v) return -1;
    return carve_bin_scratch(nullptr, pair_capacity, n_bins(n_lists(v))).bytes;
}

int gsplat_project(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, void* project_state, void* scratch,
                   int64_t scratch_bytes, gsplat_counts* counts_host, void* counts_event, int32_t flags, void* stream_) {
    bool fused = false;
    int rc = check_gaussians(g, &fused);
    if (rc) return rc;
    if ((rc = check_view(v))) return rc;
    if (!c2w || !project_state) return fail(GSPLAT_ERR_BAD_ARG, "c2w / project_state is NULL");
    if (!scratch || scratch_bytes < (int64_t)sizeof(CounterBlock)) return fail(GSPLAT_ERR_WORKSPACE, "project scratch (counter block) too small");
    if (reinterpret_cast<uintptr_t>(scratch) & 63u) return fail(GSPLAT_ERR_BAD_ARG, "project scratch must be 64-byte aligned");
    hipStream_t st = (hipStream_t)stream_;
    const int64_t n = g->n;
    const int64_t nl = n_lists(v), nb = n_bins(nl);
    ProjectState ps = carve_project(project_state, n > 0 ? n : 1, nl);
    const ViewK vk = make_viewk(*v);
    const bool mapped = (flags & GSPLAT_PROJECT_COUNTS_MAPPED) != 0;
    const bool colour_inside = !fused || (flags & GSPLAT_PROJECT_COLOUR_FUSED) != 0;
    const bool jac = fused && (flags & GSPLAT_PROJECT_SAVE_SH_JACOBIAN) != 0;
    const bool late = (flags & GSPLAT_PROJECT_COUNTS_LATE) != 0 && n > 0;      // counters totalled by bin_count_kernel
    if (n > 0) {
#ifdef GSPLAT_DIAGNOSTICS
        Records out{ps.rec, ps.rect, ps.depth, ps.tiles, ps.mask, g_ref_rect, g_ref_tiles};
#else
        Records out{ps.rec, ps.rect, ps.depth, ps.tiles, ps.mask, nullptr, nullptr};
#endif
        DevCounts* cm = mapped ? (DevCounts*)counts_host : nullptr;
        CounterBlock* cb = (CounterBlock*)scratch;
        const dim3 grid(blocks64(n)), block(64);
#define LAUNCH_PROJECT(F, C, J, KJ)                                                                                                     \
    do {                                                                                                                                \
        if (late) hipLaunchKernelGGL((project_kernel<F, C, J, false>), grid, block, 0, st, *g, c2w, ps.cam, vk, out, cb, ps.counts, cm,  \
                                     ps.bin_total, (int)nb, KJ, ps.big_flag);                                                           \
        else hipLaunchKernelGGL((project_kernel<F, C, J, true>), grid, block, 0, st, *g, c2w, ps.cam, vk, out, cb, ps.counts, cm,        \
                                ps.bin_total, (int)nb, KJ, ps.big_flag);                                                                \
    } while (0)
        if (!fused) LAUNCH_PROJECT(false, true, false, nullptr);
        else if (colour_inside && jac) LAUNCH_PROJECT(true, true, true, ps.kj);
        else if (colour_inside) LAUNCH_PROJECT(true, true, false, nullptr);
        else LAUNCH_PROJECT(true, false, false, nullptr);
#undef LAUNCH_PROJECT
        LAUNCH_CHECK("project_kernel");
        if (counts_host && !mapped && !late) HIP_TRY(hipMemcpyAsync(counts_host, ps.counts, sizeof(gsplat_counts), hipMemcpyDeviceToHost, st));
    } else {                        // no kernel runs: the counters are zero by definition
        HIP_TRY(hipMemsetAsync(ps.counts, 0, sizeof(DevCounts), st));
        HIP_TRY(hipMemsetAsync(ps.bin_total, 0, 3 * nb * sizeof(uint32_t), st));
        if (counts_host) HIP_TRY(hipMemsetAsync(counts_host, 0, sizeof(gsplat_counts), st));
    }
    if (counts_event && !late) HIP_TRY(hipEventRecord((hipEvent_t)counts_event, st));
    if (n > 0) {                    // these need no pair buffer: queued behind the event, they run while a waiting host sizes the buffers
        hipLaunchKernelGGL(bin_count_kernel, dim3((unsigned)(n_bin_blocks(n) + big_count_blocks(n))), dim3(256), bin_lds_bytes(nb, 256, 4), st, n, ps.rect, ps.tiles, ps.mask, vk.lists_x, (int)nb,
                           ps.bin_total, ps.block_off, ps.list_count, ps.ranges, (int)nl, late ? (CounterBlock*)scratch : nullptr, ps.counts,
                           late && mapped ? (DevCounts*)counts_host : nullptr, ps.rec, ps.big_flag, (uint32_t)n_bin_blocks(n), bin_batches(n));
        LAUNCH_CHECK("bin_count_kernel");
        if (late) {                 // the counters exist only now
            if (counts_host && !mapped) HIP_TRY(hipMemcpyAsync(counts_host, ps.counts, sizeof(gsplat_counts), hipMemcpyDeviceToHost, st));
            if (counts_event) HIP_TRY(hipEventRecord((hipEvent_t)counts_event, st));
        }
        if (!colour_inside) {
            if (jac) hipLaunchKernelGGL(colour_kernel<true>, dim3(blocks64(n)), dim3(64), 0, st, *g, ps.cam, ps.tiles, ps.rec, ps.kj);
            else hipLaunchKernelGGL(colour_kernel<false>, dim3(blocks64(n)), dim3(64), 0, st, *g, ps.cam, ps.tiles, ps.rec, nullptr);
            LAUNCH_CHECK("colour_kernel");
        }
    }
    return GSPLAT_OK;
}

int gsplat_bin(int64_t n, int64_t pair_capacity, const gsplat_view* v, const void* project_state, void* bin_state, void* scratch,
               int64_t scratch_bytes, void* stream_) {
    int rc = check_view(v);
    if (rc) return rc;
    const int64_t n_binned = pair_capacity;       // what the buffers hold; the actual count is read on the device (counts->n_binned)
    if (n < 0 || n_binned < 0 || n_binned > 0xFFFFFFFFLL) return fail(GSPLAT_ERR_BAD_ARG, "n / pair_capacity out of range");
    if (!project_state || !bin_state) return fail(GSPLAT_ERR_BAD_ARG, "state is NULL");
    hipStream_t st = (hipStream_t)stream_;
    const ViewK vk = make_viewk(*v);
    const int64_t nl = n_lists(v), nb = n_bins(nl);
    ProjectState ps = carve_project((void*)project_state, n > 0 ? n : 1, nl);
    if (n_binned == 0 || n == 0) {
        HIP_TRY(hipMemsetAsync(ps.ranges, 0, nl * sizeof(uint2), st));
        hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, st, (int)nl, ps.ranges, ps.order, ps.class_bounds);
        LAUNCH_CHECK("plan_kernel");
        return GSPLAT_OK;
    }
    BinScratch sc = carve_bin_scratch(scratch, n_binned, nb);
    if (!scratch || sc.bytes > scratch_bytes) return fail(GSPLAT_ERR_WORKSPACE, "bin scratch too small");
    uint32_t* sorted_ids = (uint32_t*)bin_state;
    hipLaunchKernelGGL(bin_scatter_kernel, dim3((unsigned)(n_bin_blocks(n) + big_bin_blocks(n))), dim3(256), bin_lds_bytes(nb, 9 * 1024), st, n, ps.rect, ps.tiles, ps.mask, ps.depth, vk.lists_x,
                       (int)nb, ps.bin_total, ps.block_off, ps.bin_start, (uint32_t)n_binned, sc.bvals, ps.rec, ps.big_flag,
                       (uint32_t)n_bin_blocks(n), bin_batches(n));
    LAUNCH_CHECK("bin_scatter_kernel");
    hipLaunchKernelGGL(split_count_kernel, dim3((unsigned)n_chunks(n_binned)), dim3(256), 0, st, (int)nb, ps.bin_start, sc.bvals,
                       (uint32_t)n_binned, ps.counts, ps.list_count, sc.seg_off);
    LAUNCH_CHECK("split_count_kernel");
    hipLaunchKernelGGL(split_scatter_kernel, dim3((unsigned)n_chunks(n_binned) + 1u), dim3(SS_THREADS), 0, st, (int)nl, (int)nb, ps.bin_start, sc.bvals,
                       (uint32_t)n_binned, ps.counts, ps.list_count, sc.seg_off, ps.ranges, sc.vals, ps.order, ps.class_bounds);
    LAUNCH_CHECK("split_scatter_kernel");
    // F9 + F12: per-list sort by (depth, index); one launch per size class, grids bounded by what the class can hold
    uint64_t* vals = sc.vals;
    const auto cap = [&](int64_t min_len) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(nl, n_binned / min_len)); };
    // lists of 4096+ entries: 100 KB of LDS per workgroup (8192+ fall back to global memory inside).  An empty launch of that kernel
    // still costs ~4 us, so it is made only where the AVERAGE list has 256 entries (lists 16x the average are the far tail: the longest
    // list of config 3 is 12x its average of 200); elsewhere the launch below takes such a list too.
    const bool class0 = n_binned >= 4096 && n_binned >= 256 * nl;
    if (class0) {
        hipLaunchKernelGGL((list_sort_kernel<1024, 8, 13, 0>), dim3(std::min(cap(4096), 256u)), dim3(1024), 0, st, ps.order, ps.class_bounds, ps.ranges,
                           vals, sorted_ids, 0);
        LAUNCH_CHECK("list_sort_kernel<8192>");
    }
    if (n_binned >= 1024) {       // lists of 1024 .. 4095 entries: 50 KB per workgroup
        hipLaunchKernelGGL((list_sort_kernel<512, 8, 12, 1>), dim3(std::min(cap(1024), 768u)), dim3(512), 0, st, ps.order, ps.class_bounds, ps.ranges,
                           vals, sorted_ids, class0 ? 0 : 1);
        LAUNCH_CHECK("list_sort_kernel<4096>");
    }
    {
        const unsigned mid_blocks = n_binned >= 256 ? std::min(cap(256), 4096u) : 0u;
        const unsigned small_blocks = std::min((cap(1) + 3u) / 4u, 16384u);
        hipLaunchKernelGGL(list_sort_small_kernel, dim3(mid_blocks + small_blocks), dim3(256), 0, st, ps.order, ps.class_bounds, mid_blocks,
                           ps.ranges, vals, sorted_ids);
        LAUNCH_CHECK("list_sort_small_kernel");
    }
    return GSPLAT_OK;
}

int gsplat_rasterize_forward(int64_t n, int64_t n_binned, const gsplat_view* v, const void* project_state, void* bin_state,
                             float* image, float* accum, float* grad2d, void* stream_) {
    int rc = check_view(v);
    if (rc) return rc;
    if (!project_state || !bin_state || !image) return fail(GSPLAT_ERR_BAD_ARG, "state / image is NULL");
    hipStream_t st = (hipStream_t)stream_;
    const ViewK vk = make_viewk(*v);
    const int64_t nl = n_lists(v);
    ProjectState ps = carve_project((void*)project_state, n > 0 ? n : 1, nl);
    if (accum)      // a backward pass will follow: exact sub-tile masks, saved per pair
        hipLaunchKernelGGL(raster_forward_kernel<true>, dim3((unsigned)nl), dim3(64), 0, st, ps.ranges, (const uint32_t*)bin_state, ps.rec,
                           ps.order, vk.lists_x, vk.H, vk.W, vk.chi_clip, vk.alpha_max, vk.alpha_cutoff, image, accum, STATS_FWD,
                           (uint32_t)(n > 0 ? n - 1 : 0), grad2d, grad2d ? n : 0, pair_mask_of(bin_state, n_binned));
    else
        hipLaunchKernelGGL(raster_forward_kernel<false>, dim3((unsigned)nl), dim3(64), 0, st, ps.ranges, (const uint32_t*)bin_state, ps.rec,
                           ps.order, vk.lists_x, vk.H, vk.W, vk.chi_clip, vk.alpha_max, vk.alpha_cutoff, image, accum, STATS_FWD,
                           (uint32_t)(n > 0 ? n - 1 : 0), grad2d, grad2d ? n : 0, nullptr);
    LAUNCH_CHECK("raster_forward_kernel");
    return GSPLAT_OK;
}

// (helpers inside the extern "C" block must be `static`: an anonymous namespace does not stop a function with C linkage from
//  being exported, and the product library exports nothing but the gsplat_* entry points -- tests/test_abi_cpu.py)
struct DetScratch { float* part; uint32_t* pair_base; uint32_t* block_sum; int64_t bytes; };
static DetScratch carve_det(void* base, int64_t n, int64_t capacity) {
    DetScratch d;
    char* p = (char*)base;
    int64_t o = 0;
    d.part = (float*)(p + o); o += up((capacity > 0 ? capacity : 1) * 36);
    d.pair_base = (uint32_t*)(p + o); o += up((n > 0 ? n : 1) * 4);
    d.block_sum = (uint32_t*)(p + o); o += up(((n > 0 ? n : 1) + PB_BLOCK - 1) / PB_BLOCK * 4);
    d.bytes = o;
    return d;
}

int64_t gsplat_rasterize_backward_scratch_bytes(int64_t n, int64_t pair_capacity) { return carve_det(nullptr, n, pair_capacity).bytes; }

int gsplat_rasterize_backward(int64_t n, int64_t n_binned, const gsplat_view* v, const void* project_state, const void* bin_state,
                              const float* accum, const float* grad_image, float* grad2d, int32_t grad2d_zeroed,
                              void* det_scratch, int64_t det_scratch_bytes, void* stream_) {
    int rc = check_view(v);
    if (rc) return rc;
    if (!project_state || !bin_state || !accum || !grad_image || !grad2d) return fail(GSPLAT_ERR_BAD_ARG, "NULL argument");
    hipStream_t st = (hipStream_t)stream_;
    const ViewK vk = make_viewk(*v);
    const int64_t nl = n_lists(v);
    ProjectState ps = carve_project((void*)project_state, n > 0 ? n : 1, nl);
    const bool det = det_scratch != nullptr;
    if (!grad2d_zeroed && !det) HIP_TRY(hipMemsetAsync(grad2d, 0, (size_t)(n > 0 ? n : 0) * 16 * sizeof(float), st));
    if (n == 0) return GSPLAT_OK;
    if (n_binned == 0) {
        if (det) HIP_TRY(hipMemsetAsync(grad2d, 0, (size_t)n * 16 * sizeof(float), st));
        return GSPLAT_OK;
    }
    if (!det) {
        hipLaunchKernelGGL(raster_backward_kernel<false>, dim3((unsigned)nl), dim3(64), 0, st, ps.ranges, (const uint32_t*)bin_state, ps.rec,
                           ps.order, vk.lists_x, vk.H, vk.W, vk.chi_clip, vk.alpha_max, vk.alpha_cutoff, accum, grad_image,
                           grad2d, STATS_BWD, (uint32_t)(n - 1), DetArgs{}, pair_mask_of(bin_state, n_binned));
        LAUNCH_CHECK("raster_backward_kernel");
        return GSPLAT_OK;
    }
    // deterministic: rows stored per (list, Gaussian) pair, then added per Gaussian in a fixed order
    DetScratch ds = carve_det(det_scratch, n, n_binned);
    if (ds.bytes > det_scratch_bytes) return fail(GSPLAT_ERR_WORKSPACE, "deterministic-backward scratch too small");
    const unsigned pb_blocks = (unsigned)((n + PB_BLOCK - 1) / PB_BLOCK);
    HIP_TRY(hipMemsetAsync(ds.part, 0, (size_t)n_binned * 36, st));          // rows of entries a saturated list never reaches
    hipLaunchKernelGGL(tile_block_sum_kernel, dim3(pb_blocks), dim3(256), 0, st, n, ps.tiles, ds.block_sum);
    LAUNCH_CHECK("tile_block_sum_kernel");
    hipLaunchKernelGGL(pair_base_kernel, dim3(pb_blocks), dim3(256), 0, st, n, ps.tiles, ds.block_sum, ds.pair_base);
    LAUNCH_CHECK("pair_base_kernel");
    hipLaunchKernelGGL(raster_backward_kernel<true>, dim3((unsigned)nl), dim3(64), 0, st, ps.ranges, (const uint32_t*)bin_state, ps.rec,
                       ps.order, vk.lists_x, vk.H, vk.W, vk.chi_clip, vk.alpha_max, vk.alpha_cutoff, accum, grad_image,
                       grad2d, STATS_BWD, (uint32_t)(n - 1), DetArgs{ps.rect, ps.mask, ps.tiles, ds.pair_base, ds.part, (uint32_t)n_binned},
                       pair_mask_of(bin_state, n_binned));
    LAUNCH_CHECK("raster_backward_kernel<deterministic>");
    hipLaunchKernelGGL(pair_reduce_kernel, dim3(blocks256(n)), dim3(256), 0, st, n, ps.tiles, ds.pair_base, ds.part, (uint32_t)n_binned, grad2d);
    LAUNCH_CHECK("pair_reduce_kernel");
    return GSPLAT_OK;
}

static int project_backward_impl(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, const void* project_state,
                                 const float* grad2d, const gsplat_gaussian_grads* out, int32_t flags, void* stream_, const AdamRest* ar) {
    bool fused = false;
    int rc = check_gaussians(g, &fused);
    if (rc) return rc;
    if ((rc = check_view(v))) return rc;
    if (!c2w || !project_state || !grad2d || !out) return fail(GSPLAT_ERR_BAD_ARG, "NULL argument");
    if (g->n == 0) return GSPLAT_OK;
    if (!out->pos || !out->opacity_raw) return fail(GSPLAT_ERR_BAD_ARG, "grad pos / opacity_raw is NULL");
    // fused inputs, f_dc and f_rest NULL, color given: hand out the colour-logit gradients instead of the SH gradients
    const bool factored = fused && !out->f_dc && !out->f_rest;
    const bool acc = (flags & GSPLAT_BACKWARD_ACCUMULATE) != 0;
    if (acc && (ar || !(fused && !factored && (flags & GSPLAT_BACKWARD_SH_JACOBIAN))))
        return fail(GSPLAT_ERR_BAD_ARG, "GSPLAT_BACKWARD_ACCUMULATE needs fused inputs, the saved SH Jacobian and SH gradients (no factored exchange, no in-place step)");
    if (ar && !(fused && !factored && (flags & GSPLAT_BACKWARD_SH_JACOBIAN)))
        return fail(GSPLAT_ERR_BAD_ARG, "the in-place f_rest step needs fused inputs, the saved SH Jacobian and SH gradients (no factored exchange)");
    if (fused && !factored && !(out->scale_raw && out->q_raw && out->f_dc && (out->f_rest || ar))) return fail(GSPLAT_ERR_BAD_ARG, "fused grads incomplete");
    if (factored && !(out->scale_raw && out->q_raw)) return fail(GSPLAT_ERR_BAD_ARG, "fused grads incomplete");
    if (!fused && !(out->color && out->sigma)) return fail(GSPLAT_ERR_BAD_ARG, "grad color / sigma is NULL");
    hipStream_t st = (hipStream_t)stream_;
    ProjectState ps = carve_project((void*)project_state, g->n, n_lists(v));
    const ViewK vk = make_viewk(*v);
    const AdamRest none = {nullptr, nullptr, nullptr, {0.f, 0.f, 0.f, 0.f, 0.f}, nullptr, 0};
    if (ar) {
        AdamRest a = *ar;
        a.counts = ps.counts;
        hipLaunchKernelGGL((project_backward_kernel<true, true, true>), dim3(blocks64(g->n)), dim3(64), 0, st, *g, ps.cam, vk, ps.tiles, grad2d, *out, false, ps.kj, a);
    }
    else if (acc)
        hipLaunchKernelGGL((project_backward_kernel<true, true, false, true>), dim3(blocks64(g->n)), dim3(64), 0, st, *g, ps.cam, vk, ps.tiles, grad2d, *out, false, ps.kj, none);
    else if (fused && (flags & GSPLAT_BACKWARD_SH_JACOBIAN))
        hipLaunchKernelGGL((project_backward_kernel<true, true>), dim3(blocks64(g->n)), dim3(64), 0, st, *g, ps.cam, vk, ps.tiles, grad2d, *out, factored, ps.kj, none);
    else if (fused)
        hipLaunchKernelGGL((project_backward_kernel<true, false>), dim3(blocks64(g->n)), dim3(64), 0, st, *g, ps.cam, vk, ps.tiles, grad2d, *out, factored, nullptr, none);
    else
        hipLaunchKernelGGL((project_backward_kernel<false, false>), dim3(blocks64(g->n)), dim3(64), 0, st, *g, ps.cam, vk, ps.tiles, grad2d, *out, false, nullptr, none);
    LAUNCH_CHECK("project_backward_kernel");
    return GSPLAT_OK;
}

int gsplat_project_backward(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, const void* project_state,
                            const float* grad2d, const gsplat_gaussian_grads* out, int32_t flags, void* stream_) {
    return project_backward_impl(g, c2w, v, project_state, grad2d, out, flags, stream_, nullptr);
}

// ---- one call per direction (include/gsplat_mi355x.h: "composite entries") ----------------------------------------------
// The frame arena: project_state | bin_state | accum | grad2d, each part 256-byte aligned.  Host-side arithmetic only.
struct FrameParts { int64_t project_state, bin_state, accum, grad2d, total; };
static FrameParts frame_parts(int64_t n, int64_t pair_capacity, const gsplat_view* v, int32_t flags) {
    FrameParts f;
    int64_t o = 0;
    f.project_state = o; o += up(carve_project(nullptr, n > 0 ? n : 1, n_lists(v)).bytes);
    f.bin_state = o; o += up(gsplat_bin_state_bytes(pair_capacity, v));
    f.accum = f.grad2d = -1;
    if (flags & GSPLAT_FRAME_BACKWARD) {
        f.accum = o; o += up((int64_t)v->H * v->W * 3 * (int64_t)sizeof(float));
        f.grad2d = o; o += up((n > 0 ? n : 1) * 16 * (int64_t)sizeof(float));
    }
    f.total = o;
    return f;
}
// the forward pass clears grad2d on the side unless there are so few lists that a wave's share would be long
static bool forward_clears_grad2d(int64_t n, const gsplat_view* v) { return n <= 256 * n_lists(v); }

int64_t gsplat_frame_bytes(int64_t n, int64_t pair_capacity, const gsplat_view* v, int32_t flags) {
    if (!v || v->H <= 0 || v->W <= 0 || n < 0 || pair_capacity < 0) return -1;
    return frame_parts(n, pair_capacity, v, flags).total;
}

int gsplat_forward_deferred(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, void* frame, int64_t frame_bytes,
                            int64_t pair_capacity, void* counters, int64_t counters_bytes, void* bin_scratch, int64_t bin_scratch_bytes,
                            gsplat_counts* counts_host, void* counts_event, float* image, int32_t flags, void* stream_) {
    if (!g || !v) return fail(GSPLAT_ERR_BAD_ARG, "gaussians / view is NULL");
    int rc = check_view(v);
    if (rc) return rc;
    if (!frame || !image) return fail(GSPLAT_ERR_BAD_ARG, "frame / image is NULL");
    if (reinterpret_cast<uintptr_t>(frame) & 255u) return fail(GSPLAT_ERR_BAD_ARG, "frame must be 256-byte aligned");
    if (pair_capacity < 1) return fail(GSPLAT_ERR_BAD_ARG, "pair_capacity must be positive (a capacity kept from earlier frames)");
    const FrameParts f = frame_parts(g->n, pair_capacity, v, flags);
    if (f.total > frame_bytes) return fail(GSPLAT_ERR_WORKSPACE, "frame arena too small (gsplat_frame_bytes)");
    char* base = (char*)frame;
    const bool bwd = (flags & GSPLAT_FRAME_BACKWARD) != 0;
    const bool fused = g->scale_raw != nullptr;
    int32_t pf = GSPLAT_PROJECT_COLOUR_FUSED | GSPLAT_PROJECT_COUNTS_LATE | (counts_host ? GSPLAT_PROJECT_COUNTS_MAPPED : 0);
    if (bwd && fused && !(flags & GSPLAT_FRAME_NO_SH_JACOBIAN)) pf |= GSPLAT_PROJECT_SAVE_SH_JACOBIAN;
    if ((rc = gsplat_project(g, c2w, v, base + f.project_state, counters, counters_bytes, counts_host, counts_event, pf, stream_))) return rc;
    if ((rc = gsplat_bin(g->n, pair_capacity, v, base + f.project_state, base + f.bin_state, bin_scratch, bin_scratch_bytes, stream_))) return rc;
    float* accum = bwd ? (float*)(base + f.accum) : nullptr;
    float* grad2d = bwd && forward_clears_grad2d(g->n, v) ? (float*)(base + f.grad2d) : nullptr;
    return gsplat_rasterize_forward(g->n, pair_capacity, v, base + f.project_state, base + f.bin_state, image, accum, grad2d, stream_);
}

static int backward_impl(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, void* frame, int64_t frame_bytes,
                         int64_t pair_capacity, const float* grad_image, const gsplat_gaussian_grads* out, float* grad_logit,
                         void* det_scratch, int64_t det_scratch_bytes, int32_t flags, void* stream_, const AdamRest* ar) {
    if (!g || !v) return fail(GSPLAT_ERR_BAD_ARG, "gaussians / view is NULL");
    int rc = check_view(v);
    if (rc) return rc;
    if (!frame) return fail(GSPLAT_ERR_BAD_ARG, "frame is NULL");
    const FrameParts f = frame_parts(g->n, pair_capacity, v, GSPLAT_FRAME_BACKWARD);
    if (f.total > frame_bytes) return fail(GSPLAT_ERR_WORKSPACE, "frame arena too small: was it made with GSPLAT_FRAME_BACKWARD?");
    char* base = (char*)frame;
    float* grad2d = (float*)(base + f.grad2d);
    const bool both = !(flags & (GSPLAT_BACKWARD_PHASE_RASTER | GSPLAT_BACKWARD_PHASE_PROJECT));
    if (both || (flags & GSPLAT_BACKWARD_PHASE_RASTER)) {
        if (!grad_image) return fail(GSPLAT_ERR_BAD_ARG, "grad_image is NULL");
        const int32_t zeroed = forward_clears_grad2d(g->n, v) && !(flags & GSPLAT_BACKWARD_GRAD2D_DIRTY);
        if ((rc = gsplat_rasterize_backward(g->n, pair_capacity, v, base + f.project_state, base + f.bin_state, (const float*)(base + f.accum),
                                            grad_image, grad2d, zeroed, det_scratch, det_scratch_bytes, stream_))) return rc;
        if (grad_logit && (rc = gsplat_logit_grad(g->n, v, base + f.project_state, grad2d, grad_logit, stream_))) return rc;
    }
    if (both || (flags & GSPLAT_BACKWARD_PHASE_PROJECT)) {
        if (!out) return fail(GSPLAT_ERR_BAD_ARG, "grads is NULL");
        if ((rc = project_backward_impl(g, c2w, v, base + f.project_state, grad2d, out,
                                        flags & (GSPLAT_BACKWARD_SH_JACOBIAN | GSPLAT_BACKWARD_ACCUMULATE), stream_, ar))) return rc;
    }
    return GSPLAT_OK;
}

int gsplat_backward(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, void* frame, int64_t frame_bytes,
                    int64_t pair_capacity, const float* grad_image, const gsplat_gaussian_grads* out, float* grad_logit,
                    void* det_scratch, int64_t det_scratch_bytes, int32_t flags, void* stream_) {
    return backward_impl(g, c2w, v, frame, frame_bytes, pair_capacity, grad_image, out, grad_logit, det_scratch, det_scratch_bytes, flags,
                         stream_, nullptr);
}

int gsplat_backward_adam_rest(const gsplat_gaussians* g, const float* c2w, const gsplat_view* v, void* frame, int64_t frame_bytes,
                              int64_t pair_capacity, const float* grad_image, const gsplat_gaussian_grads* out, void* det_scratch,
                              int64_t det_scratch_bytes, int32_t flags, const gsplat_adam_group* f_rest, float beta1, float beta2, float eps,
                              void* stream_) {
    if (!g || !f_rest) return fail(GSPLAT_ERR_BAD_ARG, "gaussians / f_rest update is NULL");
    if (flags & (GSPLAT_BACKWARD_PHASE_RASTER | GSPLAT_BACKWARD_PHASE_PROJECT)) return fail(GSPLAT_ERR_BAD_ARG, "the in-place step runs the whole backward pass");
    if (f_rest->n != g->n * 45 || f_rest->step < 1 || !f_rest->param || !f_rest->exp_avg || !f_rest->exp_avg_sq || f_rest->grad_scale ||
        f_rest->param != g->f_rest)
        return fail(GSPLAT_ERR_BAD_ARG, "f_rest update: param must be the f_rest the frame was rendered from (45 n values), moments given, no grad_scale");
    if ((reinterpret_cast<uintptr_t>(f_rest->param) | reinterpret_cast<uintptr_t>(f_rest->exp_avg) | reinterpret_cast<uintptr_t>(f_rest->exp_avg_sq)) & 15u)
        return fail(GSPLAT_ERR_BAD_ARG, "f_rest update: 16-byte aligned arrays");
    const double bc1 = 1.0 - pow((double)beta1, (double)f_rest->step), bc2 = 1.0 - pow((double)beta2, (double)f_rest->step);
    const AdamRest ar = {f_rest->param, f_rest->exp_avg, f_rest->exp_avg_sq,
                         {(float)((double)f_rest->lr / bc1), (float)(1.0 / sqrt(bc2)), beta1, beta2, eps}, nullptr, (long long)pair_capacity};
    return backward_impl(g, c2w, v, frame, frame_bytes, pair_capacity, grad_image, out, nullptr, det_scratch, det_scratch_bytes, flags, stream_, &ar);
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
    if (!aligned16(f_dc) || !aligned16(f_rest) || !aligned16(points) || !aligned16(grad_color) || !aligned16(grad_f_dc) ||
        !aligned16(grad_f_rest) || !aligned16(grad_points))
        return fail(GSPLAT_ERR_BAD_ARG, "arrays must be 16-byte aligned");
    hipLaunchKernelGGL(evaluate_sh_backward_kernel, dim3(blocks64(n)), dim3(64), 0, (hipStream_t)stream_, n, f_dc, f_rest, points, c2w,
                       grad_color, grad_f_dc, grad_f_rest, grad_points);
    LAUNCH_CHECK("evaluate_sh_backward_kernel");
    return GSPLAT_OK;
}

int gsplat_logit_grad(int64_t n, const gsplat_view* v, const void* project_state, const float* grad2d, float* grad_logit, void* stream_) {
    int rc = check_view(v);
    if (rc) return rc;
    if (n < 0) return fail(GSPLAT_ERR_BAD_ARG, "n < 0");
    if (n == 0) return GSPLAT_OK;
    if (!project_state || !grad2d || !grad_logit) return fail(GSPLAT_ERR_BAD_ARG, "NULL argument");
    ProjectState ps = carve_project((void*)project_state, n, n_lists(v));
    hipLaunchKernelGGL(logit_grad_kernel, dim3(blocks256(n)), dim3(256), 0, (hipStream_t)stream_, n, ps.tiles, ps.rec, grad2d, grad_logit);
    LAUNCH_CHECK("logit_grad_kernel");
    return GSPLAT_OK;
}

int gsplat_sh_accumulate(int64_t n, int32_t n_views, const float* pos, const float* eyes, const float* grad_logit, float scale,
                         float* grad_f_dc, float* grad_f_rest, void* stream_) {
    if (n < 0 || n_views < 0) return fail(GSPLAT_ERR_BAD_ARG, "n / n_views < 0");
    if (n == 0) return GSPLAT_OK;
    if (!pos || !grad_f_dc || !grad_f_rest || (n_views > 0 && (!eyes || !grad_logit))) return fail(GSPLAT_ERR_BAD_ARG, "NULL argument");
    if (!aligned16(pos) || !aligned16(grad_f_dc) || !aligned16(grad_f_rest)) return fail(GSPLAT_ERR_BAD_ARG, "arrays must be 16-byte aligned");
    hipLaunchKernelGGL(sh_accumulate_kernel, dim3(blocks64(n)), dim3(64), 0, (hipStream_t)stream_, n, (int)n_views, pos, eyes, grad_logit,
                       scale, grad_f_dc, grad_f_rest);
    LAUNCH_CHECK("sh_accumulate_kernel");
    return GSPLAT_OK;
}

}  // extern "C"
