// slam_hip.hip — C ABI (include/slam_hip.h) over the gfx950 kernels in kernels.h.
// Compile with hipcc --offload-arch=gfx950 -ffp-contract=off (see slam_amd/build.py).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <emmintrin.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <rocprim/rocprim.hpp>
#include <string>
#include <vector>

#include "../../include/slam_hip.h"
#include "clearance.h"
#include "cone_gpu.h"
#include "kernels.h"
#include "sortk.h"

#define SLAM_API extern "C" __attribute__((visibility("default")))

namespace
{
thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                                       \
    do                                                                                                     \
    {                                                                                                      \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(SLAM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                                          ":" + std::to_string(__LINE__) + ")");                           \
    } while (0)

enum
{
    G_PREDICT = 0,
    G_WEIGHTS = 1,
    G_RESAMPLE = 2,
    G_SORT = 3,
    G_CLONE = 4,
    G_COUNT = 5
};

struct TimedSpan
{
    int group;
    hipEvent_t a, b;
};

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
}  // namespace

struct slam_mcl
{
    slam_mcl_params prm;
    hipStream_t stream = nullptr;
    int64_t n = 0;
    int rows = 0, cols = 0;
    size_t map_stride = 0;
    // resident state (double buffered for the resample gather)
    // packed [x | y | theta | weight], n doubles each; x/y/th/w point into `state`
    double *state = nullptr, *state2 = nullptr;
    double *x = nullptr, *y = nullptr, *th = nullptr, *w = nullptr;
    int64_t n_glob = 0;  // particles of the whole filter (all shards)

    void bind()
    {
        x = state;
        y = state + n;
        th = state + 2 * n;
        w = state + 3 * n;
    }
    uint8_t *grid = nullptr, *clr = nullptr;     // localization: raw map + clearance planes
    size_t clr_stride = 0;                       // bytes between the 8 cone planes (0: one Chebyshev plane)
    uint8_t *maps = nullptr, *maps2 = nullptr;   // SLAM: per-particle canvases
    uint8_t* luts = nullptr;
    bool have_map = false;
    double *wlin = nullptr, *wmax = nullptr;  // SLAM_WEIGHT_LOG: exp(lw - max) of all shards, max-reduction levels
    // SLAM_MAPS_TILED
    bool tiled = false;
    bool collect_stats = true;  // slam_mcl_collect_stats
    int tile_T = 0, tile_tc = 0;       // tiles per canvas, per canvas row
    int64_t tile_cap = 0;              // tiles in the pool
    uint8_t* tile_pool = nullptr;
    int32_t *tile_tab = nullptr, *tile_tab2 = nullptr, *tile_free = nullptr;
    int32_t *tile_wl_p = nullptr, *tile_wl_t = nullptr;  // work list of k_tile_plan / k_tile_copy
    int32_t* tile_rng = nullptr;                          // per particle: its range of the work list
    int *tile_ref = nullptr, *tile_who = nullptr, *tile_shared = nullptr;  // per tile: live mark, winning entry, shared mark
    int* tile_ctrl = nullptr;                                              // cursor, free count, error flag, cursor before the plan
    uint8_t* tile_dense = nullptr;     // one dense canvas (import / export staging)
    float* tile_sector = nullptr;      // 256 sectors of the scan frame: how far the current scan can write
    int tile_gen = 0;  // number of the last collection: a tile is referenced iff its mark equals it
    int *pack_stamp = nullptr, *pack_ctr = nullptr;  // sparse canvas exchange: per tile the number of the pack that listed it
    int32_t* pack_slot = nullptr;                    // ... and its position in that pack's payload
    int* dev_bad = nullptr;  // set by a kernel that met an out-of-range entry in a caller's device index array
    bool bad_armed = false;  // an entry point that takes such an array has run since the last check
    int pack_gen = 0;
    int64_t pack_tiles = -1, pack_count = -1;        // what the last slam_mcl_pack_tiles_count found, for whom
    const int32_t* pack_idx = nullptr;
    bool maps_pending = false;  // SLAM-mode shard: poses resampled, canvases not yet (shard_select done)
    // scan
    int Bcap = 0;
    double* scan_dev = nullptr;  // [angle | dist^2 | sqrt(dist^2)], Bcap doubles each (packed at B)
    double *angles = nullptr, *z2 = nullptr, *z = nullptr;
    std::vector<double> scan_cache;
    // pinned staging ring of the scan upload: a copy from pageable memory would make the host wait for the stream to
    // drain on every update.  The slots form groups of SCAN_GROUP; one event per group, recorded behind the consumer
    // of the group's last slot and waited for when the ring comes round to the group again (an event record costs
    // ~10 us of idle GPU: one per update was 1 % of a headline step).  A filled slot stays `scan_pending` until
    // do_weights either hands it to k_order_keys (which copies it to scan_dev as a side job: no copy kernel at all) or
    // copies it itself.
    enum { SCAN_RING = 8, SCAN_GROUP = 4 };
    double* scan_pin[SCAN_RING] = {};
    hipEvent_t scan_ev[SCAN_RING / SCAN_GROUP] = {};
    int scan_slot = SCAN_RING - 1;
    const double* scan_pending = nullptr;  // pinned slot whose content scan_dev does not hold yet
    size_t scan_pending_n = 0;
    int last_B = 0;
    // noise
    double *inj_normals = nullptr, *inj_uniforms = nullptr;
    bool have_normals = false, have_uniforms = false;
    uint64_t predict_step = 0, resample_step = 0;
    // resampling
    std::vector<int64_t> lvl_n;
    std::vector<double*> lvl_s, lvl_P, lvl_tot;  // lvl_tot[l] = input array of level l+1
    bool prefix_partial = false;                 // the last selection was k_select<true>: lvl_P[0] is not written yet (complete_prefix)
    int32_t *idx = nullptr, *perm = nullptr, *slots = nullptr, *src = nullptr;
    double *keys = nullptr, *keys2 = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    // diagnostics
    bool record_hits = false;
    double* hit_xy = nullptr;
    int32_t* hit_cell = nullptr;
    size_t hit_cap = 0;
    unsigned long long* stats = nullptr;
    double* partials = nullptr;
    bool sparse_last = false;  // the kernel shape the last pass chose from the cloud's box (hysteresis)
    // deferred canvases (TiledDefer): the weighting pass of slam_mcl_update left the canvases untouched; the resample brings
    // the selected particles' canvases up to date before it hands their tables on
    bool deferred = false;
    int rp_B = 0;
    double rp_stddev = 0, rp_max_dist = 0, rp_off[3] = {0, 0, 0};
    int* surv_stamp = nullptr;      // per particle: number of the last resample that selected it
    int32_t* surv_list = nullptr;   // the distinct selected particles of the current resample
    int* surv_count = nullptr;      // device: how many
    int* surv_count_host = nullptr; // pinned: the count of an earlier resample (read late, decides the mode only)
    int surv_gen = 0;
    uint8_t* edge_flag = nullptr;   // deferred pass: 1 = this particle took the ordinary pass at once (slam_defer_safe)
    int32_t* edge_list = nullptr;   // those particles
    int* edge_count = nullptr;
    bool edge_valid = false;        // edge_flag belongs to the pending deferred pass
    int64_t n_updates = 0, n_deferred = 0;  // slam_mcl_get_defer_stats
    double* lw = nullptr;  // scratch of the particle-lane weight kernels
    size_t lw_cap = 0;     // doubles
    uint32_t *okeys = nullptr, *okeys2 = nullptr;  // pose-coherent order of the particle-lane kernels
    int* obox = nullptr;                            // ... two bounding boxes of the cloud (k_order_keys)
    int* obox_host = nullptr;                       // ... and the latest one in pinned host memory (kernel shape, do_weights)
    int order_pass = 0;
    bool poses_reset = true;  // set_poses / seed_* since the last weighting pass
    int32_t *oident = nullptr, *order = nullptr;
    void* osort_tmp = nullptr;
    size_t osort_tmp_bytes = 0;
    void* ss_tmp = nullptr;  // scratch of the sample sort (sortk.h), shared by both sorts of a step
    int profile = 0;  // 0 off, 1 every group (an event pair per group), 2 the weighting group only, on the kernels' own dispatch timestamps where the path allows
    bool ext_timing = false;  // (set by do_weights while it times the pass through hipExtLaunchKernelGGL)
    std::vector<TimedSpan> spans;
    double acc_ms[G_COUNT] = {0, 0, 0, 0, 0};
    int64_t acc_launches[G_COUNT] = {0, 0, 0, 0, 0};

    struct Span
    {
        slam_mcl* h;
        TimedSpan s;
        bool on;
        Span(slam_mcl* h_, int group) : h(h_), on(h_->profile == 1 || (h_->profile == 2 && group == G_WEIGHTS && !h_->ext_timing))
        {
            if (!on) return;
            s.group = group;
            hipEventCreate(&s.a);
            hipEventCreate(&s.b);
            hipEventRecord(s.a, h->stream);
        }
        ~Span()
        {
            if (!on) return;
            hipEventRecord(s.b, h->stream);
            h->spans.push_back(s);
        }
    };
};

namespace
{
template <class T>
int dev_alloc(T** p, size_t count)
{
    *p = nullptr;
    if (count == 0) return SLAM_OK;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T));
    if (e != hipSuccess)
        return fail(SLAM_ERR_NOMEM, std::string("hipMalloc of ") + std::to_string(count * sizeof(T)) + " bytes: " +
                                        hipGetErrorString(e));
    return SLAM_OK;
}

int finish(slam_mcl* h)
{
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->bad_armed)
    {
        int bad = 0;
        HIPCHK(hipMemcpy(&bad, h->dev_bad, sizeof(int), hipMemcpyDeviceToHost));
        h->bad_armed = false;
        if (bad)
        {
            HIPCHK(hipMemset(h->dev_bad, 0, sizeof(int)));
            return fail(SLAM_ERR_ARG, "a device index array (local_idx_dev / pos_dev / tabs_dev) held an entry out of range: "
                                      "those entries were skipped, the canvases are incomplete");
        }
    }
    if (h->tiled)
    {
        int ctrl[3];
        HIPCHK(hipMemcpy(ctrl, h->tile_ctrl, sizeof(ctrl), hipMemcpyDeviceToHost));
        if (ctrl[2])
            return fail(SLAM_ERR_NOMEM, "tile pool exhausted (" + std::to_string(h->tile_cap) +
                                            " tiles): canvases are incomplete, recreate the handle with a larger tile_pool_tiles");
    }
    return SLAM_OK;
}

// Tuning knobs of the particle-lane weighting pass, read once from the environment (experiments and
// tools/; the defaults are what the measurements in DESIGN.md chose).
struct Tuning
{
    int order_mode;  // SLAM_PL_ORDER: 0 = no pose-coherent order, else kind | (tile_shift + 1) << 4 | heading_bits << 8
    int order_bits;  // SLAM_PL_ORDER_BITS: bits of the order key that are sorted
    int seg_beams;   // SLAM_PL_SEG: beams per work item (0 = chosen from the launch size)
    int waves;       // SLAM_PL_WAVES: wavefronts per SIMD the ray kernel is compiled for (4, 5, 6 or 8)
    int direct_min;  // SLAM_PL_DIRECT_MIN: from this many 64-particle blocks on, the ray kernel sums the beams itself (no slab)
    int cone_planes; // SLAM_CONE_PLANES: 1 = per-octant cone-clearance planes (default), 0 = one Chebyshev-clearance plane
    int scan_fuse;   // SLAM_SCAN_FUSE: 1 = k_order_keys copies the new scan from its pinned slot (default), 0 = a copy of its own
    int s_kernel;    // SLAM_S_KERNEL: 0 = default (two-phase SLAM-mode kernel), 1 = legacy wave-per-ray kernel,
                     // 2 = two-phase kernel with every batch on its literal fallback (tests)
    int sort_impl;   // SLAM_SORT_IMPL: bit 0 = weight sort, bit 1 = order sort by the sample sort of sortk.h (16k..256k particles); 0 = rocPRIM
    int order_coarse;// SLAM_ORDER_COARSE: 1 = the pose-coherent order is sorted between the sample sort's buckets only, 0 = fully
    int ss_lds_cap;  // SLAM_SS_LDS_CAP: largest group of buckets the sample sort ranks in LDS (tests force the path from memory)
    int fast_div;    // SLAM_FAST_DIV: 1 = the particle-lane kernel forms xarg / stddev from the reciprocal (same bits), 0 = divides
    int clamp_limit; // SLAM_RAY_CLAMP_LIMIT: 1 = the localization hit search stops beyond z + 4 sigma (default), 0 = always to max_dist
    int cone_build;  // SLAM_CONE_BUILD: 1 = the cone-clearance planes are built on the device (default), 0 = on the host
    int order_wide;  // SLAM_PL_ORDER_WIDE: order mode for a cloud spread over more than 2^22 cells (0 = never switch)
    int order_sparse;// SLAM_PL_ORDER_SPARSE: order mode of the wave-per-particle kernel on a sparse cloud (0 = index order)
    int beam_cells;  // SLAM_BEAM_CELLS: cells of the cloud's bounding box per particle above which the wave-per-particle kernel runs (0 = never)
    int tile_fuse;   // SLAM_TILE_FUSE: 1 = the two-phase kernel makes the planned tile copies itself (default), 0 = k_tile_copy before it
    int defer;       // SLAM_DEFER: grid-SLAM update on tiles, "weights now, canvases of the survivors later" (TiledDefer, slam_s2.h):
                     // 0 = never, 1 = when the last resample kept fewer than half of the particles (default), 2 = whenever the scan allows
    int select_fuse; // SLAM_SELECT_FUSE: 1 = 4097..262144 particles: k_select rebuilds the upper levels of the running sum itself and the weight
                     // sort's last kernel gathers the states (default: three launches fewer per step), 0 = k_scan_upper + k_downsweep, k_gather
    int defer_fast;  // SLAM_DEFER_FAST: 1 = the weights-only kernel k_weights_S2F for the first half (default), 0 = k_weights_S2<DEFER>
    int order_adapt; // SLAM_PL_ORDER_ADAPT: 1 = heading buckets of the pose-coherent order sized from the cloud's extent (default), 0 = 10 bits
};
const Tuning& tuning()
{
    auto env_int = [](const char* name, int dflt) {
        const char* v = getenv(name);
        return v ? atoi(v) : dflt;
    };
    static const Tuning t = {env_int("SLAM_PL_ORDER", 3), env_int("SLAM_PL_ORDER_BITS", 32), env_int("SLAM_PL_SEG", 0),
                             env_int("SLAM_PL_WAVES", 5), env_int("SLAM_PL_DIRECT_MIN", 8192), env_int("SLAM_CONE_PLANES", 1), env_int("SLAM_SCAN_FUSE", 1), env_int("SLAM_S_KERNEL", 0),
                             env_int("SLAM_SORT_IMPL", 3), env_int("SLAM_ORDER_COARSE", 1), env_int("SLAM_SS_LDS_CAP", sortk::GROUP_CAP), env_int("SLAM_FAST_DIV", 1), env_int("SLAM_RAY_CLAMP_LIMIT", 1), env_int("SLAM_CONE_BUILD", 1), env_int("SLAM_PL_ORDER_WIDE", 4 | (7 << 4) | (8 << 8)), env_int("SLAM_PL_ORDER_SPARSE", 4 | (7 << 4) | (6 << 8)), env_int("SLAM_BEAM_CELLS", 4), env_int("SLAM_TILE_FUSE", 1), env_int("SLAM_DEFER", 1), env_int("SLAM_SELECT_FUSE", 1), env_int("SLAM_DEFER_FAST", 1), env_int("SLAM_PL_ORDER_ADAPT", 1)};
    return t;
}

// The sample sort serves the sizes at which rocPRIM's sort is a chain of launch-bound merge passes.

// The two sorts of a step (which = 1: particles by weight, slam/mcl.cpp:232-239; 2: the pose-coherent order of the ray kernels),
// all sizes on the library's own kernels (sortk.h): one workgroup up to 4096 pairs, the sample sort up to 262 144, the radix
// sort beyond.  rocPRIM stays behind SLAM_SORT_IMPL (bit cleared) as the sort the tests compare these with; its scratch is
// allocated only then.
int rocprim_scratch(void** tmp, size_t* have, size_t need)
{
    if (*tmp && *have >= need) return SLAM_OK;
    if (*tmp) hipFree(*tmp);
    *tmp = nullptr;
    uint8_t* t = nullptr;
    int rc = dev_alloc(&t, need ? need : 16);
    if (rc) return rc;
    *tmp = t;
    *have = need;
    return SLAM_OK;
}

// gather: the sample sort's last kernel also gathers the particle states (slamk::GatherEpilogue); *gathered says whether it did
int sort_by_weight(slam_mcl* h, int64_t n, const slamk::GatherEpilogue* gather = nullptr, bool* gathered = nullptr)
{
    if (gathered) *gathered = false;
    if (tuning().sort_impl & 1)
    {
        if (n <= sortk::SMALL_MAX)
            sortk::small_sort_pairs(h->keys, h->slots, h->perm, n, h->stream);
        else if (n <= sortk::MAX_N && gather && gathered && tuning().select_fuse != 0)
        {
            sortk::sort_pairs(h->ss_tmp, h->keys, h->slots, h->perm, n, false, h->stream, tuning().ss_lds_cap, *gather);
            *gathered = true;
        }
        else if (n <= sortk::MAX_N)
            sortk::sort_pairs(h->ss_tmp, h->keys, h->slots, h->perm, n, false, h->stream, tuning().ss_lds_cap);
        else
            sortk::radix_sort_pairs<double, uint64_t, int32_t>(h->ss_tmp, h->keys, h->slots, h->perm, n, 64, h->stream);
        HIPCHK(hipGetLastError());
        return SLAM_OK;
    }
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs_desc(nullptr, bytes, h->keys, h->keys2, h->slots, h->perm, (size_t)n, 0, 64, h->stream));
    int rc = rocprim_scratch(&h->sort_tmp, &h->sort_tmp_bytes, bytes);
    if (rc) return rc;
    HIPCHK(rocprim::radix_sort_pairs_desc(h->sort_tmp, bytes, h->keys, h->keys2, h->slots, h->perm, (size_t)n, 0, 64, h->stream));
    return SLAM_OK;
}

int sort_order_keys(slam_mcl* h, int order_bits)
{
    const int64_t n = h->n;
    if (tuning().sort_impl & 2)
    {
        if (n <= sortk::SMALL_MAX)
            sortk::small_sort_pairs(h->okeys, h->oident, h->order, n, h->stream);
        else if (n <= sortk::MAX_N)
            sortk::sort_pairs(h->ss_tmp, h->okeys, h->oident, h->order, n, tuning().order_coarse != 0 && order_bits == 32, h->stream);
        else
            sortk::radix_sort_pairs<uint32_t, uint32_t, int32_t>(h->ss_tmp, h->okeys, h->oident, h->order, n, order_bits, h->stream);
        HIPCHK(hipGetLastError());
        return SLAM_OK;
    }
    size_t bytes = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, bytes, h->okeys, h->okeys2, h->oident, h->order, (size_t)n, 0, order_bits, h->stream));
    int rc = rocprim_scratch(&h->osort_tmp, &h->osort_tmp_bytes, bytes);
    if (rc) return rc;
    HIPCHK(rocprim::radix_sort_pairs(h->osort_tmp, bytes, h->okeys, h->okeys2, h->oident, h->order, (size_t)n, 0, order_bits, h->stream));
    return SLAM_OK;
}

// The cloud's bounding boxes (k_order_keys / k_order_box): two on the device, the latest one also in pinned host memory.
// ... and the buffers of the pose-coherent order (keys, identity, order, rocPRIM scratch).
int ensure_order_box(slam_mcl* h)
{
    if (h->obox && h->okeys) return SLAM_OK;
    int rc;
    if (!h->okeys)
    {
        if ((rc = dev_alloc(&h->okeys, h->n)) || (rc = dev_alloc(&h->okeys2, h->n)) || (rc = dev_alloc(&h->oident, h->n)) ||
            (rc = dev_alloc(&h->order, h->n)))
            return rc;
    }
    if (h->obox) return SLAM_OK;
    if ((rc = dev_alloc(&h->obox, 16))) return rc;
    // both boxes empty (max < min): the first pass keeps the default order and the default kernel shape; [8], [9]: heading extents
    const int empty[16] = {0x7fffffff, 0x7fffffff, -0x7fffffff - 1, -0x7fffffff - 1, 0x7fffffff, 0x7fffffff, -0x7fffffff - 1, -0x7fffffff - 1,
                           65536,      65536,      0,               0,               0,          0,          0,               0};
    HIPCHK(hipMemcpy(h->obox, empty, sizeof(empty), hipMemcpyHostToDevice));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->obox_host), 4 * sizeof(int), hipHostMallocDefault));
    for (int k = 0; k < 4; ++k) h->obox_host[k] = empty[k];
    return SLAM_OK;
}

// After the poses were set or seeded (set-up calls, which may wait): the cloud's box on the device and on the host, so that
// the very first weighting pass already gets the order and the kernel shape of the cloud it meets.
int refresh_cloud_box(slam_mcl* h)
{
    if (h->prm.mode != SLAM_MODE_LOCALIZATION || h->prm.raycaster != SLAM_RAYCAST_FAST || h->n < 4096 || tuning().order_mode <= 0)
        return SLAM_OK;
    int rc = ensure_order_box(h);
    if (rc) return rc;
    hipLaunchKernelGGL(slamk::k_order_box, dim3(1), dim3(256), 0, h->stream, h->n, h->x, h->y, h->th, h->obox, (h->order_pass + 1) & 1, h->obox_host);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    h->poses_reset = false;
    return SLAM_OK;
}

// SLAM_MAPS_TILED: new tables (src != nullptr: table p = old table src[p]), census of the tiles (who
// owns what, exactly), free list of the unreferenced ones, allocation cursor restarted.
int tile_collect(slam_mcl* h, const int32_t* src)
{
    const int64_t total = (int64_t)h->n * h->tile_T;
    const int live = ++h->tile_gen;  // marks are never cleared: a tile is referenced iff its mark is this number
    HIPCHK(hipMemsetAsync(h->tile_ctrl, 0, 2 * sizeof(int), h->stream));  // cursor and count; the error flag stays
    hipLaunchKernelGGL(slamk::k_tab_gather, dim3((unsigned)h->n), dim3(256), 0, h->stream, h->tile_tab, h->tile_tab2, src,
                       h->tile_T, h->tile_ref, h->tile_who, live);
    std::swap(h->tile_tab, h->tile_tab2);
    hipLaunchKernelGGL(slamk::k_tile_shared, dim3((unsigned)cdiv(total, 1024)), dim3(256), 0, h->stream, h->tile_tab, total,
                       h->tile_who, h->tile_shared, live);
    hipLaunchKernelGGL(slamk::k_tile_own, dim3((unsigned)cdiv(total, 1024)), dim3(256), 0, h->stream, h->tile_tab, total,
                       h->tile_who, h->tile_shared, live);
    hipLaunchKernelGGL(slamk::k_tile_free, dim3((unsigned)cdiv(h->tile_cap, 8192)), dim3(256), 0, h->stream, h->tile_ref,
                       live, (int)h->tile_cap, h->tile_free, h->tile_ctrl);
    HIPCHK(hipGetLastError());
    return SLAM_OK;
}

// The two update tables of slam/raycast.cpp:196-211, evaluated as written there (double
// arithmetic, truncating casts; this translation unit is compiled without contraction or
// fast-math, so the values equal a strict-IEEE build of the reference).
void build_luts(uint8_t* t)
{
    const double L0 = 0.5, Locc = 0.40, Lfree = 0.60;
    for (int c = 0; c < 256; ++c)
    {
        double p = static_cast<double>(static_cast<unsigned char>(c)) / 255;
        p *= Lfree / L0;
        p = p < 1.0 ? p : 1.0;
        unsigned char v = static_cast<unsigned char>(p * 255);
        t[c] = v < 1 ? 1 : v;
        double q = static_cast<double>(static_cast<unsigned char>(c)) / 255;
        q *= Locc / L0;
        unsigned char u = static_cast<unsigned char>(q * 255);
        t[256 + c] = u < 1 ? 1 : u;
    }
}

int check_handle(slam_mcl* h)
{
    if (!h) return fail(SLAM_ERR_ARG, "null handle");
    hipError_t e = hipSetDevice(h->prm.device);
    if (e != hipSuccess) return fail(SLAM_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return SLAM_OK;
}

int upload_scan(slam_mcl* h, const double* angles, const double* dists, int B)
{
    if (!angles || !dists || B <= 0) return fail(SLAM_ERR_ARG, "scan: null pointer or n_beams <= 0");
    for (int b = 0; b < B; ++b)
        if (!std::isfinite(angles[b]) || !std::isfinite(dists[b])) return fail(SLAM_ERR_ARG, "scan: non-finite entry");
    if (B > h->Bcap)
    {
        HIPCHK(hipStreamSynchronize(h->stream));
        hipFree(h->scan_dev);
        h->scan_dev = nullptr;
        h->Bcap = 0;
        int rc = dev_alloc(&h->scan_dev, 3 * (size_t)B);
        if (rc) return rc;
        for (int k = 0; k < slam_mcl::SCAN_RING; ++k)
        {
            if (h->scan_pin[k]) hipHostFree(h->scan_pin[k]);
            h->scan_pin[k] = nullptr;
            HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->scan_pin[k]), sizeof(double) * 3 * (size_t)B, hipHostMallocDefault));
        }
        for (int g = 0; g < slam_mcl::SCAN_RING / slam_mcl::SCAN_GROUP; ++g)
            if (!h->scan_ev[g]) HIPCHK(hipEventCreateWithFlags(&h->scan_ev[g], hipEventDisableTiming));
        h->scan_pending = nullptr;
        h->Bcap = B;
        h->scan_cache.clear();
    }
    // host copy: angle, dist^2 (slam/mcl.cpp:72 passes dist*dist), sqrt(dist^2) (slam/raycast.cpp:236,241)
    std::vector<double> pack(3 * (size_t)B);
    for (int b = 0; b < B; ++b)
    {
        pack[b] = angles[b];
        const double d2 = dists[b] * dists[b];
        pack[B + b] = d2;
        pack[2 * (size_t)B + b] = std::sqrt(d2);
    }
    h->angles = h->scan_dev;
    h->z2 = h->scan_dev + B;
    h->z = h->scan_dev + 2 * (size_t)B;
    if (h->last_B == B && h->scan_cache == pack) return SLAM_OK;  // same scan re-applied (benchmark/mcl.cpp:58-64)
    // into the next pinned slot; kernels already queued on the stream still see the old scan in scan_dev, and the host
    // does not wait for them
    // (a slot that was filled but never consumed, after a failed update, has no reader on the stream: it is filled again)
    const int k = h->scan_pending ? h->scan_slot : (h->scan_slot = (h->scan_slot + 1) % slam_mcl::SCAN_RING);
    if (k % slam_mcl::SCAN_GROUP == 0 && !h->scan_pending)
        HIPCHK(hipEventSynchronize(h->scan_ev[k / slam_mcl::SCAN_GROUP]));  // the group's slots (24+ uploads ago) have been consumed
    std::memcpy(h->scan_pin[k], pack.data(), sizeof(double) * 3 * (size_t)B);
    h->scan_pending = h->scan_pin[k];
    h->scan_pending_n = 3 * (size_t)B;
    h->scan_cache.swap(pack);
    h->last_B = B;
    return SLAM_OK;
}

// Called behind the launch (or copy) that reads the pending slot.
int scan_consumed(slam_mcl* h)
{
    h->scan_pending = nullptr;
    if (h->scan_slot % slam_mcl::SCAN_GROUP == slam_mcl::SCAN_GROUP - 1)
        HIPCHK(hipEventRecord(h->scan_ev[h->scan_slot / slam_mcl::SCAN_GROUP], h->stream));
    return SLAM_OK;
}

int flush_scan(slam_mcl* h)
{
    if (!h->scan_pending) return SLAM_OK;
    HIPCHK(hipMemcpyAsync(h->scan_dev, h->scan_pending, sizeof(double) * h->scan_pending_n, hipMemcpyHostToDevice, h->stream));
    return scan_consumed(h);
}

// pass_mode 0: the pass as the reference runs it (canvases updated).  1 (slam_mcl_update, which resamples right away): may
// leave the canvases untouched and set h->deferred (TiledDefer).  2: the second half of that, called by do_resample: the
// ordinary pass on the particles of h->surv_list only, nothing recorded.
int do_weights(slam_mcl* h, int B, double stddev, double max_dist, const double off[3], int pass_mode = 0)
{
    if (!(stddev > 0) || !std::isfinite(stddev) || !(max_dist >= 0) || !std::isfinite(max_dist))
        return fail(SLAM_ERR_ARG, "stddev must be > 0 and max_dist >= 0, both finite");
    if (!off || !std::isfinite(off[0]) || !std::isfinite(off[1]) || !std::isfinite(off[2]))
        return fail(SLAM_ERR_ARG, "scanner_offset: null or non-finite");
    const bool slam_mode = h->prm.mode == SLAM_MODE_SLAM;
    if (!slam_mode && !h->have_map) return fail(SLAM_ERR_STATE, "localization mode: call slam_mcl_set_map_u8 first");
    if (h->maps_pending) return fail(SLAM_ERR_STATE, "canvases of the last slam_mcl_shard_select are still pending (slam_mcl_shard_clone_maps)");
    const bool replay = pass_mode == 2;
    if (h->record_hits && !replay)
    {
        const size_t need = (size_t)h->n * B;
        if (need > h->hit_cap)
        {
            hipFree(h->hit_xy);
            hipFree(h->hit_cell);
            h->hit_cap = 0;
            int rc;
            if ((rc = dev_alloc(&h->hit_xy, 2 * need)) || (rc = dev_alloc(&h->hit_cell, need))) return rc;
            h->hit_cap = need;
        }
    }
    slamk::WeightArgs a;
    a.n = h->n;
    a.x = h->x;
    a.y = h->y;
    a.th = h->th;
    a.map_stride = h->map_stride;
    a.rows = h->rows;
    a.cols = h->cols;
    a.B = B;
    a.angles = h->angles;
    a.z2 = h->z2;
    a.z = h->z;
    a.zmax2 = max_dist * max_dist;      // slam/mcl.cpp:68
    a.zmax = std::sqrt(a.zmax2);        // slam/raycast.cpp:236
    a.stddev = stddev;
    a.clamp4 = 4.0 * stddev;
    a.inv_stddev = (stddev >= 1.0e-100 && stddev <= 1.0e100 && tuning().fast_div) ? 1.0 / stddev : 0.0;
    a.norm = 1.0 / (stddev * std::sqrt(2 * M_PI));  // slam/util.cpp:11
    a.log_eps = dm_log(0.0 + 0.1);                   // pdf clamped to 0 (slam/util.cpp:16) + EPSILON, slam/mcl.cpp:73
    a.lw = nullptr;
    a.p_first = 0;
    a.npblocks = 0;
    a.nseg = a.seg_beams = 0;
    a.order = nullptr;
    a.log_domain = h->prm.weight_domain == SLAM_WEIGHT_LOG;
    a.defer_edge = 0.0;
    // slam/mcl.cpp:80-86 with the int-returning squared distance of slam/util.h:138-146
    {
        const double ddx = 0 - off[0], ddy = 0 - off[1];
        const double sq = ddx * ddx + ddy * ddy;
        if (!(sq < 2147483648.0)) return fail(SLAM_ERR_ARG, "scanner_offset too large");
        a.sd = std::sqrt(static_cast<double>(static_cast<int>(sq)));
        a.stheta = std::atan2(off[1], off[0]);
        a.srot = off[2];
    }
    int ks = (int)std::floor(std::fmin(a.zmax, 1.0e6)) - 1;
    a.k_safe = ks < 0 ? 0 : ks;
    a.max_iter = (int)std::ceil(std::fmin(a.zmax, 1.0e6)) + 8;
    a.w = h->w;
    a.hit_xy = h->record_hits && !replay ? h->hit_xy : nullptr;
    a.hit_cell = h->record_hits && !replay ? h->hit_cell : nullptr;
    a.clamp_limit = !h->record_hits && tuning().clamp_limit;  // recorded hits are the reference's, wherever they lie
    a.stats = h->stats;
    a.luts = h->luts;
    const int order_mode = tuning().order_mode;
    // A cloud so sparse that no two particles are neighbours (more than SLAM_BEAM_CELLS cells of its bounding box per
    // particle: a uniformly seeded one) gains nothing from lanes over particles: the wave-per-particle kernel, whose
    // lanes are consecutive beams of ONE particle, is 1.1-1.6x faster there (and 2.2x slower on a dense cloud).  The box
    // is the one a kernel of an earlier pass left in pinned host memory: no wait, a pass or two late.
    bool sparse_cloud = false;
    if (!slam_mode && h->prm.raycaster == SLAM_RAYCAST_FAST && h->n >= 4096 && h->obox_host && tuning().beam_cells > 0)
    {
        // (the device writes the four numbers as one 16-byte store; one 16-byte load here: never a mix of two passes)
        int bx[4];
        _mm_storeu_si128(reinterpret_cast<__m128i*>(bx), _mm_load_si128(reinterpret_cast<const __m128i*>(h->obox_host)));
        const int x0 = bx[0], y0 = bx[1], x1 = bx[2], y1 = bx[3];
        if (x1 >= x0 && y1 >= y0)
        {
            // hysteresis: the shape changes above beam_cells cells per particle and changes back below three quarters of it,
            // so that a cloud near the threshold does not flip between the kernels from pass to pass
            const double cells = ((double)x1 - x0 + 1.0) * ((double)y1 - y0 + 1.0), per = (double)tuning().beam_cells * (double)h->n;
            sparse_cloud = cells > (h->sparse_last ? 0.75 * per : per);
        }
        h->sparse_last = sparse_cloud;
    }
    const bool particle_lanes = !slam_mode && (h->prm.raycaster == SLAM_RAYCAST_FAST_PARTICLE_LANES ||
                                               h->prm.raycaster == SLAM_RAYCAST_FAST_PARTICLE_LANES_DIRECT ||
                                               (h->prm.raycaster == SLAM_RAYCAST_FAST && h->n >= 4096 && !sparse_cloud));
    if (!(particle_lanes && order_mode > 0 && h->n >= 3) && !replay)  // k_order_keys zeroes the counters itself
        HIPCHK(hipMemsetAsync(h->stats, 0, 3 * sizeof(unsigned long long), h->stream));
    if (!(particle_lanes && order_mode > 0 && tuning().scan_fuse))  // k_order_keys also brings the new scan to scan_dev
    {
        int rc = flush_scan(h);
        if (rc) return rc;
    }
    // device time of the weighting kernels proper; tiled canvases first make their deferred copies
    // (the map clone of slam/mcl.cpp:147-155, accounted to that group)
    // (profile level 2 on the particle-lane path: no event records in the stream, which cost ~10 us of idle GPU each;
    // the first and the last kernel of the pass carry a start / a stop event on their own dispatch instead)
    TimedSpan ext = {};
    h->ext_timing = h->profile == 2 && particle_lanes;
    if (h->ext_timing)
    {
        ext.group = G_WEIGHTS;
        HIPCHK(hipEventCreate(&ext.a));
        HIPCHK(hipEventCreate(&ext.b));
    }
    std::unique_ptr<slam_mcl::Span> wspan;
    if (!(slam_mode && h->tiled)) wspan.reset(new slam_mcl::Span(h, G_WEIGHTS));
    const bool stats_on = h->collect_stats && !replay;
    h->ext_timing = false;
    if (slam_mode)
    {
        a.grid = h->maps;
        // wave per particle (64 samples of a ray per pass) while that still fills the chip; one
        // thread per particle beyond (SLAM_RAYCAST_* force a shape: 2 = wave, 3 = thread)
        const bool wave = h->prm.raycaster == SLAM_RAYCAST_FAST_BEAM_LANES ||
                          (h->prm.raycaster != SLAM_RAYCAST_FAST_PARTICLE_LANES && h->n < 262144);
        slamk::TileArgs ta = {h->tile_pool, h->tile_tab, h->tile_T, h->tile_tc, h->tile_free, h->tile_ctrl, 0};
        // two-phase kernel (k_weights_S2): the default wherever the wave form would run, as long as the
        // geometry bits of 64 rays (and the table window) fit the LDS of one wavefront
        const int npass = (a.max_iter + 63) / 64;
        const int win_w = 2 * npass + 2;
        const size_t s2_lds = 512 + (size_t)npass * 1024 + (h->tiled ? (size_t)win_w * win_w * 4 : 0);
        const bool two_phase = s2_lds <= 60 * 1024 && tuning().s_kernel != 1 &&
                               (h->prm.raycaster == SLAM_RAYCAST_TWO_PHASE || h->prm.raycaster == SLAM_RAYCAST_TWO_PHASE_LITERAL ||
                                (h->prm.raycaster == SLAM_RAYCAST_FAST && (wave || h->tiled)));  // tiles: at any particle count
        const int s2_literal = (tuning().s_kernel == 2 || h->prm.raycaster == SLAM_RAYCAST_TWO_PHASE_LITERAL) ? 1 : 0;
        if ((h->prm.raycaster == SLAM_RAYCAST_TWO_PHASE || h->prm.raycaster == SLAM_RAYCAST_TWO_PHASE_LITERAL) && !two_phase)
            return fail(SLAM_ERR_ARG, "SLAM_RAYCAST_TWO_PHASE: max_dist too large for the LDS of one wavefront");
        // "weights now, canvases later": possible when the pass's own writes that a later ray can meet fit an LDS window
        int dk = 0, dw = 0;
        bool defer = false;
        if (pass_mode == 1 && h->tiled && two_phase && tuning().tile_fuse != 0 && tuning().defer != 0 && h->prm.raycaster == SLAM_RAYCAST_FAST &&
            h->surv_list)
        {
            // smallest angular gap between two beams of the scan (directions, modulo a turn)
            std::vector<double> ang(h->scan_cache.begin(), h->scan_cache.begin() + B);
            for (double& v : ang)
            {
                v = std::fmod(v, 2 * M_PI);
                if (v < 0) v += 2 * M_PI;
            }
            std::sort(ang.begin(), ang.end());
            double gap = B > 1 ? ang[0] + 2 * M_PI - ang[B - 1] : M_PI;
            for (int b = 1; b < B; ++b) gap = std::fmin(gap, ang[b] - ang[b - 1]);
            // two rays from one origin, `gap` apart, pass farther than sqrt(2) from each other beyond sqrt(2) / sin(gap)
            // cells from it: no common cell there; one cell of margin for the accumulated positions, one for the sample count
            const double rstar = gap >= M_PI / 2 ? 1.4143 : (gap > 1.0e-6 ? 1.4143 / std::sin(gap) : 1.0e9);
            a.defer_edge = 2.0 * rstar + 2.0;
            if (rstar < 60.0)
            {
                dk = (int)std::ceil(rstar + 1.0) + 1;
                dw = ((2 * (dk + 2) + 1 + 14) + 7) & ~7;
                const bool few_survive = h->surv_count_host && *(volatile int*)h->surv_count_host >= 0 &&
                                         2 * (int64_t)(*(volatile int*)h->surv_count_host) < h->n;
                // (below 8192 particles the step is one chain of dependent rays long: a second pass over the survivors costs
                //  more than the copies it saves; SLAM_DEFER=2 defers regardless, for the tests)
                defer = s2_lds + (size_t)dw * dw <= 60 * 1024 && (tuning().defer == 2 || (few_survive && h->n >= 8192));
            }
        }
        if (h->tiled && defer)
        {
            // the canvases stay as they are; do_resample runs the ordinary pass on the particles it selects.  (Particles too
            // close to row 0 / column 0 for that, slam_defer_safe, are listed and take the ordinary pass below.)
            h->deferred = true;
            h->edge_valid = true;
            ++h->n_deferred;
            HIPCHK(hipMemsetAsync(h->edge_count, 0, sizeof(int), h->stream));
            hipLaunchKernelGGL(slamk::k_defer_split, dim3((unsigned)cdiv(h->n, 256)), dim3(256), 0, h->stream, a, h->edge_flag, h->edge_list, h->edge_count);
            h->rp_B = B;
            h->rp_stddev = stddev;
            h->rp_max_dist = max_dist;
            for (int k = 0; k < 3; ++k) h->rp_off[k] = off[k];
            wspan.reset(new slam_mcl::Span(h, G_WEIGHTS));
            const slamk::S2Subset all = {nullptr, nullptr};
            const size_t lds = s2_lds + (size_t)dw * dw;
            if (stats_on)
                hipLaunchKernelGGL((slamk::k_weights_S2<true, true, true>), dim3((unsigned)h->n), dim3(64), lds, h->stream, a, ta, npass, win_w,
                                   s2_literal, all, dk, dw);
            else if (tuning().defer_fast)
                hipLaunchKernelGGL(slamk::k_weights_S2F, dim3((unsigned)h->n), dim3(64), (size_t)512 + 1024 + (size_t)win_w * win_w * 4 + (size_t)dw * dw,
                                   h->stream, a, ta, win_w, s2_literal, dk, dw);
            else
                hipLaunchKernelGGL((slamk::k_weights_S2<true, false, true>), dim3((unsigned)h->n), dim3(64), lds, h->stream, a, ta, npass, win_w,
                                   s2_literal, all, dk, dw);
            wspan.reset();
        }
        if (h->tiled)
        {
            // make every tile within sensor range exclusive up front (parallel), then walk
            const slamk::S2Subset sub = {defer ? h->edge_list : (replay ? h->surv_list : nullptr), defer ? h->edge_count : (replay ? h->surv_count : nullptr)};
            std::unique_ptr<slam_mcl::Span> copies(new slam_mcl::Span(h, G_CLONE));
            // (no farther than the scan's farthest write: min(z, zmax) + 2 over its beams, plus a cell of margin)
            double range = 0.0;
            {
                const double* zz = h->scan_cache.data() + 2 * (size_t)B;
                for (int b = 0; b < B; ++b) range = std::fmax(range, std::fmin(zz[b], a.zmax));
                range = std::fmin(range, a.zmax) + 3.0;
            }
            // (bounding square in tiles; capped: what it misses is copied inside the kernel)
            const int span = 2 * std::max(h->tile_T / h->tile_tc, h->tile_tc) + 1;
            const int side = std::min(std::min(2 * (int)std::ceil(std::fmin(range, 1.0e6) / 64.0) + 3, span), 31);
            const int64_t per = std::min<int64_t>((int64_t)side * side, h->tile_T);
            const int64_t blocks = std::min<int64_t>(std::min<int64_t>(h->n * per, h->tile_cap), 16384);
            slamk::TilePlan pl = {h->tile_wl_p, h->tile_wl_t, h->tile_rng};
            const bool fuse = two_phase && tuning().tile_fuse != 0;
            {
                // per sector of the scan frame (sector k covers angles [-pi + k 2pi/256, ...)): the farthest write
                float sec[256] = {0.0f};
                slamk::SectorTable dil;
                const double* ang = h->scan_cache.data();
                const double* zz = ang + 2 * (size_t)B;
                for (int b = 0; b < B; ++b)
                {
                    double al = std::fmod(ang[b], 2 * M_PI);
                    if (al >= M_PI) al -= 2 * M_PI;
                    if (al < -M_PI) al += 2 * M_PI;
                    const int k = (int)std::floor((al + M_PI) * (256.0 / (2 * M_PI))) & 255;
                    const float r = (float)(std::fmin(zz[b], a.zmax) + 2.0) * 1.0001f;
                    if (r > sec[k]) sec[k] = r;
                }
                for (int k = 0; k < 256; ++k) dil.r[k] = std::max(sec[k], std::max(sec[(k + 1) & 255], sec[(k + 255) & 255]));
                // travels as a kernel argument: a copy from pageable memory would make the host wait for the stream
                hipLaunchKernelGGL(slamk::k_set_sectors, dim3(1), dim3(256), 0, h->stream, dil, h->tile_sector);
            }
            if (!fuse) hipLaunchKernelGGL(slamk::k_tile_mark, dim3(1), dim3(1), 0, h->stream, h->tile_ctrl);  // (where k_tile_copy starts)
            hipLaunchKernelGGL(slamk::k_tile_plan, dim3((unsigned)cdiv(h->n, 16 * slamk::TILE_PLAN_PW)), dim3(1024), 0, h->stream, h->n, h->x, h->y, h->th, a.sd,
                               a.stheta, a.srot, range, h->tile_sector, h->rows, h->cols, side, ta, pl, sub.list, sub.count);
            if (fuse)
            {
                ta.plan_rng = h->tile_rng;
                ta.plan_t = h->tile_wl_t;
            }
            else
                hipLaunchKernelGGL(slamk::k_tile_copy, dim3((unsigned)blocks), dim3(256), 0, h->stream, ta, pl);
            copies.reset();
            wspan.reset(new slam_mcl::Span(h, replay ? G_CLONE : G_WEIGHTS));  // (the second half of a deferred pass IS the deferred deep copy)
            size_t lds = 0;
            if ((size_t)h->tile_T * 16 <= 48 * 1024)
            {
                ta.lds_entries = h->tile_T;
                lds = (size_t)h->tile_T * 16;
            }
            if (two_phase && stats_on)
                hipLaunchKernelGGL((slamk::k_weights_S2<true, true>), dim3((unsigned)h->n), dim3(64), s2_lds, h->stream, a, ta, npass, win_w,
                                   s2_literal, sub, 0, 0);
            else if (two_phase)
                hipLaunchKernelGGL((slamk::k_weights_S2<true, false>), dim3((unsigned)h->n), dim3(64), s2_lds, h->stream, a, ta, npass, win_w,
                                   s2_literal, sub, 0, 0);
            else
                hipLaunchKernelGGL(slamk::k_weights_S_wave<true>, dim3((unsigned)cdiv(h->n, 4)), dim3(256), lds, h->stream, a, ta);
        }
        else if (two_phase && stats_on)
            hipLaunchKernelGGL((slamk::k_weights_S2<false, true>), dim3((unsigned)h->n), dim3(64), s2_lds, h->stream, a, ta, npass, win_w,
                               s2_literal, slamk::S2Subset{nullptr, nullptr}, 0, 0);
        else if (two_phase)
            hipLaunchKernelGGL((slamk::k_weights_S2<false, false>), dim3((unsigned)h->n), dim3(64), s2_lds, h->stream, a, ta, npass, win_w,
                               s2_literal, slamk::S2Subset{nullptr, nullptr}, 0, 0);
        else if (wave)
            hipLaunchKernelGGL(slamk::k_weights_S_wave<false>, dim3((unsigned)cdiv(h->n, 4)), dim3(256), 0, h->stream, a, ta);
        else
            hipLaunchKernelGGL(slamk::k_weights_S, dim3((unsigned)cdiv(h->n, 64)), dim3(64), 0, h->stream, a);
    }
    else if (particle_lanes)
    {
        // lane = particle; (64-particle block, beam segment) work items; see k_rays_pl
        a.grid = h->clr;
        a.map_stride = h->clr_stride;
        const size_t cap_max = (size_t)64 << 20;  // 512 MiB of scratch at most (two launches with slabs that fit the 256 MB cache: 4 % slower)
        const int64_t pblocks_all = cdiv(h->n, 64);
        // enough particle blocks to fill the chip several times over: one launch, whole scans per lane, no slab
        const bool direct = h->prm.raycaster == SLAM_RAYCAST_FAST_PARTICLE_LANES_DIRECT ||
                            (h->prm.raycaster == SLAM_RAYCAST_FAST && pblocks_all >= tuning().direct_min);
        int64_t pb_per_launch = (int64_t)(cap_max / ((size_t)B * 64));
        if (pb_per_launch < 1) pb_per_launch = 1;
        if (pb_per_launch > pblocks_all || direct) pb_per_launch = pblocks_all;
        const size_t need = direct ? 0 : (size_t)pb_per_launch * B * 64;
        if (need > h->lw_cap)
        {
            hipFree(h->lw);
            h->lw_cap = 0;
            int rc = dev_alloc(&h->lw, need);
            if (rc) return rc;
            h->lw_cap = need;
        }
        a.lw = h->lw;
        const int order_bits = tuning().order_bits;
        // mean range at which the scan's rays end: sizes the heading buckets of the pose-coherent order (k_order_keys)
        float scan_range = 0.0f;
        if (h->scan_cache.size() >= 3 * (size_t)B)
        {
            double sum = 0.0;
            const double* zz = h->scan_cache.data() + 2 * (size_t)B;
            for (int b = 0; b < B; ++b) sum += zz[b] < max_dist ? zz[b] : max_dist;
            scan_range = (float)(sum / B);
        }
        bool ext_started = false;
        if (order_mode > 0)
        {
            {
                int rc = ensure_order_box(h);
                if (rc) return rc;
            }
            if (h->poses_reset)
            {  // poses set or seeded since the last pass: the box this pass reads comes from the new cloud
                hipLaunchKernelGGL(slamk::k_order_box, dim3(1), dim3(256), 0, h->stream, h->n, h->x, h->y, h->th, h->obox, (h->order_pass + 1) & 1, h->obox_host);
                h->poses_reset = false;
            }
            hipExtLaunchKernelGGL(slamk::k_order_keys, dim3((unsigned)cdiv(h->n, 256)), dim3(256), 0, h->stream, ext.a, nullptr, 0, h->n,
                                  h->x, h->y, h->th, order_mode, order_bits, h->okeys, h->oident, h->stats, h->scan_pending,
                                  h->scan_dev, (int)(h->scan_pending ? h->scan_pending_n : 0), h->obox, h->order_pass++ & 1,
                                  order_bits == 32 ? tuning().order_wide : 0, h->obox_host, tuning().order_adapt ? scan_range : 0.0f);
            ext_started = true;
            if (h->scan_pending)
            {
                int rc = scan_consumed(h);
                if (rc) return rc;
            }
            {
                int rc = sort_order_keys(h, order_bits);
                if (rc) return rc;
            }
            a.order = h->order;
        }
        for (int64_t pb0 = 0; pb0 < pblocks_all; pb0 += pb_per_launch)
        {
            a.p_first = pb0 * 64;
            a.npblocks = std::min(pb_per_launch, pblocks_all - pb0);
            // enough work items to keep 8 waves on each of the 1024 SIMDs through the tail
            int64_t want = cdiv(4 * 8192, a.npblocks);
            if (want < 1) want = 1;
            if (want > B) want = B;
            a.seg_beams = (int)cdiv(B, want);
            const int seg_env = tuning().seg_beams;
            if (seg_env > 0) a.seg_beams = seg_env < B ? seg_env : B;
            if (direct) a.seg_beams = B;
            a.nseg = (int)cdiv(B, a.seg_beams);
            const dim3 grid((unsigned)cdiv(a.npblocks * a.nseg, 4));
            const int pl_waves = tuning().waves;
            hipEvent_t ev_a = ext_started ? nullptr : ext.a;  // (null events: a plain launch)
            ext_started = true;
            if (direct)
            {
                if (a.hit_xy)
                    hipExtLaunchKernelGGL((slamk::k_rays_pl<4, true, true>), grid, dim3(256), 0, h->stream, ev_a, ext.b, 0, a);
                else
                    hipExtLaunchKernelGGL((slamk::k_rays_pl<5, true>), grid, dim3(256), 0, h->stream, ev_a, ext.b, 0, a);
                break;
            }
            if (a.hit_xy)
                hipExtLaunchKernelGGL((slamk::k_rays_pl<4, false, true>), grid, dim3(256), 0, h->stream, ev_a, nullptr, 0, a);
            else if (pl_waves >= 8)
                hipExtLaunchKernelGGL(slamk::k_rays_pl<8>, grid, dim3(256), 0, h->stream, ev_a, nullptr, 0, a);
            else if (pl_waves >= 6)
                hipExtLaunchKernelGGL(slamk::k_rays_pl<6>, grid, dim3(256), 0, h->stream, ev_a, nullptr, 0, a);
            else if (pl_waves >= 5)
                hipExtLaunchKernelGGL(slamk::k_rays_pl<5>, grid, dim3(256), 0, h->stream, ev_a, nullptr, 0, a);
            else
                hipExtLaunchKernelGGL(slamk::k_rays_pl<4>, grid, dim3(256), 0, h->stream, ev_a, nullptr, 0, a);
            const bool last = pb0 + pb_per_launch >= pblocks_all;
            hipExtLaunchKernelGGL(slamk::k_sum_lw, dim3((unsigned)cdiv(a.npblocks * 64, 256)), dim3(256), 0, h->stream, nullptr,
                                  last ? ext.b : nullptr, 0, a);
        }
        if (ext.a) h->spans.push_back(ext);
    }
    else if (h->prm.raycaster == SLAM_RAYCAST_FAST || h->prm.raycaster == SLAM_RAYCAST_FAST_BEAM_LANES)
    {
        a.grid = h->clr;
        a.map_stride = h->clr_stride;
        if (sparse_cloud && h->okeys && tuning().order_bits == 32 && tuning().order_sparse)
        {
            // the particles in place order (keys and sort as for lanes over particles, but Morton order of 64-cell tiles:
            // with one particle per wavefront a square neighbourhood beats a band, 340 -> 299 ms on the 16384^2 shard;
            // k_order_keys also leaves this pass's bounding box): wavefronts that run together read one part of the planes
            hipLaunchKernelGGL(slamk::k_order_keys, dim3((unsigned)cdiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->x, h->y, h->th,
                               tuning().order_sparse, 32, h->okeys, h->oident, h->stats, (const double*)nullptr, h->scan_dev, 0, h->obox,
                               h->order_pass++ & 1, tuning().order_sparse, h->obox_host, 0.0f);
            h->poses_reset = false;
            {
                int rc = sort_order_keys(h, 32);
                if (rc) return rc;
            }
            a.order = h->order;
        }
        else if (sparse_cloud)
        {  // (no order: the box of this pass from a launch of its own)
            hipLaunchKernelGGL(slamk::k_order_box, dim3(1), dim3(256), 0, h->stream, h->n, h->x, h->y, h->th, h->obox, h->order_pass++ & 1, h->obox_host);
            h->poses_reset = false;
        }
        // (6: the register budget of six wavefronts per SIMD, 80 VGPRs; measured on the 16384^2 shard: 95 VGPRs / 5 wavefronts
        // 225.6 ms, 6: 204.3, 7 with 48 bytes of scratch: 215.7, 8 with 80 bytes: 220.2)
        hipLaunchKernelGGL((slamk::k_weights_L<0, 6>), dim3((unsigned)cdiv(h->n, 4)), dim3(256), 0, h->stream, a);
    }
    else
    {
        a.grid = h->grid;
        hipLaunchKernelGGL(slamk::k_weights_L<1>, dim3((unsigned)cdiv(h->n, 4)), dim3(256), 0, h->stream, a);
    }
    HIPCHK(hipGetLastError());
    return SLAM_OK;
}

// Canonical blocked running sum of `in` (nullptr: the weight column of `view`) into u.P[0].
// u.n[l] = entries of level l (n[0] = all weights, n[l+1] = ceil(n[l] / 64), the last level holds <= 64);
// the top level's P is its s (same buffer).  Launches only; allocates nothing.
// part: 0 = all of it; 1 = the scans inside the 64-chunks of the weights only (k_select<true> rebuilds the rest for itself);
// 2 = what part 1 left out (slam_mcl_get_prefix after such a selection).  Parts 1 and 2: three-level hierarchies only.
void launch_blocked_prefix(hipStream_t stream, const double* in, const slamk::StateView& view, const slamk::UpperLevels& u, int part = 0)
{
    const int L = u.L;
    if (L >= 3 && L <= 8 && u.n[1] <= 65536)
    {
        // the small upper levels in one launch (3 launches for 100k particles instead of 5)
        if (part != 2)
            hipLaunchKernelGGL(slamk::k_scan_chunks, dim3((unsigned)cdiv(cdiv(u.n[0], 64), 4)), dim3(256), 0, stream, in, view, u.n[0],
                               u.s[0], u.tot[0]);
        if (part == 1) return;
        hipLaunchKernelGGL(slamk::k_scan_upper, dim3(1), dim3(1024), 0, stream, u);
        hipLaunchKernelGGL(slamk::k_downsweep, dim3((unsigned)cdiv(u.n[0], 256)), dim3(256), 0, stream, u.s[0], u.P[1], u.n[0],
                           u.P[0]);
        return;
    }
    for (int l = 0; l < L; ++l)
    {
        hipLaunchKernelGGL(slamk::k_scan_chunks, dim3((unsigned)cdiv(cdiv(u.n[l], 64), 4)), dim3(256), 0, stream, in, view, u.n[l],
                           u.s[l], u.tot[l]);
        in = u.tot[l];
    }
    // top level: P = s (same buffer); lower levels add the earlier chunks' sum
    for (int l = L - 2; l >= 0; --l)
        hipLaunchKernelGGL(slamk::k_downsweep, dim3((unsigned)cdiv(u.n[l], 256)), dim3(256), 0, stream, u.s[l], u.P[l + 1], u.n[l],
                           u.P[l]);
}

// running sums of the weights of all shards (n_glob entries) into lvl_P[0], in the configured order
bool select_fusable(const slam_mcl* h)
{
    return tuning().select_fuse != 0 && h->prm.prefix_order == SLAM_PREFIX_BLOCKED && h->lvl_n.size() == 3 &&
           h->lvl_n[1] <= slamk::SELECT_FUSED_MAX_N1;
}
slamk::UpperLevels upper_levels(const slam_mcl* h)
{
    slamk::UpperLevels u;
    u.L = (int)h->lvl_n.size();
    for (int l = 0; l < u.L; ++l)
    {
        u.n[l] = h->lvl_n[l];
        u.s[l] = h->lvl_s[l];
        u.tot[l] = h->lvl_tot[l];
        u.P[l] = h->lvl_P[l];
    }
    return u;
}
// lvl_P[0] in full after a selection that did without it (readers of the running sums: tests, tools, device_state)
int complete_prefix(slam_mcl* h)
{
    if (!h->prefix_partial) return SLAM_OK;
    launch_blocked_prefix(h->stream, nullptr, slamk::StateView{}, upper_levels(h), 2);
    HIPCHK(hipGetLastError());
    h->prefix_partial = false;
    return SLAM_OK;
}
int do_prefix(slam_mcl* h, const slamk::StateView& view, bool for_fused_select = false)
{
    const double* lin = nullptr;  // SLAM_WEIGHT_LOG: the selection runs on exp(lw - max lw)
    if (h->prm.weight_domain == SLAM_WEIGHT_LOG)
    {
        int64_t m = h->n_glob;
        const double* src = nullptr;
        double* dst = h->wmax;
        for (;;)
        {
            const int64_t nb = cdiv(m, 256);
            hipLaunchKernelGGL(slamk::k_max_partials, dim3((unsigned)nb), dim3(256), 0, h->stream, src, view, m, dst);
            if (nb == 1) break;
            src = dst;
            dst += nb;
            m = nb;
        }
        hipLaunchKernelGGL(slamk::k_exp_shift, dim3((unsigned)cdiv(h->n_glob, 256)), dim3(256), 0, h->stream, view,
                           h->n_glob, dst, h->wlin);
        HIPCHK(hipGetLastError());
        lin = h->wlin;
    }
    if (h->prm.prefix_order == SLAM_PREFIX_REFERENCE)
    {
        // single shard only (checked by the caller): the weight column is contiguous
        hipLaunchKernelGGL(slamk::k_prefix_seq, dim3(1), dim3(64), 0, h->stream, lin ? lin : view.base + 3 * view.n_loc,
                           h->n_glob, h->lvl_P[0]);
        HIPCHK(hipGetLastError());
        return SLAM_OK;
    }
    launch_blocked_prefix(h->stream, lin, view, upper_levels(h), for_fused_select ? 1 : 0);
    h->prefix_partial = for_fused_select;
    HIPCHK(hipGetLastError());
    return SLAM_OK;
}

// n_src: canvases behind `from` (entries of src outside it are skipped and reported by finish())
void launch_clone(slam_mcl* h, const uint8_t* from, uint8_t* to, const int32_t* src, int64_t count, int64_t n_src)
{
    const unsigned chunks = (unsigned)std::min<size_t>(cdiv((int64_t)(h->map_stride / 16), 256), 64);
    hipLaunchKernelGGL(slamk::k_clone_maps, dim3((unsigned)count, chunks), dim3(256), 0, h->stream, from, to, src,
                       h->map_stride, n_src, h->dev_bad);
}

// `gathered`: packed states of all shards ([n_shards][4][n]); nullptr = this handle alone.
// `gathered_w` != nullptr: split form, gathered = [n_shards][3][n] poses, gathered_w = [n_shards][n] weights.
// clone_local = false: SLAM-mode shards; the caller exchanges the canvases and finishes with
// slam_mcl_shard_clone_maps.
int do_resample(slam_mcl* h, const double* gathered, int n_shards, bool clone_local = true, const double* gathered_w = nullptr)
{
    const bool injected = h->prm.noise == SLAM_NOISE_INJECTED;
    if (injected && !h->have_uniforms) return fail(SLAM_ERR_STATE, "injected noise: no uniforms set for this update");
    const int64_t n = h->n;
    if ((int64_t)n_shards * n != h->n_glob)
        return fail(SLAM_ERR_STATE, "resample: shards * n_particles != global_n (sharded handles must use slam_mcl_shard_resample)");
    if (n_shards > 1 && (h->prm.sort != SLAM_SORT_NONE || h->prm.prefix_order != SLAM_PREFIX_BLOCKED))
        return fail(SLAM_ERR_STATE, "sharded resampling needs SLAM_SORT_NONE and SLAM_PREFIX_BLOCKED");
    if (n_shards > 1 && h->prm.mode == SLAM_MODE_SLAM && clone_local)
        return fail(SLAM_ERR_STATE, "SLAM-mode shards: use slam_mcl_shard_select + slam_mcl_shard_clone_maps");
    slamk::StateView view;
    view.base = gathered ? gathered : h->state;
    view.n_loc = n;
    view.wbase = gathered ? gathered_w : nullptr;
    {
        slam_mcl::Span span(h, G_RESAMPLE);
        const bool fused = select_fusable(h);
        h->prefix_partial = false;
        int rc = do_prefix(h, view, fused);
        if (rc) return rc;
        slamk::SelectArgs s;
        s.s0 = h->lvl_s[0];
        s.tot0 = h->lvl_tot[0];
        s.n1 = (int)(h->lvl_n.size() > 1 ? h->lvl_n[1] : 0);
        s.P = h->lvl_P[0];
        s.n_total = h->n_glob;
        s.slot_first = h->prm.global_first;
        s.n_slots = n;
        s.injected = injected ? h->inj_uniforms : nullptr;
        s.seed = h->prm.seed;
        s.step = h->resample_step;
        s.resampler = h->prm.resampler;
        s.idx = h->idx;
        s.view = view;
        s.keys = h->prm.sort == SLAM_SORT_FULL ? h->keys : nullptr;
        s.slots = h->slots;
        if (fused)
            hipLaunchKernelGGL(slamk::k_select<true>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream, s);
        else
            hipLaunchKernelGGL(slamk::k_select<false>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream, s);
        HIPCHK(hipGetLastError());
    }
    if (h->surv_list && n_shards == 1 && clone_local)
    {
        // the distinct particles this resample selected: how many (read by a later update, which then decides whether to
        // defer its canvases) and, after a deferred weighting pass, which: their canvases are brought up to date now, by
        // the ordinary pass on them alone, before their tables are handed to their children below (slam/mcl.cpp:147-155
        // copies the maps the update has just drawn)
        slam_mcl::Span span(h, G_CLONE);
        const int gen = ++h->surv_gen;
        HIPCHK(hipMemsetAsync(h->surv_count, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL(slamk::k_mark_sources, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream, h->idx, n, (int64_t)0, n, gen, h->surv_stamp);
        hipLaunchKernelGGL(slamk::k_list_sources, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream, h->surv_stamp, n, gen, h->surv_list, h->surv_count,
                           h->deferred && h->edge_valid ? h->edge_flag : nullptr);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h->surv_count_host, h->surv_count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    }
    if (h->deferred)
    {
        h->deferred = false;
        h->edge_valid = false;
        if (!(h->surv_list && n_shards == 1 && clone_local)) return fail(SLAM_ERR_STATE, "resample: deferred canvases without a local resample (library bug)");
        int rc = do_weights(h, h->rp_B, h->rp_stddev, h->rp_max_dist, h->rp_off, 2);
        if (rc) return rc;
    }
    const int32_t* perm = nullptr;
    bool states_moved = false;
    if (h->prm.sort == SLAM_SORT_FULL)
    {
        slam_mcl::Span span(h, G_SORT);
        // (weight descending, slot ascending): any stable sort gives this order
        const slamk::GatherEpilogue ge = {view, h->idx, n, h->state2, h->src};
        int rc = sort_by_weight(h, n, &ge, &states_moved);
        if (rc) return rc;
        perm = h->perm;
    }
    if (!states_moved)
    {
        slam_mcl::Span span(h, G_RESAMPLE);
        hipLaunchKernelGGL(slamk::k_gather, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream, view, h->idx, perm, n,
                           h->state2, h->src);
        HIPCHK(hipGetLastError());
    }
    std::swap(h->state, h->state2);
    h->bind();
    if (h->prm.mode == SLAM_MODE_SLAM && clone_local && h->tiled)
    {
        // slam/mcl.cpp:147-155 without the deep copy: the selected particles share their source's
        // tiles until one of them writes
        slam_mcl::Span span(h, G_CLONE);
        int rc = tile_collect(h, h->src);
        if (rc) return rc;
    }
    else if (h->prm.mode == SLAM_MODE_SLAM && clone_local)
    {
        slam_mcl::Span span(h, G_CLONE);
        launch_clone(h, h->maps, h->maps2, h->src, n, n);
        HIPCHK(hipGetLastError());
        std::swap(h->maps, h->maps2);
    }
    h->have_uniforms = false;
    h->resample_step++;
    return SLAM_OK;
}

int maybe_finish(slam_mcl* h)
{
    if (h->prm.async_launch) return SLAM_OK;
    return finish(h);
}
}  // namespace

SLAM_API const char* slam_last_error(void) { return g_err.c_str(); }
SLAM_API int slam_version(void) { return 1; }

SLAM_API void slam_mcl_destroy(slam_mcl* h)
{
    if (!h) return;
    hipSetDevice(h->prm.device);
    hipStreamSynchronize(h->stream);
    for (auto& s : h->spans)
    {
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    void* ptrs[] = {h->state, h->state2,
                    h->grid,  h->clr,    h->maps, h->maps2,    h->luts,         h->scan_dev,
                    h->inj_normals, h->inj_uniforms, h->idx, h->perm, h->slots, h->src,   h->keys, h->keys2,
                    h->sort_tmp, h->hit_xy, h->hit_cell, h->stats, h->partials, h->lw, h->surv_stamp, h->surv_list, h->surv_count, h->edge_flag, h->edge_list, h->edge_count,
                    h->okeys, h->okeys2, h->obox, h->oident, h->order, h->osort_tmp, h->ss_tmp, h->wlin, h->wmax,
                    h->tile_pool, h->tile_tab, h->tile_tab2, h->tile_free, h->tile_ref, h->tile_who, h->tile_shared, h->tile_ctrl, h->tile_dense, h->tile_sector,
                    h->tile_wl_p, h->tile_wl_t, h->tile_rng, h->pack_stamp, h->pack_ctr, h->pack_slot, h->dev_bad};
    for (void* p : ptrs)
        if (p) hipFree(p);
    for (int k = 0; k < slam_mcl::SCAN_RING; ++k)
        if (h->scan_pin[k]) hipHostFree(h->scan_pin[k]);
    if (h->obox_host) hipHostFree(h->obox_host);
    if (h->surv_count_host) hipHostFree(h->surv_count_host);
    for (int g = 0; g < slam_mcl::SCAN_RING / slam_mcl::SCAN_GROUP; ++g)
        if (h->scan_ev[g]) hipEventDestroy(h->scan_ev[g]);
    for (size_t l = 0; l < h->lvl_s.size(); ++l)
    {
        if (h->lvl_s[l]) hipFree(h->lvl_s[l]);
        if (h->lvl_P[l] && h->lvl_P[l] != h->lvl_s[l]) hipFree(h->lvl_P[l]);
        if (h->lvl_tot[l]) hipFree(h->lvl_tot[l]);
    }
    delete h;
}

SLAM_API int slam_mcl_create(const slam_mcl_params* params, slam_mcl** out)
{
    if (!params || !out) return fail(SLAM_ERR_ARG, "null params/out");
    *out = nullptr;
    if (params->struct_size != (int32_t)sizeof(slam_mcl_params))
        return fail(SLAM_ERR_ARG, "slam_mcl_params.struct_size mismatch (header/library version skew)");
    if (params->n_particles <= 0 || params->n_particles >= 2147483647LL)
        return fail(SLAM_ERR_ARG, "n_particles must be in [1, 2^31-1) (the reference indexes with int)");
    if (params->rows <= 0 || params->cols <= 0 || params->rows > (1 << 20) || params->cols > (1 << 20) ||
        (int64_t)params->rows * params->cols >= 2147483647LL)
        return fail(SLAM_ERR_ARG, "rows/cols out of range");
    if (params->mode != SLAM_MODE_LOCALIZATION && params->mode != SLAM_MODE_SLAM) return fail(SLAM_ERR_ARG, "bad mode");
    if (params->raycaster < SLAM_RAYCAST_FAST || params->raycaster > SLAM_RAYCAST_FAST_PARTICLE_LANES_DIRECT)
        return fail(SLAM_ERR_ARG, "bad raycaster");
    if (params->mode == SLAM_MODE_SLAM && params->raycaster == SLAM_RAYCAST_FAST_PARTICLE_LANES_DIRECT)
        return fail(SLAM_ERR_ARG, "SLAM_RAYCAST_FAST_PARTICLE_LANES_DIRECT is a localization-mode kernel shape");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (params->device < 0 || params->device >= ndev) return fail(SLAM_ERR_ARG, "device ordinal out of range");
    HIPCHK(hipSetDevice(params->device));

    slam_mcl* h = new (std::nothrow) slam_mcl();
    if (!h) return fail(SLAM_ERR_NOMEM, "host allocation failed");
    h->prm = *params;
    if (h->prm.global_n <= 0)
    {
        h->prm.global_n = params->n_particles;
        h->prm.global_first = 0;
    }
    h->stream = reinterpret_cast<hipStream_t>(params->stream);
    h->n = params->n_particles;
    h->rows = params->rows;
    h->cols = params->cols;
    const int64_t n = h->n;
    int rc = SLAM_OK;
#define TRY(e)                  \
    if (!rc) rc = (e)
    h->n_glob = h->prm.global_n;
    if (h->prm.global_first < 0 || h->prm.global_first + n > h->n_glob || h->n_glob >= 2147483647LL)
        rc = fail(SLAM_ERR_ARG, "global_first/global_n do not contain this shard");
    TRY(dev_alloc(&h->state, 4 * n));
    TRY(dev_alloc(&h->state2, 4 * n));
    if (!rc) h->bind();
    TRY(dev_alloc(&h->idx, n));
    TRY(dev_alloc(&h->dev_bad, 1));
    if (!rc && hipMemsetAsync(h->dev_bad, 0, sizeof(int), h->stream) != hipSuccess) rc = fail(SLAM_ERR_HIP, "memset");
    TRY(dev_alloc(&h->perm, n));
    TRY(dev_alloc(&h->slots, n));
    TRY(dev_alloc(&h->src, n));
    TRY(dev_alloc(&h->keys, n));
    TRY(dev_alloc(&h->keys2, n));
    TRY(dev_alloc(&h->stats, 3));
    TRY(dev_alloc(&h->luts, 512));
    TRY(dev_alloc(&h->partials, 6 * (size_t)cdiv(n, 256)));  // average_pose: 4 per block; shard_summary: + 2
    if (h->prm.weight_domain != SLAM_WEIGHT_LINEAR && h->prm.weight_domain != SLAM_WEIGHT_LOG)
        rc = fail(SLAM_ERR_ARG, "weight_domain must be SLAM_WEIGHT_LINEAR or SLAM_WEIGHT_LOG");
    if (h->prm.weight_domain == SLAM_WEIGHT_LOG)
    {
        TRY(dev_alloc(&h->wlin, (size_t)h->n_glob));
        TRY(dev_alloc(&h->wmax, 2 * (size_t)cdiv(h->n_glob, 256) + 64));
    }
    if (!rc && h->n_glob != n && (h->prm.sort != SLAM_SORT_NONE || h->prm.prefix_order != SLAM_PREFIX_BLOCKED))
        rc = fail(SLAM_ERR_ARG, "a shard (global_n > n_particles) needs SLAM_SORT_NONE and SLAM_PREFIX_BLOCKED");
    if (h->prm.noise == SLAM_NOISE_INJECTED)
    {
        TRY(dev_alloc(&h->inj_normals, 3 * n));
        TRY(dev_alloc(&h->inj_uniforms, n));
    }
    // running-sum hierarchy
    if (!rc)
    {
        int64_t nl = h->n_glob;  // every shard builds the running sum of the whole filter
        for (;;)
        {
            h->lvl_n.push_back(nl);
            if (nl <= 64) break;
            nl = cdiv(nl, 64);
        }
        const size_t L = h->lvl_n.size();
        h->lvl_s.assign(L, nullptr);
        h->lvl_P.assign(L, nullptr);
        h->lvl_tot.assign(L, nullptr);
        for (size_t l = 0; l < L && !rc; ++l)
        {
            TRY(dev_alloc(&h->lvl_s[l], h->lvl_n[l]));
            TRY(dev_alloc(&h->lvl_tot[l], cdiv(h->lvl_n[l], 64)));
            if (l + 1 == L && h->prm.prefix_order == SLAM_PREFIX_BLOCKED)
                h->lvl_P[l] = h->lvl_s[l];
            else
                TRY(dev_alloc(&h->lvl_P[l], h->lvl_n[l]));
        }
    }
    if (!rc && n >= sortk::MIN_N)
    {
        // scratch of the library's own sorts, shared by both sorts of a step: the sample sort's, or the radix sort's with
        // 64-bit keys (the order sort's 32-bit form needs less)
        uint8_t* tmp = nullptr;
        TRY(dev_alloc(&tmp, n <= sortk::MAX_N ? sortk::scratch_bytes(n) : sortk::radix_scratch_bytes<uint64_t>(n)));
        h->ss_tmp = tmp;
    }
    if (!rc)
    {
        uint8_t t[512];
        build_luts(t);
        if (hipMemcpy(h->luts, t, 512, hipMemcpyHostToDevice) != hipSuccess) rc = fail(SLAM_ERR_HIP, "lut upload");
    }
    if (!rc && (h->prm.map_storage < SLAM_MAPS_AUTO || h->prm.map_storage > SLAM_MAPS_TILED))
        rc = fail(SLAM_ERR_ARG, "map_storage must be SLAM_MAPS_AUTO, SLAM_MAPS_DENSE or SLAM_MAPS_TILED");
    if (!rc && h->prm.map_storage == SLAM_MAPS_AUTO)
    {
        // dense canvases only where the one-thread-per-particle kernel takes over (262144 particles and more) AND the
        // two copies of every canvas fit a quarter of the HBM; maze-sized canvases never do: tiles
        const double dense_bytes = 2.0 * (double)n * (double)(((size_t)h->rows * h->cols + 255) / 256 * 256);
        h->prm.map_storage = (n == h->n_glob && (n < 262144 || dense_bytes > 64.0 * 1073741824.0)) ? SLAM_MAPS_TILED : SLAM_MAPS_DENSE;
    }
    if (!rc && h->prm.mode == SLAM_MODE_SLAM && h->prm.map_storage == SLAM_MAPS_TILED)
    {
        h->tiled = true;
        h->map_stride = ((size_t)h->rows * h->cols + 255) / 256 * 256;
        h->tile_tc = (h->cols + 63) / 64;
        h->tile_T = ((h->rows + 63) / 64) * h->tile_tc;
        const int64_t dflt = (3 * n * h->tile_T + 1) / 2 + 1;
        h->tile_cap = h->prm.tile_pool_tiles > 0 ? h->prm.tile_pool_tiles + 1 : dflt;  // + the blank tile 0
        if (h->tile_cap >= (int64_t)slamk::TILE_OWN || n * h->tile_T >= 2147483647LL)
            rc = fail(SLAM_ERR_ARG, "tiled canvases: pool or tables too large");
        TRY(dev_alloc(&h->tile_pool, (size_t)h->tile_cap * slamk::TILE_BYTES));
        TRY(dev_alloc(&h->tile_tab, (size_t)n * h->tile_T));
        TRY(dev_alloc(&h->tile_tab2, (size_t)n * h->tile_T));
        TRY(dev_alloc(&h->tile_free, (size_t)h->tile_cap));
        TRY(dev_alloc(&h->tile_wl_p, (size_t)h->tile_cap));
        TRY(dev_alloc(&h->tile_wl_t, (size_t)h->tile_cap));
        TRY(dev_alloc(&h->tile_rng, 2 * (size_t)n));
        if (h->n_glob == n)
        {  // (one handle holds the whole filter: the resample is local, canvases may be deferred)
            TRY(dev_alloc(&h->surv_stamp, (size_t)n));
            TRY(dev_alloc(&h->surv_list, (size_t)n));
            TRY(dev_alloc(&h->surv_count, 1));
            TRY(dev_alloc(&h->edge_flag, (size_t)n));
            TRY(dev_alloc(&h->edge_list, (size_t)n));
            TRY(dev_alloc(&h->edge_count, 1));
            if (!rc && hipMemset(h->surv_stamp, 0, sizeof(int) * (size_t)n) != hipSuccess) rc = fail(SLAM_ERR_HIP, "hipMemset");
            if (!rc && hipHostMalloc(reinterpret_cast<void**>(&h->surv_count_host), sizeof(int), hipHostMallocDefault) != hipSuccess)
                rc = fail(SLAM_ERR_HIP, "hipHostMalloc");
            if (!rc) *h->surv_count_host = -1;  // no resample yet: the first update runs the ordinary pass
        }
        TRY(dev_alloc(&h->tile_ref, (size_t)h->tile_cap));
        TRY(dev_alloc(&h->tile_who, (size_t)h->tile_cap));
        TRY(dev_alloc(&h->tile_shared, (size_t)h->tile_cap));
        TRY(dev_alloc(&h->tile_ctrl, 4));
        TRY(dev_alloc(&h->tile_dense, h->map_stride));
        TRY(dev_alloc(&h->tile_sector, 256 * slamk::SECTOR_LEVELS));
        if (!rc)
        {
            // slam/mcl.cpp:30: every canvas is the blank tile 0 (128), nothing else is referenced
            bool ok = hipMemsetAsync(h->tile_pool, 128, slamk::TILE_BYTES, h->stream) == hipSuccess &&
                      hipMemsetAsync(h->tile_tab, 0, sizeof(int32_t) * (size_t)n * h->tile_T, h->stream) == hipSuccess &&
                      hipMemsetAsync(h->tile_ctrl, 0, 4 * sizeof(int), h->stream) == hipSuccess &&
                      hipMemsetAsync(h->tile_ref, 0, sizeof(int) * (size_t)h->tile_cap, h->stream) == hipSuccess &&
                      hipMemsetAsync(h->tile_who, 0, sizeof(int) * (size_t)h->tile_cap, h->stream) == hipSuccess &&
                      hipMemsetAsync(h->tile_shared, 0, sizeof(int) * (size_t)h->tile_cap, h->stream) == hipSuccess &&
                      hipMemsetAsync(h->tile_wl_p, 0xff, sizeof(int32_t) * (size_t)h->tile_cap, h->stream) == hipSuccess;
            if (!ok) rc = fail(SLAM_ERR_HIP, "tile pool initialisation");
            TRY(tile_collect(h, nullptr));
        }
    }
    else if (!rc && h->prm.mode == SLAM_MODE_SLAM)
    {
        h->map_stride = ((size_t)h->rows * h->cols + 255) / 256 * 256;
        TRY(dev_alloc(&h->maps, h->map_stride * (size_t)n));
        TRY(dev_alloc(&h->maps2, h->map_stride * (size_t)n));
        // slam/mcl.cpp:30: blank canvas of 128
        if (!rc && hipMemsetAsync(h->maps, 128, h->map_stride * (size_t)n, h->stream) != hipSuccess)
            rc = fail(SLAM_ERR_HIP, "canvas fill");
        if (!rc && hipMemsetAsync(h->maps2, 128, h->map_stride * (size_t)n, h->stream) != hipSuccess)
            rc = fail(SLAM_ERR_HIP, "canvas fill");
    }
    if (!rc)
    {
        // slam/mcl.cpp:22-25,31-36: canvas centre, heading pi/2, uniform weight
        hipLaunchKernelGGL(slamk::k_fill_poses, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, h->stream, n, h->x, h->y,
                           h->th, h->w, h->cols / 2.0, h->rows / 2.0, M_PI / 2, 1.0 / (double)h->prm.global_n);
        rc = finish(h);
    }
#undef TRY
    if (rc)
    {
        std::string keep = g_err;
        slam_mcl_destroy(h);
        g_err = keep;
        return rc;
    }
    *out = h;
    return SLAM_OK;
}

SLAM_API int slam_mcl_set_map_u8(slam_mcl* h, const uint8_t* map, int32_t rows, int32_t cols)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (h->prm.mode != SLAM_MODE_LOCALIZATION) return fail(SLAM_ERR_STATE, "set_map_u8 is for localization mode");
    if (!map || rows != h->rows || cols != h->cols) return fail(SLAM_ERR_ARG, "map null or size differs from create()");
    const size_t bytes = (size_t)rows * cols;
    // the 8 cone-clearance planes with their frames, or ONE Chebyshev plane (a map too large for 32-bit plane offsets;
    // SLAM_CONE_PLANES=0 forces it: tests)
    const bool cone = slam_cone_planes_fit(rows, cols) && tuning().cone_planes;
    const size_t stride = cone ? slam_cone_plane_stride(rows, cols) : 0;
    const size_t pbytes = cone ? 8 * stride : (size_t)(rows + 4) * (size_t)(cols + 4);
    // each buffer under its own test: a call that failed between the two allocations leaves a state the next call completes
    // (the plane layout is the same for every call on a handle: the map size is fixed at create and the tuning is read once)
    if (!h->grid && (rc = dev_alloc(&h->grid, bytes))) return rc;
    if (!h->clr && (rc = dev_alloc(&h->clr, pbytes))) return rc;
    h->clr_stride = stride;  // only once both buffers exist
    h->have_map = false;     // until the planes below are complete
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->grid, map, bytes, hipMemcpyHostToDevice));
    if (cone && tuning().cone_build)
    {
        // the planes from the resident raw map (cone_gpu.h): only the map itself crosses PCIe
        hipError_t e = slam_build_cone_planes_device(h->grid, rows, cols, h->clr, h->clr_stride, h->stream);
        if (e != hipSuccess) return fail(SLAM_ERR_HIP, std::string("cone-clearance planes: ") + hipGetErrorString(e));
    }
    else
    {
        std::vector<uint8_t> E(pbytes);
        if (cone)
            slam_build_cone_clearance_padded(map, rows, cols, E.data());
        else
            slam_build_clearance_padded(map, rows, cols, E.data());
        HIPCHK(hipMemcpy(h->clr, E.data(), pbytes, hipMemcpyHostToDevice));
    }
    h->have_map = true;
    return SLAM_OK;
}

SLAM_API int slam_mcl_get_clearance(slam_mcl* h, uint8_t* out, int64_t capacity_bytes, int32_t* n_planes, int64_t* plane_stride)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (h->prm.mode != SLAM_MODE_LOCALIZATION || !h->have_map) return fail(SLAM_ERR_STATE, "get_clearance: no map set");
    const bool cone = h->clr_stride != 0;
    const int64_t stride = cone ? (int64_t)h->clr_stride : (int64_t)(h->rows + 4) * (int64_t)(h->cols + 4);
    const int planes = cone ? 8 : 1;
    if (n_planes) *n_planes = planes;
    if (plane_stride) *plane_stride = stride;
    if (!out) return SLAM_OK;
    if (capacity_bytes < planes * stride) return fail(SLAM_ERR_ARG, "get_clearance: buffer too small");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, h->clr, (size_t)(planes * stride), hipMemcpyDeviceToHost));
    return SLAM_OK;
}

SLAM_API int slam_mcl_set_poses(slam_mcl* h, const double* x, const double* y, const double* theta, int64_t first,
                                int64_t count)
{
    int rc = check_handle(h);
    if (rc) return rc;
    h->poses_reset = true;
    if (!x || !y || !theta || first < 0 || count < 0 || first + count > h->n) return fail(SLAM_ERR_ARG, "set_poses: bad range");
    for (int64_t k = 0; k < count; ++k)
        if (!std::isfinite(x[k]) || !std::isfinite(y[k]) || !std::isfinite(theta[k]))
            return fail(SLAM_ERR_ARG, "set_poses: non-finite pose");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->x + first, x, sizeof(double) * count, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->y + first, y, sizeof(double) * count, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->th + first, theta, sizeof(double) * count, hipMemcpyHostToDevice));
    return refresh_cloud_box(h);
}

SLAM_API int slam_mcl_set_weights(slam_mcl* h, const double* w, int64_t first, int64_t count)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!w || first < 0 || count < 0 || first + count > h->n) return fail(SLAM_ERR_ARG, "set_weights: bad range");
    // linear weights are likelihoods (>= 0); a SLAM_WEIGHT_LOG handle holds sums of logs in that column (any finite value)
    const bool log_domain = h->prm.weight_domain == SLAM_WEIGHT_LOG;
    for (int64_t k = 0; k < count; ++k)
        if (!std::isfinite(w[k]) || (!log_domain && !(w[k] >= 0)))
            return fail(SLAM_ERR_ARG, log_domain ? "set_weights: log-domain weights must be finite" : "set_weights: weights must be finite and >= 0");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->w + first, w, sizeof(double) * count, hipMemcpyHostToDevice));
    return SLAM_OK;
}

SLAM_API int slam_mcl_seed_gaussian(slam_mcl* h, const double mean[3], const double sigma[3])
{
    int rc = check_handle(h);
    if (rc) return rc;
    h->poses_reset = true;
    if (!mean || !sigma) return fail(SLAM_ERR_ARG, "seed_gaussian: null");
    for (int k = 0; k < 3; ++k)
        if (!std::isfinite(mean[k]) || !std::isfinite(sigma[k]) || sigma[k] < 0) return fail(SLAM_ERR_ARG, "seed_gaussian: bad value");
    hipLaunchKernelGGL(slamk::k_seed_gaussian, dim3((unsigned)cdiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->x, h->y,
                       h->th, h->w, mean[0], mean[1], mean[2], sigma[0], sigma[1], sigma[2], h->prm.seed,
                       (uint64_t)h->prm.global_first, 1.0 / (double)h->prm.global_n);
    if ((rc = refresh_cloud_box(h))) return rc;
    return finish(h);
}

SLAM_API int slam_mcl_seed_uniform_free(slam_mcl* h)
{
    int rc = check_handle(h);
    if (rc) return rc;
    h->poses_reset = true;
    if (h->prm.mode != SLAM_MODE_LOCALIZATION || !h->have_map)
        return fail(SLAM_ERR_STATE, "seed_uniform_free needs localization mode and a map");
    hipLaunchKernelGGL(slamk::k_seed_uniform_free, dim3((unsigned)cdiv(h->n, 256)), dim3(256), 0, h->stream, h->n, h->x, h->y,
                       h->th, h->w, h->grid, h->rows, h->cols, h->prm.seed, (uint64_t)h->prm.global_first,
                       1.0 / (double)h->prm.global_n);
    if ((rc = refresh_cloud_box(h))) return rc;
    return finish(h);
}

SLAM_API int slam_mcl_set_maps_u8(slam_mcl* h, const uint8_t* maps, int64_t first, int64_t count)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (h->prm.mode != SLAM_MODE_SLAM) return fail(SLAM_ERR_STATE, "set_maps_u8 is for SLAM mode");
    if (!maps || first < 0 || count < 0 || first + count > h->n) return fail(SLAM_ERR_ARG, "set_maps_u8: bad range");
    const size_t bytes = (size_t)h->rows * h->cols;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->tiled)
    {
        // drop the old tiles of these particles, then import every canvas as fresh owned tiles
        HIPCHK(hipMemsetAsync(h->tile_tab + (size_t)first * h->tile_T, 0, sizeof(int32_t) * (size_t)count * h->tile_T, h->stream));
        if ((rc = tile_collect(h, nullptr))) return rc;
        for (int64_t k = 0; k < count; ++k)
        {
            HIPCHK(hipMemcpyAsync(h->tile_dense, maps + bytes * (size_t)k, bytes, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(slamk::k_tile_import, dim3((unsigned)h->tile_T), dim3(256), 0, h->stream, h->tile_dense, h->rows,
                               h->cols, h->tile_tab + (size_t)(first + k) * h->tile_T, h->tile_tc, h->tile_pool, h->tile_free,
                               h->tile_ctrl);
            HIPCHK(hipStreamSynchronize(h->stream));  // tile_dense is reused
        }
        return finish(h);
    }
    HIPCHK(hipMemcpy2D(h->maps + h->map_stride * (size_t)first, h->map_stride, maps, bytes, bytes, (size_t)count,
                       hipMemcpyHostToDevice));
    return SLAM_OK;
}

SLAM_API int slam_mcl_set_noise(slam_mcl* h, const double* unit_normals, const double* canonical_uniforms)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (h->prm.noise != SLAM_NOISE_INJECTED) return fail(SLAM_ERR_STATE, "handle was not created with SLAM_NOISE_INJECTED");
    HIPCHK(hipStreamSynchronize(h->stream));
    if (unit_normals)
    {
        for (int64_t k = 0; k < 3 * h->n; ++k)
            if (!std::isfinite(unit_normals[k])) return fail(SLAM_ERR_ARG, "set_noise: non-finite normal");
        HIPCHK(hipMemcpy(h->inj_normals, unit_normals, sizeof(double) * 3 * h->n, hipMemcpyHostToDevice));
        h->have_normals = true;
    }
    if (canonical_uniforms)
    {
        for (int64_t k = 0; k < h->n; ++k)
            if (!(canonical_uniforms[k] >= 0.0 && canonical_uniforms[k] < 1.0))
                return fail(SLAM_ERR_ARG, "set_noise: uniforms must lie in [0,1)");
        HIPCHK(hipMemcpy(h->inj_uniforms, canonical_uniforms, sizeof(double) * h->n, hipMemcpyHostToDevice));
        h->have_uniforms = true;
    }
    return SLAM_OK;
}

SLAM_API int slam_mcl_predict(slam_mcl* h, const double odom[3], const double alphas[4])
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!odom || !alphas) return fail(SLAM_ERR_ARG, "predict: null");
    for (int k = 0; k < 3; ++k)
        if (!std::isfinite(odom[k])) return fail(SLAM_ERR_ARG, "predict: non-finite odometry");
    for (int k = 0; k < 4; ++k)
        if (!std::isfinite(alphas[k]) || alphas[k] < 0) return fail(SLAM_ERR_ARG, "predict: alphas must be finite and >= 0");
    const bool injected = h->prm.noise == SLAM_NOISE_INJECTED;
    if (injected && !h->have_normals) return fail(SLAM_ERR_STATE, "injected noise: no normals set for this predict");
    slamk::PredictArgs a;
    a.n = h->n;
    a.x = h->x;
    a.y = h->y;
    a.th = h->th;
    a.r1 = odom[0];
    a.t = odom[1];
    a.r2 = odom[2];
    // slam/motion.cpp:15-25 (std::pow(v, 2) is v*v)
    a.s1 = std::sqrt(alphas[0] * (a.r1 * a.r1) + alphas[1] * (a.t * a.t));
    a.st = std::sqrt(alphas[2] * (a.t * a.t) + alphas[3] * ((a.r1 * a.r1) + (a.r2 * a.r2)));
    a.s2 = std::sqrt(alphas[0] * (a.r2 * a.r2) + alphas[1] * (a.t * a.t));
    a.seed = h->prm.seed;
    a.step = h->predict_step;
    a.global_first = (uint64_t)h->prm.global_first;
    a.injected = injected ? h->inj_normals : nullptr;
    {
        slam_mcl::Span span(h, G_PREDICT);
        hipLaunchKernelGGL(slamk::k_predict, dim3((unsigned)cdiv(h->n, 256)), dim3(256), 0, h->stream, a);
    }
    HIPCHK(hipGetLastError());
    h->predict_step++;
    h->have_normals = false;
    return maybe_finish(h);
}

SLAM_API int slam_mcl_predict_velocity(slam_mcl* h, const double vel[2], double dt, const double alphas[6])
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!vel || !alphas) return fail(SLAM_ERR_ARG, "predict_velocity: null");
    if (!std::isfinite(vel[0]) || !std::isfinite(vel[1]) || !std::isfinite(dt))
        return fail(SLAM_ERR_ARG, "predict_velocity: non-finite velocity or dt");
    for (int k = 0; k < 6; ++k)
        if (!std::isfinite(alphas[k]) || alphas[k] < 0)
            return fail(SLAM_ERR_ARG, "predict_velocity: alphas must be finite and >= 0");
    const bool injected = h->prm.noise == SLAM_NOISE_INJECTED;
    if (injected && !h->have_normals) return fail(SLAM_ERR_STATE, "injected noise: no normals set for this predict");
    slamk::PredictVelArgs a;
    a.n = h->n;
    a.x = h->x;
    a.y = h->y;
    a.th = h->th;
    a.v = vel[0];
    a.w = vel[1];
    a.dt = dt;
    // slam/motion.cpp:40-45 (std::pow(v, 2) is v*v)
    a.sv = std::sqrt(alphas[0] * (a.v * a.v) + alphas[1] * (a.w * a.w));
    a.sw = std::sqrt(alphas[2] * (a.w * a.w) + alphas[3] * (a.v * a.v));
    a.sg = std::sqrt(alphas[4] * (a.v * a.v) + alphas[5] * (a.w * a.w));
    a.seed = h->prm.seed;
    a.step = h->predict_step;
    a.global_first = (uint64_t)h->prm.global_first;
    a.injected = injected ? h->inj_normals : nullptr;
    {
        slam_mcl::Span span(h, G_PREDICT);
        hipLaunchKernelGGL(slamk::k_predict_velocity, dim3((unsigned)cdiv(h->n, 256)), dim3(256), 0, h->stream, a);
    }
    HIPCHK(hipGetLastError());
    h->predict_step++;
    h->have_normals = false;
    return maybe_finish(h);
}

SLAM_API int slam_mcl_update_weights(slam_mcl* h, const double* angles, const double* dists, int32_t n_beams,
                                     double stddev, double max_dist, const double scanner_offset[3])
{
    int rc = check_handle(h);
    if (rc) return rc;
    if ((rc = upload_scan(h, angles, dists, n_beams))) return rc;
    if ((rc = do_weights(h, n_beams, stddev, max_dist, scanner_offset))) return rc;
    return maybe_finish(h);
}

SLAM_API int slam_mcl_resample(slam_mcl* h)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if ((rc = do_resample(h, nullptr, 1))) return rc;
    return maybe_finish(h);
}

SLAM_API int slam_mcl_shard_resample(slam_mcl* h, const double* gathered_state_dev, int32_t n_shards)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!gathered_state_dev || n_shards < 1) return fail(SLAM_ERR_ARG, "shard_resample: null buffer or n_shards < 1");
    if ((rc = do_resample(h, gathered_state_dev, n_shards))) return rc;
    return maybe_finish(h);
}

SLAM_API int slam_mcl_shard_resample_split(slam_mcl* h, const double* gathered_poses_dev, const double* gathered_weights_dev,
                                           int32_t n_shards)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!gathered_poses_dev || !gathered_weights_dev || n_shards < 1)
        return fail(SLAM_ERR_ARG, "shard_resample_split: null buffer or n_shards < 1");
    if ((rc = do_resample(h, gathered_poses_dev, n_shards, true, gathered_weights_dev))) return rc;
    return maybe_finish(h);
}

SLAM_API int slam_mcl_shard_select(slam_mcl* h, const double* gathered_state_dev, int32_t n_shards)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!gathered_state_dev || n_shards < 1) return fail(SLAM_ERR_ARG, "shard_select: null buffer or n_shards < 1");
    if (h->prm.mode != SLAM_MODE_SLAM) return fail(SLAM_ERR_STATE, "shard_select is the SLAM-mode form of shard_resample");
    if (h->prm.sort != SLAM_SORT_NONE || h->prm.prefix_order != SLAM_PREFIX_BLOCKED)
        return fail(SLAM_ERR_STATE, "sharded resampling needs SLAM_SORT_NONE and SLAM_PREFIX_BLOCKED");
    if ((rc = do_resample(h, gathered_state_dev, n_shards, false))) return rc;
    h->maps_pending = true;
    return maybe_finish(h);
}

SLAM_API int slam_mcl_pack_maps(slam_mcl* h, const int32_t* local_idx_dev, int64_t count, uint8_t* out_dev)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (h->prm.mode != SLAM_MODE_SLAM) return fail(SLAM_ERR_STATE, "pack_maps: SLAM mode only");
    if (count < 0 || count > 65535LL * 32768 || (count > 0 && (!local_idx_dev || !out_dev)))
        return fail(SLAM_ERR_ARG, "pack_maps: bad count or null pointer");
    if (count > 0 && h->tiled)
    {
        // tiled canvases travel as dense ones (row-major, map_stride bytes apart)
        slam_mcl::Span span(h, G_CLONE);
        const int64_t cells = (int64_t)h->rows * h->cols;
        for (int64_t c0 = 0; c0 < count; c0 += 65535)
        {
            const int64_t cn = std::min<int64_t>(65535, count - c0);
            hipLaunchKernelGGL(slamk::k_untile_many, dim3((unsigned)cdiv(cells, 256), (unsigned)cn), dim3(256), 0, h->stream,
                               h->tile_pool, h->tile_tab, h->tile_T, h->tile_tc, h->rows, h->cols, local_idx_dev + c0,
                               h->map_stride, out_dev + h->map_stride * (size_t)c0, h->n, h->dev_bad);
        }
        HIPCHK(hipGetLastError());
    }
    else if (count > 0)
    {
        slam_mcl::Span span(h, G_CLONE);
        launch_clone(h, h->maps, out_dev, local_idx_dev, count, h->n);
        HIPCHK(hipGetLastError());
    }
    h->bad_armed = true;
    return maybe_finish(h);
}

SLAM_API int slam_mcl_shard_clone_maps(slam_mcl* h, const uint8_t* pool_dev, int64_t pool_count, const int32_t* pos_dev)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (h->prm.mode != SLAM_MODE_SLAM) return fail(SLAM_ERR_STATE, "shard_clone_maps: SLAM mode only");
    if (!h->maps_pending) return fail(SLAM_ERR_STATE, "shard_clone_maps: no slam_mcl_shard_select pending");
    if (!pool_dev || !pos_dev || pool_count < 1) return fail(SLAM_ERR_ARG, "shard_clone_maps: null pointer or empty pool");
    h->bad_armed = true;
    if (h->tiled)
    {
        // every tile of the old canvases becomes free, then each slot imports its received canvas (blank
        // tiles map to tile 0); the census then finds all imported tiles exclusively owned
        slam_mcl::Span span(h, G_CLONE);
        HIPCHK(hipMemsetAsync(h->tile_tab, 0, sizeof(int32_t) * (size_t)h->n * h->tile_T, h->stream));
        if ((rc = tile_collect(h, nullptr))) return rc;
        for (int64_t p0 = 0; p0 < h->n; p0 += 65535)
        {
            const int64_t pn = std::min<int64_t>(65535, h->n - p0);
            hipLaunchKernelGGL(slamk::k_tile_import_many, dim3((unsigned)h->tile_T, (unsigned)pn), dim3(256), 0, h->stream, pool_dev,
                               pos_dev + p0, h->map_stride, h->rows, h->cols, h->tile_tab + (size_t)p0 * h->tile_T, h->tile_T,
                               h->tile_tc, h->tile_pool, h->tile_free, h->tile_ctrl, pool_count, h->dev_bad);
        }
        HIPCHK(hipGetLastError());
        if ((rc = tile_collect(h, nullptr))) return rc;
        h->maps_pending = false;
        return maybe_finish(h);
    }
    {
        slam_mcl::Span span(h, G_CLONE);
        launch_clone(h, pool_dev, h->maps2, pos_dev, h->n, pool_count);
        HIPCHK(hipGetLastError());
    }
    std::swap(h->maps, h->maps2);
    h->maps_pending = false;
    return maybe_finish(h);
}

SLAM_API int slam_mcl_pack_tiles_count(slam_mcl* h, const int32_t* local_idx_dev, int64_t count, int64_t* n_tiles_out)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (h->prm.mode != SLAM_MODE_SLAM || !h->tiled) return fail(SLAM_ERR_STATE, "pack_tiles: SLAM mode with tiled canvases only");
    if (!n_tiles_out || count < 0 || (count > 0 && !local_idx_dev) || count * h->tile_T >= (int64_t)256 * 2147483647LL)
        return fail(SLAM_ERR_ARG, "pack_tiles_count: bad count or null pointer");
    if (!h->pack_stamp)
    {
        rc = dev_alloc(&h->pack_stamp, (size_t)h->tile_cap);
        if (!rc) rc = dev_alloc(&h->pack_slot, (size_t)h->tile_cap);
        if (!rc) rc = dev_alloc(&h->pack_ctr, 1);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(h->pack_stamp, 0, sizeof(int) * (size_t)h->tile_cap, h->stream));
    }
    int found = 0;
    if (count > 0)
    {
        slam_mcl::Span span(h, G_CLONE);
        const int64_t total = count * h->tile_T;
        HIPCHK(hipMemsetAsync(h->pack_ctr, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL(slamk::k_pack_mark, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream, h->tile_tab, h->tile_T,
                           local_idx_dev, total, ++h->pack_gen, h->pack_stamp, h->pack_slot, h->tile_wl_t, h->pack_ctr, h->n, h->dev_bad);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&found, h->pack_ctr, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    h->bad_armed = true;
    h->pack_tiles = *n_tiles_out = found;
    h->pack_count = count;
    h->pack_idx = local_idx_dev;
    return SLAM_OK;
}

SLAM_API int slam_mcl_pack_tiles(slam_mcl* h, const int32_t* local_idx_dev, int64_t count, int32_t* tabs_out_dev,
                                 uint8_t* tiles_out_dev, int64_t n_tiles)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (h->prm.mode != SLAM_MODE_SLAM || !h->tiled) return fail(SLAM_ERR_STATE, "pack_tiles: SLAM mode with tiled canvases only");
    if (count != h->pack_count || n_tiles != h->pack_tiles || local_idx_dev != h->pack_idx)
        return fail(SLAM_ERR_STATE, "pack_tiles: call slam_mcl_pack_tiles_count with the same canvases first");
    if ((count > 0 && !tabs_out_dev) || (n_tiles > 0 && !tiles_out_dev)) return fail(SLAM_ERR_ARG, "pack_tiles: null pointer");
    h->pack_count = h->pack_tiles = -1;
    if (count > 0)
    {
        slam_mcl::Span span(h, G_CLONE);
        const int64_t total = count * h->tile_T;
        hipLaunchKernelGGL(slamk::k_pack_tabs, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream, h->tile_tab, h->tile_T,
                           local_idx_dev, total, h->pack_slot, tabs_out_dev, h->n);
        if (n_tiles > 0)
            hipLaunchKernelGGL(slamk::k_tiles_copy, dim3((unsigned)n_tiles), dim3(256), 0, h->stream, h->tile_pool, h->tile_wl_t,
                               tiles_out_dev, (const int32_t*)nullptr);
        HIPCHK(hipGetLastError());
    }
    return maybe_finish(h);
}

SLAM_API int slam_mcl_shift_tables(slam_mcl* h, int32_t* tabs_dev, int64_t n_tabs, int32_t first_tile)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (h->prm.mode != SLAM_MODE_SLAM || !h->tiled) return fail(SLAM_ERR_STATE, "shift_tables: SLAM mode with tiled canvases only");
    if (n_tabs < 0 || (n_tabs > 0 && !tabs_dev) || first_tile < 0) return fail(SLAM_ERR_ARG, "shift_tables: bad argument");
    const int64_t total = n_tabs * h->tile_T;
    if (total > 0 && first_tile > 0)
    {
        hipLaunchKernelGGL(slamk::k_shift_tabs, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream, tabs_dev, total, first_tile);
        HIPCHK(hipGetLastError());
    }
    return SLAM_OK;
}

SLAM_API int slam_mcl_shard_import_tiles(slam_mcl* h, const int32_t* tabs_dev, int64_t n_tabs, const uint8_t* tiles_dev,
                                         int64_t n_tiles, const int32_t* pos_dev)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (h->prm.mode != SLAM_MODE_SLAM || !h->tiled)
        return fail(SLAM_ERR_STATE, "shard_import_tiles: SLAM mode with tiled canvases only");
    if (!h->maps_pending) return fail(SLAM_ERR_STATE, "shard_import_tiles: no slam_mcl_shard_select pending");
    if (!tabs_dev || !pos_dev || n_tabs < 1 || n_tiles < 0 || (n_tiles > 0 && !tiles_dev))
        return fail(SLAM_ERR_ARG, "shard_import_tiles: null pointer or no tables");
    h->bad_armed = true;
    if (n_tiles > h->tile_cap - 1)
        return fail(SLAM_ERR_NOMEM, "shard_import_tiles: " + std::to_string(n_tiles) + " tiles arrive, the pool holds " +
                                        std::to_string(h->tile_cap - 1) + ": recreate the handle with a larger tile_pool_tiles");
    slam_mcl::Span span(h, G_CLONE);
    // every tile of the old canvases becomes free; payload tile q lands in the q-th free tile; slot p
    // takes the table of its source; the census then settles who owns what (clones share)
    HIPCHK(hipMemsetAsync(h->tile_tab, 0, sizeof(int32_t) * (size_t)h->n * h->tile_T, h->stream));
    if ((rc = tile_collect(h, nullptr))) return rc;
    if (n_tiles > 0)
        hipLaunchKernelGGL(slamk::k_tiles_copy, dim3((unsigned)n_tiles), dim3(256), 0, h->stream, tiles_dev, (const int32_t*)nullptr,
                           h->tile_pool, h->tile_free);
    const int64_t total = (int64_t)h->n * h->tile_T;
    hipLaunchKernelGGL(slamk::k_import_tabs, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, h->stream, tabs_dev, pos_dev, h->tile_T,
                       total, (int)n_tiles, h->tile_free, h->tile_tab, n_tabs, h->dev_bad);
    HIPCHK(hipGetLastError());
    // the free list is rebuilt by the census, which reads the tables just written
    if ((rc = tile_collect(h, nullptr))) return rc;
    h->maps_pending = false;
    return maybe_finish(h);
}

SLAM_API int slam_mcl_get_tile_stats(slam_mcl* h, int64_t out[3])
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!out) return fail(SLAM_ERR_ARG, "null");
    if (!h->tiled) return fail(SLAM_ERR_STATE, "get_tile_stats: the handle holds dense canvases");
    HIPCHK(hipStreamSynchronize(h->stream));
    int ctrl[3];
    HIPCHK(hipMemcpy(ctrl, h->tile_ctrl, sizeof(ctrl), hipMemcpyDeviceToHost));
    out[0] = h->tile_cap - 1;
    out[1] = h->tile_cap - 1 - ctrl[1];
    out[2] = ctrl[0] < ctrl[1] ? ctrl[0] : ctrl[1];
    return SLAM_OK;
}

SLAM_API int slam_mcl_get_defer_stats(slam_mcl* h, int64_t out[3])
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!out) return fail(SLAM_ERR_ARG, "null");
    if (!h->tiled) return fail(SLAM_ERR_STATE, "get_defer_stats: the handle holds dense canvases");
    out[0] = h->n_updates;
    out[1] = h->n_deferred;
    out[2] = 0;
    if (h->n_deferred && h->edge_count)
    {
        HIPCHK(hipStreamSynchronize(h->stream));
        int c = 0;
        HIPCHK(hipMemcpy(&c, h->edge_count, sizeof(int), hipMemcpyDeviceToHost));
        out[2] = c;
    }
    return SLAM_OK;
}

SLAM_API int slam_mcl_device_maps_get(slam_mcl* h, slam_mcl_device_maps* out)
{
    if (!h || !out) return fail(SLAM_ERR_ARG, "null");
    out->maps_dev = h->maps;
    out->idx_dev = h->idx;
    out->map_stride = (int64_t)h->map_stride;
    out->n_particles = h->n;
    return SLAM_OK;
}

SLAM_API int slam_mcl_update(slam_mcl* h, const double* angles, const double* dists, int32_t n_beams, double stddev,
                             double max_dist, const double scanner_offset[3])
{
    int rc = check_handle(h);
    if (rc) return rc;
    if ((rc = upload_scan(h, angles, dists, n_beams))) return rc;
    ++h->n_updates;
    if ((rc = do_weights(h, n_beams, stddev, max_dist, scanner_offset, 1))) return rc;  // (resampled right away: canvases may wait)
    if ((rc = do_resample(h, nullptr, 1))) return rc;
    return maybe_finish(h);
}

SLAM_API int slam_mcl_get_poses(slam_mcl* h, double* x, double* y, double* theta, int64_t first, int64_t count)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!x || !y || !theta || first < 0 || count < 0 || first + count > h->n) return fail(SLAM_ERR_ARG, "get_poses: bad range");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(x, h->x + first, sizeof(double) * count, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(y, h->y + first, sizeof(double) * count, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(theta, h->th + first, sizeof(double) * count, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

SLAM_API int slam_mcl_get_weights(slam_mcl* h, double* w, int64_t first, int64_t count)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!w || first < 0 || count < 0 || first + count > h->n) return fail(SLAM_ERR_ARG, "get_weights: bad range");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(w, h->w + first, sizeof(double) * count, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

SLAM_API int slam_mcl_get_map_u8(slam_mcl* h, int64_t particle, uint8_t* map)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (h->prm.mode != SLAM_MODE_SLAM) return fail(SLAM_ERR_STATE, "get_map_u8 is for SLAM mode");
    if (!map || particle < 0 || particle >= h->n) return fail(SLAM_ERR_ARG, "get_map_u8: bad particle");
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->tiled)
    {
        const int64_t cells = (int64_t)h->rows * h->cols;
        hipLaunchKernelGGL(slamk::k_untile, dim3((unsigned)cdiv(cells, 256)), dim3(256), 0, h->stream, h->tile_pool,
                           h->tile_tab + (size_t)particle * h->tile_T, h->tile_tc, h->rows, h->cols, h->tile_dense);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(map, h->tile_dense, (size_t)cells, hipMemcpyDeviceToHost));
        return SLAM_OK;
    }
    HIPCHK(hipMemcpy(map, h->maps + h->map_stride * (size_t)particle, (size_t)h->rows * h->cols, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

SLAM_API int slam_mcl_get_last_indices(slam_mcl* h, int32_t* idx)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!idx) return fail(SLAM_ERR_ARG, "null");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(idx, h->idx, sizeof(int32_t) * h->n, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

SLAM_API int slam_mcl_get_last_order(slam_mcl* h, int32_t* perm)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!perm) return fail(SLAM_ERR_ARG, "null");
    if (h->prm.sort != SLAM_SORT_FULL) return fail(SLAM_ERR_STATE, "handle was created without the sort");
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(perm, h->perm, sizeof(int32_t) * h->n, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

SLAM_API int slam_mcl_get_prefix(slam_mcl* h, double* prefix, double* sum)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if ((rc = complete_prefix(h))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (prefix) HIPCHK(hipMemcpy(prefix, h->lvl_P[0], sizeof(double) * h->n_glob, hipMemcpyDeviceToHost));
    if (sum) HIPCHK(hipMemcpy(sum, h->lvl_P[0] + (h->n_glob - 1), sizeof(double), hipMemcpyDeviceToHost));
    return SLAM_OK;
}

SLAM_API int slam_mcl_record_hits(slam_mcl* h, int32_t enable)
{
    if (!h) return fail(SLAM_ERR_ARG, "null handle");
    h->record_hits = enable != 0;
    return SLAM_OK;
}

SLAM_API int slam_mcl_get_hits(slam_mcl* h, double* hit_xy, int32_t* hit_cell)
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!h->record_hits || !h->hit_xy || h->last_B <= 0) return fail(SLAM_ERR_STATE, "no hits recorded");
    const size_t m = (size_t)h->n * h->last_B;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (hit_xy) HIPCHK(hipMemcpy(hit_xy, h->hit_xy, sizeof(double) * 2 * m, hipMemcpyDeviceToHost));
    if (hit_cell) HIPCHK(hipMemcpy(hit_cell, h->hit_cell, sizeof(int32_t) * m, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

SLAM_API int slam_mcl_average_pose(slam_mcl* h, double out[3])
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!out) return fail(SLAM_ERR_ARG, "null");
    const int64_t nb = cdiv(h->n, 256);
    hipLaunchKernelGGL(slamk::k_pose_partials, dim3((unsigned)nb), dim3(256), 0, h->stream, h->n, h->x, h->y, h->th,
                       h->partials);
    HIPCHK(hipGetLastError());
    std::vector<double> part(4 * (size_t)nb);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(part.data(), h->partials, sizeof(double) * part.size(), hipMemcpyDeviceToHost));
    double s[4] = {0, 0, 0, 0};
    for (int64_t b = 0; b < nb; ++b)
        for (int q = 0; q < 4; ++q) s[q] += part[4 * (size_t)b + q];
    out[0] = s[0] / (double)h->n;  // slam/util.cpp:80-82
    out[1] = s[1] / (double)h->n;
    out[2] = std::atan2(s[3], s[2]);
    return SLAM_OK;
}

SLAM_API int slam_mcl_shard_summary(slam_mcl* h, double out[SLAM_SHARD_SUMMARY_DOUBLES])
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!out) return fail(SLAM_ERR_ARG, "null");
    const int64_t nb = cdiv(h->n, 256);
    hipLaunchKernelGGL(slamk::k_pose_partials, dim3((unsigned)nb), dim3(256), 0, h->stream, h->n, h->x, h->y, h->th,
                       h->partials);
    hipLaunchKernelGGL(slamk::k_best_partials, dim3((unsigned)nb), dim3(256), 0, h->stream, h->n, h->w, h->partials + 4 * (size_t)nb);
    HIPCHK(hipGetLastError());
    std::vector<double> part(6 * (size_t)nb);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(part.data(), h->partials, sizeof(double) * part.size(), hipMemcpyDeviceToHost));
    double s[4] = {0, 0, 0, 0};
    for (int64_t b = 0; b < nb; ++b)
        for (int q = 0; q < 4; ++q) s[q] += part[4 * (size_t)b + q];
    const double* best = part.data() + 4 * (size_t)nb;
    double bw = best[0], bi = best[1];
    for (int64_t b = 1; b < nb; ++b)  // blocks in slot order: a later block wins only with a larger weight
        if (best[2 * b] > bw || (bw != bw && best[2 * b] == best[2 * b]))
        {
            bw = best[2 * b];
            bi = best[2 * b + 1];
        }
    for (int q = 0; q < 4; ++q) out[q] = s[q];
    out[4] = (double)h->n;
    out[5] = bw;
    out[6] = (double)h->prm.global_first + bi;
    out[7] = bi;
    return SLAM_OK;
}

SLAM_API int slam_shard_summary_combine(const double* summaries, int32_t n_shards, double avg_pose[3], double* best_weight,
                                        int64_t* best_global_slot, int32_t* best_shard, int64_t* best_local_slot)
{
    if (!summaries || n_shards < 1) return fail(SLAM_ERR_ARG, "shard_summary_combine: no summaries");
    double s[4] = {0, 0, 0, 0}, count = 0;
    int win = 0;
    for (int r = 0; r < n_shards; ++r)
    {
        const double* q = summaries + (size_t)r * SLAM_SHARD_SUMMARY_DOUBLES;
        for (int k = 0; k < 4; ++k) s[k] += q[k];  // shards in rank order: the same bits on every rank
        count += q[4];
        const double* b = summaries + (size_t)win * SLAM_SHARD_SUMMARY_DOUBLES;
        // (weight descending, global slot ascending): the front of slam/mcl.cpp:236-238's order
        if (q[5] > b[5] || (q[5] == b[5] && q[6] < b[6]) || (b[5] != b[5] && q[5] == q[5])) win = r;
    }
    if (!(count >= 1)) return fail(SLAM_ERR_ARG, "shard_summary_combine: no particles");
    const double* b = summaries + (size_t)win * SLAM_SHARD_SUMMARY_DOUBLES;
    if (avg_pose)
    {
        avg_pose[0] = s[0] / count;  // slam/util.cpp:80-82
        avg_pose[1] = s[1] / count;
        avg_pose[2] = std::atan2(s[3], s[2]);
    }
    if (best_weight) *best_weight = b[5];
    if (best_global_slot) *best_global_slot = (int64_t)b[6];
    if (best_shard) *best_shard = win;
    if (best_local_slot) *best_local_slot = (int64_t)b[7];
    return SLAM_OK;
}

SLAM_API int slam_mcl_collect_stats(slam_mcl* h, int32_t enable)
{
    int rc = check_handle(h);
    if (rc) return rc;
    h->collect_stats = enable != 0;
    return SLAM_OK;
}

SLAM_API int slam_mcl_get_stats(slam_mcl* h, int64_t out[3])
{
    int rc = check_handle(h);
    if (rc) return rc;
    if (!out) return fail(SLAM_ERR_ARG, "null");
    if (!h->collect_stats) return fail(SLAM_ERR_STATE, "get_stats: the counters are switched off (slam_mcl_collect_stats)");
    HIPCHK(hipStreamSynchronize(h->stream));
    unsigned long long t[3];
    HIPCHK(hipMemcpy(t, h->stats, sizeof(t), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) out[k] = (int64_t)t[k];
    return SLAM_OK;
}

SLAM_API int slam_mcl_profile(slam_mcl* h, int32_t enable)
{
    if (!h) return fail(SLAM_ERR_ARG, "null handle");
    h->profile = enable == 2 ? 2 : (enable != 0);
    return SLAM_OK;
}

SLAM_API int slam_mcl_get_kernel_ms(slam_mcl* h, double ms[5], int64_t launches[5], int32_t reset)
{
    int rc = check_handle(h);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    for (auto& s : h->spans)
    {
        float t = 0;
        if (hipEventElapsedTime(&t, s.a, s.b) == hipSuccess)
        {
            h->acc_ms[s.group] += t;
            h->acc_launches[s.group]++;
        }
        hipEventDestroy(s.a);
        hipEventDestroy(s.b);
    }
    h->spans.clear();
    for (int k = 0; k < G_COUNT; ++k)
    {
        if (ms) ms[k] = h->acc_ms[k];
        if (launches) launches[k] = h->acc_launches[k];
        if (reset)
        {
            h->acc_ms[k] = 0;
            h->acc_launches[k] = 0;
        }
    }
    return SLAM_OK;
}

SLAM_API int slam_mcl_synchronize(slam_mcl* h)
{
    int rc = check_handle(h);
    if (rc) return rc;
    return finish(h);
}

SLAM_API int slam_mcl_device_state_get(slam_mcl* h, slam_mcl_device_state* out)
{
    if (!h || !out) return fail(SLAM_ERR_ARG, "null");
    if (int rc = complete_prefix(h)) return rc;  // (enqueued on the handle's stream, like everything that writes these arrays)
    out->state_dev = h->state;
    out->prefix_dev = h->lvl_P[0];
    out->n_particles = h->n;
    out->global_n = h->n_glob;
    return SLAM_OK;
}

// ------------------------------------------------------------------------------------------------
// Resident map for the single-map raycasts (slam/raycast.cpp:7-223, slam/fake_lidar.cpp:15-33): the
// grid is uploaded once, every call then costs one small upload of the rays, one launch and one small
// download.  The stateless entry points below are one-shot wrappers over it.
struct slam_map
{
    int device = 0;
    int rows = 0, cols = 0;
    uint8_t* grid = nullptr;
    uint8_t* luts = nullptr;   // update tables (slam_map_raycast_mapping)
    double* rays = nullptr;    // [x | y | theta | z2 | hit x, hit y], grown on demand
    int64_t ray_cap = 0;
    double* pin = nullptr;     // pinned staging of the same size
    hipStream_t stream = nullptr;
};

namespace
{
int map_reserve(slam_map* m, int64_t n)
{
    if (n <= m->ray_cap) return SLAM_OK;
    if (m->rays) hipFree(m->rays);
    if (m->pin) hipHostFree(m->pin);
    m->rays = nullptr;
    m->pin = nullptr;
    m->ray_cap = 0;
    const int64_t cap = n < 1024 ? 1024 : n;
    int rc = dev_alloc(&m->rays, 6 * (size_t)cap);
    if (rc) return rc;
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&m->pin), sizeof(double) * 6 * (size_t)cap, hipHostMallocDefault));
    m->ray_cap = cap;
    return SLAM_OK;
}

int check_map(slam_map* m)
{
    if (!m) return fail(SLAM_ERR_ARG, "null map handle");
    hipError_t e = hipSetDevice(m->device);
    if (e != hipSuccess) return fail(SLAM_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return SLAM_OK;
}

// rays: n x (x, y, theta) given by fill(k, &x, &y, &theta); z2 optional; kind 1 / 2: read-only, 3: raycast_mapping
template <class FILL>
int map_cast(slam_map* m, int kind, int64_t n, FILL fill, const double* z2, double max2, double step, double* hits)
{
    if (n <= 0 || !hits || !(step > 0) || !std::isfinite(step) || !(max2 >= 0) || !std::isfinite(max2))
        return fail(SLAM_ERR_ARG, "raycast: bad argument");
    int rc = map_reserve(m, n);
    if (rc) return rc;
    double *px = m->pin, *py = m->pin + n, *pth = m->pin + 2 * n, *pz = m->pin + 3 * n;
    for (int64_t k = 0; k < n; ++k)
    {
        fill(k, &px[k], &py[k], &pth[k]);
        pz[k] = z2 ? z2[k] : 0.0;
        if (!std::isfinite(px[k]) || !std::isfinite(py[k]) || !std::isfinite(pth[k]) || !std::isfinite(pz[k]) || pz[k] < 0 ||
            pz[k] > max2)
            return fail(SLAM_ERR_ARG, "raycast: non-finite pose, or a range outside [0, max]");
    }
    HIPCHK(hipMemcpyAsync(m->rays, m->pin, sizeof(double) * 4 * (size_t)n, hipMemcpyHostToDevice, m->stream));
    double* dh = m->rays + 4 * n;
    const double reach = std::fmin(std::sqrt(max2) / step, 1.0e8);
    const int max_iter = (int)std::ceil(reach) + 8;
    if (kind == 1)
        hipLaunchKernelGGL(slamk::k_raycast_ro<1>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, m->stream, m->grid, m->rows, m->cols, n,
                           m->rays, m->rays + n, m->rays + 2 * n, max2, step, max_iter, dh);
    else if (kind == 2)
        hipLaunchKernelGGL(slamk::k_raycast_ro<2>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, m->stream, m->grid, m->rows, m->cols, n,
                           m->rays, m->rays + n, m->rays + 2 * n, max2, step, max_iter, dh);
    else
        hipLaunchKernelGGL(slamk::k_raycast_mapping_seq, dim3(1), dim3(64), 0, m->stream, m->grid, m->rows, m->cols, n, m->rays,
                           m->rays + n, m->rays + 2 * n, m->rays + 3 * n, max2, step, max_iter, m->luts, dh);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(m->pin + 4 * n, dh, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    std::memcpy(hits, m->pin + 4 * n, sizeof(double) * 2 * (size_t)n);
    return SLAM_OK;
}
}  // namespace

SLAM_API void slam_map_destroy(slam_map* m)
{
    if (!m) return;
    hipSetDevice(m->device);
    if (m->stream) hipStreamSynchronize(m->stream);
    if (m->grid) hipFree(m->grid);
    if (m->luts) hipFree(m->luts);
    if (m->rays) hipFree(m->rays);
    if (m->pin) hipHostFree(m->pin);
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
}

SLAM_API int slam_map_upload(slam_map* m, const uint8_t* map)
{
    int rc = check_map(m);
    if (rc) return rc;
    if (!map) return fail(SLAM_ERR_ARG, "map_upload: null");
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipMemcpy(m->grid, map, (size_t)m->rows * m->cols, hipMemcpyHostToDevice));
    return SLAM_OK;
}

SLAM_API int slam_map_download(slam_map* m, uint8_t* map)
{
    int rc = check_map(m);
    if (rc) return rc;
    if (!map) return fail(SLAM_ERR_ARG, "map_download: null");
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipMemcpy(map, m->grid, (size_t)m->rows * m->cols, hipMemcpyDeviceToHost));
    return SLAM_OK;
}

SLAM_API int slam_map_create(int32_t device, const uint8_t* map, int32_t rows, int32_t cols, slam_map** out)
{
    if (!out) return fail(SLAM_ERR_ARG, "map_create: null out");
    *out = nullptr;
    if (!map || rows <= 0 || cols <= 0 || rows > (1 << 20) || cols > (1 << 20) || (int64_t)rows * cols >= 2147483647LL)
        return fail(SLAM_ERR_ARG, "map_create: null map or rows/cols out of range");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(SLAM_ERR_ARG, "device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    slam_map* m = new (std::nothrow) slam_map();
    if (!m) return fail(SLAM_ERR_NOMEM, "host allocation failed");
    m->device = device;
    m->rows = rows;
    m->cols = cols;
    int rc = dev_alloc(&m->grid, (size_t)rows * cols);
    if (!rc) rc = dev_alloc(&m->luts, 512);
    if (!rc && hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) rc = fail(SLAM_ERR_HIP, "hipStreamCreate");
    if (!rc)
    {
        uint8_t t[512];
        build_luts(t);
        if (hipMemcpy(m->luts, t, 512, hipMemcpyHostToDevice) != hipSuccess) rc = fail(SLAM_ERR_HIP, "lut upload");
    }
    if (!rc) rc = slam_map_upload(m, map);
    if (rc)
    {
        std::string keep = g_err;
        slam_map_destroy(m);
        g_err = keep;
        return rc;
    }
    *out = m;
    return SLAM_OK;
}

SLAM_API int slam_map_fake_lidar_scan(slam_map* m, const double* poses3, int32_t n_poses, double start, double stop, double max_dist,
                                      int32_t n_rays, double* hits_xy)
{
    int rc = check_map(m);
    if (rc) return rc;
    if (!poses3 || n_poses <= 0 || n_rays <= 0 || !std::isfinite(start) || !std::isfinite(stop) || !std::isfinite(max_dist))
        return fail(SLAM_ERR_ARG, "fake_lidar_scan: bad argument");
    // slam/fake_lidar.cpp:17-30: theta starts at pose.theta - range/2 and ACCUMULATES by += step
    const double range = stop - start;
    const double step = range / n_rays;
    std::vector<double> th((size_t)n_poses * n_rays);
    for (int p = 0; p < n_poses; ++p)
    {
        double theta = poses3[3 * p + 2];
        theta -= range / 2;
        for (int r = 0; r < n_rays; ++r)
        {
            th[(size_t)p * n_rays + r] = theta;
            theta += step;
        }
    }
    auto fill = [&](int64_t k, double* x, double* y, double* t) {
        const int64_t p = k / n_rays;
        *x = poses3[3 * p];
        *y = poses3[3 * p + 1];
        *t = th[(size_t)k];
    };
    return map_cast(m, 2, (int64_t)n_poses * n_rays, fill, nullptr, max_dist * max_dist, 0.5, hits_xy);
}

SLAM_API int slam_map_raycast_batch(slam_map* m, int32_t kind, const double* poses3, int64_t n, double max_distance_squared,
                                    double step_size, double* hits_xy)
{
    int rc = check_map(m);
    if (rc) return rc;
    if (!poses3 || n <= 0 || (kind != 1 && kind != 2)) return fail(SLAM_ERR_ARG, "raycast_batch: bad argument");
    auto fill = [&](int64_t k, double* x, double* y, double* t) {
        *x = poses3[3 * k];
        *y = poses3[3 * k + 1];
        *t = poses3[3 * k + 2];
    };
    return map_cast(m, kind, n, fill, nullptr, max_distance_squared, step_size, hits_xy);
}

SLAM_API int slam_map_raycast_mapping(slam_map* m, const double* poses3, const double* z_squared, int64_t n, double z_max_squared,
                                      double step_size, double* hits_xy)
{
    int rc = check_map(m);
    if (rc) return rc;
    if (!poses3 || !z_squared || n <= 0) return fail(SLAM_ERR_ARG, "raycast_mapping: bad argument");
    auto fill = [&](int64_t k, double* x, double* y, double* t) {
        *x = poses3[3 * k];
        *y = poses3[3 * k + 1];
        *t = poses3[3 * k + 2];
    };
    return map_cast(m, 3, n, fill, z_squared, z_max_squared, step_size, hits_xy);
}

namespace
{
struct TempMap
{
    slam_map* m = nullptr;
    ~TempMap() { slam_map_destroy(m); }
};
}  // namespace

SLAM_API int slam_fake_lidar_scan(int32_t device, const uint8_t* map01, int32_t rows, int32_t cols, const double* poses3,
                                  int32_t n_poses, double start, double stop, double max_dist, int32_t n_rays,
                                  double* hits_xy)
{
    TempMap t;
    int rc = slam_map_create(device, map01, rows, cols, &t.m);
    if (rc) return rc;
    return slam_map_fake_lidar_scan(t.m, poses3, n_poses, start, stop, max_dist, n_rays, hits_xy);
}

SLAM_API int slam_raycast_u8(int32_t device, const uint8_t* map, int32_t rows, int32_t cols, const double* poses3,
                             int64_t n, double max_distance_squared, double step_size, double* hits_xy)
{
    TempMap t;
    int rc = slam_map_create(device, map, rows, cols, &t.m);
    if (rc) return rc;
    return slam_map_raycast_batch(t.m, 1, poses3, n, max_distance_squared, step_size, hits_xy);
}

SLAM_API int slam_raycast_batch(int32_t device, int32_t kind, const uint8_t* map, int32_t rows, int32_t cols,
                                const double* poses3, int64_t n, double max_distance_squared, double step_size,
                                double* hits_xy)
{
    TempMap t;
    int rc = slam_map_create(device, map, rows, cols, &t.m);
    if (rc) return rc;
    return slam_map_raycast_batch(t.m, kind, poses3, n, max_distance_squared, step_size, hits_xy);
}

// ------------------------------------------------------------------------------------------------
// The entry point that allocates nothing (include/slam_hip.h: slam_resample_select_static).
// The running-sum hierarchy of the selection (levels s / P / tot of launch_blocked_prefix, what
// slam_mcl_create allocates per handle) lives in ONE file-scope device array per GPU; it is held
// until the process ends.  `scratch` itself is only a host-side stand-in: kernels get the address
// hipGetSymbolAddress returned for the current device, never the array's name.
namespace
{
constexpr int64_t kStaticMaxN = SLAM_SELECT_STATIC_MAX;
constexpr size_t static_round(size_t doubles) { return (doubles * sizeof(double) + 255) / 256 * 256; }
constexpr size_t static_level_bytes(int64_t n, bool top)
{
    // s[n], P[n] (the top level's P is its s), tot[ceil(n / 64)]
    return static_round((size_t)n) * (top ? 1 : 2) + static_round((size_t)((n + 63) / 64));
}
constexpr size_t static_total_bytes(int64_t n) { return n <= 64 ? static_level_bytes(n, true) : static_level_bytes(n, false) + static_total_bytes((n + 63) / 64); }
constexpr size_t kStaticBytes = static_total_bytes(kStaticMaxN);
}  // namespace

static __device__ __align__(256) unsigned char scratch[kStaticBytes];

namespace
{
constexpr int kMaxDevices = 64;
unsigned char* g_scratch_ptr[kMaxDevices];  // per device: the address of that device's copy of `scratch`
bool g_scratch_ok[kMaxDevices];

// once per device, with that device current
int static_scratch(int device, unsigned char** out)
{
    if (device < 0 || device >= kMaxDevices) return fail(SLAM_ERR_ARG, "device ordinal out of range");
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!g_scratch_ok[device])
    {
        void* p = nullptr;
        HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(SLAM_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
        g_scratch_ok[device] = true;
    }
    *out = g_scratch_ptr[device];
    return SLAM_OK;
}
}  // namespace

SLAM_API int slam_static_init(int32_t device)
{
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(SLAM_ERR_ARG, "device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    unsigned char* p = nullptr;
    return static_scratch(device, &p);
}

SLAM_API int slam_resample_select_static(int32_t device, void* stream, const double* weights_dev, int64_t n, int32_t resampler,
                                         const double* uniforms_dev, uint64_t seed, uint64_t step, int32_t* idx_dev)
{
    if (!weights_dev || !idx_dev) return fail(SLAM_ERR_ARG, "select_static: null pointer");
    if (n < 1 || n > kStaticMaxN)
        return fail(SLAM_ERR_ARG, "select_static: n must be in [1, " + std::to_string(kStaticMaxN) +
                                      "] (SLAM_SELECT_STATIC_MAX: the size of the static scratch array); use slam_mcl_resample beyond");
    if (resampler != SLAM_RESAMPLE_MULTINOMIAL && resampler != SLAM_RESAMPLE_SYSTEMATIC)
        return fail(SLAM_ERR_ARG, "select_static: bad resampler");
    int rc = slam_static_init(device);
    if (rc) return rc;
    unsigned char* base = nullptr;
    if ((rc = static_scratch(device, &base))) return rc;
    // carve the levels out of the array exactly as static_total_bytes counted them
    slamk::UpperLevels u;
    u.L = 0;
    size_t at = 0;
    for (int64_t nl = n;; nl = cdiv(nl, 64))
    {
        const int l = u.L++;
        const bool top = nl <= 64;
        u.n[l] = nl;
        u.s[l] = reinterpret_cast<double*>(base + at);
        at += static_round((size_t)nl);
        u.P[l] = u.s[l];
        if (!top)
        {
            u.P[l] = reinterpret_cast<double*>(base + at);
            at += static_round((size_t)nl);
        }
        u.tot[l] = reinterpret_cast<double*>(base + at);
        at += static_round((size_t)cdiv(nl, 64));
        if (top) break;
    }
    if (at > kStaticBytes) return fail(SLAM_ERR_STATE, "select_static: scratch layout exceeds the static array (library bug)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    slamk::StateView none = {nullptr, 1, nullptr};
    // (every scratch entry that is read has been written by this call: nothing needs clearing)
    launch_blocked_prefix(st, weights_dev, none, u);
    slamk::SelectArgs s;
    s.s0 = s.tot0 = nullptr;
    s.n1 = 0;
    s.P = u.P[0];
    s.n_total = n;
    s.slot_first = 0;
    s.n_slots = n;
    s.injected = uniforms_dev;
    s.seed = seed;
    s.step = step;
    s.resampler = resampler;
    s.idx = idx_dev;
    s.view = none;
    s.keys = nullptr;
    s.slots = nullptr;
    hipLaunchKernelGGL(slamk::k_select<false>, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, st, s);
    HIPCHK(hipGetLastError());
    return SLAM_OK;
}
