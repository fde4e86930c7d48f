/* slam_hip.h — C ABI of libslam_hip.so, the MI355X (gfx950) particle-filter step.
 *
 * The reference (ferasboulala/slam) has no FFI or plugin layer: its boundary for this path is the
 * statically linked C++ class slam::MCL (slam/mcl.h:12-46) plus the free functions of
 * slam/motion.h:9-11 and slam/raycast.h:13-28.  This header is what a C++ shim of that class (see
 * slam_amd/host/ and INTEGRATION.md) binds instead of the CPU implementation in slam/mcl.cpp.
 * Every entry point cites the reference code it replaces.
 *
 * Conventions: plain C types; caller-allocated buffers; host pointers unless a name ends in
 * _dev; int status return (0 = ok, negative = error, text from slam_last_error()); no exceptions
 * cross the boundary; a handle is used from one thread at a time; every call is complete
 * (synchronised) on return unless documented otherwise.  Particle state stays resident in HBM
 * between calls.
 */
#ifndef SLAM_HIP_H
#define SLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slam_mcl slam_mcl;

enum
{
    SLAM_OK = 0,
    SLAM_ERR_ARG = -1,     /* bad argument (NULL, non-finite pose, size mismatch) */
    SLAM_ERR_HIP = -2,     /* a HIP runtime call failed */
    SLAM_ERR_STATE = -3,   /* call not valid in this mode / before a map was set */
    SLAM_ERR_NOMEM = -4
};

enum { SLAM_MODE_LOCALIZATION = 0, SLAM_MODE_SLAM = 1 };
enum { SLAM_NOISE_PHILOX = 0, SLAM_NOISE_INJECTED = 1 };
enum { SLAM_PREFIX_BLOCKED = 0, SLAM_PREFIX_REFERENCE = 1 };
enum { SLAM_RESAMPLE_MULTINOMIAL = 0, SLAM_RESAMPLE_SYSTEMATIC = 1 };
enum { SLAM_SORT_FULL = 0, SLAM_SORT_NONE = 1 };
enum { SLAM_WEIGHT_LINEAR = 0, SLAM_WEIGHT_LOG = 1 };
enum { SLAM_MAPS_AUTO = 0, SLAM_MAPS_DENSE = 1, SLAM_MAPS_TILED = 2 };
enum
{
    SLAM_RAYCAST_FAST = 0,                /* clearance-skipping search; kernel shape chosen by N */
    SLAM_RAYCAST_WALK = 1,                /* literal cell-by-cell walk (counts cells) */
    SLAM_RAYCAST_FAST_BEAM_LANES = 2,     /* force: one wavefront per particle, lanes over beams */
    SLAM_RAYCAST_FAST_PARTICLE_LANES = 3, /* force: lane = particle, work item = (64 particles, beam segment), summed by a
                                             second kernel in beam order */
    SLAM_RAYCAST_TWO_PHASE = 4,           /* SLAM mode, force: geometry of 64 rays at once (lane = beam), then the rays
                                             in order, 64 samples per pass (what SLAM_RAYCAST_FAST picks in SLAM mode
                                             below 262144 particles and on tiles; 2 forces the older wave-per-ray form) */
    SLAM_RAYCAST_TWO_PHASE_LITERAL = 5,   /* test hook: the two-phase kernel with every batch of beams on its literal
                                             fallback loop (taken by itself only when rounding makes a cell index move by
                                             two in one sample) */
    SLAM_RAYCAST_FAST_PARTICLE_LANES_DIRECT = 6 /* force: lane = particle, work item = 64 particles and the WHOLE scan; the
                                             lane adds its beams' log-likelihoods itself, no scratch slab and no summing
                                             kernel (what SLAM_RAYCAST_FAST picks from 524 288 particles per handle on) */
};

/* Construction parameters.  Zero-initialise, set struct_size = sizeof(slam_mcl_params), fill in.
 * Replaces the constructor arguments of slam::MCL (slam/mcl.h:15, slam/mcl.cpp:16-39). */
typedef struct slam_mcl_params
{
    int32_t struct_size;
    int32_t mode;          /* SLAM_MODE_SLAM: one uint8 canvas per particle, updated along every
                              ray — exactly slam::MCL.  SLAM_MODE_LOCALIZATION: one shared
                              read-only map (SURVEY.md §8a row aL). */
    int64_t n_particles;   /* particles held by THIS handle (one shard) */
    int32_t rows, cols;    /* canvas size (SLAM) or map size (localization) */
    int32_t device;        /* HIP device ordinal */
    int32_t noise;         /* SLAM_NOISE_*: where the normal/uniform streams come from */
    uint64_t seed;         /* Philox key */
    int32_t prefix_order;  /* SLAM_PREFIX_BLOCKED: canonical radix-64 blocked order (parallel,
                              independent of GPU count); SLAM_PREFIX_REFERENCE: the reference's
                              sequential running sum (slam/mcl.cpp:170-176), one thread */
    int32_t resampler;     /* SLAM_RESAMPLE_MULTINOMIAL is the reference's rule */
    int32_t sort;          /* SLAM_SORT_FULL keeps the reference's post-update invariant
                              (slam/mcl.cpp:236-238); order is (weight desc, slot asc) */
    int32_t raycaster;     /* SLAM_RAYCAST_FAST: clearance-skipping walk (localization only);
                              SLAM_RAYCAST_WALK: cell-by-cell walk, counts cells */
    int64_t global_first;  /* index of this shard's first particle in the whole filter */
    int64_t global_n;      /* particles in the whole filter (0 = n_particles) */
    void* stream;          /* hipStream_t to launch on (NULL = the null stream) */
    int32_t async_launch;  /* 0: predict/update* return after the stream has drained (default).
                              1: they return after enqueueing; the caller orders later work on
                              `stream` or calls slam_mcl_synchronize */
    int32_t weight_domain; /* SLAM_WEIGHT_LINEAR: weight = exp(sum_b log w_b), slam/mcl.cpp:67-75 (underflows
                              to 0 for every particle beyond ~434 beams).
                              SLAM_WEIGHT_LOG (opt-in, no reference counterpart): the weight column
                              holds sum_b log w_b and the selection of slam/mcl.cpp:157-203 runs on
                              exp(lw - max lw), max over the whole filter (all shards) */
    int32_t map_storage;   /* SLAM mode; the results do not depend on it.  SLAM_MAPS_DENSE: every particle
                              holds rows*cols bytes and the resample deep-copies them, as
                              slam/mcl.cpp:147-155 does.  SLAM_MAPS_TILED: canvases are tables of 64x64
                              copy-on-write tiles in a shared pool; the resample copies tables, a tile
                              is copied when a particle that shares it is about to write it
                              (SURVEY.md §8f row 3).  SLAM_MAPS_AUTO (0): tiled; dense for shards and
                              for n_particles >= 262144 whose dense canvases fit 64 GiB */
    int32_t reserved;
    int64_t tile_pool_tiles; /* SLAM_MAPS_TILED: tiles (4 KiB each) in the pool; 0 = 1.5 x particles x
                                tiles per canvas, which can never run out */
} slam_mcl_params;

const char* slam_last_error(void);
int slam_version(void);

/* slam::MCL::MCL + reset_particles (slam/mcl.cpp:16-39): SLAM mode starts every particle at the
 * canvas centre (cols/2, rows/2, pi/2), weight 1/N, canvas filled with 128.  Localization mode
 * starts with the same poses and no map (set one, then seed). */
int slam_mcl_create(const slam_mcl_params* params, slam_mcl** out);
void slam_mcl_destroy(slam_mcl* h);

/* Localization mode: the shared occupancy grid, row-major uint8, hit test `< 128` as in
 * slam/raycast.cpp:183-184.  Uploads the map and builds the clearance planes used by SLAM_RAYCAST_FAST
 * on the device (41 ms for a 16384 x 16384 map). */
int slam_mcl_set_map_u8(slam_mcl* h, const uint8_t* map, int32_t rows, int32_t cols);
/* The clearance planes set_map_u8 built (no reference counterpart: inspection and tests): n_planes x plane_stride
 * bytes, each plane (rows + 4) x (cols + 4) with a two-cell frame; 8 cone-clearance planes (one per octant of the ray
 * direction), or 1 Chebyshev plane for maps too large for 32-bit plane offsets.  out == NULL: sizes only. */
int slam_mcl_get_clearance(slam_mcl* h, uint8_t* out, int64_t capacity_bytes, int32_t* n_planes, int64_t* plane_stride);

/* Particle seeding / direct state access (no reference counterpart beyond reset_particles). */
int slam_mcl_set_poses(slam_mcl* h, const double* x, const double* y, const double* theta, int64_t first, int64_t count);
int slam_mcl_seed_gaussian(slam_mcl* h, const double mean[3], const double sigma[3]);
int slam_mcl_seed_uniform_free(slam_mcl* h); /* uniform over the free cells of the map, heading uniform in (-pi, pi] */
int slam_mcl_set_weights(slam_mcl* h, const double* w, int64_t first, int64_t count);
int slam_mcl_set_maps_u8(slam_mcl* h, const uint8_t* maps, int64_t first, int64_t count); /* SLAM mode */

/* Noise injection (SLAM_NOISE_INJECTED): unit normals z[3*n_particles] consumed by the next
 * predict (order rotation_1, translation, rotation_2 per particle: slam/motion.cpp:13-25) and
 * canonical uniforms c[n_particles] consumed by the next update (slam/mcl.cpp:181).  Either
 * pointer may be NULL to leave that stream unchanged. */
int slam_mcl_set_noise(slam_mcl* h, const double* unit_normals, const double* canonical_uniforms);

/* slam::MCL::predict (slam/mcl.cpp:41-47) = sample_motion_model_odometry (slam/motion.cpp:9-32)
 * for every particle.  odom = {rotation_1, translation, rotation_2}. */
int slam_mcl_predict(slam_mcl* h, const double odom[3], const double alphas[4]);

/* sample_motion_model_velocity (slam/motion.cpp:34-56) for every particle: vel = {v, w}.  The
 * reference has no filter-level caller for it (MCL::predict takes Odometry); same noise seam and
 * streams as slam_mcl_predict, draws in the order v, w, gamma. */
int slam_mcl_predict_velocity(slam_mcl* h, const double vel[2], double dt, const double alphas[6]);

/* slam::MCL::update (slam/mcl.cpp:102-145): weights from raycast_mapping + measurement_model_beam
 * (slam/raycast.cpp:143-242, slam/mcl.cpp:49-78), then resample_particles (slam/mcl.cpp:157-239).
 * scan is (angle, dist) per beam as built by the callers (benchmark/mcl.cpp:37-56). */
int slam_mcl_update(slam_mcl* h, const double* angles, const double* dists, int32_t n_beams, double stddev,
                    double max_dist, const double scanner_offset[3]);

/* The two halves of update, for sharded (multi-GPU) drivers and tests. */
int slam_mcl_update_weights(slam_mcl* h, const double* angles, const double* dists, int32_t n_beams, double stddev,
                            double max_dist, const double scanner_offset[3]);
int slam_mcl_resample(slam_mcl* h);

/* Read-back (replaces get_particles(), slam/mcl.h:26). */
int slam_mcl_get_poses(slam_mcl* h, double* x, double* y, double* theta, int64_t first, int64_t count);
int slam_mcl_get_weights(slam_mcl* h, double* w, int64_t first, int64_t count);
int slam_mcl_get_map_u8(slam_mcl* h, int64_t particle, uint8_t* map);      /* SLAM mode */
int slam_mcl_get_last_indices(slam_mcl* h, int32_t* idx);                  /* selection before the sort */
int slam_mcl_get_last_order(slam_mcl* h, int32_t* perm);                   /* sort permutation */
int slam_mcl_get_prefix(slam_mcl* h, double* prefix, double* sum);        /* running sum used by the last resample */
int slam_mcl_get_hits(slam_mcl* h, double* hit_xy, int32_t* hit_cell);    /* per particle-beam, last update_weights;
                                                                              recorded only after slam_mcl_record_hits(h, 1) */
int slam_mcl_record_hits(slam_mcl* h, int32_t enable);
/* slam::average_pose (slam/util.cpp:66-85) reduced on the device. */
int slam_mcl_average_pose(slam_mcl* h, double out[3]);
/* Counters of the last update_weights: {cells read, cells written, steps}; SLAM_RAYCAST_WALK and SLAM mode only. */
int slam_mcl_get_stats(slam_mcl* h, int64_t out[3]);
/* The counters are instrumentation (the reference keeps none) and cost the grid-SLAM kernel about 5 %: on by default,
 * enable = 0 switches them off (slam_mcl_get_stats then fails with SLAM_ERR_STATE), as slam_mcl_record_hits does for hits. */
int slam_mcl_collect_stats(slam_mcl* h, int32_t enable);
/* Accumulated device time (HIP events on the handle's stream) per kernel group since the last
 * reset: out[0] predict, [1] weights (the ray kernels), [2] resample (scan+select+gather), [3] sort, [4] map clone
 * (dense: the deep copy after the selection; tiles: table gather + census, and the copy-on-write copies made just
 * before the next weighting pass);
 * launches[k] = number of launches accumulated.  Enable with slam_mcl_profile(h, 1).  Every bracketed group costs
 * two event records in the stream (about 10 us of idle GPU each); slam_mcl_profile(h, 2) times only out[1], and on
 * the large-N localization path does so with a start event on the first and a stop event on the last kernel of the
 * pass (hipExtLaunchKernel), i.e. without any record between the kernels. */
int slam_mcl_profile(slam_mcl* h, int32_t enable);
int slam_mcl_get_kernel_ms(slam_mcl* h, double ms[5], int64_t launches[5], int32_t reset);

/* Device pointers of the resident state (for RCCL collectives issued by the caller).  The state
 * of a shard is one packed buffer [x | y | theta | weight], n_particles doubles each; the pointer
 * changes after every resample (double buffering), so query it again after each update. */
typedef struct slam_mcl_device_state
{
    double* state_dev;       /* 4 * n_particles doubles */
    double* prefix_dev;      /* global_n inclusive running sums of the last resample before this call (a later resample of
                              * 4097..262144 particles selects without writing them out: call again, or slam_mcl_get_prefix) */
    int64_t n_particles;
    int64_t global_n;
} slam_mcl_device_state;
int slam_mcl_device_state_get(slam_mcl* h, slam_mcl_device_state* out);

/* Sharded resampling for one-process-per-GPU drivers.  Every shard holds
 * n_particles = global_n / n_shards particles (any count: the canonical blocked order runs over the
 * global index, so its 64-chunks may straddle shards).  After update_weights the caller
 * all-gathers the packed state buffers of all shards, in shard order, into gathered_state_dev
 * ([n_shards][4][n_particles] doubles, i.e. the plain concatenation all-gather produces).  This call
 * then builds the running sum of ALL weights in the canonical blocked order, draws this shard's
 * output slots (global slot index = global_first + k) with the rule of slam/mcl.cpp:179-203, and
 * copies the selected particles out of the gathered buffer.  The result is bit-identical to a single
 * handle with n_particles = global_n and SLAM_SORT_NONE, for any number of shards.
 * Requires localization mode, SLAM_SORT_NONE, SLAM_PREFIX_BLOCKED. */
int slam_mcl_shard_resample(slam_mcl* h, const double* gathered_state_dev, int32_t n_shards);
/* The same with the exchange split in two, so that three quarters of it leave the critical path: the
 * poses are final after slam_mcl_predict and can be gathered while the weighting pass runs
 * (gathered_poses_dev = [n_shards][3][n_particles]: the first 3*n doubles of every shard's
 * state_dev); only the weights (gathered_weights_dev = [n_shards][n_particles]: the last n doubles)
 * are gathered after slam_mcl_update_weights.  Same result as slam_mcl_shard_resample. */
int slam_mcl_shard_resample_split(slam_mcl* h, const double* gathered_poses_dev, const double* gathered_weights_dev,
                                  int32_t n_shards);

/* What the caller reads after every step, across shards: slam::average_pose over ALL particles
 * (slam/util.cpp:66-85, called at apps/grid_slam.cpp:119) and get_particles().front(), the heaviest particle
 * (apps/grid_slam.cpp:112; front of the order slam/mcl.cpp:236-238 establishes: weight descending, and the lowest
 * global slot among equals, which is what SLAM_SORT_FULL puts first on one handle).  Sharded handles do not sort
 * (SLAM_SORT_NONE), so: every shard reduces its own particles with slam_mcl_shard_summary
 *   out = { sum x, sum y, sum cos(theta), sum sin(theta), n_particles, best weight, its global slot, its local slot },
 * the driver all-gathers the SLAM_SHARD_SUMMARY_DOUBLES doubles of every shard in rank order (64 B per rank, one
 * collective), and slam_shard_summary_combine (host arithmetic only, the same on every rank) adds the sums in rank
 * order, divides (slam/util.cpp:80-82) and picks the winner.  The shard that owns the winner serves its pose and
 * canvas (slam_mcl_get_poses / slam_mcl_get_map_u8 at best_local_slot).  The sums are floating point: the mean pose
 * agrees with the unsharded slam_mcl_average_pose to rounding (different summation order; tests: 1e-12 relative),
 * the winner is exact.  Any output pointer of the combine call may be NULL. */
#define SLAM_SHARD_SUMMARY_DOUBLES 8
int slam_mcl_shard_summary(slam_mcl* h, double out[SLAM_SHARD_SUMMARY_DOUBLES]);
int slam_shard_summary_combine(const double* summaries /* [n_shards][SLAM_SHARD_SUMMARY_DOUBLES] */, int32_t n_shards,
                               double avg_pose[3], double* best_weight, int64_t* best_global_slot, int32_t* best_shard,
                               int64_t* best_local_slot);

/* Sharded resampling in SLAM mode (per-particle canvases, slam/mcl.cpp:147-155,205-227): the
 * selected particles' canvases live on the shard that owns the source, so the step has an exchange
 * of whole canvases between the two halves:
 *   1. slam_mcl_shard_select: as slam_mcl_shard_resample for poses and weights; the global source
 *      index of every local slot is left in idx_dev (slam_mcl_device_maps_get), canvases untouched.
 *   2. the caller tells every owner which of its canvases are wanted, the owners pack them with
 *      slam_mcl_pack_maps (out_dev[c] = canvas local_idx_dev[c], map_stride bytes each) and the
 *      packed canvases travel (all-to-all over RCCL, slam_amd/dist.py);
 *   3. slam_mcl_shard_clone_maps: canvas of local slot k = pool_dev[pos_dev[k]] (pool_count canvases of
 *      map_stride bytes), which completes the resample.
 * The index arrays live on the device, so the library cannot check them before launching: an entry
 * out of range is skipped by the kernels and the next synchronising call returns SLAM_ERR_ARG.
 * predict / update_weights before step 3 would see stale canvases; update_weights refuses. */
int slam_mcl_shard_select(slam_mcl* h, const double* gathered_state_dev, int32_t n_shards);
int slam_mcl_pack_maps(slam_mcl* h, const int32_t* local_idx_dev, int64_t count, uint8_t* out_dev);
int slam_mcl_shard_clone_maps(slam_mcl* h, const uint8_t* pool_dev, int64_t pool_count, const int32_t* pos_dev);
/* The same exchange for SLAM_MAPS_TILED handles.  slam_mcl_pack_maps / slam_mcl_shard_clone_maps work
 * on them as well (canvases then travel dense; blank tiles stay shared on arrival), but a tiled
 * shard normally holds far more canvases than would fit dense, so steps 2 and 3 have a sparse
 * form: a requested canvas travels as its table (SLAM_TILES_PER_CANVAS(rows, cols) int32: 0 = blank
 * tile, k > 0 = the k-th tile of the payload) and every distinct SLAM_TILE_BYTES tile those tables
 * name travels once, copy-on-write sharing between the requested canvases included.
 *   2'. slam_mcl_pack_tiles_count (how many tiles the request needs: the caller sizes the buffers),
 *       then slam_mcl_pack_tiles with the same arguments: tabs_out_dev[c] = table of canvas
 *       local_idx_dev[c], tiles_out_dev = the n_tiles payload tiles;
 *   3'. slam_mcl_shard_import_tiles: tabs_dev/tiles_dev = the n_tabs received tables and n_tiles tiles of all owners
 *       (the caller has shifted each owner's table entries by the position of that owner's first tile
 *       in tiles_dev); canvas of local slot k = table pos_dev[k].  Slots cloned from one source share
 *       its tiles until they write.  SLAM_ERR_NOMEM if the pool is smaller than n_tiles.
 * No reference counterpart beyond the resample loop cited above: the reference keeps whole canvases. */
#define SLAM_TILE_SIDE 64
#define SLAM_TILE_BYTES 4096
#define SLAM_TILES_PER_CANVAS(rows, cols) ((((rows) + 63) / 64) * (((cols) + 63) / 64))
int slam_mcl_pack_tiles_count(slam_mcl* h, const int32_t* local_idx_dev, int64_t count, int64_t* n_tiles_out);
int slam_mcl_pack_tiles(slam_mcl* h, const int32_t* local_idx_dev, int64_t count, int32_t* tabs_out_dev,
                        uint8_t* tiles_out_dev, int64_t n_tiles);
int slam_mcl_shard_import_tiles(slam_mcl* h, const int32_t* tabs_dev, int64_t n_tabs, const uint8_t* tiles_dev,
                                int64_t n_tiles, const int32_t* pos_dev);
/* The shift of step 3' on the device, for drivers without a tensor library (slam::ShardedMCL): every entry > 0 of the
 * n_tabs tables at tabs_dev (one owner's answer) grows by first_tile, the position of that owner's first tile in the
 * concatenated payload.  Enqueued on the handle's stream. */
int slam_mcl_shift_tables(slam_mcl* h, int32_t* tabs_dev, int64_t n_tabs, int32_t first_tile);
typedef struct slam_mcl_device_maps
{
    uint8_t* maps_dev;       /* n_particles canvases, map_stride bytes apart (rows*cols valid bytes each); NULL if tiled */
    const int32_t* idx_dev;  /* global source index per local slot, from the last select */
    int64_t map_stride;
    int64_t n_particles;
} slam_mcl_device_maps;
int slam_mcl_device_maps_get(slam_mcl* h, slam_mcl_device_maps* out);
/* SLAM_MAPS_TILED: {tiles in the pool, tiles referenced after the last resample (the blank tile
 * excluded), tiles handed out since}.  A pool that runs dry makes the call that notices it return
 * SLAM_ERR_NOMEM (canvases are then incomplete: recreate the handle with a larger pool). */
int slam_mcl_get_tile_stats(slam_mcl* h, int64_t out[3]);
/* SLAM_MAPS_TILED: {slam_mcl_update calls so far, how many of them left their canvases for the resample that follows
 * ("weights now, canvases of the survivors later": DESIGN.md), particles of the last such update that took the ordinary
 * pass at once because their sensor was too close to row 0 / column 0}.  Results do not depend on which way an update
 * went; the numbers are for tests and benchmarks.  No counterpart in the reference (slam/mcl.cpp:102-144 always draws). */
int slam_mcl_get_defer_stats(slam_mcl* h, int64_t out[3]);
int slam_mcl_synchronize(slam_mcl* h);

/* FakeLidar::scan (slam/fake_lidar.cpp:15-33) + raycast<int> (slam/raycast.cpp:100-141), batched
 * on the device: n_poses sensor poses x n_rays rays on a CV_32S-style map given as uint8 (0 = wall).
 * hits_xy[(p*n_rays + r)*2 ..] = hit position or (-1,-1). */
int slam_fake_lidar_scan(int32_t device, const uint8_t* map01, int32_t rows, int32_t cols, const double* poses3,
                         int32_t n_poses, double start, double stop, double max_dist, int32_t n_rays, double* hits_xy);

/* raycast<unsigned char> (slam/raycast.cpp:53-98), batched: n rays (x, y, theta). */
int slam_raycast_u8(int32_t device, const uint8_t* map, int32_t rows, int32_t cols, const double* poses3, int64_t n,
                    double max_distance_squared, double step_size, double* hits_xy);

/* Any of the read-only raycasts of slam/raycast.cpp:7-141, batched: kind 1 = unsigned char map
 * (tests distance, bounds, then `< 128`; a double map thresholded at 0.5 by the caller follows the
 * same order), kind 2 = int map given as uint8 0/1 (tests bounds, distance, then `== 0`). */
int slam_raycast_batch(int32_t device, int32_t kind, const uint8_t* map, int32_t rows, int32_t cols, const double* poses3,
                       int64_t n, double max_distance_squared, double step_size, double* hits_xy);

/* Resident single-map handle for the raycasts above (and for raycast_mapping on ONE canvas): the grid
 * is uploaded once (slam_map_create / slam_map_upload), every call afterwards costs one small upload
 * of the rays, one launch and one small download.  The three stateless calls above are one-shot
 * wrappers over it.  FakeLidar::scan / raycast<T> of the C++ shim keep a handle per map
 * (slam/fake_lidar.cpp:15-33, slam/raycast.cpp:7-141). */
typedef struct slam_map slam_map;
int slam_map_create(int32_t device, const uint8_t* map, int32_t rows, int32_t cols, slam_map** out);
void slam_map_destroy(slam_map* m);
int slam_map_upload(slam_map* m, const uint8_t* map);    /* replace the grid (same size) */
int slam_map_download(slam_map* m, uint8_t* map);        /* the grid as slam_map_raycast_mapping left it */
int slam_map_fake_lidar_scan(slam_map* m, const double* poses3, int32_t n_poses, double start, double stop, double max_dist,
                             int32_t n_rays, double* hits_xy);
int slam_map_raycast_batch(slam_map* m, int32_t kind, const double* poses3, int64_t n, double max_distance_squared,
                           double step_size, double* hits_xy);
/* slam::raycast_mapping (slam/raycast.cpp:143-223) for n rays (x, y, theta; measured range squared
 * z_squared[k] <= z_max_squared) applied IN ORDER to the resident uint8 canvas: hit search and
 * occupancy update exactly as the filter's own kernels do them; hits_xy[2k..] = hit or (-1,-1).
 * measurement_model_beam (slam/raycast.cpp:225-242) is this plus two host scalars (the shim). */
int slam_map_raycast_mapping(slam_map* m, const double* poses3, const double* z_squared, int64_t n, double z_max_squared,
                             double step_size, double* hits_xy);

/* The one entry point that allocates nothing.  It has NO counterpart in the reference; it exists for
 * callers that cannot call hipMalloc: code under HIP graph capture, tight serving loops, harnesses
 * that forbid allocation.  It is the selection half of slam_mcl_resample (slam/mcl.cpp:157-203: running
 * sum of the weights in the canonical blocked order of SLAM_PREFIX_BLOCKED, then the reference's
 * bisection per output slot) as a stateless call on device buffers: weights_dev[n] in (finite, >= 0:
 * not checked, they live on the device), idx_dev[n] out; identical indices to a handle of n particles
 * given the same weights and uniforms.  uniforms_dev: n canonical uniforms in [0,1) (systematic: one),
 * or NULL for the Philox stream (seed, step, slot) a handle would draw.  The running-sum levels live in
 * a static device array of the library (about 4.3 MB per GPU, held until the process ends), so
 * n <= SLAM_SELECT_STATIC_MAX; larger n is rejected with SLAM_ERR_ARG.  The call only enqueues on
 * `stream` (no synchronisation, no hipMalloc/hipFree); every call on one GPU shares that GPU's array,
 * so calls that overlap on one GPU (different streams) race: order them.  slam_static_init(device)
 * does the one-time per-device set-up (looks up the array's address) ahead of time, e.g. before a
 * graph capture; otherwise the first call on a device does it. */
#define SLAM_SELECT_STATIC_MAX 262144
int slam_static_init(int32_t device);
int slam_resample_select_static(int32_t device, void* stream, const double* weights_dev, int64_t n, int32_t resampler,
                                const double* uniforms_dev, uint64_t seed, uint64_t step, int32_t* idx_dev);

#ifdef __cplusplus
}
#endif
#endif /* SLAM_HIP_H */
