/*
 * ufomap_hip.h -- C ABI of the MI355X-native scan-integration path for UFOMap.
 *
 * The reference (UnknownFreeOccupied/ufomap v1) has no plugin/FFI interface: its boundary is the
 * C++ member-template surface of ufo::map::OccupancyMapBase<DATA_TYPE>.  This header is the thin
 * C-ABI layer the host-side mirrors (include/ufomap_amd/occupancy_map.hpp for C++,
 * ufomap_amd/occupancy_map.py for Python) forward to; each entry point cites the reference
 * interface it replaces (paths relative to ufomap/include/ufo/map/ in the reference tree).
 *
 * Conventions: plain pointers and sizes only; every function that can fail returns int
 * (0 = UFOMAP_OK, <0 = error) and leaves a message for ufomap_last_error(); nothing throws.
 * The reference's hot path has no error reporting at all, so UFOMAP_OK is the only
 * outcome a reference-valid call can produce.  A map handle owns four HIP streams (cloud upload + first
 * kernel of a scan, scan half, tree update, read-back); with async != 0 up to two tree updates may be in flight while the next scan is cast -- the
 * reference has one `integrate_` future (occupancy_map_base.h:315, 405, 1553) and overlaps only its head loop
 * with it; results are identical either way.  Not thread-safe per handle (neither is the reference).
 *
 * There is NO CPU fallback behind this ABI: without a HIP device every call fails loudly.
 *
 * Environment: UFOMAP_NO_GATES=1 -- hand-overs between the handle's streams through HIP events instead of one-wave gate
 * kernels (chosen automatically under tools that serialise kernels across streams: rocprofv3 --pmc, rocprof v1/v2,
 * AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING); UFOMAP_RCCL_LIB -- name of the RCCL library for ufomap_comm_*;
 * UFOMAP_MERGE_PHASES -- see ufomap_map_set_option. Results never depend on them.
 */
#ifndef UFOMAP_HIP_H
#define UFOMAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UFOMAP_OK 0
#define UFOMAP_ERR_INVALID -1     /* bad argument (e.g. depth_levels outside [2,21]: octree.h:931-935) */
#define UFOMAP_ERR_DEVICE -2      /* HIP runtime error / no device */
#define UFOMAP_ERR_UNSUPPORTED -3 /* what the reference itself does not compile (a coloured cloud in continuous mode), early_stopping > 0 on
                                    * a ray box whose first-ray array exceeds the scratch limit, insert depth > 0 in a multi-GPU step */
#define UFOMAP_ERR_RUNAWAY -4     /* a clipped ray left the map cube (reference walks ~2^31 cells); map unchanged */
#define UFOMAP_ERR_CAPACITY -5    /* scan grid or node table exceeded the configured memory limit */

typedef struct ufomap_map ufomap_map;

/* ---- library ---------------------------------------------------------------------------- */
const char* ufomap_last_error(void);
int ufomap_device_count(void);
const char* ufomap_version(void);

/* ---- map lifetime: OccupancyMap / OccupancyMapColor constructors
 *      (occupancy_map.h:60-63, occupancy_map_color.h:60-64; defaults occupancy_map_base.h:859-862).
 *      has_color selects OccupancyMapColor.  device = HIP device ordinal. Returns NULL on error. */
ufomap_map* ufomap_map_create(double resolution, unsigned depth_levels, int automatic_pruning,
                              double occupied_thres, double free_thres, double prob_hit,
                              double prob_miss, double clamping_thres_min,
                              double clamping_thres_max, int has_color, int device);
void ufomap_map_destroy(ufomap_map* m);
/* Octree::clear() (octree.h:541): back to a single unknown root leaf. */
int ufomap_map_clear(ufomap_map* m);
/* Pre-size the node table for about n 8-child node blocks: a HINT (optional; the table grows on demand). The table is tile-major --
 * 73 slots per depth-3 tile that holds anything -- so what n blocks need depends on how dense the map is inside its tiles: the call
 * sizes for n / 40 tile groups and n / 200 blocks above depth 3, which fits a dense map (a 2 mm RGB-D frame: 47 live blocks per
 * group); a sparse one (LiDAR: ~10 per group) may still see a re-hash later. */
int ufomap_map_reserve(ufomap_map* m, size_t n_blocks);
/* Upper bound in bytes for a scan's dense dedup grid (default 16 GiB). A scan whose bounding box needs more keeps its ray
 * cells in a sparse set of node blocks instead (bounded by the cells the rays touch, as the reference's CodeMap is,
 * code.h:568-785; slower per step); the set itself must fit the limit too. The limit bounds the scratch of ONE scan: scratch buffers
 * belong to the hand-over set of the scan that uses them, grow-only, and a handle fed asynchronously keeps two to three sets busy (up
 * to eight exist) -- e.g. the volume path's brick grids of a 2 mm RGB-D frame are 4.5 GB per set, 9 GB for a pipelined stream of such
 * frames. They are released with the handle (ufomap_map_destroy). */
int ufomap_map_set_scratch_limit(ufomap_map* m, size_t bytes);

/* ---- sensor model setters (occupancy_map_base.h:746-773), probabilities, not log-odds ------- */
int ufomap_map_set_sensor_model(ufomap_map* m, double occupied_thres, double free_thres,
                                double prob_hit, double prob_miss, double clamping_thres_min,
                                double clamping_thres_max);

/* ---- THE HOT PATH ---------------------------------------------------------------------------
 * insertPointCloud          (discrete=0: occupancy_map_base.h:270-327)
 * insertPointCloudDiscrete  (discrete=1: occupancy_map_base.h:340-417;
 *                            with rgb != NULL the colour overload occupancy_map_color.h:177-267)
 * xyz: n x 3 float64 (AoS, like std::vector<Point3>), rgb: n x 3 uint8 or NULL.
 * max_range < 0 = unlimited; depth = DepthType at which free space is cleared;
 * async != 0 returns after enqueueing (the reference's std::async, occupancy_map_base.h:318);
 * inputs are copied/consumed before the call returns either way (SURVEY.md 8b ownership).
 * The call first joins any previous integration (occupancy_map_base.h:315).
 * ufomap_map_insert takes HOST pointers (H2D copy included); ufomap_map_insert_device takes
 * DEVICE pointers already resident in HBM. Either way the caller's buffers are not referenced after the call has
 * returned, asynchronous or not: the steady-state path reads a device cloud with its first kernel only, keeps the
 * points it needs for a possible repeat of the scan, and the call returns when that kernel has run (normally before
 * the rest of the scan has been enqueued); the general path works on a device-to-device copy. */
int ufomap_map_insert(ufomap_map* m, const double sensor_origin[3], const double* xyz,
                      const uint8_t* rgb, size_t n, double max_range, unsigned depth, int discrete,
                      int simple_ray_casting, unsigned early_stopping, int async);
int ufomap_map_insert_device(ufomap_map* m, const double sensor_origin[3], const double* d_xyz,
                             const uint8_t* d_rgb, size_t n, double max_range, unsigned depth,
                             int discrete, int simple_ray_casting, unsigned early_stopping,
                             int async);
/* insertPointCloudWait / insertPointCloudDone (occupancy_map_base.h:430-443).
 * wait returns the status of the joined integration; done returns 1/0 (or <0 on error). */
/* The three lines in front of the hot path in the reference's server (ufomap_mapping/src/server.cpp:114-120):
 *   rosToUfo(msg, cloud)            ufomap_ros/src/conversions.cpp:98-138   float32 fields -> Point3Color, NaN dropped
 *   cloud.transform(transform)      map/point_cloud.h:157-166, math/pose6.h:114-125, math/quaternion.h:253-286
 *   map.insertPointCloudDiscrete(transform.translation(), cloud, ...)
 * on the raw records of a sensor_msgs/PointCloud2: `data` = n_points records of point_step bytes (host memory, or
 * device memory if data_on_device), float32 x / y / z at off_x / off_y / off_z, bytes r / g / b at off_r / off_g /
 * off_b (all three -1: no colour; the sub-bytes of a packed "rgb" field at offset o are b = o, g = o+1, r = o+2).
 * Conversion, NaN filter and transform run inside the first kernel of the scan, in the reference's operation
 * order (same keys, bit for bit); the float64 cloud is never materialised. discrete = 0 gives insertPointCloud. */
int ufomap_map_insert_pointcloud2(ufomap_map* m, const double translation[3], const double rotation_wxyz[4], const void* data,
                                  int data_on_device, size_t n_points, uint32_t point_step, int off_x, int off_y, int off_z,
                                  int off_r, int off_g, int off_b, double max_range, unsigned depth, int discrete,
                                  int simple_ray_casting, unsigned early_stopping, int async);

/* OccupancyMapBase::setValueVolume(ufo::geometry::AABB(aabb_min, aabb_max), occupancy_value, min_depth)
 * (occupancy_map_base.h:492-518, 986-1031): every node of depth min_depth (voxel for 0) whose box intersects the
 * volume gets clamp(toLogit(occupancy_value)), expanded nodes among them lose their subtrees, ancestors are
 * re-evaluated and pruned exactly as the reference's recursion does. The server calls it after every scan to
 * clear the robot's own volume (ufomap_mapping/src/server.cpp:137-168). AABB volumes only. */
int ufomap_map_set_value_volume(ufomap_map* m, const double aabb_min[3], const double aabb_max[3], double occupancy_value,
                                unsigned min_depth);
/* Batch of point queries: per coordinate (n x 3 doubles, host memory or device memory if xyz_on_device) at `depth`
 *   logodds[i]  occupancy log-odds of the node Octree::getNode(toCode(coord, depth)) returns (map/octree.h:974-985);
 *               getOccupancy(coord, depth) is toProb of it (occupancy_map_base.h:599-613)
 *   state[i]    bit 0 occupied | bit 1 free | bit 2 unknown = getState (619-634, hence isOccupied/isFree/isUnknown,
 *               637-680; containsOccupied = isOccupied, 686-691), bit 3 containsFree, bit 4 containsUnknown (693-728)
 * getNode's loop stops one level early (the node of depth + 1 answers for a fully expanded path); this is
 * reproduced, so every answer equals the reference's. Outputs are host arrays. */
int ufomap_map_query(ufomap_map* m, const double* xyz, int xyz_on_device, size_t n, unsigned depth, float* logodds, uint8_t* state);
/* getClampingThresMin() / getClampingThresMax() (occupancy_map_base.h:742-744), the value the server passes */
int ufomap_map_clamping_thres(ufomap_map* m, double* thres_min, double* thres_max);

int ufomap_map_wait(ufomap_map* m);
int ufomap_map_done(ufomap_map* m);

/* ---- what the reference's callers use around the hot path (ufomap_mapping/src/server.cpp) ------------------------------
 * Bounding volumes are AABBs given as (centre[3], half_size[3]) -- the members of ufo::geometry::AABB (geometry/aabb.h:
 * 50-72), which the reference's intersection tests read directly; NULL pointers = no bounding volume. */

/* Octree::clear(new_resolution, new_depth_levels) (octree.h:544-575): back to a single unknown root at a new geometry. */
int ufomap_map_clear_to(ufomap_map* m, double resolution, unsigned depth_levels);
/* getOccupiedThres / getFreeThres / getProbHit / getProbMiss / getClampingThresMin / getClampingThresMax
 * (occupancy_map_base.h:734-744): toProb(LogitType = float) of the stored logits, in this order. */
int ufomap_map_get_sensor_model(ufomap_map* m, double out[6]);
/* setProbHit (which = 2) / setProbMiss (3) / setClampingThresMin (4) / setClampingThresMax (5)
 * (occupancy_map_base.h:761-773; server.cpp:468-471): stores toLogit(probability); the tree is not touched. */
int ufomap_map_set_model_value(ufomap_map* m, int which, double probability);
/* setOccupiedFreeThres (occupancy_map_base.h:746-759): the reference writes the tree to a stream, changes the
 * thresholds and reads it back, so that every inner node's contains_free / contains_unknown (and its max, and the
 * pruning) is re-evaluated; done the same way here, on the device (ufomap_map_write_ex + ufomap_map_read_data). */
int ufomap_map_set_occupied_free_thres(ufomap_map* m, double occupied_thres, double free_thres);
/* setValueVolume with the AABB as the reference holds it (centre, half size): what a BoundingVar carries. */
int ufomap_map_set_value_volume_ch(ufomap_map* m, const double aabb_center[3], const double aabb_half[3], double occupancy_value,
                                   unsigned min_depth);

/* Change detection (occupancy_map_base.h:779-791): while enabled, every leaf update that changes a value records its
 * code (occupancy_map_base.h:1070-1072, 1094-1108; occupancy_map_color.h:278-280) in a set. changes() returns the set
 * as (code >> 3*depth, depth) sorted by (depth, code); the total is returned, at most cap entries are written. */
int ufomap_map_enable_change_detection(ufomap_map* m, int enable);
int ufomap_map_reset_change_detection(ufomap_map* m);
size_t ufomap_map_changes(ufomap_map* m, uint64_t* codes, uint8_t* depths, size_t cap);
/* enableMinMaxChangeDetection (occupancy_map_base.h:791-797): the change AABB only grows while enabled (default here:
 * enabled, as the server sets it, server.cpp:74). */
int ufomap_map_enable_minmax_change_detection(ufomap_map* m, int enable);

/* beginLeaves / beginTree (occupancy_map_base.h:93-165) run to the end: the nodes OccupancyMapIterator
 * (iterator/occupancy_map.h:168-208 over iterator/octree.h:186-300) returns for a bounding volume, the three state
 * switches, `contains` and min_depth, in the iterator's (pre-order) sequence. Evaluated level by level on the
 * device; only the matching nodes are copied back. flags: bit 0 contains_free, bit 1 contains_unknown, bit 2 the
 * node is a leaf. Returns the total (may exceed cap); (size_t)-1 on error. Any output pointer may be NULL. */
size_t ufomap_map_iterate(ufomap_map* m, const double* aabb_center, const double* aabb_half, int occupied_space, int free_space,
                          int unknown_space, int contains, unsigned min_depth, int only_leaves, uint64_t* codes, uint8_t* depths,
                          float* logodds, uint8_t* rgb, uint8_t* flags, size_t cap);

/* Octree::write / writeData with all their arguments (octree.h:779-917; occupancy_map_base.h:1457-1533): bounding
 * volume, LZ4 compression (LZ4_compress_fast with acceleration_level, or LZ4_compress_HC when compression_level > 0:
 * octree.h:1430-1458; liblz4 is loaded at run time), min_depth (nodes at min_depth are written as leaves). header != 0:
 * write() = text header + data; header == 0: writeData() = data only, which is what ufomap_msgs::ufoToMsg puts
 * into a UFOMap message (ufomap_msgs/conversions.h:162-186; server.cpp:200, 305, 333). *uncompressed_size (may be
 * NULL) receives writeData's return value. Returns the number of bytes, written only if cap is large enough. */
size_t ufomap_map_write_ex(ufomap_map* m, const double* aabb_center, const double* aabb_half, int compress, unsigned min_depth,
                           int compression_acceleration_level, int compression_level, int header, uint8_t* buf, size_t cap,
                           long long* uncompressed_size);
/* Octree::read(std::istream&) (octree.h:701-735): header + data as write() produces them; the map takes the file's
 * resolution and depth_levels (returned through the two pointers, which may be NULL). */
int ufomap_map_read(ufomap_map* m, const uint8_t* buf, size_t n, double* resolution, unsigned* depth_levels);
/* Octree::readData (octree.h:737-777) -> readNodes (occupancy_map_base.h:1379-1455): the node stream of a UFOMap
 * message / file merged into the tree -- nodes inside the bounding volume are replaced by the stream's, their
 * ancestors re-evaluated (updateNode, pruning). */
int ufomap_map_read_data(ufomap_map* m, const uint8_t* data, size_t n, const double* aabb_center, const double* aabb_half,
                         double resolution, unsigned depth_levels, int uncompressed_data_size, int compressed);


/* ---- read-back (what the reference exposes through beginLeaves()/beginTree(),
 *      occupancy_map_base.h:93-137; iterator/octree.h:133-158) -------------------------------
 * Canonical dump: node = (code >> 3*depth, depth); output sorted by (depth, code).
 * Return the total count (may exceed cap; only cap entries are written); (size_t)-1 on error.
 * Any output pointer may be NULL.  rgb: 3 bytes per node (zeros for a non-colour map).
 * flags: bit0 contains_free, bit1 contains_unknown (occupancy_map_node.h:171-176). */
size_t ufomap_map_export_leaves(ufomap_map* m, int include_unknown, uint64_t* codes,
                                uint8_t* depths, float* logodds, uint8_t* rgb, size_t cap);
size_t ufomap_map_export_inner(ufomap_map* m, uint64_t* codes, uint8_t* depths, float* logodds,
                               uint8_t* flags, uint8_t* rgb, size_t cap);

/* Order-independent fingerprint of the two dumps above, computed on the device in one pass (no export, no sort):
 * per record h = mix64(mix64(code >> 3*depth | depth << 58) ^ (float32 bits | rgb24 << 32 | flags << 56)), mix64 =
 * the splitmix64 finaliser; out[0..2] = number of leaves, sum and xor of their h (flags = 0); out[3..5] the same
 * over the inner nodes. Equal dumps <=> equal digests (up to 2^-64); tests/golden_util.py computes the same value
 * from a reference dump. Used for maps too large to export (config C3 at insert depth 0: 3.4e8 leaves) and by
 * replicas of one map on several GPUs to check each other. Not part of the reference's surface. */
int ufomap_map_digest(ufomap_map* m, int include_unknown, uint64_t out[6]);

/* Octree::write(std::ostream&) (octree.h:833-868) with compress=false, min_depth=0 and no bounding volume:
 * the text header followed by the pre-order node stream of OccupancyMapBase::writeNodes
 * (occupancy_map_base.h:1457-1533). Byte-identical to what the reference writes for the same map, so the
 * result can be loaded by the reference (`Octree::read`), wrapped into a ufomap_msgs/UFOMap message
 * (ufomap_msgs conversions.h:162-186 sends exactly this node stream) or saved as a .ufo file.
 * Returns the total size in bytes (header + data), writing only if cap is large enough; (size_t)-1 on error. */
size_t ufomap_map_write(ufomap_map* m, uint8_t* buf, size_t cap);

/* min/max change AABB: minChange()/maxChange()/resetMinMaxChangeDetection
 * (occupancy_map_base.h:793-822). Always tracked. */
int ufomap_map_minmax_change(ufomap_map* m, double mn[3], double mx[3]);
int ufomap_map_reset_minmax_change(ufomap_map* m);

/* getNumInnerNodes / getNumLeafNodes / memoryUsage analogues (octree.h:411-443): live node blocks,
 * leaves reachable from them, bytes of HBM held by the map. */
int ufomap_map_stats(ufomap_map* m, uint64_t* n_inner, uint64_t* n_leaf, uint64_t* bytes);

/* ---- stage-level outputs of the last integration (parity tests; SURVEY.md section 4 iii) ----
 * hits: unique hit voxel codes at depth 0 (`occupied_hits`, occupancy_map_base.h:296);
 * misses: unique free cells as code >> 3*depth (`free_hits`, occupancy_map_base.h:1356-1365);
 * both sorted ascending. counts: [0]=points in, [1]=rays cast, [2]=DDA steps, [3]=unique hits,
 * [4]=unique miss cells (filled by last_misses), [5]=node blocks touched, [6]=node blocks created,
 * [7]=cells dropped because their key fell outside [0,2^L) (clipped rays only; the reference aliases
 * such keys to the opposite face of the map). */
size_t ufomap_map_last_hits(ufomap_map* m, uint64_t* codes, size_t cap);
size_t ufomap_map_last_misses(ufomap_map* m, uint64_t* codes, size_t cap);
int ufomap_map_last_counts(ufomap_map* m, uint64_t counts[8]);

/* ---- measurement: per-kernel HIP-event timing on the map's own stream ------------------------
 * With profiling on, every kernel launch of the hot path is bracketed by hipEvents on the
 * stream it is launched on. kernel_times returns, for up to cap kernels, the name, number of
 * launches and total milliseconds since the last reset. Returns the number of kernels. */
int ufomap_map_set_profiling(ufomap_map* m, int on);
int ufomap_map_kernel_times(ufomap_map* m, const char** names, uint64_t* launches, double* total_ms,
                            int cap);
int ufomap_map_reset_kernel_times(ufomap_map* m);

/* ---- multi-GPU batched scans: the path split at its only exchange point -----------
 * The reference integrates scans one after the other into one tree. For a batch of scans taken by
 * different sensors (BASELINE config C4) every GPU ray-casts ITS scan (scan_keys: the ~80 % of the work
 * that never reads the map) and produces the scan's update list; the lists are exchanged (RCCL
 * all-gather, done by the caller) and every replica applies all lists in scan order (apply_keys), which
 * reproduces sequential integration bit-exactly -- clamping makes summed log-odds deltas inexact.
 *
 * An update list is n_hit + n_miss records of 16 bytes, hit records first:
 *   u64 location key of an 8-child node block | u8 hit mask | u8 miss mask | u8 level |
 *   u8 child updated last | u32 point index of that update (hits: cloud order)
 * scan_keys leaves the list in the map's scratch memory (valid until the next call on this map);
 * get_keys copies it (device to device) into caller memory, e.g. a torch tensor used as RCCL buffer.
 * d_xyz / d_dst / d_entries are DEVICE pointers. */
typedef struct ufomap_keys_info {
	uint32_t n_hit, n_miss; /* records */
	int32_t nb_hit[3];      /* extent of the scan's hit grid in node blocks (sizes the node table) */
	int32_t nb_miss[3];
	uint32_t depth;         /* insert depth of the scan: miss records are level depth+1 */
	uint32_t reserved;      /* bit 1: a colour section follows the records (ufomap_map_scan_keys_rgb);
	                           bit 0: merged list (depth 0): n_hit records that carry the hit AND the miss mask of
	                           their block, n_miss = 0 -- what scan_keys produces for depth-0 scans */
} ufomap_keys_info;
int ufomap_map_scan_keys(ufomap_map* m, const double sensor_origin[3], const double* d_xyz, size_t n,
                         double max_range, unsigned depth, int discrete, int simple_ray_casting,
                         ufomap_keys_info* info);
/* Colour maps (occupancy_map_color.h:177-267, discrete integrator, insert depth 0): the list gets a colour section behind
 * its records -- 8 x uint32 (r | g<<8 | b<<16) per hit record, the colour of the first point of every hit voxel, which
 * is what the reference blends in (225-233) -- and bit 1 of `reserved` says so. get_keys copies it too (cap_entries counts
 * 16-byte units: n_hit + n_miss + 2 * n_hit); apply_keys / apply_keys_batch on a colour map require it. */
int ufomap_map_scan_keys_rgb(ufomap_map* m, const double sensor_origin[3], const double* d_xyz, const uint8_t* d_rgb, size_t n,
                             double max_range, unsigned depth, int discrete, int simple_ray_casting, ufomap_keys_info* info);
int ufomap_map_get_keys(ufomap_map* m, void* d_dst, size_t cap_entries, const ufomap_keys_info* info);
int ufomap_map_apply_keys(ufomap_map* m, const void* d_entries, const ufomap_keys_info* info);
/* The update lists of n_lists scans (insert depth 0; list j = what get_keys returned for scan j) applied in
 * order 0..n_lists-1 with ONE walk of the tree for the whole batch. Same map as n_lists calls of
 * ufomap_map_apply_keys, i.e. as the reference integrating the scans one after the other
 * (occupancy_map_base.h:340-417 called n_lists times). With ufomap_map_set_option(m, "async_apply", 1) the call
 * returns after enqueueing (the lists must stay valid until the next call on this map that joins the update:
 * apply_keys*, insert*, wait, any reader); ufomap_map_scan_keys of the NEXT batch does not join it -- ray casting
 * never reads the map -- so it overlaps with the update. */
int ufomap_map_apply_keys_batch(ufomap_map* m, const void* const* d_lists, const ufomap_keys_info* infos, int n_lists);

/* ---- batched multi-sensor integration across GPUs (BASELINE config C4) ------------------------------
 * One process per GPU. Every rank calls ufomap_map_insert_batch with ITS scan of the batch; every replica then equals
 * the reference's map after insertPointCloudDiscrete / insertPointCloud (occupancy_map_base.h:270-417) of scan 0, 1, ...,
 * world-1 in that order. Steady state (plain map, the ranks share a ray grid): the scan is ray-cast on that grid and
 * travels as two bit grids + a tile bitmap + its control block (~0.2 MB) in ONE RCCL all-gather; ONE walk of the tree
 * applies all ranks' scans in rank order, enqueued behind the previous step's (option "async_apply": the call returns
 * after enqueueing). First steps, colour maps (ufomap_map_scan_keys_rgb: a colour section travels with the records) and
 * grids beyond LDS exchange update lists (ufomap_map_scan_keys / ufomap_map_apply_keys_batch) in fixed-size slots that
 * grow alike on all ranks; a step whose common grid a rank's scan does not fit is repeated in that form by ALL ranks when
 * it is joined. Contract: every rank makes the same sequence of calls on its map with the same map parameters and
 * options (joins happen at fixed points of that sequence); cloud, size (0 allowed), pose and max_range are the rank's
 * own. A rank whose scan fails still takes part in the collective and every rank returns the error. Change detection: the
 * min / max change box of every replica grows by every rank's scan (the boxes travel with the step); per-code change
 * detection makes every step take the list form. Insert depth 0: the
 * bit-grid form, up to "batch_depth" (3) steps in flight with option "async_apply"; depth > 0: the list form, joined first.
 * d_xyz / d_rgb are consumed when the call returns. librccl is loaded at run time (a copy already in the process is
 * preferred; UFOMAP_RCCL_LIB names the library to use instead -- then that one only).
 *   ufomap_comm_unique_id   on ONE rank; the 128 bytes reach the others by the host's own means (file, socket, MPI...)
 *   ufomap_comm_create      ncclCommInitRank on `device` (collective: all ranks call it)
 *   ufomap_comm_from_nccl   wrap a communicator the host already has (ncclComm_t); not destroyed by ufomap_comm_destroy
 *   ufomap_comm_stats       out[0] world, [1] rank, [2] slot capacity in bytes, [3] times the capacity had to grow */
#define UFOMAP_COMM_ID_BYTES 128
typedef struct ufomap_comm ufomap_comm;
int ufomap_comm_unique_id(uint8_t id[UFOMAP_COMM_ID_BYTES]);
ufomap_comm* ufomap_comm_create(const uint8_t id[UFOMAP_COMM_ID_BYTES], int world, int rank, int device);
ufomap_comm* ufomap_comm_from_nccl(void* nccl_comm, int world, int rank, int device);
void ufomap_comm_destroy(ufomap_comm* c);
int ufomap_comm_stats(const ufomap_comm* c, uint64_t out[4]);
/* out[0] steps of ufomap_map_insert_batch that exchanged bit grids and applied all ranks' scans with one walk (the steady
 * state), [1] steps that were repeated in update-list form (a rank's scan did not fit the ranks' common ray grid),
 * [2] 1 if the ranks currently share a ray grid, [3] reserved */
int ufomap_comm_counters(const ufomap_comm* c, uint64_t out[4]);
int ufomap_map_insert_batch(ufomap_map* m, ufomap_comm* c, const double sensor_origin[3], const double* d_xyz,
                            const uint8_t* d_rgb /* colour maps: 3 bytes per point; else NULL */, size_t n, double max_range,
                            unsigned depth, int discrete);
/* ... with the two remaining arguments of insertPointCloud / insertPointCloudDiscrete (occupancy_map_base.h:270-273, 340-344):
 * simple_ray_casting (freeSpaceSimple, 1303-1339) and early_stopping (1289-1298, 1327-1333). Both have to be the same on every
 * rank (like depth and discrete); a step with either takes the update-list form. Insert depth > 0 (occupancy_map_base.h:378-386,
 * 1085-1120; round 6): the update-list form too, the ranks' lists applied one by one in rank order -- each its hits, then its misses at
 * the insert depth, as the reference does scan after scan (plain maps; a colour map's lists are built at depth 0 only). */
int ufomap_map_insert_batch_ex(ufomap_map* m, ufomap_comm* c, const double sensor_origin[3], const double* d_xyz, const uint8_t* d_rgb, size_t n,
                               double max_range, unsigned depth, int discrete, int simple_ray_casting, unsigned early_stopping);

/* Diagnostic overrides for tests: "dda_mode" (-1 auto; 1 / 2 force the LDS-filter / direct variants of
 * the ray kernel on grids that would fit in LDS), "entry_guess" (cap of the guessed update-list size, to
 * exercise the exact-size retry), "dda_seg" (0: one lane per ray), "merge_phases" (0: hits and misses of a
 * depth-0 scan as two passes over the tree instead of one; the environment variable UFOMAP_MERGE_PHASES
 * sets the default for new maps), "spec" (0: always read a scan's bounding boxes back before sizing its ray
 * grid; default 1: a depth-0 scan is enqueued on the grid predicted from the previous scan and repeated if it
 * does not fit), "fast" (0: never take the steady-state path of fast_kernels.h), "fast_color" (0: colour maps keep to the
 * general path), "big" (0: so do scans whose ray grid does not fit in LDS), "solo" (0: a synchronous call with nothing in
 * flight uses the three pipeline streams like an asynchronous one instead of the map stream alone), "lazy_done" (0: the end
 * of a scan half is published by a kernel of its own instead of the next scan's gate kernel), "cast_wgs" (workgroups of the
 * steady-state ray kernel; default: one per CU for a synchronous call, on three CUs in four for asynchronous calls in a
 * row), "cast_threads" / "cast_batch" / "cast_qcap" / "cast_prio" (its workgroup size, rays per round, segment queue
 * entries, wave priority), "tstamps" (1: the hand-over kernels record the device clock, ufomap_map_timeline),
 * "batch_max" (scans one walk of the tree may take when scans
 * have queued up behind the map stream: 1 .. 16, default 8), "hold" (test aid: a slot on the map stream for every hold-th
 * scan only, so that walks over several scans happen whatever the timing), "gate_us" (a stream hand-over gives up after
 * this long and the handle uses events from then on; default 20000), "cast_global" (0: grids
 * beyond LDS through k_dda_seg instead of k_cast<2>; 2-4: force the box / filter variants on small grids), "sparse_set"
 * (1: every scan's ray cells through the sparse set), "phase_limit" / "scan_id" (when the per-phase tags restart),
 * "async_apply" (apply_keys_batch / insert_batch return after enqueueing); round 4: "vol" (0: depth-0 scans whose ray box is
 * beyond the steady-state path -- a 2 mm RGB-D frame -- keep to the general path instead of the volume path of vol_kernels.h;
 * 2: the volume path also for boxes the steady-state path would take), "vol_pregrow" (0: no growth of the node table before a
 * volume-path walk: it runs out of its reserve and the table is exchanged in the middle of the walk), "vol_keep" (0: the merged
 * ray cells of a volume-path scan are not kept for ufomap_map_last_misses), "vol_mode" (A/B switches of its ray kernels: bit 1 blocks in launch order, bit 2
 * rays in the cloud's order, bit 3 no write-combining table, bit 4 one lane per ray (k_vdda) instead of the segmented walk); round 5:
 * "vol_seg" (cells per segment of a ray on the volume path, default 192), "vol_walk_blocks" (workgroups of its walk per eighth of the
 * scan), "vol_async" (0: an asynchronous call returns a finished integration on the volume path instead of leaving its walk
 * enqueued), "vol_color" (0: colour maps keep to the general path), "cast_fused" (0: the steady-state path's ray kernel in its round-3
 * form, k_fcast), "cast2_k" (cells per segment of the fused ray kernel, default 64), "fast_simple" (0: fixed-step ray casting keeps to the general path), "gather_stream" (1: the all-gather of a batch step on
 * a stream of its own), "fail_scan" (test aid: the scan half of the next batch steps fails on this rank before the collective).
 * Results never depend on these. */
int ufomap_map_set_option(ufomap_map* m, const char* key, long long value);
/* Diagnostics: std::exp(float) as the device evaluates it inside toProb (occupancy_map_base.h:911 with LogitType = float;
 * ufomap_amd/csrc/expf_ref.h), on n host values. tests/: equal to the host's libm expf, which is what the reference calls. */
int ufomap_dev_expf(const float* x, float* out, size_t n, int device);
/* Diagnostics (option "tstamps" = 1): the device clock (100 MHz) at the hand-overs of the steady-state pipeline, 8 words per
 * fast-path scan, slot = scan number mod 4096: [0] first-point pass done, [1] gate entered, [2] gate open, [3] scan half
 * published, [4] claim entered, [5] claim done, [6] tree update done, [7] scans that walk took. Joins everything first;
 * *newest = number of the newest fast-path scan. scripts/dev/dev_timeline.py prints where each stream waits. */
int ufomap_map_timeline(ufomap_map* m, unsigned long long* out, size_t n_words, unsigned long long* newest);

/* Diagnostics: up to 64 raw 64-bit words written by the last integration's kernels (per-level
 * wall_clock64 stamps of the propagation tails: [level] hits phase, [32+level] misses phase, [31]/[63]
 * end stamps; 100 MHz clock); [62] scans enqueued on a predicted ray grid, [63] of which had to be repeated.
 * Not part of the reference's surface. */
int ufomap_map_debug(ufomap_map* m, uint64_t* out, int n);

/* Diagnostics: device allocations of this process so far -- out[0] hipMalloc calls, [1] hipFree calls, [2] bytes asked
 * for, [3] host nanoseconds spent inside them, [4] re-hashes of a node table. A warm scan must see none (bench.py reports
 * the counts inside every timed call of its bandwidth-bound leg). Not part of the reference's surface. */
void ufomap_alloc_counters(uint64_t out[5]);

/* Raw HIP stream of the map (hipStream_t), for callers that need to order their own work. */
void* ufomap_map_stream(ufomap_map* m);

#ifdef __cplusplus
}
#endif
#endif /* UFOMAP_HIP_H */
