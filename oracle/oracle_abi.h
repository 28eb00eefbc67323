/*
 * oracle_abi.h -- C ABI shared by the two CPU checkers of this repo.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load these libraries, and only as
 * the checker / the reported CPU baseline -- never as the thing measured or shipped.
 *
 * Two libraries export exactly this interface:
 *   oracle/_ref/libufo_ref.so   the UNMODIFIED reference (UnknownFreeOccupied/ufomap v1) compiled
 *                               from /root/reference by oracle/Makefile (ref_harness.cpp is the
 *                               only file of ours in that build; no reference source is copied);
 *   oracle/libufo_oracle.so     our own CPU restatement of the integration path (ufo_oracle.cpp),
 *                               validated against libufo_ref.so and the golden vectors.
 *
 * Canonical dump format: a node is identified by (code >> 3*depth, depth); leaves
 * are returned sorted by (depth, shifted code).
 */
#ifndef UFO_ORACLE_ABI_H
#define UFO_ORACLE_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ufo_oracle_map ufo_oracle_map;

/* Constructor arguments mirror OccupancyMapBase (occupancy_map_base.h:859-862). */
ufo_oracle_map* ufo_oracle_create(double resolution, unsigned depth_levels, int automatic_pruning,
                                  double occupied_thres, double free_thres, double prob_hit,
                                  double prob_miss, double clamp_min, double clamp_max, int color);
void ufo_oracle_destroy(ufo_oracle_map* m);

/* insertPointCloud (discrete=0, occupancy_map_base.h:270) / insertPointCloudDiscrete (discrete=1,
 * occupancy_map_base.h:340; colour overload occupancy_map_color.h:177). rgb may be NULL.
 * Returns 0 on success, <0 when the combination is not supported by this checker. */
int ufo_oracle_insert(ufo_oracle_map* m, const double origin[3], const double* xyz,
                      const uint8_t* rgb, size_t n, double max_range, unsigned depth, int discrete,
                      int simple_ray_casting, unsigned early_stopping);

/* Leaves in canonical order. include_unknown=0 skips leaves whose state is "unknown".
 * Returns the number of leaves (may exceed cap; only cap entries are written). Any output
 * pointer may be NULL. rgb is 3 bytes per leaf (zeros for a non-colour map). */
size_t ufo_oracle_export_leaves(const ufo_oracle_map* m, int include_unknown, uint64_t* codes,
                                uint8_t* depths, float* logodds, uint8_t* rgb, size_t cap);

/* Inner nodes that have children (depth >= 1), canonical order. flags bit0 = contains_free,
 * bit1 = contains_unknown. */
size_t ufo_oracle_export_inner(const ufo_oracle_map* m, uint64_t* codes, uint8_t* depths,
                               float* logodds, uint8_t* flags, uint8_t* rgb, size_t cap);

/* The map as the reference's byte stream: Octree::write(std::ostream&, compress=false) (octree.h:833-868):
 * text header + pre-order node stream (occupancy_map_base.h:1457-1533). Returns the size; writes only if
 * cap is large enough. */
size_t ufo_oracle_write(const ufo_oracle_map* m, uint8_t* buf, size_t cap);

/* Min/max change AABB (occupancy_map_base.h:305-308, 388-398, 1367-1372). Returns 0 if enabled. */
int ufo_oracle_minmax_change(const ufo_oracle_map* m, double mn[3], double mx[3]);

/* Stage-level outputs of the LAST insert (restatement only; the reference build returns
 * (size_t)-1 for these): unique hit codes (depth 0), ray end points, unique miss codes
 * (shifted by 3*depth), total DDA steps. */
size_t ufo_oracle_last_hits(const ufo_oracle_map* m, uint64_t* codes, size_t cap);
size_t ufo_oracle_last_rays(const ufo_oracle_map* m, double* ends_xyz, size_t cap);
size_t ufo_oracle_last_misses(const ufo_oracle_map* m, uint64_t* codes, size_t cap);
uint64_t ufo_oracle_last_steps(const ufo_oracle_map* m);
/* Number of hit/miss keys of the last insert with a coordinate outside [0, 2^L): the reference
 * aliases such keys into the tree (bits above 3L are ignored by getChildIdx, code.h:245-248) and can
 * apply one voxel twice; the HIP path drops them. Inputs with a non-zero count are outside the
 * parity contract (only reachable when a segment is clipped at the map cube). */
uint64_t ufo_oracle_last_oob(const ufo_oracle_map* m);

/* "reference" or "port". */
/* Robot clearing: OccupancyMapBase::setValueVolume(AABB(mn, mx), occupancy_value, min_depth)
 * (map/occupancy_map_base.h:492-518, 986-1031; geometry/aabb.h:62-65; collision_checks.cpp:256-264) as the server
 * calls it after every scan (ufomap_mapping/src/server.cpp:152-155). occupancy_value is a probability. */
int ufo_oracle_set_value_volume(ufo_oracle_map* m, const double mn[3], const double mx[3], double occupancy_value,
                                unsigned min_depth);
/* getClampingThresMin() / getClampingThresMax() (occupancy_map_base.h:742-744): toProb of the stored logits */
void ufo_oracle_clamping_thres(const ufo_oracle_map* m, double* thres_min, double* thres_max);

/* Point queries: for each coordinate, code = toCode(coord, depth) and
 * (node, d) = Octree::getNode(code) (map/octree.h:974-985). NOTE the reference's loop stops one level early:
 * on a fully expanded path the node returned is the one at depth code.depth + 1 (reported as code.depth); where
 * the path ends in a leaf earlier, that leaf with its true depth. Everything below is evaluated on THAT node:
 *   logodds[i]  node->value.occupancy (getOccupancy(code) is toProb of it, OMB:599-613)
 *   state[i]    bit 0 occupied / bit 1 free / bit 2 unknown  = getState(code) (OMB:619-634)
 *               bit 3 containsFree(code), bit 4 containsUnknown(code) (OMB:693-728, 953-968) */
void ufo_oracle_query(const ufo_oracle_map* m, const double* xyz, size_t n, unsigned depth, float* logodds, uint8_t* state);

/* Ingest in front of the hot path: rosToUfo (ufomap_ros/ufomap_ros/src/conversions.cpp:
 * 98-138: float32 x, y, z [+ r, g, b bytes] at byte offsets inside records of `step` bytes; points with a NaN
 * coordinate are dropped) followed by PointCloud::transform (map/point_cloud.h:157-166 -> math/pose6.h:114-125
 * -> math/quaternion.h:253-286: q v q^-1, then + translation, all in double). Returns the number of points
 * kept; xyz_out[3k..], rgb_out[3k..] (rgb_out may be NULL; off_r < 0: colour 0,0,0 as Point3Color(x,y,z)). */
size_t ufo_oracle_ingest(const uint8_t* data, size_t n, uint32_t step, int off_x, int off_y, int off_z, int off_r, int off_g,
                         int off_b, const double rot_wxyz[4], const double trans[3], double* xyz_out, uint8_t* rgb_out);


/* ---- round 2: what the reference's callers use around the hot path. Provided by the REFERENCE build only
 * (the port returns -1 / (size_t)-1): the parity tests for these rows need oracle/_ref/libufo_ref.so, which is built
 * where /root/reference exists and travels to the GPU box. Bounding volumes: AABB as (centre[3], half_size[3]) or
 * NULL, NULL. ---- */
/* beginLeaves (only_leaves) / beginTree run to the end (occupancy_map_base.h:93-165): the nodes in iteration order.
 * flags: bit 0 contains_free, bit 1 contains_unknown, bit 2 leaf. */
size_t ufo_oracle_iterate(const ufo_oracle_map* m, const double* aabb_center, const double* aabb_half, int occupied_space, int free_space,
                          int unknown_space, int contains, unsigned min_depth, int only_leaves, uint64_t* codes, uint8_t* depths,
                          float* logodds, uint8_t* rgb, uint8_t* flags, size_t cap);
/* enableChangeDetection / resetChangeDetection / changesBegin..End (occupancy_map_base.h:779-791), sorted by (depth, code) */
int ufo_oracle_enable_change_detection(ufo_oracle_map* m, int enable);
int ufo_oracle_reset_change_detection(ufo_oracle_map* m);
size_t ufo_oracle_changes(const ufo_oracle_map* m, uint64_t* codes, uint8_t* depths, size_t cap);
int ufo_oracle_enable_minmax_change_detection(ufo_oracle_map* m, int enable);
/* write (header != 0) / writeData (header == 0) with all arguments (octree.h:779-917) */
size_t ufo_oracle_write_ex(const ufo_oracle_map* m, const double* aabb_center, const double* aabb_half, int compress, unsigned min_depth,
                           int accel, int level, int header, uint8_t* buf, size_t cap, long long* uncompressed_size);
/* read(std::istream&) / readData(...) (octree.h:701-777) */
int ufo_oracle_read(ufo_oracle_map* m, const uint8_t* buf, size_t n);
int ufo_oracle_read_data(ufo_oracle_map* m, const uint8_t* data, size_t n, const double* aabb_center, const double* aabb_half,
                         double resolution, unsigned depth_levels, int uncompressed_data_size, int compressed);
/* getters in the order occupied, free, hit, miss, clamp min, clamp max (occupancy_map_base.h:734-744); setters 746-773 */
int ufo_oracle_get_sensor_model(const ufo_oracle_map* m, double out[6]);
int ufo_oracle_set_model_value(ufo_oracle_map* m, int which, double probability);
int ufo_oracle_set_occupied_free_thres(ufo_oracle_map* m, double occupied_thres, double free_thres);
int ufo_oracle_clear_to(ufo_oracle_map* m, double resolution, unsigned depth_levels);
/* setValueVolume with the AABB given as (centre, half size) */
int ufo_oracle_set_value_volume_ch(ufo_oracle_map* m, const double c[3], const double h[3], double occupancy_value, unsigned min_depth);

const char* ufo_oracle_kind(void);

#ifdef __cplusplus
}
#endif
#endif
