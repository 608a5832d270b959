/*
 * orbslam3_hip_host.h -- C entry points of the C++ host layer (csrc/host/), for tests and bindings.
 *
 * The drop-in itself is C++: ORB_SLAM3::Optimizer::LocalBundleAdjustment (include/Optimizer.h) and
 * ORB_SLAM3::ORBmatcher::SearchByProjection (include/ORBmatcher.h), same signatures as the reference.
 * These C wrappers build a KeyFrame / MapPoint / Map (or Frame) graph from flat arrays, call the C++
 * entry points and read the graph back, so the boundary can be exercised from ctypes.
 */
#ifndef ORBSLAM3_HIP_HOST_H
#define ORBSLAM3_HIP_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct osh_host_graph osh_host_graph;

/* Wall time (ms) of the product call (Optimizer:: / ORBmatcher:: entry point) inside the last harness wrapper on this thread,
 * without the construction of the KeyFrame / MapPoint / Frame objects around it. */
double osh_host_last_call_ms(void);

/* Build a map: n_kf keyframes (ids kf_id, poses Tcw as qx qy qz qw tx ty tz floats, one shared pinhole
 * fx fy cx cy + bf), n_mp map points, n_obs observations (keyframe index, point index, u v ur with ur<0 for
 * monocular, octave).  Keypoint k of a keyframe is its k-th observation in array order. */
osh_host_graph* osh_host_graph_create(int32_t n_kf, const int64_t* kf_id, const float* kf_pose_qt, const float* cam5,
                                      const float* inv_level_sigma2, int32_t n_levels,
                                      int32_t n_mp, const int64_t* mp_id, const float* mp_pos,
                                      int32_t n_obs, const int32_t* obs_kf, const int32_t* obs_mp, const float* obs_uvr,
                                      const int32_t* obs_octave, int64_t init_kf_id, int32_t inertial);
/* Turn the map into a fisheye STEREO rig (call after osh_host_graph_set_fisheye): right camera cam2 = fx fy cx cy k1..k4,
 * Trl = qx qy qz qw tx ty tz, and n_obs right-camera observations (keyframe index, map point index, u v, octave) which become
 * mvKeysRight entries observed through the right slot of MapPoint::GetObservations()'s tuples. */
int osh_host_graph_set_rig(osh_host_graph* g, const float cam2[8], const float trl_qt[7], int32_t n_obs, const int32_t* obs_kf,
                           const int32_t* obs_mp, const float* obs_uv, const int32_t* obs_octave);
void osh_host_graph_destroy(osh_host_graph* g);
/* Camera model of the window the last osh_host_pack_lba / _gba / _welding call built: 1 + k1..k4 for KannalaBrandt8, else 0. */
int osh_host_last_pack_kb8(osh_host_graph* g, double k[4]);
/* Same for the rig description of a fisheye stereo window: returns 1 and fills cam2[8] / trl[7], else 0. */
int osh_host_last_pack_rig(osh_host_graph* g, double cam2[8], double trl[7]);
/* Switch the map's camera to a KannalaBrandt8 (same fx fy cx cy, coefficients k1..k4): a monocular fisheye map. */
void osh_host_graph_set_fisheye(osh_host_graph* g, const float k[4]);
/* covisibility list returned by KeyFrame::GetVectorCovisibleKeyFrames() of keyframe kf_index */
int osh_host_graph_set_covisible(osh_host_graph* g, int32_t kf_index, int32_t n, const int32_t* kf_indices);

/* Steps 1-6 of Optimizer::LocalBundleAdjustment only (graph -> osh_lba_problem arrays); no GPU needed.
 * sizes = {P, F, L, E, num_fixedKF}; every array may be NULL.  Returns 0, 1 (no fixed keyframe) or <0. */
int osh_host_pack_lba(osh_host_graph* g, int32_t kf_index, int32_t sizes[5], double* pose_qt, double* pose_cam,
                      double* points, int32_t* edge_pose, int32_t* edge_point, uint8_t* edge_kind, double* edge_obs,
                      double* edge_info, int64_t* pose_kf_id, int64_t* point_mp_id);
/* ORB_SLAM3::Optimizer::LocalBundleAdjustment(kf, stop_flag, map, ...); counts = num_fixedKF, num_OptKF, num_MPs, num_edges */
int osh_host_run_lba(osh_host_graph* g, int32_t kf_index, unsigned char* stop_flag, int32_t counts[4]);

void osh_host_get_kf_pose(osh_host_graph* g, int32_t kf_index, float out_qt[7]);
void osh_host_get_mp_pos(osh_host_graph* g, int32_t mp_index, float out[3]);
int  osh_host_mp_num_observations(osh_host_graph* g, int32_t mp_index);   /* size of GetObservations() */
int  osh_host_mp_is_bad(osh_host_graph* g, int32_t mp_index);
int  osh_host_kf_num_matches(osh_host_graph* g, int32_t kf_index);        /* non-null GetMapPointMatches() */
int  osh_host_kf_observes(osh_host_graph* g, int32_t kf_index, int32_t mp_index);
int  osh_host_map_change_index(osh_host_graph* g);
int  osh_host_kf_pose_sets(osh_host_graph* g, int32_t kf_index);

/* ---- inertial ---- */
/* Attach IMU state to the first n keyframes listed: previous keyframe (index or -1), velocity, bias (bax bay baz bwx bwy bwz),
 * the preintegration from the previous keyframe (OSH_PREINT_FLOATS record, dT == 0: none) with its 15x15 covariance, and the
 * camera-IMU calibration T_bc (qx qy qz qw tx ty tz).  Every keyframe of the map gets the calibration. */
int osh_host_graph_set_inertial(osh_host_graph* g, int32_t n, const int32_t* kf_index, const int32_t* prev_index, const float* vel,
                                const float* bias6, const float* preint, const float* cov225, const float* Tbc_qt);
/* Window selection + packing of Optimizer::LocalInertialBA (no GPU): fills *out with pointers into storage owned by the
 * graph (valid until the next call).  pose_kf_id / point_mp_id (may be NULL) receive the ids in problem order. */
struct osh_liba_problem;
int osh_host_pack_liba(osh_host_graph* g, int32_t kf_index, int32_t b_large, int32_t b_rec_init, struct osh_liba_problem* out,
                       int64_t* pose_kf_id, int64_t* point_mp_id);
/* ORB_SLAM3::Optimizer::LocalInertialBA(kf, NULL, map, ..., bLarge, bRecInit) */
int osh_host_run_liba(osh_host_graph* g, int32_t kf_index, int32_t b_large, int32_t b_rec_init);
/* Optimizer::FullInertialBA(&map, its, bFixLocal, nLoopId, NULL, bInit) (src/Optimizer.cc:393-814): the flat problem it solves
 * (pack; *n_idle = keyframes no edge touches) and the call itself.  -3: a case the device path does not take (message on stderr). */
int osh_host_pack_full_inertial(osh_host_graph* g, int32_t its, int32_t fix_local, int32_t b_init, float prior_g, float prior_a,
                                struct osh_liba_problem* out, int64_t* pose_kf_id, int64_t* point_mp_id, int32_t* n_idle);
int osh_host_run_full_inertial(osh_host_graph* g, int32_t its, int32_t fix_local, int64_t loop_id, int32_t b_init, float prior_g, float prior_a);
int64_t osh_host_get_kf_inertial_gba(osh_host_graph* g, int32_t i, float vel[3], float bias6[6]);   /* mVwbGBA, mBiasGBA; returns mnBAGlobalForKF */
/* Optimizer::MergeInertialBA(curr, merge, NULL, &map, corrPoses) (src/Optimizer.cc:3956-4498).  n_sets = {temporal, covisible}
 * keyframe counts with their ids in the reference's order; run returns corrPoses.size() and copies {id; qx qy qz qw tx ty tz s}. */
int osh_host_pack_merge_inertial(osh_host_graph* g, int32_t curr, int32_t merge, struct osh_liba_problem* out, int64_t* pose_kf_id,
                                 int64_t* point_mp_id, int32_t n_sets[2], int64_t* temporal_kf_id, int64_t* cov_kf_id);
int osh_host_run_merge_inertial(osh_host_graph* g, int32_t curr, int32_t merge, int32_t max_corr, int64_t* corr_kf_id, double* corr_sim3);
void osh_host_get_kf_velocity(osh_host_graph* g, int32_t kf_index, float out[3]);
void osh_host_get_kf_bias(osh_host_graph* g, int32_t kf_index, float out6[6]);
/* IMU::Preintegrated: integrate n measurements (float32 recursion) -> record + covariance */
int osh_host_preintegrate(int32_t n, const float* acc, const float* gyr, float dt, const float* bias6, const float* nga6,
                          const float* walk6, float* rec_out, float* cov225_out);
/* EdgeInertial information from a 15x15 preintegration covariance (inverse, symmetrise, eigenvalue clamp) */
int osh_host_inertial_information(const float* cov225, double* info81_out);

/* ---- matcher ---- */
/* Optimizer::GlobalBundleAdjustemnt(&map, n_iterations, stop_flag, n_loop_kf, robust) on every keyframe and point of the graph
 * (src/Optimizer.cc:53-392).  pack: the osh_lba_problem the host layer builds (sizes = {n_free, n_fixed, n_points, n_edges,
 * points without any edge}); the GBA getters return mnBAGlobalForKF and fill mTcwGBA / mPosGBA. */
int osh_host_pack_gba(osh_host_graph* g, int32_t sizes[5], double* pose_qt, double* pose_cam, double* points,
                      int32_t* edge_pose, int32_t* edge_point, uint8_t* edge_kind, double* edge_obs, double* edge_info,
                      int64_t* pose_kf_id, int64_t* point_mp_id);
/* The welding Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, pbStopFlag) (src/Optimizer.cc:3506-3955); pack =
 * the problem of its first optimisation. */
int osh_host_pack_welding(osh_host_graph* g, int32_t main_index, int32_t n_adj, const int32_t* adj, int32_t n_fix,
                          const int32_t* fix, int32_t sizes[5], double* pose_qt, double* pose_cam, double* points,
                          int32_t* edge_pose, int32_t* edge_point, uint8_t* edge_kind, double* edge_obs, double* edge_info,
                          int64_t* pose_kf_id, int64_t* point_mp_id);
int osh_host_run_welding(osh_host_graph* g, int32_t main_index, int32_t n_adj, const int32_t* adj, int32_t n_fix,
                         const int32_t* fix, unsigned char* stop_flag);
int osh_host_run_gba(osh_host_graph* g, int32_t n_iterations, unsigned char* stop_flag, int64_t n_loop_kf, int32_t robust);
int64_t osh_host_get_kf_pose_gba(osh_host_graph* g, int32_t kf_index, float pose_qt[7]);
int64_t osh_host_get_mp_pos_gba(osh_host_graph* g, int32_t mp_index, float pos[3]);
int osh_host_mp_normal_updates(osh_host_graph* g, int32_t mp_index);
void osh_host_set_bad(osh_host_graph* g, int32_t kf_index, int32_t mp_index);   /* marks a keyframe and/or a point bad (-1: none) */

typedef struct osh_host_frame osh_host_frame;
/* A frame with n keypoints (x y, octave, angle, uRight (<=0: none), 32-byte descriptor), pose Tcw. */
osh_host_frame* osh_host_frame_create(int32_t n, const float* kp_xy, const int32_t* octave, const float* angle,
                                      const float* uright, const uint8_t* desc, const float pose_qt[7],
                                      const float cam4[4], float mbf, float mb, int32_t n_levels, float scale_factor);
/* Fisheye STEREO frame: the first n_left keypoints become the left camera's, the rest the right camera's (descriptor rows
 * [n_left, N)); left_to_right[n_left] / right_to_left[N - n_left] are Frame::mvLeftToRightMatch / mvRightToLeftMatch. */
int osh_host_frame_set_rig(osh_host_frame* f, int32_t n_left, const int32_t* left_to_right, const int32_t* right_to_left, const float trl_qt[7]);
/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) on such a frame (src/ORBmatcher.cc:43-213, both camera
 * passes); per map point the tracking fields of both cameras; assignment[N] = map point stored in each keypoint slot or -1. */
int osh_host_search_local_points_rig(osh_host_frame* f, int32_t n_mp, const uint8_t* mp_desc, const uint8_t* in_left,
                                     const float* proj_left, const int32_t* level_left, const float* viewcos_left,
                                     const uint8_t* in_right, const float* proj_right, const int32_t* level_right,
                                     const float* viewcos_right, const int32_t* n_observations, float nnratio, float th,
                                     int32_t* assignment);
int osh_host_frame_set_camera2(osh_host_frame* f, const float cam2[8]);
int osh_host_search_by_bow_kf(int32_t n1, const uint8_t* desc1, const float* angle1, const uint8_t* has_mp1, int32_t nodes1,
                              const int32_t* node_id1, const int32_t* node_off1, const int32_t* node_feat1, int32_t n2,
                              const uint8_t* desc2, const float* angle2, const uint8_t* has_mp2, int32_t nodes2, const int32_t* node_id2,
                              const int32_t* node_off2, const int32_t* node_feat2, float nnratio, int32_t check_ori, int32_t* match12);
int osh_host_search_by_bow(osh_host_frame* f, int32_t n_kf, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_has_mp,
                           int32_t kf_nodes, const int32_t* kf_node_id, const int32_t* kf_node_off, const int32_t* kf_node_feat,
                           int32_t f_nodes, const int32_t* f_node_id, const int32_t* f_node_off, const int32_t* f_node_feat,
                           float nnratio, int32_t check_ori, int32_t* assignment);
void osh_host_frame_destroy(osh_host_frame* f);
/* Switch the frame's camera to a KannalaBrandt8 (same fx fy cx cy, coefficients k1..k4). */
void osh_host_frame_set_fisheye(osh_host_frame* f, const float k[4]);
/* ORBmatcher(nnratio).SearchByProjection(F, vpMapPoints, th): map points given by their tracking scratch
 * (mTrackProjX/Y/XR, mnTrackScaleLevel, mTrackViewCos, mTrackDepth), descriptor and Observations().
 * assignment[k] = index of the map point stored in F.mvpMapPoints[k] or -1.  Returns nmatches (<0: error). */
int osh_host_search_local_points(osh_host_frame* f, int32_t n_mp, const uint8_t* mp_desc, const float* proj_xy,
                                 const float* proj_xr, const int32_t* level, const float* viewcos, const float* depth,
                                 const int32_t* n_observations, float nnratio, float th, int32_t* assignment);
/* Frame::isInFrustum(pMP, viewing_cos_limit) for every map point (the loop of Tracking::SearchLocalPoints,
 * src/Tracking.cc:3411-3432) as one device launch; outputs are the MapPoint tracking fields afterwards.  With
 * assignment != NULL it continues with ORBmatcher(nnratio).SearchByProjection(F, vpMapPoints, th) (:3460) like
 * osh_host_search_local_points.  Returns the number of points in view (<0: error). */
int osh_host_frame_search_local_points_projected(osh_host_frame* f, int32_t n_mp, const float* mp_pos, const float* mp_normal,
                                                 const float* mp_min_dist, const float* mp_max_dist, float viewing_cos_limit,
                                                 uint8_t* in_view, float* proj_xy, float* proj_xr, float* depth, float* view_cos,
                                                 int32_t* level, const uint8_t* mp_desc, const int32_t* n_observations,
                                                 float nnratio, float th, int32_t* assignment, int32_t* n_matches);
/* ORBmatcher(nnratio, check_ori).SearchByProjection(Current, Last, th, bMono): last_mp[k] = map point index held by
 * keypoint k of the last frame (-1 none); map points given by world position + descriptor. */
int osh_host_search_last_frame(osh_host_frame* cur, osh_host_frame* last, const int32_t* last_mp, int32_t n_mp,
                               const float* mp_pos, const uint8_t* mp_desc, float th, int32_t b_mono, int32_t check_ori,
                               int32_t* assignment);
/* ORBmatcher(0.9, check_ori).SearchByProjection(Current, pKF, sAlreadyFound, th, ORBdist) (relocalisation,
 * src/ORBmatcher.cc:1889-2010): the keyframe is given by its keypoint angles and kf_mp[k] = map point held by keypoint k
 * (-1 none); map points by world position, descriptor, {mfMinDistance, mfMaxDistance}, membership of sAlreadyFound, isBad();
 * cur_mp[k] = map point already held by keypoint k of the current frame (-1 none). */
int osh_host_search_keyframe(osh_host_frame* cur, int32_t n_kf, const float* kf_angle, const int32_t* kf_mp, int32_t n_mp,
                             const float* mp_pos, const uint8_t* mp_desc, const float* mp_min_max_dist,
                             const uint8_t* mp_found, const uint8_t* mp_bad, const int32_t* cur_mp, float th,
                             int32_t orb_dist, int32_t check_ori, int32_t* assignment);
/* ORBmatcher::SearchByProjection(KeyFrame*, Sim3f&, vpPoints[, vpPointsKFs], vpMatched[, vpMatchedKF], th, ratioHamming)
 * (src/ORBmatcher.cc:427-646): the keyframe is made from `f`; scw = unit quaternion (x y z w), translation, scale;
 * map points by world position, descriptor, {mfMinDistance, mfMaxDistance}, normal, isBad(); matched_in[k] = map point
 * already matched to keypoint k (-1 none).  with_keyframes selects the second overload (point j comes from keyframe j). */
int osh_host_search_sim3(osh_host_frame* f, const float scw[8], int32_t n_mp, const float* mp_pos, const uint8_t* mp_desc,
                         const float* mp_min_max_dist, const float* mp_normal, const uint8_t* mp_bad,
                         const int32_t* matched_in, int32_t th, float ratio_hamming, int32_t with_keyframes,
                         int32_t* matched_out, int32_t* matched_kf_out);
/* ORBmatcher(nnratio, check_ori).SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (src/ORBmatcher.cc:648-763):
 * prev_xy[n1][2] is vbPrevMatched (updated in place), matches12[n1] = vnMatches12. */
int osh_host_search_for_initialization(osh_host_frame* f1, osh_host_frame* f2, float* prev_xy, int32_t window, float nnratio, int32_t check_ori,
                                       int32_t* matches12);
/* ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (src/ORBmatcher.cc:907-1146) on two pinhole
 * keyframes: kp[n][4] = x, y, angle, uright (< 0: monocular), vocabulary nodes as (id, offsets, features), poses as unit quaternion
 * (x y z w) + translation of Tcw.  match12[i] = feature of keyframe 2 paired with feature i of keyframe 1 (-1 none). */
int osh_host_search_for_triangulation(const float cam4[4], int32_t n_levels, float scale_factor, int32_t n1, const float* kp1,
                                      const int32_t* octave1, const uint8_t* desc1, const uint8_t* has_mp1, const float pose1_qt[7],
                                      int32_t nodes1, const int32_t* node_id1, const int32_t* node_off1, const int32_t* node_feat1,
                                      int32_t n2, const float* kp2, const int32_t* octave2, const uint8_t* desc2, const uint8_t* has_mp2,
                                      const float pose2_qt[7], int32_t nodes2, const int32_t* node_id2, const int32_t* node_off2,
                                      const int32_t* node_feat2, int32_t only_stereo, int32_t coarse, int32_t check_ori, int32_t* match12);
/* ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:1148-1338): the keyframe is made from `f` (its pose,
 * keypoints, grid, mvuRight).  Candidate map points j (null_mask[j]: a null entry) with world position, descriptor, {mfMinDistance,
 * mfMaxDistance}, normal, isBad(), Observations(); resident map points r already sitting in keypoint slots (slot_res[k] = r or -1).
 * Ids in the outputs: candidate j -> j, resident r -> 100000 + r, none -> -1.  slot_out[k]: the keyframe's match at keypoint k
 * afterwards; *_replaced_out: GetReplaced(); *_nobs_out: Observations() afterwards.  Returns nFused. */
int osh_host_fuse(osh_host_frame* f, int32_t n_mp, const float* mp_pos, const uint8_t* mp_desc, const float* mp_min_max_dist,
                  const float* mp_normal, const uint8_t* mp_bad, const uint8_t* null_mask, const int32_t* mp_nobs,
                  int32_t n_res, const int32_t* slot_res, const int32_t* res_nobs, const uint8_t* res_bad, float th,
                  int32_t* slot_out, uint8_t* cand_bad_out, int32_t* cand_replaced_out, int32_t* cand_nobs_out,
                  uint8_t* res_bad_out, int32_t* res_replaced_out, int32_t* res_nobs_out);
/* ORBmatcher::Fuse(KeyFrame*, Sim3f&, const vector<MapPoint*>&, th, vpReplacePoint) (src/ORBmatcher.cc:1340-1455): as osh_host_fuse;
 * found_slot[j] >= 0: candidate j already sits in the keyframe at that keypoint (spAlreadyFound); replace_out[j]: id of
 * vpReplacePoint[j] (-1 null). */
int osh_host_fuse_sim3(osh_host_frame* f, const float scw[8], int32_t n_mp, const float* mp_pos, const uint8_t* mp_desc,
                       const float* mp_min_max_dist, const float* mp_normal, const uint8_t* mp_bad, const int32_t* mp_nobs,
                       const int32_t* found_slot, int32_t n_res, const int32_t* slot_res, const uint8_t* res_bad, float th,
                       int32_t* slot_out, int32_t* replace_out, int32_t* cand_nobs_out);
/* Optimizer::PoseOptimization(&frame) (src/Optimizer.cc:815-1114): kp_mp[k] = map point matched to keypoint k (-1 none);
 * returns the inlier count, the optimised pose and mvbOutlier (keypoints without a match keep the value 1 they are preset to). */
int osh_host_frame_pose_optimization(osh_host_frame* f, int32_t n_mp, const float* mp_pos, const int32_t* kp_mp,
                                     const float* inv_level_sigma2, int32_t n_levels, float pose_out[7], uint8_t* outlier_out);
/* ---- Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame on a test frame (csrc/host/harness.cc) */
typedef struct osh_host_posei osh_host_posei;
osh_host_posei* osh_host_posei_create(int32_t mode, int32_t n_kp, const float* kp_xy, const int32_t* octave, const float* uright,
                                      int32_t n_left, const float pose_qt[7], const float cam5[5], const float* kb8,
                                      const float* cam2_8, const float* trl_qt, const float* inv_level_sigma2, int32_t n_levels,
                                      const float* mp_pos, const uint8_t* mp_close, const float Tbc_qt[7], const float vel[3],
                                      const float bias6[6], const float prev_pose_qt[7], const float prev_vel[3], const float prev_bias6[6],
                                      const float* preint72, const float* cov225, const double* prior_Rwb, const double* prior_twb,
                                      const double* prior_vel, const double* prior_bg, const double* prior_ba, const double* prior_H);
void osh_host_posei_destroy(osh_host_posei* h);
int osh_host_posei_pack(osh_host_posei* h, int32_t rec_init, osh_posei_problem* out, int32_t* kp_of_edge);
int osh_host_posei_run(osh_host_posei* h, int32_t rec_init, float pose_out[7], float Rwb_out[9], float twb_out[3], float vel_out[3],
                       float bias6_out[6], uint8_t* outlier_out, double* H225_out, int32_t* prev_cpi_deleted);

/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (src/ORBmatcher.cc:1457-1674), two frames standing in for the keyframes.
 * slot_mp_k[i]: index of the map point in keypoint slot i of keyframe k (into its own point list) or -1; matches_in[i1]: keyframe-2
 * point already matched to slot i1 or -1; matches_out[i1]: keyframe-2 point index or -1.  s12 = qx qy qz qw tx ty tz s. */
int osh_host_search_by_sim3(osh_host_frame* f1, osh_host_frame* f2, const float s12[8], float th,
                            int32_t n_mp1, const float* mp_pos1, const uint8_t* mp_desc1, const float* mp_min_max1, const uint8_t* mp_bad1,
                            const int32_t* slot_mp1, int32_t n_mp2, const float* mp_pos2, const uint8_t* mp_desc2, const float* mp_min_max2,
                            const uint8_t* mp_bad2, const int32_t* slot_mp2, const int32_t* matches_in, int32_t* matches_out);

#ifdef __cplusplus
}
#endif
#endif
