// harness.cc -- extern "C" wrappers around the C++ host layer (include/orbslam3_hip_host.h).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <set>
#include <vector>

#include "Frame.h"
#include "ORBmatcher.h"
#include "Optimizer.h"
#include "host_pack.h"
#include "orbslam3_hip.h"
#include "orbslam3_hip_host.h"

using namespace ORB_SLAM3;

struct osh_host_graph {
  Map map;
  std::unique_ptr<GeometricCamera> cam, cam2;
  std::vector<std::unique_ptr<KeyFrame>> kfs;
  std::vector<std::unique_ptr<MapPoint>> mps;
  std::vector<std::unique_ptr<IMU::Preintegrated>> preints;
  LibaPack liba;   // storage behind osh_host_pack_liba
  bool last_has_kb8 = false;   // camera model of the last osh_host_pack_lba / _gba / _welding
  double last_kb8[4] = {0, 0, 0, 0};
  bool last_has_rig = false;
  double last_cam2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_trl[7] = {0, 0, 0, 1, 0, 0, 0};
};

static Sophus::SE3f pose_from(const float* qt) {
  return Sophus::SE3f(Eigen::Quaternionf(qt[3], qt[0], qt[1], qt[2]), Eigen::Vector3f(qt[4], qt[5], qt[6]));
}

// Wall time of the product call inside the last harness wrapper on this thread (the wrappers build KeyFrame / MapPoint / Frame
// objects around it; bench.py reports the call alone).
static thread_local double g_last_call_ms = 0.0;
struct CallTimer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ~CallTimer() { g_last_call_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
extern "C" double osh_host_last_call_ms(void) { return g_last_call_ms; }

extern "C" osh_host_graph* osh_host_graph_create(int32_t n_kf, const int64_t* kf_id, const float* kf_pose_qt, const float* cam5,
                                                 const float* inv_level_sigma2, int32_t n_levels, int32_t n_mp,
                                                 const int64_t* mp_id, const float* mp_pos, int32_t n_obs, const int32_t* obs_kf,
                                                 const int32_t* obs_mp, const float* obs_uvr, const int32_t* obs_octave,
                                                 int64_t init_kf_id, int32_t inertial) {
  osh_host_graph* g = new osh_host_graph();
  g->map.mnInitKFid = (unsigned long)init_kf_id;
  g->map.mbIsInertial = inertial != 0;
  g->cam.reset(new Pinhole(std::vector<float>{cam5[0], cam5[1], cam5[2], cam5[3]}));
  for (int i = 0; i < n_kf; ++i) {
    std::unique_ptr<KeyFrame> kf(new KeyFrame((unsigned long)kf_id[i], &g->map));
    kf->SetPose(pose_from(kf_pose_qt + 7 * i));
    kf->mnPoseSets = 0;
    kf->fx = cam5[0]; kf->fy = cam5[1]; kf->cx = cam5[2]; kf->cy = cam5[3]; kf->mbf = cam5[4];
    kf->mpCamera = g->cam.get();
    kf->mvInvLevelSigma2.assign(inv_level_sigma2, inv_level_sigma2 + n_levels);
    g->kfs.push_back(std::move(kf));
  }
  for (int j = 0; j < n_mp; ++j)
    g->mps.emplace_back(new MapPoint((unsigned long)mp_id[j], Eigen::Vector3f(mp_pos[3 * j], mp_pos[3 * j + 1], mp_pos[3 * j + 2]), &g->map));
  for (int o = 0; o < n_obs; ++o) {
    KeyFrame* kf = g->kfs[obs_kf[o]].get();
    MapPoint* mp = g->mps[obs_mp[o]].get();
    cv::KeyPoint kp;
    kp.pt.x = obs_uvr[3 * o]; kp.pt.y = obs_uvr[3 * o + 1]; kp.octave = obs_octave[o];
    const int idx = (int)kf->mvKeysUn.size();
    kf->mvKeysUn.push_back(kp);
    kf->mvuRight.push_back(obs_uvr[3 * o + 2]);
    kf->mvpMapPoints.push_back(mp);
    kf->N = idx + 1;
    mp->AddObservation(kf, idx);
  }
  // Map::GetAllKeyFrames / GetOriginKF: the origin keyframe is the one with the map's initial id (else the first one)
  for (auto& kf : g->kfs) {
    g->map.mvpKeyFrames.push_back(kf.get());
    g->map.mnMaxKFid = std::max(g->map.mnMaxKFid, kf->mnId);
    if (kf->mnId == g->map.mnInitKFid) g->map.mpKFinitial = kf.get();
  }
  if (!g->map.mpKFinitial && !g->kfs.empty()) g->map.mpKFinitial = g->kfs[0].get();
  for (auto& mp : g->mps) g->map.mvpMapPoints.push_back(mp.get());
  return g;
}

// Replace the map's camera by a KannalaBrandt8 with the same fx fy cx cy and coefficients k[4] (a monocular fisheye map).
extern "C" void osh_host_graph_set_fisheye(osh_host_graph* g, const float k[4]) {
  if (!g || g->kfs.empty()) return;
  KeyFrame* k0 = g->kfs[0].get();
  g->cam.reset(new KannalaBrandt8(std::vector<float>{k0->fx, k0->fy, k0->cx, k0->cy, k[0], k[1], k[2], k[3]}));
  for (auto& kf : g->kfs) kf->mpCamera = g->cam.get();
}

// Turn the map into a fisheye STEREO rig (after osh_host_graph_set_fisheye): every keyframe gets mpCamera2 = KannalaBrandt8(cam2),
// Trl, NLeft = its number of left keypoints, and the right-camera observations obs_kf/obs_mp/obs_uv/obs_octave are appended to
// mvKeysRight and registered with MapPoint::AddObservation(kf, NLeft + index) (src/MapPoint.cc:140-165).
extern "C" int osh_host_graph_set_rig(osh_host_graph* g, const float cam2[8], const float trl_qt[7], int32_t n_obs, const int32_t* obs_kf,
                                      const int32_t* obs_mp, const float* obs_uv, const int32_t* obs_octave) {
  if (!g || g->kfs.empty()) return -1;
  g->cam2.reset(new KannalaBrandt8(std::vector<float>(cam2, cam2 + 8)));
  for (auto& kf : g->kfs) {
    kf->mpCamera2 = g->cam2.get();
    kf->mTrl = pose_from(trl_qt);
    kf->NLeft = (int)kf->mvKeysUn.size();
  }
  for (int o = 0; o < n_obs; ++o) {
    KeyFrame* kf = g->kfs[obs_kf[o]].get();
    MapPoint* mp = g->mps[obs_mp[o]].get();
    cv::KeyPoint kp;
    kp.pt.x = obs_uv[2 * o]; kp.pt.y = obs_uv[2 * o + 1]; kp.octave = obs_octave[o];
    const int ridx = (int)kf->mvKeysRight.size();
    kf->mvKeysRight.push_back(kp);
    // the keyframe's match table covers left and right keypoints (indices >= NLeft are right ones)
    if ((int)kf->mvpMapPoints.size() < kf->NLeft + ridx + 1) kf->mvpMapPoints.resize(kf->NLeft + ridx + 1, nullptr);
    kf->mvpMapPoints[kf->NLeft + ridx] = mp;
    kf->mvuRight.resize(kf->mvpMapPoints.size(), -1.f);
    mp->AddObservation(kf, kf->NLeft + ridx);
  }
  return 0;
}

extern "C" void osh_host_graph_destroy(osh_host_graph* g) { delete g; }

extern "C" int osh_host_graph_set_covisible(osh_host_graph* g, int32_t kf_index, int32_t n, const int32_t* kf_indices) {
  if (!g || kf_index < 0 || kf_index >= (int)g->kfs.size()) return -1;
  auto& v = g->kfs[kf_index]->mvpOrderedConnectedKeyFrames;
  v.clear();
  for (int i = 0; i < n; ++i) v.push_back(g->kfs[kf_indices[i]].get());
  return 0;
}

extern "C" int osh_host_pack_lba(osh_host_graph* g, int32_t kf_index, int32_t sizes[5], double* pose_qt, double* pose_cam,
                                 double* points, int32_t* edge_pose, int32_t* edge_point, uint8_t* edge_kind, double* edge_obs,
                                 double* edge_info, int64_t* pose_kf_id, int64_t* point_mp_id) {
  if (!g || kf_index < 0 || kf_index >= (int)g->kfs.size()) return -1;
  LbaPack pk;
  const bool ok = PackLocalBA(g->kfs[kf_index].get(), &g->map, pk);
  // PackLocalBA marks the graph like the reference does; undo so the call can be repeated
  for (auto& kf : g->kfs) { kf->mnBALocalForKF = 0; kf->mnBAFixedForKF = 0; }
  for (auto& mp : g->mps) mp->mnBALocalForKF = 0;
  sizes[0] = pk.n_free; sizes[1] = pk.n_fixed; sizes[2] = (int32_t)pk.vPointMPs.size(); sizes[3] = (int32_t)pk.edge_pose.size();
  sizes[4] = pk.num_fixedKF;
  if (!ok) return 1;
  if (pk.unsupported) return -3;
  g->last_has_kb8 = pk.has_kb8; for (int k = 0; k < 4; ++k) g->last_kb8[k] = pk.kb8[k];
  g->last_has_rig = pk.has_rig; for (int k = 0; k < 8; ++k) g->last_cam2[k] = pk.cam2[k]; for (int k = 0; k < 7; ++k) g->last_trl[k] = pk.trl[k];
  auto cp = [](auto* dst, const auto& src) { if (dst) std::copy(src.begin(), src.end(), dst); };
  cp(pose_qt, pk.pose_qt); cp(pose_cam, pk.pose_cam); cp(points, pk.points); cp(edge_pose, pk.edge_pose);
  cp(edge_point, pk.edge_point); cp(edge_kind, pk.edge_kind); cp(edge_obs, pk.edge_obs); cp(edge_info, pk.edge_info);
  if (pose_kf_id) for (size_t i = 0; i < pk.vPoseKFs.size(); ++i) pose_kf_id[i] = (int64_t)pk.vPoseKFs[i]->mnId;
  if (point_mp_id) for (size_t j = 0; j < pk.vPointMPs.size(); ++j) point_mp_id[j] = (int64_t)pk.vPointMPs[j]->mnId;
  return 0;
}

extern "C" int osh_host_run_lba(osh_host_graph* g, int32_t kf_index, unsigned char* stop_flag, int32_t counts[4]) {
  CallTimer timed;
  if (!g || kf_index < 0 || kf_index >= (int)g->kfs.size()) return -1;
  static_assert(sizeof(bool) == 1, "bool is one byte");
  int a = -1, b = -1, c = -1, d = -1;
  Optimizer::LocalBundleAdjustment(g->kfs[kf_index].get(), reinterpret_cast<bool*>(stop_flag), &g->map, a, b, c, d);
  counts[0] = a; counts[1] = b; counts[2] = c; counts[3] = d;
  return 0;
}

// Camera model of the window the last pack call produced: returns 1 and fills k[4] for a KannalaBrandt8 window, else 0.
extern "C" int osh_host_last_pack_kb8(osh_host_graph* g, double k[4]) {
  if (!g || !g->last_has_kb8) return 0;
  for (int i = 0; i < 4; ++i) k[i] = g->last_kb8[i];
  return 1;
}

// Rig of the window the last pack call produced: returns 1 and fills cam2[8], trl[7] for a fisheye stereo rig window, else 0.
extern "C" int osh_host_last_pack_rig(osh_host_graph* g, double cam2[8], double trl[7]) {
  if (!g || !g->last_has_rig) return 0;
  for (int i = 0; i < 8; ++i) cam2[i] = g->last_cam2[i];
  for (int i = 0; i < 7; ++i) trl[i] = g->last_trl[i];
  return 1;
}

// ---- Optimizer::GlobalBundleAdjustemnt (csrc/host/OptimizerGlobal.cc)
extern "C" int osh_host_pack_gba(osh_host_graph* g, int32_t sizes[5], double* pose_qt, double* pose_cam, double* points,
                                 int32_t* edge_pose, int32_t* edge_point, uint8_t* edge_kind, double* edge_obs, double* edge_info,
                                 int64_t* pose_kf_id, int64_t* point_mp_id) {
  if (!g) return -1;
  LbaPack pk;
  std::vector<bool> notIncluded;
  PackBundleAdjustment(g->map.GetAllKeyFrames(), g->map.GetAllMapPoints(), pk, notIncluded);
  sizes[0] = pk.n_free; sizes[1] = pk.n_fixed; sizes[2] = (int32_t)pk.vPointMPs.size(); sizes[3] = (int32_t)pk.edge_pose.size();
  sizes[4] = (int32_t)std::count(notIncluded.begin(), notIncluded.end(), true);
  if (pk.unsupported) return -3;
  g->last_has_kb8 = pk.has_kb8; for (int k = 0; k < 4; ++k) g->last_kb8[k] = pk.kb8[k];
  g->last_has_rig = pk.has_rig; for (int k = 0; k < 8; ++k) g->last_cam2[k] = pk.cam2[k]; for (int k = 0; k < 7; ++k) g->last_trl[k] = pk.trl[k];
  auto cp = [](auto* dst, const auto& src) { if (dst) std::copy(src.begin(), src.end(), dst); };
  cp(pose_qt, pk.pose_qt); cp(pose_cam, pk.pose_cam); cp(points, pk.points); cp(edge_pose, pk.edge_pose);
  cp(edge_point, pk.edge_point); cp(edge_kind, pk.edge_kind); cp(edge_obs, pk.edge_obs); cp(edge_info, pk.edge_info);
  if (pose_kf_id) for (size_t i = 0; i < pk.vPoseKFs.size(); ++i) pose_kf_id[i] = (int64_t)pk.vPoseKFs[i]->mnId;
  if (point_mp_id) for (size_t j = 0; j < pk.vPointMPs.size(); ++j) point_mp_id[j] = (int64_t)pk.vPointMPs[j]->mnId;
  return 0;
}

// ---- welding Optimizer::LocalBundleAdjustment(pMainKF, vpAdjustKF, vpFixedKF, pbStopFlag)
static void kf_lists(osh_host_graph* g, int32_t n_adj, const int32_t* adj, int32_t n_fix, const int32_t* fix,
                     std::vector<KeyFrame*>& A, std::vector<KeyFrame*>& F) {
  for (int i = 0; i < n_adj; ++i) A.push_back(g->kfs[adj[i]].get());
  for (int i = 0; i < n_fix; ++i) F.push_back(g->kfs[fix[i]].get());
}
extern "C" int osh_host_pack_welding(osh_host_graph* g, int32_t main_index, int32_t n_adj, const int32_t* adj, int32_t n_fix,
                                     const int32_t* fix, int32_t sizes[5], double* pose_qt, double* pose_cam, double* points,
                                     int32_t* edge_pose, int32_t* edge_point, uint8_t* edge_kind, double* edge_obs, double* edge_info,
                                     int64_t* pose_kf_id, int64_t* point_mp_id) {
  if (!g) return -1;
  std::vector<KeyFrame*> A, F;
  kf_lists(g, n_adj, adj, n_fix, fix, A, F);
  LbaPack pk;
  std::vector<MapPoint*> vpMPs;
  PackWeldingBA(g->kfs[main_index].get(), A, F, pk, vpMPs);
  for (auto& kf : g->kfs) kf->mnBALocalForMerge = 0;   // undo the marks so the call can be repeated
  for (auto& mp : g->mps) mp->mnBALocalForMerge = 0;
  sizes[0] = pk.n_free; sizes[1] = pk.n_fixed; sizes[2] = (int32_t)pk.vPointMPs.size(); sizes[3] = (int32_t)pk.edge_pose.size(); sizes[4] = 0;
  if (pk.unsupported) return -3;
  g->last_has_kb8 = pk.has_kb8; for (int k = 0; k < 4; ++k) g->last_kb8[k] = pk.kb8[k];
  g->last_has_rig = pk.has_rig; for (int k = 0; k < 8; ++k) g->last_cam2[k] = pk.cam2[k]; for (int k = 0; k < 7; ++k) g->last_trl[k] = pk.trl[k];
  auto cp = [](auto* dst, const auto& src) { if (dst) std::copy(src.begin(), src.end(), dst); };
  cp(pose_qt, pk.pose_qt); cp(pose_cam, pk.pose_cam); cp(points, pk.points); cp(edge_pose, pk.edge_pose);
  cp(edge_point, pk.edge_point); cp(edge_kind, pk.edge_kind); cp(edge_obs, pk.edge_obs); cp(edge_info, pk.edge_info);
  if (pose_kf_id) for (size_t i = 0; i < pk.vPoseKFs.size(); ++i) pose_kf_id[i] = (int64_t)pk.vPoseKFs[i]->mnId;
  if (point_mp_id) for (size_t j = 0; j < pk.vPointMPs.size(); ++j) point_mp_id[j] = (int64_t)pk.vPointMPs[j]->mnId;
  return 0;
}
extern "C" int osh_host_run_welding(osh_host_graph* g, int32_t main_index, int32_t n_adj, const int32_t* adj, int32_t n_fix,
                                    const int32_t* fix, unsigned char* stop_flag) {
  if (!g) return -1;
  std::vector<KeyFrame*> A, F;
  kf_lists(g, n_adj, adj, n_fix, fix, A, F);
  for (auto& kf : g->kfs) kf->mnBALocalForMerge = 0;
  for (auto& mp : g->mps) mp->mnBALocalForMerge = 0;
  Optimizer::LocalBundleAdjustment(g->kfs[main_index].get(), A, F, reinterpret_cast<bool*>(stop_flag));
  return 0;
}

extern "C" int osh_host_run_gba(osh_host_graph* g, int32_t n_iterations, unsigned char* stop_flag, int64_t n_loop_kf, int32_t robust) {
  if (!g) return -1;
  Optimizer::GlobalBundleAdjustemnt(&g->map, n_iterations, reinterpret_cast<bool*>(stop_flag), (unsigned long)n_loop_kf, robust != 0);
  return 0;
}

extern "C" int64_t osh_host_get_kf_pose_gba(osh_host_graph* g, int32_t i, float o[7]) {
  const Sophus::SE3f T = g->kfs[i]->mTcwGBA;
  o[0] = T.unit_quaternion().x(); o[1] = T.unit_quaternion().y(); o[2] = T.unit_quaternion().z(); o[3] = T.unit_quaternion().w();
  o[4] = T.translation()(0); o[5] = T.translation()(1); o[6] = T.translation()(2);
  return (int64_t)g->kfs[i]->mnBAGlobalForKF;
}
extern "C" int64_t osh_host_get_mp_pos_gba(osh_host_graph* g, int32_t j, float o[3]) {
  const Eigen::Vector3f p = g->mps[j]->mPosGBA;
  o[0] = p(0); o[1] = p(1); o[2] = p(2);
  return (int64_t)g->mps[j]->mnBAGlobalForKF;
}
extern "C" int osh_host_mp_normal_updates(osh_host_graph* g, int32_t j) { return g->mps[j]->mnNormalUpdates; }
extern "C" void osh_host_set_bad(osh_host_graph* g, int32_t kf_index, int32_t mp_index) {
  if (kf_index >= 0) g->kfs[kf_index]->mbBad = true;
  if (mp_index >= 0) g->mps[mp_index]->mbBad = true;
}

extern "C" void osh_host_get_kf_pose(osh_host_graph* g, int32_t i, float o[7]) {
  const Sophus::SE3f T = g->kfs[i]->GetPose();
  o[0] = T.unit_quaternion().x(); o[1] = T.unit_quaternion().y(); o[2] = T.unit_quaternion().z(); o[3] = T.unit_quaternion().w();
  o[4] = T.translation()(0); o[5] = T.translation()(1); o[6] = T.translation()(2);
}
extern "C" void osh_host_get_mp_pos(osh_host_graph* g, int32_t j, float o[3]) {
  const Eigen::Vector3f p = g->mps[j]->GetWorldPos();
  o[0] = p(0); o[1] = p(1); o[2] = p(2);
}
extern "C" int osh_host_mp_num_observations(osh_host_graph* g, int32_t j) { return (int)g->mps[j]->GetObservations().size(); }
extern "C" int osh_host_mp_is_bad(osh_host_graph* g, int32_t j) { return g->mps[j]->isBad() ? 1 : 0; }
extern "C" int osh_host_kf_num_matches(osh_host_graph* g, int32_t i) {
  int n = 0;
  for (MapPoint* p : g->kfs[i]->GetMapPointMatches()) n += p != nullptr;
  return n;
}
extern "C" int osh_host_kf_observes(osh_host_graph* g, int32_t i, int32_t j) {
  return g->mps[j]->GetObservations().count(g->kfs[i].get()) ? 1 : 0;
}
extern "C" int osh_host_map_change_index(osh_host_graph* g) { return g->map.GetMapChangeIndex(); }
extern "C" int osh_host_kf_pose_sets(osh_host_graph* g, int32_t i) { return g->kfs[i]->mnPoseSets; }

// ------------------------------------------------------------------------------------------ inertial
extern "C" int osh_host_graph_set_inertial(osh_host_graph* g, int32_t n, const int32_t* kf_index, const int32_t* prev_index,
                                           const float* vel, const float* bias6, const float* preint, const float* cov225,
                                           const float* Tbc_qt) {
  if (!g) return -1;
  IMU::Calib calib(pose_from(Tbc_qt), 0.f, 0.f, 0.f, 0.f);
  g->map.mnKeyFrames = g->kfs.size();
  g->map.mbIsInertial = true;
  for (auto& kf : g->kfs) { kf->mImuCalib = calib; kf->SetPose(kf->GetPose()); kf->mnPoseSets = 0; }
  for (int i = 0; i < n; ++i) {
    KeyFrame* kf = g->kfs[kf_index[i]].get();
    kf->bImu = true;
    kf->mPrevKF = prev_index[i] >= 0 ? g->kfs[prev_index[i]].get() : nullptr;
    if (kf->mPrevKF) kf->mPrevKF->mNextKF = kf;
    kf->SetVelocity(Eigen::Vector3f(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]));
    kf->mImuBias = IMU::Bias(bias6[6 * i], bias6[6 * i + 1], bias6[6 * i + 2], bias6[6 * i + 3], bias6[6 * i + 4], bias6[6 * i + 5]);
    const float* r = preint + (size_t)i * OSH_PREINT_FLOATS;
    if (r[0] > 0.f) {
      g->preints.emplace_back(new IMU::Preintegrated(IMU::Bias(r[61], r[62], r[63], r[64], r[65], r[66]), calib));
      IMU::Preintegrated* P = g->preints.back().get();
      P->dT = r[0];
      for (int a = 0; a < 9; ++a) { P->dR.v[a] = r[1 + a]; P->JRg.v[a] = r[16 + a]; P->JVg.v[a] = r[25 + a]; P->JVa.v[a] = r[34 + a]; P->JPg.v[a] = r[43 + a]; P->JPa.v[a] = r[52 + a]; }
      for (int a = 0; a < 3; ++a) { P->dV(a) = r[10 + a]; P->dP(a) = r[13 + a]; }
      for (int a = 0; a < 225; ++a) P->C.v[a] = cov225[(size_t)i * 225 + a];
      kf->mpImuPreintegrated = P;
    }
  }
  return 0;
}

static void reset_marks(osh_host_graph* g) {
  for (auto& kf : g->kfs) { kf->mnBALocalForKF = 0; kf->mnBAFixedForKF = 0; }
  for (auto& mp : g->mps) mp->mnBALocalForKF = 0;
}

extern "C" int osh_host_pack_liba(osh_host_graph* g, int32_t kf_index, int32_t b_large, int32_t b_rec_init, osh_liba_problem* out,
                                  int64_t* pose_kf_id, int64_t* point_mp_id) {
  if (!g || !out || kf_index < 0 || kf_index >= (int)g->kfs.size()) return -1;
  const bool ok = PackLocalInertialBA(g->kfs[kf_index].get(), &g->map, b_large != 0, b_rec_init != 0, g->liba);
  reset_marks(g);
  if (!ok) return 1;
  if (g->liba.unsupported) return -3;
  g->liba.fill(*out);
  if (pose_kf_id) for (size_t i = 0; i < g->liba.vPoseKFs.size(); ++i) pose_kf_id[i] = (int64_t)g->liba.vPoseKFs[i]->mnId;
  if (point_mp_id) for (size_t j = 0; j < g->liba.vPointMPs.size(); ++j) point_mp_id[j] = (int64_t)g->liba.vPointMPs[j]->mnId;
  return 0;
}

extern "C" int osh_host_run_liba(osh_host_graph* g, int32_t kf_index, int32_t b_large, int32_t b_rec_init) {
  if (!g || kf_index < 0 || kf_index >= (int)g->kfs.size()) return -1;
  int a = -1, b = -1, c = -1, d = -1;
  CallTimer timed;
  Optimizer::LocalInertialBA(g->kfs[kf_index].get(), nullptr, &g->map, a, b, c, d, b_large != 0, b_rec_init != 0);
  return (a == -1 && b == -1 && c == -1 && d == -1) ? 0 : 2;   // the reference never assigns the num_* outputs
}

// ------------------------------------------------------------------------------------------ FullInertialBA / MergeInertialBA
static void export_pack(osh_host_graph* g, osh_liba_problem* out, int64_t* pose_kf_id, int64_t* point_mp_id) {
  g->liba.fill(*out);
  out->max_iterations = g->liba.opt_it;
  if (pose_kf_id) for (int i = 0; i < out->n_opt + out->n_fixed_imu + out->n_fixed; ++i) pose_kf_id[i] = -1;   // -1: a virtual keyframe (bInit)
  if (pose_kf_id) for (size_t i = 0; i < g->liba.vPoseKFs.size(); ++i) pose_kf_id[i] = (int64_t)g->liba.vPoseKFs[i]->mnId;
  if (point_mp_id) for (size_t j = 0; j < g->liba.vPointMPs.size(); ++j) point_mp_id[j] = (int64_t)g->liba.vPointMPs[j]->mnId;
}

// the problem Optimizer::FullInertialBA(&map, its, bFixLocal, ., ., bInit) solves; *n_idle = keyframes no edge touches
extern "C" int osh_host_pack_full_inertial(osh_host_graph* g, int32_t its, int32_t fix_local, int32_t b_init, float prior_g, float prior_a,
                                           osh_liba_problem* out, int64_t* pose_kf_id, int64_t* point_mp_id, int32_t* n_idle) {
  if (!g || !out) return -1;
  std::vector<KeyFrame*> idle;
  std::vector<MapPoint*> all;
  const bool ok = PackFullInertialBA(&g->map, its, fix_local != 0, b_init != 0, prior_g, prior_a, g->liba, idle, all, nullptr);
  if (g->liba.unsupported) return -3;
  if (!ok) return 1;
  export_pack(g, out, pose_kf_id, point_mp_id);
  if (n_idle) *n_idle = (int32_t)idle.size();
  return 0;
}

extern "C" int osh_host_run_full_inertial(osh_host_graph* g, int32_t its, int32_t fix_local, int64_t loop_id, int32_t b_init, float prior_g, float prior_a) {
  if (!g) return -1;
  Optimizer::FullInertialBA(&g->map, its, fix_local != 0, (unsigned long)loop_id, nullptr, b_init != 0, prior_g, prior_a);
  return 0;
}

// what FullInertialBA leaves beside the live state when nLoopId != 0: returns mnBAGlobalForKF
extern "C" int64_t osh_host_get_kf_inertial_gba(osh_host_graph* g, int32_t i, float vel[3], float bias6[6]) {
  const KeyFrame* k = g->kfs[i].get();
  for (int a = 0; a < 3; ++a) vel[a] = k->mVwbGBA(a);
  bias6[0] = k->mBiasGBA.bax; bias6[1] = k->mBiasGBA.bay; bias6[2] = k->mBiasGBA.baz; bias6[3] = k->mBiasGBA.bwx; bias6[4] = k->mBiasGBA.bwy; bias6[5] = k->mBiasGBA.bwz;
  return (int64_t)k->mnBAGlobalForKF;
}

// the problem Optimizer::MergeInertialBA(curr, merge, ...) solves; n_sets = {temporal keyframes, covisible keyframes}
extern "C" int osh_host_pack_merge_inertial(osh_host_graph* g, int32_t curr, int32_t merge, osh_liba_problem* out, int64_t* pose_kf_id,
                                            int64_t* point_mp_id, int32_t n_sets[2], int64_t* temporal_kf_id, int64_t* cov_kf_id) {
  if (!g || !out || curr < 0 || curr >= (int)g->kfs.size() || merge < 0 || merge >= (int)g->kfs.size()) return -1;
  std::vector<KeyFrame*> cov;
  const bool ok = PackMergeInertialBA(g->kfs[curr].get(), g->kfs[merge].get(), g->liba, cov);
  reset_marks(g);
  if (g->liba.unsupported) return -3;
  if (!ok) return 1;
  export_pack(g, out, pose_kf_id, point_mp_id);
  out->lambda_init = 1e3;
  if (n_sets) { n_sets[0] = (int32_t)g->liba.vpOptimizableKFs.size(); n_sets[1] = (int32_t)cov.size(); }
  if (temporal_kf_id) for (size_t i = 0; i < g->liba.vpOptimizableKFs.size(); ++i) temporal_kf_id[i] = (int64_t)g->liba.vpOptimizableKFs[i]->mnId;
  if (cov_kf_id) for (size_t i = 0; i < cov.size(); ++i) cov_kf_id[i] = (int64_t)cov[i]->mnId;
  return 0;
}

// Optimizer::MergeInertialBA; returns the size of corrPoses and copies up to max_corr entries {keyframe id; qx qy qz qw tx ty tz s}
extern "C" int osh_host_run_merge_inertial(osh_host_graph* g, int32_t curr, int32_t merge, int32_t max_corr, int64_t* corr_kf_id, double* corr_sim3) {
  if (!g || curr < 0 || curr >= (int)g->kfs.size() || merge < 0 || merge >= (int)g->kfs.size()) return -1;
  LoopClosing::KeyFrameAndPose corr;
  Optimizer::MergeInertialBA(g->kfs[curr].get(), g->kfs[merge].get(), nullptr, &g->map, corr);
  reset_marks(g);
  int n = 0;
  for (const auto& kv : corr) {
    if (n < max_corr) {
      corr_kf_id[n] = (int64_t)kv.first->mnId;
      double* o = corr_sim3 + 8 * (size_t)n;
      o[0] = kv.second.rotation().x(); o[1] = kv.second.rotation().y(); o[2] = kv.second.rotation().z(); o[3] = kv.second.rotation().w();
      for (int a = 0; a < 3; ++a) o[4 + a] = kv.second.translation()(a);
      o[7] = kv.second.scale();
    }
    ++n;
  }
  return n;
}

extern "C" void osh_host_get_kf_velocity(osh_host_graph* g, int32_t i, float o[3]) {
  const Eigen::Vector3f v = g->kfs[i]->GetVelocity();
  o[0] = v(0); o[1] = v(1); o[2] = v(2);
}
extern "C" void osh_host_get_kf_bias(osh_host_graph* g, int32_t i, float o[6]) {
  const IMU::Bias b = g->kfs[i]->GetImuBias();
  o[0] = b.bax; o[1] = b.bay; o[2] = b.baz; o[3] = b.bwx; o[4] = b.bwy; o[5] = b.bwz;
}

extern "C" int osh_host_preintegrate(int32_t n, const float* acc, const float* gyr, float dt, const float* bias6, const float* nga6,
                                     const float* walk6, float* r, float* cov225_out) {
  IMU::Calib calib;
  for (int i = 0; i < 6; ++i) { calib.Cov[i] = nga6[i]; calib.CovWalk[i] = walk6[i]; }
  IMU::Preintegrated P(IMU::Bias(bias6[0], bias6[1], bias6[2], bias6[3], bias6[4], bias6[5]), calib);
  for (int k = 0; k < n; ++k)
    P.IntegrateNewMeasurement(Eigen::Vector3f(acc[3 * k], acc[3 * k + 1], acc[3 * k + 2]), Eigen::Vector3f(gyr[3 * k], gyr[3 * k + 1], gyr[3 * k + 2]), dt);
  for (int a = 0; a < OSH_PREINT_FLOATS; ++a) r[a] = 0.f;
  r[0] = P.dT;
  for (int a = 0; a < 9; ++a) { r[1 + a] = P.dR.v[a]; r[16 + a] = P.JRg.v[a]; r[25 + a] = P.JVg.v[a]; r[34 + a] = P.JVa.v[a]; r[43 + a] = P.JPg.v[a]; r[52 + a] = P.JPa.v[a]; }
  for (int a = 0; a < 3; ++a) { r[10 + a] = P.dV(a); r[13 + a] = P.dP(a); }
  r[61] = P.b.bax; r[62] = P.b.bay; r[63] = P.b.baz; r[64] = P.b.bwx; r[65] = P.b.bwy; r[66] = P.b.bwz;
  for (int a = 0; a < 225; ++a) cov225_out[a] = P.C.v[a];
  return 0;
}

extern "C" int osh_host_inertial_information(const float* cov225, double* info81_out) {
  Eigen::Matrix<float, 15, 15> C;
  for (int a = 0; a < 225; ++a) C.v[a] = cov225[a];
  InertialInformation(C, info81_out);
  return 0;
}

// ------------------------------------------------------------------------------------------ matcher
struct osh_host_frame {
  Frame F;
  std::unique_ptr<GeometricCamera> cam, cam2;
  Map map;
};

extern "C" osh_host_frame* osh_host_frame_create(int32_t n, const float* kp_xy, const int32_t* octave, const float* angle,
                                                 const float* uright, const uint8_t* desc, const float pose_qt[7],
                                                 const float cam4[4], float mbf, float mb, int32_t n_levels, float scale_factor) {
  osh_host_frame* f = new osh_host_frame();
  Frame& F = f->F;
  f->cam.reset(new Pinhole(std::vector<float>{cam4[0], cam4[1], cam4[2], cam4[3]}));
  F.mpCamera = f->cam.get();
  F.N = n; F.mbf = mbf; F.mb = mb;
  F.mTcw = pose_from(pose_qt);
  F.UpdatePoseMatrices();
  F.fx = cam4[0]; F.fy = cam4[1]; F.cx = cam4[2]; F.cy = cam4[3];
  F.mvScaleFactors.assign(n_levels, 1.0f);
  for (int l = 1; l < n_levels; ++l) F.mvScaleFactors[l] = F.mvScaleFactors[l - 1] * scale_factor;  // src/ORBextractor.cc:414-422
  F.mnScaleLevels = n_levels;
  F.mfLogScaleFactor = std::log(scale_factor);   // src/Frame.cc:75 (float log of the float factor)
  F.mDescriptors = cv::Mat(n, 32);
  for (int i = 0; i < n; ++i) {
    cv::KeyPoint kp;
    kp.pt.x = kp_xy[2 * i]; kp.pt.y = kp_xy[2 * i + 1]; kp.octave = octave[i]; kp.angle = angle ? angle[i] : 0.f;
    F.mvKeys.push_back(kp); F.mvKeysUn.push_back(kp);
    F.mvuRight.push_back(uright ? uright[i] : -1.f);
    std::memcpy(F.mDescriptors.ptr<uint8_t>(i), desc + 32 * (size_t)i, 32);
  }
  F.mvpMapPoints.assign(n, nullptr);
  F.mvbOutlier.assign(n, false);
  F.AssignFeaturesToGrid();
  return f;
}
// Replace the frame's camera by a KannalaBrandt8 with the same fx fy cx cy and coefficients k[4] (monocular fisheye frame).
extern "C" void osh_host_frame_set_fisheye(osh_host_frame* f, const float k[4]) {
  if (!f) return;
  f->cam.reset(new KannalaBrandt8(std::vector<float>{f->cam->getParameter(0), f->cam->getParameter(1), f->cam->getParameter(2),
                                                     f->cam->getParameter(3), k[0], k[1], k[2], k[3]}));
  f->F.mpCamera = f->cam.get();
}
// Re-interpret the frame as a fisheye STEREO frame: the first n_left keypoints are the left camera's (mvKeys), the remaining
// ones the right camera's (mvKeysRight, descriptor rows [n_left, N)); stereo matches between the two sets and Trl as given.
extern "C" int osh_host_frame_set_rig(osh_host_frame* f, int32_t n_left, const int32_t* left_to_right, const int32_t* right_to_left,
                                      const float trl_qt[7]) {
  if (!f || n_left < 0 || n_left > f->F.N) return -1;
  Frame& F = f->F;
  F.Nleft = n_left; F.Nright = F.N - n_left;
  F.mvKeysRight.assign(F.mvKeys.begin() + n_left, F.mvKeys.end());
  F.mvKeys.resize(n_left);
  F.mvLeftToRightMatch.assign(left_to_right, left_to_right + n_left);
  F.mvRightToLeftMatch.assign(right_to_left, right_to_left + F.Nright);
  F.mTrl = pose_from(trl_qt);
  F.mpCamera2 = F.mpCamera;
  F.AssignFeaturesToGrid();
  return 0;
}

// The right camera of a fisheye stereo frame as its own KannalaBrandt8 (after osh_host_frame_set_rig, which shares the left one).
extern "C" int osh_host_frame_set_camera2(osh_host_frame* f, const float cam2[8]) {
  if (!f) return -1;
  f->cam2.reset(new KannalaBrandt8(std::vector<float>(cam2, cam2 + 8)));
  f->F.mpCamera2 = f->cam2.get();
  return 0;
}

extern "C" void osh_host_frame_destroy(osh_host_frame* f) { delete f; }

static std::vector<std::unique_ptr<MapPoint>> make_points(Map* map, int32_t n_mp, const uint8_t* mp_desc, const float* pos,
                                                          const int32_t* n_observations) {
  std::vector<std::unique_ptr<MapPoint>> v;
  for (int j = 0; j < n_mp; ++j) {
    Eigen::Vector3f p(0, 0, 0);
    if (pos) p = Eigen::Vector3f(pos[3 * j], pos[3 * j + 1], pos[3 * j + 2]);
    v.emplace_back(new MapPoint((unsigned long)j, p, map));
    v.back()->mDescriptor = cv::Mat(1, 32);
    std::memcpy(v.back()->mDescriptor.ptr<uint8_t>(0), mp_desc + 32 * (size_t)j, 32);
    v.back()->nObs = n_observations ? n_observations[j] : 1;
  }
  return v;
}

// SearchByProjection(Frame&, vector<MapPoint*>) on a fisheye stereo frame (after osh_host_frame_set_rig): per map point the
// tracking fields of BOTH cameras as Frame::isInFrustumChecks leaves them (src/Frame.cc:589-656).
extern "C" int osh_host_search_local_points_rig(osh_host_frame* f, int32_t n_mp, const uint8_t* mp_desc, const uint8_t* in_left,
                                                const float* proj_left, const int32_t* level_left, const float* viewcos_left,
                                                const uint8_t* in_right, const float* proj_right, const int32_t* level_right,
                                                const float* viewcos_right, const int32_t* n_observations, float nnratio, float th,
                                                int32_t* assignment) {
  if (!f || f->F.Nleft == -1) return -1;
  auto pts = make_points(&f->map, n_mp, mp_desc, nullptr, n_observations);
  std::vector<MapPoint*> vp;
  for (int j = 0; j < n_mp; ++j) {
    MapPoint* p = pts[j].get();
    p->mbTrackInView = in_left[j] != 0;
    p->mTrackProjX = proj_left[2 * j]; p->mTrackProjY = proj_left[2 * j + 1];
    p->mnTrackScaleLevel = level_left[j];
    p->mTrackViewCos = viewcos_left[j];
    p->mbTrackInViewR = in_right[j] != 0;
    p->mTrackProjXR = proj_right[2 * j]; p->mTrackProjYR = proj_right[2 * j + 1];
    p->mnTrackScaleLevelR = level_right[j];
    p->mTrackViewCosR = viewcos_right[j];
    p->mTrackDepth = 1.f;
    vp.push_back(p);
  }
  f->F.mvpMapPoints.assign(f->F.N, nullptr);
  ORBmatcher matcher(nnratio);
  int n;
  { CallTimer timed; n = matcher.SearchByProjection(f->F, vp, th); }
  for (int k = 0; k < f->F.N; ++k) assignment[k] = f->F.mvpMapPoints[k] ? (int32_t)f->F.mvpMapPoints[k]->mnId : -1;
  f->F.mvpMapPoints.assign(f->F.N, nullptr);
  return n;
}

extern "C" int osh_host_search_local_points(osh_host_frame* f, int32_t n_mp, const uint8_t* mp_desc, const float* proj_xy,
                                            const float* proj_xr, const int32_t* level, const float* viewcos, const float* depth,
                                            const int32_t* n_observations, float nnratio, float th, int32_t* assignment) {
  if (!f) return -1;
  auto pts = make_points(&f->map, n_mp, mp_desc, nullptr, n_observations);
  std::vector<MapPoint*> vp;
  for (int j = 0; j < n_mp; ++j) {
    MapPoint* p = pts[j].get();
    p->mbTrackInView = true;
    p->mTrackProjX = proj_xy[2 * j]; p->mTrackProjY = proj_xy[2 * j + 1];
    p->mTrackProjXR = proj_xr ? proj_xr[j] : 0.f;
    p->mnTrackScaleLevel = level[j];
    p->mTrackViewCos = viewcos ? viewcos[j] : 1.f;
    p->mTrackDepth = depth ? depth[j] : 1.f;
    vp.push_back(p);
  }
  f->F.mvpMapPoints.assign(f->F.N, nullptr);
  ORBmatcher matcher(nnratio);
  int n;
  { CallTimer timed; n = matcher.SearchByProjection(f->F, vp, th); }
  for (int k = 0; k < f->F.N; ++k) assignment[k] = f->F.mvpMapPoints[k] ? (int32_t)f->F.mvpMapPoints[k]->mnId : -1;
  f->F.mvpMapPoints.assign(f->F.N, nullptr);
  return n;
}

// Frame::isInFrustum over a list of map points (the loop of Tracking::SearchLocalPoints, src/Tracking.cc:3411-3432),
// optionally followed by ORBmatcher(nnratio).SearchByProjection(F, vpMapPoints, th) on the points it put in view (:3460).
extern "C" int osh_host_frame_search_local_points_projected(osh_host_frame* f, int32_t n_mp, const float* mp_pos, const float* mp_normal,
                                                            const float* mp_min_dist, const float* mp_max_dist, float viewing_cos_limit,
                                                            uint8_t* in_view, float* proj_xy, float* proj_xr, float* depth,
                                                            float* view_cos, int32_t* level, const uint8_t* mp_desc,
                                                            const int32_t* n_observations, float nnratio, float th,
                                                            int32_t* assignment, int32_t* n_matches) {
  if (!f) return -1;
  std::vector<uint8_t> zero_desc;
  if (!mp_desc) { zero_desc.assign((size_t)n_mp * 32, 0); }
  auto pts = make_points(&f->map, n_mp, mp_desc ? mp_desc : zero_desc.data(), mp_pos, n_observations);
  std::vector<MapPoint*> vp;
  for (int j = 0; j < n_mp; ++j) {
    MapPoint* p = pts[j].get();
    p->mNormalVector = Eigen::Vector3f(mp_normal[3 * j], mp_normal[3 * j + 1], mp_normal[3 * j + 2]);
    p->mfMinDistance = mp_min_dist[j]; p->mfMaxDistance = mp_max_dist[j];
    p->mnTrackScaleLevel = -1; p->mTrackViewCos = 0.f; p->mTrackProjXR = 0.f; p->mTrackDepth = 0.f;
    vp.push_back(p);
  }
  std::vector<bool> in;
  const int n_in = f->F.isInFrustum(vp, viewing_cos_limit, in);
  if (n_in < 0) return -1;
  for (int j = 0; j < n_mp; ++j) {
    const MapPoint* p = vp[j];
    in_view[j] = p->mbTrackInView ? 1 : 0;
    proj_xy[2 * j] = p->mTrackProjX; proj_xy[2 * j + 1] = p->mTrackProjY;
    proj_xr[j] = p->mTrackProjXR; depth[j] = p->mTrackDepth; view_cos[j] = p->mTrackViewCos; level[j] = p->mnTrackScaleLevel;
  }
  if (assignment) {
    f->F.mvpMapPoints.assign(f->F.N, nullptr);
    ORBmatcher matcher(nnratio);
    const int n = matcher.SearchByProjection(f->F, vp, th);
    for (int k = 0; k < f->F.N; ++k) assignment[k] = f->F.mvpMapPoints[k] ? (int32_t)f->F.mvpMapPoints[k]->mnId : -1;
    f->F.mvpMapPoints.assign(f->F.N, nullptr);
    if (n_matches) *n_matches = n;
  }
  return n_in;
}

extern "C" int osh_host_search_last_frame(osh_host_frame* cur, osh_host_frame* last, const int32_t* last_mp, int32_t n_mp,
                                          const float* mp_pos, const uint8_t* mp_desc, float th, int32_t b_mono, int32_t check_ori,
                                          int32_t* assignment) {
  if (!cur || !last) return -1;
  auto pts = make_points(&cur->map, n_mp, mp_desc, mp_pos, nullptr);
  for (int k = 0; k < last->F.N; ++k) last->F.mvpMapPoints[k] = last_mp[k] >= 0 ? pts[last_mp[k]].get() : nullptr;
  cur->F.mvpMapPoints.assign(cur->F.N, nullptr);
  ORBmatcher matcher(0.9f, check_ori != 0);
  const int n = matcher.SearchByProjection(cur->F, last->F, th, b_mono != 0);
  for (int k = 0; k < cur->F.N; ++k) assignment[k] = cur->F.mvpMapPoints[k] ? (int32_t)cur->F.mvpMapPoints[k]->mnId : -1;
  cur->F.mvpMapPoints.assign(cur->F.N, nullptr);
  last->F.mvpMapPoints.assign(last->F.N, nullptr);
  return n;
}

// ORBmatcher(0.9, check_ori).SearchByProjection(Current, pKF, sAlreadyFound, th, ORBdist)
extern "C" int osh_host_search_keyframe(osh_host_frame* cur, int32_t n_kf, const float* kf_angle, const int32_t* kf_mp, int32_t n_mp,
                                        const float* mp_pos, const uint8_t* mp_desc, const float* mp_min_max_dist,
                                        const uint8_t* mp_found, const uint8_t* mp_bad, const int32_t* cur_mp, float th,
                                        int32_t orb_dist, int32_t check_ori, int32_t* assignment) {
  if (!cur) return -1;
  auto pts = make_points(&cur->map, n_mp, mp_desc, mp_pos, nullptr);
  std::set<MapPoint*> found;
  for (int j = 0; j < n_mp; ++j) {
    pts[j]->mfMinDistance = mp_min_max_dist[2 * j]; pts[j]->mfMaxDistance = mp_min_max_dist[2 * j + 1];
    if (mp_found && mp_found[j]) found.insert(pts[j].get());
    if (mp_bad && mp_bad[j]) pts[j]->mbBad = true;
  }
  KeyFrame kf(1, &cur->map);
  kf.N = n_kf;
  kf.mvKeysUn.resize(n_kf);
  kf.mvpMapPoints.assign(n_kf, nullptr);
  for (int k = 0; k < n_kf; ++k) {
    kf.mvKeysUn[k].angle = kf_angle ? kf_angle[k] : 0.f;
    if (kf_mp[k] >= 0) kf.mvpMapPoints[k] = pts[kf_mp[k]].get();
  }
  Frame& F = cur->F;
  for (int k = 0; k < F.N; ++k) F.mvpMapPoints[k] = (cur_mp && cur_mp[k] >= 0) ? pts[cur_mp[k]].get() : nullptr;
  ORBmatcher matcher(0.9f, check_ori != 0);
  const int n = matcher.SearchByProjection(F, &kf, found, th, orb_dist);
  for (int k = 0; k < F.N; ++k) assignment[k] = F.mvpMapPoints[k] ? (int32_t)F.mvpMapPoints[k]->mnId : -1;
  F.mvpMapPoints.assign(F.N, nullptr);
  return n;
}

// ORBmatcher(0.75, true).SearchByProjection(pKF, Scw, vpPoints[, vpPointsKFs], vpMatched[, vpMatchedKF], th, ratioHamming):
// the keyframe is made from the frame's keypoints, descriptors and grid (as the KeyFrame constructor copies them).
extern "C" int osh_host_search_sim3(osh_host_frame* f, const float scw[8], int32_t n_mp, const float* mp_pos, const uint8_t* mp_desc,
                                    const float* mp_min_max_dist, const float* mp_normal, const uint8_t* mp_bad,
                                    const int32_t* matched_in, int32_t th, float ratio_hamming, int32_t with_keyframes,
                                    int32_t* matched_out, int32_t* matched_kf_out) {
  if (!f) return -1;
  Frame& F = f->F;
  auto pts = make_points(&f->map, n_mp, mp_desc, mp_pos, nullptr);
  for (int j = 0; j < n_mp; ++j) {
    pts[j]->mfMinDistance = mp_min_max_dist[2 * j]; pts[j]->mfMaxDistance = mp_min_max_dist[2 * j + 1];
    pts[j]->mNormalVector = Eigen::Vector3f(mp_normal[3 * j], mp_normal[3 * j + 1], mp_normal[3 * j + 2]);
    if (mp_bad && mp_bad[j]) pts[j]->mbBad = true;
  }
  KeyFrame kf(1, &f->map);
  kf.N = F.N;
  kf.mvKeysUn = F.mvKeysUn;
  kf.mDescriptors = F.mDescriptors;
  kf.mvScaleFactors = F.mvScaleFactors;
  kf.mnScaleLevels = F.mnScaleLevels;
  kf.mfLogScaleFactor = F.mfLogScaleFactor;
  kf.mpCamera = F.mpCamera;
  kf.fx = f->cam->getParameter(0); kf.fy = f->cam->getParameter(1); kf.cx = f->cam->getParameter(2); kf.cy = f->cam->getParameter(3);
  kf.mnGridCols = FRAME_GRID_COLS; kf.mnGridRows = FRAME_GRID_ROWS;
  kf.mfGridElementWidthInv = F.mfGridElementWidthInv; kf.mfGridElementHeightInv = F.mfGridElementHeightInv;
  kf.mnMinX = (int)F.mnMinX; kf.mnMinY = (int)F.mnMinY; kf.mnMaxX = (int)F.mnMaxX; kf.mnMaxY = (int)F.mnMaxY;
  kf.mGrid.assign(FRAME_GRID_COLS, std::vector<std::vector<size_t>>(FRAME_GRID_ROWS));
  for (int i = 0; i < FRAME_GRID_COLS; ++i)
    for (int j = 0; j < FRAME_GRID_ROWS; ++j) kf.mGrid[i][j] = F.mGrid[i][j];
  // one distinct (dummy) source keyframe per point for the second overload
  std::vector<std::unique_ptr<KeyFrame>> srcs;
  std::vector<KeyFrame*> vpPointsKFs;
  std::vector<MapPoint*> vpPoints;
  for (int j = 0; j < n_mp; ++j) {
    vpPoints.push_back(pts[j].get());
    srcs.emplace_back(new KeyFrame((unsigned long)(1000 + j), &f->map));
    vpPointsKFs.push_back(srcs.back().get());
  }
  std::vector<MapPoint*> vpMatched(F.N, nullptr);
  std::vector<KeyFrame*> vpMatchedKF(F.N, nullptr);
  for (int k = 0; k < F.N; ++k) if (matched_in && matched_in[k] >= 0) vpMatched[k] = pts[matched_in[k]].get();
  Sophus::Sim3f Scw(Eigen::Quaternionf(scw[3], scw[0], scw[1], scw[2]), Eigen::Vector3f(scw[4], scw[5], scw[6]), scw[7]);
  ORBmatcher matcher(0.75f, true);
  const int n = with_keyframes ? matcher.SearchByProjection(&kf, Scw, vpPoints, vpPointsKFs, vpMatched, vpMatchedKF, th, ratio_hamming)
                               : matcher.SearchByProjection(&kf, Scw, vpPoints, vpMatched, th, ratio_hamming);
  for (int k = 0; k < F.N; ++k) {
    matched_out[k] = vpMatched[k] ? (int32_t)vpMatched[k]->mnId : -1;
    if (matched_kf_out) matched_kf_out[k] = vpMatchedKF[k] ? (int32_t)(vpMatchedKF[k]->mnId - 1000) : -1;
  }
  return n;
}

// The frame's keypoints, descriptors, pyramid, grid and pose as a KeyFrame (what the KeyFrame constructor copies from a Frame).
static void keyframe_from_frame(KeyFrame& kf, osh_host_frame* f) {
  Frame& F = f->F;
  kf.N = F.N;
  kf.mvKeys = F.mvKeys; kf.mvKeysUn = F.mvKeysUn; kf.mvuRight = F.mvuRight;
  kf.mDescriptors = F.mDescriptors;
  kf.mvScaleFactors = F.mvScaleFactors; kf.mnScaleLevels = F.mnScaleLevels; kf.mfLogScaleFactor = F.mfLogScaleFactor;
  kf.mpCamera = F.mpCamera;
  kf.fx = F.fx; kf.fy = F.fy; kf.cx = F.cx; kf.cy = F.cy; kf.mbf = F.mbf;
  kf.mnGridCols = FRAME_GRID_COLS; kf.mnGridRows = FRAME_GRID_ROWS;
  kf.mfGridElementWidthInv = F.mfGridElementWidthInv; kf.mfGridElementHeightInv = F.mfGridElementHeightInv;
  kf.mnMinX = (int)F.mnMinX; kf.mnMinY = (int)F.mnMinY; kf.mnMaxX = (int)F.mnMaxX; kf.mnMaxY = (int)F.mnMaxY;
  kf.mGrid.assign(FRAME_GRID_COLS, std::vector<std::vector<size_t>>(FRAME_GRID_ROWS));
  for (int i = 0; i < FRAME_GRID_COLS; ++i)
    for (int j = 0; j < FRAME_GRID_ROWS; ++j) kf.mGrid[i][j] = F.mGrid[i][j];
  kf.mvpMapPoints.assign(F.N, nullptr);
  kf.SetPose(F.GetPose());
}

// ORBmatcher(0.75, true).SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (src/ORBmatcher.cc:1457-1674) with two frames standing in for
// the keyframes.  Map points of keyframe k sit in its keypoint slots (slot_mp_k[i] = index into that keyframe's point list or -1);
// matches_in[i1] = index of a keyframe-2 point already matched to slot i1 or -1.  matches_out[i1] = keyframe-2 point index or -1.
extern "C" int osh_host_search_by_sim3(osh_host_frame* f1, osh_host_frame* f2, const float s12[8], float th,
                                       int32_t n_mp1, const float* mp_pos1, const uint8_t* mp_desc1, const float* mp_min_max1, const uint8_t* mp_bad1,
                                       const int32_t* slot_mp1, int32_t n_mp2, const float* mp_pos2, const uint8_t* mp_desc2, const float* mp_min_max2,
                                       const uint8_t* mp_bad2, const int32_t* slot_mp2, const int32_t* matches_in, int32_t* matches_out) {
  if (!f1 || !f2) return -1;
  KeyFrame kf1(1, &f1->map), kf2(2, &f1->map);
  keyframe_from_frame(kf1, f1);
  keyframe_from_frame(kf2, f2);
  auto pts1 = make_points(&f1->map, n_mp1, mp_desc1, mp_pos1, nullptr);
  auto pts2 = make_points(&f1->map, n_mp2, mp_desc2, mp_pos2, nullptr);
  for (int j = 0; j < n_mp1; ++j) { pts1[j]->mfMinDistance = mp_min_max1[2 * j]; pts1[j]->mfMaxDistance = mp_min_max1[2 * j + 1]; pts1[j]->mbBad = mp_bad1 && mp_bad1[j]; }
  for (int j = 0; j < n_mp2; ++j) {
    pts2[j]->mnId = 100000 + (unsigned long)j;
    pts2[j]->mfMinDistance = mp_min_max2[2 * j]; pts2[j]->mfMaxDistance = mp_min_max2[2 * j + 1]; pts2[j]->mbBad = mp_bad2 && mp_bad2[j];
  }
  for (int i = 0; i < kf1.N; ++i) if (slot_mp1[i] >= 0) { kf1.mvpMapPoints[i] = pts1[slot_mp1[i]].get(); pts1[slot_mp1[i]]->AddObservation(&kf1, i); }
  for (int i = 0; i < kf2.N; ++i) if (slot_mp2[i] >= 0) { kf2.mvpMapPoints[i] = pts2[slot_mp2[i]].get(); pts2[slot_mp2[i]]->AddObservation(&kf2, i); }
  std::vector<MapPoint*> vpMatches12(kf1.N, nullptr);
  for (int i = 0; i < kf1.N; ++i) if (matches_in && matches_in[i] >= 0) vpMatches12[i] = pts2[matches_in[i]].get();
  Sophus::Sim3f S12(Eigen::Quaternionf(s12[3], s12[0], s12[1], s12[2]), Eigen::Vector3f(s12[4], s12[5], s12[6]), s12[7]);
  ORBmatcher matcher(0.75f, true);
  int n;
  { CallTimer timed; n = matcher.SearchBySim3(&kf1, &kf2, vpMatches12, S12, th); }
  for (int i = 0; i < kf1.N; ++i) matches_out[i] = vpMatches12[i] ? (int32_t)(vpMatches12[i]->mnId - 100000) : -1;
  return n;
}

// ORBmatcher::Fuse(pKF, vpMapPoints, th) with the frame standing in for the keyframe (pose = the frame's).  Candidates j in
// [0, n_mp) (null_mask[j]: a null entry of vpMapPoints), residents r in [0, n_res) sitting in keypoint slots (slot_res[k] = r or
// -1).  Every point starts with n_obs observations in keyframes of its own (monocular dummies), residents in this keyframe too.
// Outputs use ids: candidate j -> j, resident r -> 100000 + r, none -> -1.
extern "C" int osh_host_fuse(osh_host_frame* f, int32_t n_mp, const float* mp_pos, const uint8_t* mp_desc, const float* mp_min_max_dist,
                             const float* mp_normal, const uint8_t* mp_bad, const uint8_t* null_mask, const int32_t* mp_nobs,
                             int32_t n_res, const int32_t* slot_res, const int32_t* res_nobs, const uint8_t* res_bad, float th,
                             int32_t* slot_out, uint8_t* cand_bad_out, int32_t* cand_replaced_out, int32_t* cand_nobs_out,
                             uint8_t* res_bad_out, int32_t* res_replaced_out, int32_t* res_nobs_out) {
  if (!f) return -1;
  Frame& F = f->F;
  KeyFrame kf(1, &f->map);
  kf.N = F.N;
  kf.mvKeys = F.mvKeys; kf.mvKeysUn = F.mvKeysUn; kf.mvuRight = F.mvuRight;
  kf.mDescriptors = F.mDescriptors;
  kf.mvScaleFactors = F.mvScaleFactors; kf.mnScaleLevels = F.mnScaleLevels; kf.mfLogScaleFactor = F.mfLogScaleFactor;
  kf.mvInvLevelSigma2.assign(F.mnScaleLevels, 1.0f);
  for (int l = 0; l < F.mnScaleLevels; ++l) kf.mvInvLevelSigma2[l] = 1.0f / (F.mvScaleFactors[l] * F.mvScaleFactors[l]);   // src/ORBextractor.cc:414-422
  kf.mpCamera = F.mpCamera;
  kf.fx = F.fx; kf.fy = F.fy; kf.cx = F.cx; kf.cy = F.cy; kf.mbf = F.mbf;
  kf.mnGridCols = FRAME_GRID_COLS; kf.mnGridRows = FRAME_GRID_ROWS;
  kf.mfGridElementWidthInv = F.mfGridElementWidthInv; kf.mfGridElementHeightInv = F.mfGridElementHeightInv;
  kf.mnMinX = (int)F.mnMinX; kf.mnMinY = (int)F.mnMinY; kf.mnMaxX = (int)F.mnMaxX; kf.mnMaxY = (int)F.mnMaxY;
  kf.mGrid.assign(FRAME_GRID_COLS, std::vector<std::vector<size_t>>(FRAME_GRID_ROWS));
  for (int i = 0; i < FRAME_GRID_COLS; ++i)
    for (int j = 0; j < FRAME_GRID_ROWS; ++j) kf.mGrid[i][j] = F.mGrid[i][j];
  kf.mvpMapPoints.assign(F.N, nullptr);
  kf.SetPose(F.GetPose());
  auto cands = make_points(&f->map, n_mp, mp_desc, mp_pos, nullptr);
  std::vector<uint8_t> zero_desc((size_t)std::max(n_res, 1) * 32, 0);
  auto residents = make_points(&f->map, n_res, zero_desc.data(), nullptr, nullptr);
  std::vector<std::unique_ptr<KeyFrame>> others;
  auto give_observations = [&](MapPoint* p, int n) {   // n observations in monocular keyframes nobody else sees
    p->nObs = 0;
    for (int k = 0; k < n; ++k) {
      others.emplace_back(new KeyFrame((unsigned long)(10 + others.size()), &f->map));
      KeyFrame* o = others.back().get();
      o->N = 1; o->mvuRight.assign(1, -1.f); o->mvpMapPoints.assign(1, p);
      p->AddObservation(o, 0);
    }
  };
  for (int j = 0; j < n_mp; ++j) {
    cands[j]->mfMinDistance = mp_min_max_dist[2 * j]; cands[j]->mfMaxDistance = mp_min_max_dist[2 * j + 1];
    cands[j]->mNormalVector = Eigen::Vector3f(mp_normal[3 * j], mp_normal[3 * j + 1], mp_normal[3 * j + 2]);
    give_observations(cands[j].get(), mp_nobs[j]);
    if (mp_bad && mp_bad[j]) cands[j]->mbBad = true;
  }
  for (int r = 0; r < n_res; ++r) {
    residents[r]->mnId = 100000 + (unsigned long)r;
    give_observations(residents[r].get(), res_nobs[r]);
    if (res_bad && res_bad[r]) residents[r]->mbBad = true;
  }
  for (int k = 0; k < F.N; ++k)
    if (slot_res[k] >= 0) { kf.mvpMapPoints[k] = residents[slot_res[k]].get(); residents[slot_res[k]]->AddObservation(&kf, k); }
  std::vector<MapPoint*> vpMapPoints;
  for (int j = 0; j < n_mp; ++j) vpMapPoints.push_back((null_mask && null_mask[j]) ? nullptr : cands[j].get());
  ORBmatcher matcher(0.6f, true);
  int n;
  { CallTimer timed; n = matcher.Fuse(&kf, vpMapPoints, th); }
  for (int k = 0; k < F.N; ++k) slot_out[k] = kf.mvpMapPoints[k] ? (int32_t)kf.mvpMapPoints[k]->mnId : -1;
  for (int j = 0; j < n_mp; ++j) {
    cand_bad_out[j] = cands[j]->isBad() ? 1 : 0;
    cand_replaced_out[j] = cands[j]->GetReplaced() ? (int32_t)cands[j]->GetReplaced()->mnId : -1;
    cand_nobs_out[j] = cands[j]->Observations();
  }
  for (int r = 0; r < n_res; ++r) {
    res_bad_out[r] = residents[r]->isBad() ? 1 : 0;
    res_replaced_out[r] = residents[r]->GetReplaced() ? (int32_t)residents[r]->GetReplaced()->mnId : -1;
    res_nobs_out[r] = residents[r]->Observations();
  }
  return n;
}

// ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint): same set-up as osh_host_fuse; already_found[j]: candidate j sits in the
// keyframe at entry (slot found_slot[j]).  replace_out[j]: id of vpReplacePoint[j] (-1 null).
extern "C" int osh_host_fuse_sim3(osh_host_frame* f, const float scw[8], int32_t n_mp, const float* mp_pos, const uint8_t* mp_desc,
                                  const float* mp_min_max_dist, const float* mp_normal, const uint8_t* mp_bad, const int32_t* mp_nobs,
                                  const int32_t* found_slot, int32_t n_res, const int32_t* slot_res, const uint8_t* res_bad, float th,
                                  int32_t* slot_out, int32_t* replace_out, int32_t* cand_nobs_out) {
  if (!f) return -1;
  Frame& F = f->F;
  KeyFrame kf(1, &f->map);
  kf.N = F.N;
  kf.mvKeys = F.mvKeys; kf.mvKeysUn = F.mvKeysUn; kf.mvuRight = F.mvuRight;
  kf.mDescriptors = F.mDescriptors;
  kf.mvScaleFactors = F.mvScaleFactors; kf.mnScaleLevels = F.mnScaleLevels; kf.mfLogScaleFactor = F.mfLogScaleFactor;
  kf.mpCamera = F.mpCamera;
  kf.fx = F.fx; kf.fy = F.fy; kf.cx = F.cx; kf.cy = F.cy; kf.mbf = F.mbf;
  kf.mnGridCols = FRAME_GRID_COLS; kf.mnGridRows = FRAME_GRID_ROWS;
  kf.mfGridElementWidthInv = F.mfGridElementWidthInv; kf.mfGridElementHeightInv = F.mfGridElementHeightInv;
  kf.mnMinX = (int)F.mnMinX; kf.mnMinY = (int)F.mnMinY; kf.mnMaxX = (int)F.mnMaxX; kf.mnMaxY = (int)F.mnMaxY;
  kf.mGrid.assign(FRAME_GRID_COLS, std::vector<std::vector<size_t>>(FRAME_GRID_ROWS));
  for (int i = 0; i < FRAME_GRID_COLS; ++i)
    for (int j = 0; j < FRAME_GRID_ROWS; ++j) kf.mGrid[i][j] = F.mGrid[i][j];
  kf.mvpMapPoints.assign(F.N, nullptr);
  auto cands = make_points(&f->map, n_mp, mp_desc, mp_pos, mp_nobs);
  std::vector<uint8_t> zero_desc((size_t)std::max(n_res, 1) * 32, 0);
  auto residents = make_points(&f->map, n_res, zero_desc.data(), nullptr, nullptr);
  for (int j = 0; j < n_mp; ++j) {
    cands[j]->mfMinDistance = mp_min_max_dist[2 * j]; cands[j]->mfMaxDistance = mp_min_max_dist[2 * j + 1];
    cands[j]->mNormalVector = Eigen::Vector3f(mp_normal[3 * j], mp_normal[3 * j + 1], mp_normal[3 * j + 2]);
    if (mp_bad && mp_bad[j]) cands[j]->mbBad = true;
    if (found_slot && found_slot[j] >= 0) kf.mvpMapPoints[found_slot[j]] = cands[j].get();
  }
  for (int r = 0; r < n_res; ++r) { residents[r]->mnId = 100000 + (unsigned long)r; if (res_bad && res_bad[r]) residents[r]->mbBad = true; }
  for (int k = 0; k < F.N; ++k) if (slot_res[k] >= 0) kf.mvpMapPoints[k] = residents[slot_res[k]].get();
  std::vector<MapPoint*> vpPoints, vpReplace(n_mp, nullptr);
  for (int j = 0; j < n_mp; ++j) vpPoints.push_back(cands[j].get());
  Sophus::Sim3f Scw(Eigen::Quaternionf(scw[3], scw[0], scw[1], scw[2]), Eigen::Vector3f(scw[4], scw[5], scw[6]), scw[7]);
  ORBmatcher matcher(0.8f, true);
  const int n = matcher.Fuse(&kf, Scw, vpPoints, th, vpReplace);
  for (int k = 0; k < F.N; ++k) slot_out[k] = kf.mvpMapPoints[k] ? (int32_t)kf.mvpMapPoints[k]->mnId : -1;
  for (int j = 0; j < n_mp; ++j) { replace_out[j] = vpReplace[j] ? (int32_t)vpReplace[j]->mnId : -1; cand_nobs_out[j] = cands[j]->Observations(); }
  return n;
}

// Optimizer::PoseOptimization(&frame): kp_mp[k] = map point matched to keypoint k (-1 none), map points by position.
extern "C" int osh_host_frame_pose_optimization(osh_host_frame* f, int32_t n_mp, const float* mp_pos, const int32_t* kp_mp,
                                                const float* inv_level_sigma2, int32_t n_levels, float pose_out[7], uint8_t* outlier_out) {
  if (!f) return -1;
  Frame& F = f->F;
  std::vector<uint8_t> nodesc((size_t)n_mp * 32, 0);
  auto pts = make_points(&f->map, n_mp, nodesc.data(), mp_pos, nullptr);
  F.fx = f->cam->getParameter(0); F.fy = f->cam->getParameter(1); F.cx = f->cam->getParameter(2); F.cy = f->cam->getParameter(3);
  F.mvInvLevelSigma2.assign(inv_level_sigma2, inv_level_sigma2 + n_levels);
  for (int k = 0; k < F.N; ++k) { F.mvpMapPoints[k] = kp_mp[k] >= 0 ? pts[kp_mp[k]].get() : nullptr; F.mvbOutlier[k] = true; }
  const int n = Optimizer::PoseOptimization(&F);
  const Sophus::SE3f T = F.GetPose();
  pose_out[0] = T.unit_quaternion().x(); pose_out[1] = T.unit_quaternion().y(); pose_out[2] = T.unit_quaternion().z(); pose_out[3] = T.unit_quaternion().w();
  pose_out[4] = T.translation()(0); pose_out[5] = T.translation()(1); pose_out[6] = T.translation()(2);
  for (int k = 0; k < F.N; ++k) outlier_out[k] = F.mvbOutlier[k] ? 1 : 0;
  F.mvpMapPoints.assign(F.N, nullptr);
  return n;
}

// ORBmatcher(nnratio, check_ori).SearchByBoW(&keyframe, frame, matches): the keyframe is built from n_kf features (descriptor, angle,
// has_mp[i]: the feature holds a good map point); the DBoW2 feature vectors of keyframe and frame come as CSR (node ids ascending).
// assignment[k] (k < F.N): keyframe feature whose map point landed in vpMapPointMatches[k], or -1.
extern "C" int osh_host_search_by_bow(osh_host_frame* f, int32_t n_kf, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_has_mp,
                                      int32_t kf_nodes, const int32_t* kf_node_id, const int32_t* kf_node_off, const int32_t* kf_node_feat,
                                      int32_t f_nodes, const int32_t* f_node_id, const int32_t* f_node_off, const int32_t* f_node_feat,
                                      float nnratio, int32_t check_ori, int32_t* assignment) {
  if (!f) return -1;
  Frame& F = f->F;
  KeyFrame kf(7, &f->map);
  kf.N = n_kf;
  kf.mDescriptors = cv::Mat(n_kf, 32);
  std::vector<std::unique_ptr<MapPoint>> mps;
  kf.mvpMapPoints.assign(n_kf, nullptr);
  for (int i = 0; i < n_kf; ++i) {
    cv::KeyPoint kp;
    kp.angle = kf_angle[i];
    kf.mvKeysUn.push_back(kp); kf.mvKeys.push_back(kp);
    std::memcpy(kf.mDescriptors.ptr<uint8_t>(i), kf_desc + 32 * (size_t)i, 32);
    if (kf_has_mp[i]) {
      mps.emplace_back(new MapPoint((unsigned long)i, Eigen::Vector3f(0.f, 0.f, 1.f), &f->map));
      kf.mvpMapPoints[i] = mps.back().get();
    }
  }
  for (int a = 0; a < kf_nodes; ++a)
    kf.mFeatVec[(unsigned)kf_node_id[a]] = std::vector<unsigned int>(kf_node_feat + kf_node_off[a], kf_node_feat + kf_node_off[a + 1]);
  F.mFeatVec.clear();
  for (int b = 0; b < f_nodes; ++b)
    F.mFeatVec[(unsigned)f_node_id[b]] = std::vector<unsigned int>(f_node_feat + f_node_off[b], f_node_feat + f_node_off[b + 1]);
  std::vector<MapPoint*> matches;
  ORBmatcher matcher(nnratio, check_ori != 0);
  const int n = matcher.SearchByBoW(&kf, F, matches);
  for (int k = 0; k < F.N; ++k) {
    assignment[k] = -1;
    if (k < (int)matches.size() && matches[k]) assignment[k] = (int32_t)matches[k]->mnId;
  }
  return n;
}

// ORBmatcher(nnratio, check_ori).SearchByBoW(&kf1, &kf2, matches12): two keyframes built from flat features (descriptor, angle, has_mp)
// and CSR feature vectors; map point of feature i of keyframe 2 has id i.  match12[i] (i < n1): feature of keyframe 2 or -1.
extern "C" int osh_host_search_by_bow_kf(int32_t n1, const uint8_t* desc1, const float* angle1, const uint8_t* has_mp1, int32_t nodes1,
                                         const int32_t* node_id1, const int32_t* node_off1, const int32_t* node_feat1, int32_t n2,
                                         const uint8_t* desc2, const float* angle2, const uint8_t* has_mp2, int32_t nodes2, const int32_t* node_id2,
                                         const int32_t* node_off2, const int32_t* node_feat2, float nnratio, int32_t check_ori, int32_t* match12) {
  Map map;
  std::vector<std::unique_ptr<MapPoint>> mps;
  auto build = [&](KeyFrame& kf, int n, const uint8_t* desc, const float* angle, const uint8_t* has_mp, int nodes, const int32_t* nid,
                   const int32_t* noff, const int32_t* nfeat) {
    kf.N = n;
    kf.mDescriptors = cv::Mat(n, 32);
    kf.mvpMapPoints.assign(n, nullptr);
    for (int i = 0; i < n; ++i) {
      cv::KeyPoint kp;
      kp.angle = angle[i];
      kf.mvKeysUn.push_back(kp); kf.mvKeys.push_back(kp);
      std::memcpy(kf.mDescriptors.ptr<uint8_t>(i), desc + 32 * (size_t)i, 32);
      if (has_mp[i]) {
        mps.emplace_back(new MapPoint((unsigned long)i, Eigen::Vector3f(0.f, 0.f, 1.f), &map));
        kf.mvpMapPoints[i] = mps.back().get();
      }
    }
    for (int a = 0; a < nodes; ++a) kf.mFeatVec[(unsigned)nid[a]] = std::vector<unsigned int>(nfeat + noff[a], nfeat + noff[a + 1]);
  };
  KeyFrame kf1(7, &map), kf2(8, &map);
  build(kf1, n1, desc1, angle1, has_mp1, nodes1, node_id1, node_off1, node_feat1);
  build(kf2, n2, desc2, angle2, has_mp2, nodes2, node_id2, node_off2, node_feat2);
  std::vector<MapPoint*> matches;
  ORBmatcher matcher(nnratio, check_ori != 0);
  const int n = matcher.SearchByBoW(&kf1, &kf2, matches);
  for (int i = 0; i < n1; ++i) {
    match12[i] = -1;
    if (i < (int)matches.size() && matches[i]) match12[i] = (int32_t)matches[i]->mnId;
  }
  return n;
}

// ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize): prev_xy[n1][2] in / out.
extern "C" int osh_host_search_for_initialization(osh_host_frame* f1, osh_host_frame* f2, float* prev_xy, int32_t window, float nnratio, int32_t check_ori,
                                                  int32_t* matches12) {
  if (!f1 || !f2) return -1;
  std::vector<cv::Point2f> prev(f1->F.mvKeysUn.size());
  for (size_t i = 0; i < prev.size(); ++i) { prev[i].x = prev_xy[2 * i]; prev[i].y = prev_xy[2 * i + 1]; }
  std::vector<int> m;
  ORBmatcher matcher(nnratio, check_ori != 0);
  const int n = matcher.SearchForInitialization(f1->F, f2->F, prev, m, window);
  for (size_t i = 0; i < prev.size(); ++i) { prev_xy[2 * i] = prev[i].x; prev_xy[2 * i + 1] = prev[i].y; matches12[i] = m[i]; }
  return n;
}

// ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) on two pinhole keyframes built from flat
// features: kp[n][4] = x, y, angle, uright (< 0: monocular keypoint); match12[i] = feature of keyframe 2 paired with feature i or -1.
extern "C" int osh_host_search_for_triangulation(const float cam4[4], int32_t n_levels, float scale_factor, int32_t n1, const float* kp1,
                                                 const int32_t* octave1, const uint8_t* desc1, const uint8_t* has_mp1, const float pose1_qt[7],
                                                 int32_t nodes1, const int32_t* node_id1, const int32_t* node_off1, const int32_t* node_feat1,
                                                 int32_t n2, const float* kp2, const int32_t* octave2, const uint8_t* desc2, const uint8_t* has_mp2,
                                                 const float pose2_qt[7], int32_t nodes2, const int32_t* node_id2, const int32_t* node_off2,
                                                 const int32_t* node_feat2, int32_t only_stereo, int32_t coarse, int32_t check_ori, int32_t* match12) {
  Map map;
  Pinhole cam(std::vector<float>{cam4[0], cam4[1], cam4[2], cam4[3]});
  std::vector<std::unique_ptr<MapPoint>> mps;
  auto build = [&](KeyFrame& kf, int n, const float* kp, const int32_t* octave, const uint8_t* desc, const uint8_t* has_mp, const float* pose_qt,
                   int nodes, const int32_t* nid, const int32_t* noff, const int32_t* nfeat) {
    kf.N = n;
    kf.mpCamera = &cam;
    kf.mDescriptors = cv::Mat(n, 32);
    kf.mvpMapPoints.assign(n, nullptr);
    kf.mvScaleFactors.assign(n_levels, 1.0f);
    kf.mvLevelSigma2.assign(n_levels, 1.0f);
    for (int l = 1; l < n_levels; ++l) { kf.mvScaleFactors[l] = kf.mvScaleFactors[l - 1] * scale_factor; kf.mvLevelSigma2[l] = kf.mvScaleFactors[l] * kf.mvScaleFactors[l]; }
    for (int i = 0; i < n; ++i) {
      cv::KeyPoint k;
      k.pt.x = kp[4 * i]; k.pt.y = kp[4 * i + 1]; k.angle = kp[4 * i + 2]; k.octave = octave[i];
      kf.mvKeysUn.push_back(k); kf.mvKeys.push_back(k);
      kf.mvuRight.push_back(kp[4 * i + 3]);
      std::memcpy(kf.mDescriptors.ptr<uint8_t>(i), desc + 32 * (size_t)i, 32);
      if (has_mp[i]) {
        mps.emplace_back(new MapPoint((unsigned long)i, Eigen::Vector3f(0.f, 0.f, 1.f), &map));
        kf.mvpMapPoints[i] = mps.back().get();
      }
    }
    for (int a = 0; a < nodes; ++a) kf.mFeatVec[(unsigned)nid[a]] = std::vector<unsigned int>(nfeat + noff[a], nfeat + noff[a + 1]);
    kf.SetPose(pose_from(pose_qt));
  };
  KeyFrame kf1(7, &map), kf2(8, &map);
  build(kf1, n1, kp1, octave1, desc1, has_mp1, pose1_qt, nodes1, node_id1, node_off1, node_feat1);
  build(kf2, n2, kp2, octave2, desc2, has_mp2, pose2_qt, nodes2, node_id2, node_off2, node_feat2);
  std::vector<std::pair<size_t, size_t>> pairs;
  ORBmatcher matcher(0.6f, check_ori != 0);
  const int n = matcher.SearchForTriangulation(&kf1, &kf2, pairs, only_stereo != 0, coarse != 0);
  for (int i = 0; i < n1; ++i) match12[i] = -1;
  for (const auto& pr : pairs) match12[pr.first] = (int32_t)pr.second;
  return n;
}

// ------------------------------------------------------------------------------------------ PoseInertialOptimization*
struct osh_host_posei {
  Frame F, prevF;
  Map map;
  std::unique_ptr<KeyFrame> prevKF;
  std::unique_ptr<GeometricCamera> cam, cam2;
  std::vector<std::unique_ptr<MapPoint>> mps;
  std::unique_ptr<IMU::Preintegrated> preint;
  PoseiPack pack;
  int mode = 0;
};

// A tracked frame with n_kp matched keypoints (map point k at mp_pos[k], mTrackDepth 5 or 20 by mp_close[k]) and its IMU link to the
// last keyframe (mode 0) or to the previous frame (mode 1, with that frame's ConstraintPoseImu).  n_left < 0: monocular /
// rectified stereo (uright >= 0: stereo keypoint); n_left >= 0: fisheye stereo rig, keypoints [0, n_left) left, the rest right.
extern "C" osh_host_posei* osh_host_posei_create(int32_t mode, int32_t n_kp, const float* kp_xy, const int32_t* octave, const float* uright,
                                                 int32_t n_left, const float pose_qt[7], const float cam5[5], const float* kb8,
                                                 const float* cam2_8, const float* trl_qt, const float* inv_level_sigma2, int32_t n_levels,
                                                 const float* mp_pos, const uint8_t* mp_close, const float Tbc_qt[7], const float vel[3],
                                                 const float bias6[6], const float prev_pose_qt[7], const float prev_vel[3], const float prev_bias6[6],
                                                 const float* preint72, const float* cov225, const double* prior_Rwb, const double* prior_twb,
                                                 const double* prior_vel, const double* prior_bg, const double* prior_ba, const double* prior_H) {
  osh_host_posei* h = new osh_host_posei();
  h->mode = mode;
  Frame& F = h->F;
  if (kb8) h->cam.reset(new KannalaBrandt8(std::vector<float>{cam5[0], cam5[1], cam5[2], cam5[3], kb8[0], kb8[1], kb8[2], kb8[3]}));
  else h->cam.reset(new Pinhole(std::vector<float>{cam5[0], cam5[1], cam5[2], cam5[3]}));
  F.mpCamera = h->cam.get();
  F.fx = cam5[0]; F.fy = cam5[1]; F.cx = cam5[2]; F.cy = cam5[3]; F.mbf = cam5[4];
  F.N = n_kp;
  F.mvInvLevelSigma2.assign(inv_level_sigma2, inv_level_sigma2 + n_levels);
  const IMU::Calib calib(pose_from(Tbc_qt), 0.f, 0.f, 0.f, 0.f);
  F.mImuCalib = calib;
  F.mTcw = pose_from(pose_qt);
  F.UpdatePoseMatrices();
  F.SetVelocity(Eigen::Vector3f(vel[0], vel[1], vel[2]));
  F.mImuBias = IMU::Bias(bias6[0], bias6[1], bias6[2], bias6[3], bias6[4], bias6[5]);
  if (n_left >= 0) {
    h->cam2.reset(new KannalaBrandt8(std::vector<float>(cam2_8, cam2_8 + 8)));
    F.mpCamera2 = h->cam2.get();
    F.mTrl = pose_from(trl_qt);
    F.Nleft = n_left; F.Nright = n_kp - n_left;
  }
  for (int i = 0; i < n_kp; ++i) {
    cv::KeyPoint kp;
    kp.pt.x = kp_xy[2 * i]; kp.pt.y = kp_xy[2 * i + 1]; kp.octave = octave[i];
    if (n_left >= 0 && i >= n_left) F.mvKeysRight.push_back(kp);
    else { F.mvKeys.push_back(kp); F.mvKeysUn.push_back(kp); }
    F.mvuRight.push_back(uright ? uright[i] : -1.f);
    h->mps.emplace_back(new MapPoint((unsigned long)(1000 + i), Eigen::Vector3f(mp_pos[3 * i], mp_pos[3 * i + 1], mp_pos[3 * i + 2]), &h->map));
    h->mps.back()->mTrackDepth = mp_close[i] ? 5.f : 20.f;
    F.mvpMapPoints.push_back(h->mps.back().get());
  }
  F.mvbOutlier.assign(n_kp, true);
  h->preint.reset(new IMU::Preintegrated(IMU::Bias(preint72[61], preint72[62], preint72[63], preint72[64], preint72[65], preint72[66]), calib));
  IMU::Preintegrated* P = h->preint.get();
  P->dT = preint72[0];
  for (int a = 0; a < 9; ++a) { P->dR.v[a] = preint72[1 + a]; P->JRg.v[a] = preint72[16 + a]; P->JVg.v[a] = preint72[25 + a]; P->JVa.v[a] = preint72[34 + a]; P->JPg.v[a] = preint72[43 + a]; P->JPa.v[a] = preint72[52 + a]; }
  for (int a = 0; a < 3; ++a) { P->dV(a) = preint72[10 + a]; P->dP(a) = preint72[13 + a]; }
  for (int a = 0; a < 225; ++a) P->C.v[a] = cov225[a];
  F.mpImuPreintegrated = P;          // the random-walk informations come from here in both variants
  F.mpImuPreintegratedFrame = P;
  if (mode == 0) {
    h->prevKF.reset(new KeyFrame(99, &h->map));
    KeyFrame* kf = h->prevKF.get();
    kf->mImuCalib = calib;
    kf->SetPose(pose_from(prev_pose_qt));
    kf->bImu = true;
    kf->SetVelocity(Eigen::Vector3f(prev_vel[0], prev_vel[1], prev_vel[2]));
    kf->mImuBias = IMU::Bias(prev_bias6[0], prev_bias6[1], prev_bias6[2], prev_bias6[3], prev_bias6[4], prev_bias6[5]);
    F.mpLastKeyFrame = kf;
  } else {
    Frame& Fp = h->prevF;
    Fp.mImuCalib = calib;
    Fp.mTcw = pose_from(prev_pose_qt);
    Fp.UpdatePoseMatrices();
    Fp.SetVelocity(Eigen::Vector3f(prev_vel[0], prev_vel[1], prev_vel[2]));
    Fp.mImuBias = IMU::Bias(prev_bias6[0], prev_bias6[1], prev_bias6[2], prev_bias6[3], prev_bias6[4], prev_bias6[5]);
    Eigen::Matrix3d R; Eigen::Vector3d t, v, bg, ba; Matrix15d H;
    for (int a = 0; a < 9; ++a) R.v[a] = prior_Rwb[a];
    for (int a = 0; a < 3; ++a) { t(a) = prior_twb[a]; v(a) = prior_vel[a]; bg(a) = prior_bg[a]; ba(a) = prior_ba[a]; }
    for (int a = 0; a < 225; ++a) H.v[a] = prior_H[a];
    Fp.mpcpi = new ConstraintPoseImu(R, t, v, bg, ba, H);
    F.mpPrevFrame = &Fp;
  }
  return h;
}

extern "C" void osh_host_posei_destroy(osh_host_posei* h) {
  if (!h) return;
  delete h->F.mpcpi;
  delete h->prevF.mpcpi;
  delete h;
}

// The osh_posei_problem the host layer builds for this frame (arrays owned by the handle until the next call).
extern "C" int osh_host_posei_pack(osh_host_posei* h, int32_t rec_init, osh_posei_problem* out, int32_t* kp_of_edge) {
  if (!h || !out) return -1;
  if (!PackPoseInertial(&h->F, rec_init != 0, h->mode, h->pack)) return -3;
  h->pack.fill(*out);
  for (size_t e = 0; e < h->pack.index.size(); ++e) kp_of_edge[e] = h->pack.index[e];
  for (int k = 0; k < h->F.N; ++k) h->F.mvbOutlier[k] = true;
  return 0;
}

// Optimizer::PoseInertialOptimizationLastKeyFrame / LastFrame (&frame, bRecInit): returns the function's value; outputs the frame's
// new Tcw, IMU pose, velocity, bias, mvbOutlier, the H of its new ConstraintPoseImu and whether the previous frame's was deleted.
extern "C" int osh_host_posei_run(osh_host_posei* h, int32_t rec_init, float pose_out[7], float Rwb_out[9], float twb_out[3], float vel_out[3],
                                  float bias6_out[6], uint8_t* outlier_out, double* H225_out, int32_t* prev_cpi_deleted) {
  if (!h) return -1;
  Frame& F = h->F;
  const int n = h->mode == 0 ? Optimizer::PoseInertialOptimizationLastKeyFrame(&F, rec_init != 0)
                             : Optimizer::PoseInertialOptimizationLastFrame(&F, rec_init != 0);
  const Sophus::SE3f T = F.GetPose();
  pose_out[0] = T.unit_quaternion().x(); pose_out[1] = T.unit_quaternion().y(); pose_out[2] = T.unit_quaternion().z(); pose_out[3] = T.unit_quaternion().w();
  pose_out[4] = T.translation()(0); pose_out[5] = T.translation()(1); pose_out[6] = T.translation()(2);
  const Eigen::Matrix3f Rwb = F.GetImuRotation();
  const Eigen::Vector3f twb = F.GetImuPosition(), v = F.GetVelocity();
  for (int a = 0; a < 9; ++a) Rwb_out[a] = Rwb.v[a];
  for (int a = 0; a < 3; ++a) { twb_out[a] = twb(a); vel_out[a] = v(a); }
  bias6_out[0] = F.mImuBias.bax; bias6_out[1] = F.mImuBias.bay; bias6_out[2] = F.mImuBias.baz;
  bias6_out[3] = F.mImuBias.bwx; bias6_out[4] = F.mImuBias.bwy; bias6_out[5] = F.mImuBias.bwz;
  for (int k = 0; k < F.N; ++k) outlier_out[k] = F.mvbOutlier[k] ? 1 : 0;
  if (F.mpcpi) for (int a = 0; a < 225; ++a) H225_out[a] = F.mpcpi->H.v[a];
  *prev_cpi_deleted = (h->mode == 1 && h->prevF.mpcpi == nullptr) ? 1 : 0;
  return n;
}
