/*
 * orb_oracle.c -- CPU restatement of the Hamming search inside
 * ORB_SLAM3::ORBmatcher::SearchByProjection.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see lba_oracle.c header).
 * PARITY UNPINNED against a reference binary (OpenCV absent, reference cannot
 * be built); pinned by known-answer vectors (d(a,a)=0, d(a,~a)=256, SWAR ==
 * popcount) and hand-built tie-break / occupancy cases in tests/.
 *
 * Paths relative to /root/reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* ORBmatcher::DescriptorDistance, src/ORBmatcher.cc:2058-2074 (SWAR popcount
 * over 8 x int32). */
int oracle_descriptor_distance(const uint8_t* a, const uint8_t* b) {
  int dist = 0;
  for (int i = 0; i < 8; i++) {
    int32_t pa, pb;
    memcpy(&pa, a + 4 * i, 4);
    memcpy(&pb, b + 4 * i, 4);
    unsigned int v = (unsigned int)(pa ^ pb);
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

void oracle_distance_matrix(int n, int m, const uint8_t* a, const uint8_t* b, int32_t* out) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < m; ++j) out[(size_t)i * m + j] = oracle_descriptor_distance(a + 32 * (size_t)i, b + 32 * (size_t)j);
}

/* The candidate loop of src/ORBmatcher.cc:84-120 for one query: strict '<'
 * left-to-right scan keeping best and second best with their levels.
 * `occupied` (may be NULL) marks slots skipped by the test at :88-90. */
static void scan_one(const uint8_t* qd, const uint8_t* train_desc, const int32_t* train_level,
                     const int32_t* cand, int ncand, int n_train, const uint8_t* occupied,
                     int* bestIdx, int* bestDist, int* bestDist2, int* bestLevel, int* bestLevel2) {
  *bestDist = 256; *bestLevel = -1; *bestDist2 = 256; *bestLevel2 = -1; *bestIdx = -1;
  const int n = cand ? ncand : n_train;
  for (int c = 0; c < n; ++c) {
    const int idx = cand ? cand[c] : c;
    if (occupied && occupied[idx]) continue;
    const int dist = oracle_descriptor_distance(qd, train_desc + 32 * (size_t)idx);
    const int lev = train_level ? train_level[idx] : 0;
    if (dist < *bestDist) {
      *bestDist2 = *bestDist;
      *bestDist = dist;
      *bestLevel2 = *bestLevel;
      *bestLevel = lev;
      *bestIdx = idx;
    } else if (dist < *bestDist2) {
      *bestLevel2 = lev;
      *bestDist2 = dist;
    }
  }
}

/* Per-query best/second over (optionally windowed) candidate lists, no
 * acceptance logic: what the device kernel must reproduce bit for bit. */
void oracle_orb_search(int n_query, int n_train, const uint8_t* query_desc, const uint8_t* train_desc,
                       const int32_t* train_level, const int32_t* cand_off, const int32_t* cand_idx,
                       const uint8_t* occupied,
                       int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                       int32_t* best_level, int32_t* second_level) {
  for (int q = 0; q < n_query; ++q) {
    int bi, bd, bd2, bl, bl2;
    const int32_t* cand = cand_off ? cand_idx + cand_off[q] : NULL;
    const int nc = cand_off ? cand_off[q + 1] - cand_off[q] : n_train;
    scan_one(query_desc + 32 * (size_t)q, train_desc, train_level, cand, nc, n_train, occupied, &bi, &bd, &bd2, &bl, &bl2);
    best_idx[q] = bi; best_dist[q] = bd; second_dist[q] = bd2; best_level[q] = bl; second_level[q] = bl2;
  }
}

/* SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) for the
 * monocular/rectified-stereo case (F.Nleft == -1), src/ORBmatcher.cc:43-141:
 * sequential over queries; a slot taken by an earlier accepted query is
 * skipped (:88-90, local map points have Observations()>0); ratio test in
 * float (:125,128).  occupied[n_train] is read and updated; assignment[t] =
 * query index stored into F.mvpMapPoints[t] (caller pre-fills with -1).
 * The stereo ur window test (:92-97) is a static per-pair filter and is
 * expected to be already applied to the candidate lists. */
int oracle_orb_match_local_points(int n_query, int n_train, const uint8_t* query_desc,
                                  const uint8_t* train_desc, const int32_t* train_level,
                                  const int32_t* cand_off, const int32_t* cand_idx,
                                  float nn_ratio, int th_high, uint8_t* occupied, int32_t* assignment) {
  int nmatches = 0;
  for (int q = 0; q < n_query; ++q) {
    const int32_t* cand = cand_off ? cand_idx + cand_off[q] : NULL;
    const int nc = cand_off ? cand_off[q + 1] - cand_off[q] : n_train;
    if (cand_off && nc == 0) continue; /* vIndices.empty() :74 */
    int bi, bd, bd2, bl, bl2;
    scan_one(query_desc + 32 * (size_t)q, train_desc, train_level, cand, nc, n_train, occupied, &bi, &bd, &bd2, &bl, &bl2);
    if (bd <= th_high) {
      if (bl == bl2 && bd > nn_ratio * bd2) continue;
      if (bl != bl2 || bd <= nn_ratio * bd2) {
        assignment[bi] = q;
        occupied[bi] = 1;
        nmatches++;
      }
    }
  }
  return nmatches;
}

/* SearchByProjection(Frame&, const vector<MapPoint*>&, th, ...) for a FISHEYE STEREO frame (F.Nleft != -1), the whole of
 * src/ORBmatcher.cc:43-213: per map point first the left-camera pass (:60-141, candidates among the left keypoints, NO u_right
 * test, levels from mvKeys), then -- unless the left ratio test `continue`d the outer loop (:125-126) -- the right-camera pass
 * (:144-210, candidates among the right keypoints, window WITHOUT the th factor, levels from mvKeysRight).  An accepted left
 * match also claims its stereo partner F.mvLeftToRightMatch[best] + Nleft (:131-135, nmatches += 2), an accepted right match
 * its partner F.mvRightToLeftMatch[best] (:199-203).  Slots: assignment[0 .. n_left) = left keypoints, [n_left ..) = right ones;
 * occupied[] likewise (read and updated: local map points have Observations() > 0).
 * in_l / in_r: mbTrackInView / (mbTrackInViewR && mnTrackScaleLevelR != -1) of each map point; a point with neither is skipped
 * by the caller's common filters already.  Candidate lists index their own side (right candidates 0-based in the right set). */
int oracle_orb_match_local_points_rig(int n_query, int n_left, int n_right, const uint8_t* query_desc, const uint8_t* desc,
                                      const int32_t* level_left, const int32_t* level_right,
                                      const uint8_t* in_l, const int32_t* candl_off, const int32_t* candl_idx,
                                      const uint8_t* in_r, const int32_t* candr_off, const int32_t* candr_idx,
                                      const int32_t* left_to_right, const int32_t* right_to_left,
                                      float nn_ratio, int th_high, uint8_t* occupied, int32_t* assignment) {
  int nmatches = 0;
  const uint8_t* desc_r = desc + 32 * (size_t)n_left;
  uint8_t* occ_r = occupied + n_left;
  for (int q = 0; q < n_query; ++q) {
    const uint8_t* qd = query_desc + 32 * (size_t)q;
    if (in_l[q]) {
      const int nc = candl_off[q + 1] - candl_off[q];
      if (nc > 0) { /* !vIndices.empty() :74 */
        int bi, bd, bd2, bl, bl2;
        scan_one(qd, desc, level_left, candl_idx + candl_off[q], nc, n_left, occupied, &bi, &bd, &bd2, &bl, &bl2);
        if (bd <= th_high) {
          if (bl == bl2 && bd > nn_ratio * bd2) continue; /* skips the right-camera pass of this point too */
          if (bl != bl2 || bd <= nn_ratio * bd2) {
            assignment[bi] = q; occupied[bi] = 1;
            if (left_to_right[bi] != -1) { assignment[left_to_right[bi] + n_left] = q; occ_r[left_to_right[bi]] = 1; nmatches++; }
            nmatches++;
          }
        }
      }
    }
    if (in_r[q]) {
      const int nc = candr_off[q + 1] - candr_off[q];
      if (nc == 0) continue; /* :153-154 */
      int bi, bd, bd2, bl, bl2;
      scan_one(qd, desc_r, level_right, candr_idx + candr_off[q], nc, n_right, occ_r, &bi, &bd, &bd2, &bl, &bl2);
      if (bd <= th_high) {
        if (bl == bl2 && bd > nn_ratio * bd2) continue;
        if (right_to_left[bi] != -1) { assignment[right_to_left[bi]] = q; occupied[right_to_left[bi]] = 1; nmatches++; }
        assignment[bi + n_left] = q; occ_r[bi] = 1;
        nmatches++;
      }
    }
  }
  return nmatches;
}

/* ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:2012-2053 (on bin sizes). */
static void three_maxima(const int* sizes, int L, int* ind1, int* ind2, int* ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = sizes[i];
    if (s > max1) {
      max3 = max2; max2 = max1; max1 = s;
      *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
    } else if (s > max2) {
      max3 = max2; max2 = s;
      *ind3 = *ind2; *ind2 = i;
    } else if (s > max3) {
      max3 = s; *ind3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono),
 * src/ORBmatcher.cc:1676-1887, left-camera case: best only (:1743-1768),
 * accept <= TH_HIGH (:1770), rotation histogram with factor 1.0f/HISTO_LENGTH
 * (:1684,1784-1791), keep the three dominant bins (:1863-1884). */
int oracle_orb_match_last_frame(int n_query, int n_train, const uint8_t* query_desc,
                                const uint8_t* train_desc, const int32_t* cand_off, const int32_t* cand_idx,
                                const float* query_angle, const float* train_angle,
                                int th_high, int check_orientation, uint8_t* occupied, int32_t* assignment) {
  enum { HISTO_LENGTH = 30 };
  int nmatches = 0;
  int* hist[HISTO_LENGTH];
  int hsize[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; ++i) { hist[i] = (int*)malloc(sizeof(int) * (size_t)(n_query + 1)); hsize[i] = 0; }
  const float factor = 1.0f / HISTO_LENGTH;
  for (int q = 0; q < n_query; ++q) {
    const int32_t* cand = cand_off ? cand_idx + cand_off[q] : NULL;
    const int nc = cand_off ? cand_off[q + 1] - cand_off[q] : n_train;
    if (cand_off && nc == 0) continue;
    int bi, bd, bd2, bl, bl2;
    scan_one(query_desc + 32 * (size_t)q, train_desc, NULL, cand, nc, n_train, occupied, &bi, &bd, &bd2, &bl, &bl2);
    if (bd <= th_high) {
      assignment[bi] = q;
      occupied[bi] = 1;
      nmatches++;
      if (check_orientation) {
        float rot = query_angle[q] - train_angle[bi];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        hist[bin][hsize[bin]++] = bi;
      }
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(hsize, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (int j = 0; j < hsize[i]; j++) {
          assignment[hist[i][j]] = -1; /* slot stays "occupied" only within this call */
          nmatches--;
        }
      }
    }
  }
  for (int i = 0; i < HISTO_LENGTH; ++i) free(hist[i]);
  return nmatches;
}

/* SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) with a FISHEYE STEREO current frame
 * (CurrentFrame.Nleft != -1), src/ORBmatcher.cc:1676-1887 including the right-camera block :1794-1858.  Per query (a map point
 * of the last frame that passed the projection tests): best-only search among the left keypoints -- an EMPTY left candidate list
 * `continue`s past the right-camera block (:1731-1732) --, then best-only search among the right keypoints; accept <= TH_HIGH;
 * rotation histogram over both (right slots enter as index + n_left), three dominant bins kept.  Slots / occupied as in
 * oracle_orb_match_local_points_rig.  query_angle: angle of the last frame's keypoint; angle_left / angle_right: current frame. */
int oracle_orb_match_last_frame_rig(int n_query, int n_left, int n_right, const uint8_t* query_desc, const uint8_t* desc,
                                    const int32_t* candl_off, const int32_t* candl_idx, const int32_t* candr_off, const int32_t* candr_idx,
                                    const float* query_angle, const float* angle_left, const float* angle_right,
                                    int th_high, int check_orientation, uint8_t* occupied, int32_t* assignment) {
  enum { HISTO_LENGTH = 30 };
  int nmatches = 0;
  int* hist[HISTO_LENGTH];
  int hsize[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; ++i) { hist[i] = (int*)malloc(sizeof(int) * (size_t)(2 * n_query + 1)); hsize[i] = 0; }
  const float factor = 1.0f / HISTO_LENGTH;
  const uint8_t* desc_r = desc + 32 * (size_t)n_left;
  uint8_t* occ_r = occupied + n_left;
  for (int q = 0; q < n_query; ++q) {
    const uint8_t* qd = query_desc + 32 * (size_t)q;
    const int ncl = candl_off[q + 1] - candl_off[q];
    if (ncl == 0) continue;
    int bi, bd, bd2, bl, bl2;
    scan_one(qd, desc, NULL, candl_idx + candl_off[q], ncl, n_left, occupied, &bi, &bd, &bd2, &bl, &bl2);
    if (bd <= th_high) {
      assignment[bi] = q; occupied[bi] = 1; nmatches++;
      if (check_orientation) {
        float rot = query_angle[q] - angle_left[bi];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        hist[bin][hsize[bin]++] = bi;
      }
    }
    const int ncr = candr_off[q + 1] - candr_off[q];
    scan_one(qd, desc_r, NULL, candr_idx + candr_off[q], ncr, n_right, occ_r, &bi, &bd, &bd2, &bl, &bl2);
    if (ncr > 0 && bd <= th_high) {
      assignment[bi + n_left] = q; occ_r[bi] = 1; nmatches++;
      if (check_orientation) {
        float rot = query_angle[q] - angle_right[bi];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        hist[bin][hsize[bin]++] = bi + n_left;
      }
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(hsize, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (int j = 0; j < hsize[i]; j++) { assignment[hist[i][j]] = -1; nmatches--; }
  }
  for (int i = 0; i < HISTO_LENGTH; ++i) free(hist[i]);
  return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches), src/ORBmatcher.cc:223-420
 * (Tracking::TrackReferenceKeyFrame, Relocalization).  The two DBoW2 feature vectors are walked in step (:242-395); inside a
 * common vocabulary node every keyframe feature that holds a good map point is compared with the frame features of the node
 * that are still unmatched (:266-268), best and second best by strict '<' in list order; accept bestDist1 <= TH_LOW and
 * (float)bestDist1 < mfNNratio * (float)bestDist2 (:319-321); rotation histogram with factor 1.0f/HISTO_LENGTH, three dominant
 * bins kept (:397-417).  Fisheye stereo frame (n_left_f >= 0): left candidates (index < Nleft) and right candidates compete
 * separately; the right best is accepted without a ratio test ("|| true", :352) but only inside the branch of an accepted-range
 * left best (bestDist1 <= TH_LOW, :319).
 * Feature vectors as CSR: node ids ascending, node_off[n_nodes + 1], node_feat.  kf_has_mp[i]: keyframe feature i holds a map
 * point that is not bad.  assignment[n_f] (out): keyframe feature whose map point ends in vpMapPointMatches[slot], or -1. */
int oracle_orb_search_by_bow(int n_kf, int n_f, int n_left_f, const uint8_t* kf_desc, const uint8_t* f_desc, const uint8_t* kf_has_mp,
                             int kf_nodes, const int32_t* kf_node_id, const int32_t* kf_node_off, const int32_t* kf_node_feat,
                             int f_nodes, const int32_t* f_node_id, const int32_t* f_node_off, const int32_t* f_node_feat,
                             const float* kf_angle, const float* f_angle, float nn_ratio, int th_low, int check_orientation,
                             int32_t* assignment) {
  enum { HISTO_LENGTH = 30 };
  (void)n_kf;
  int nmatches = 0;
  int* hist[HISTO_LENGTH];
  int hsize[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; ++i) { hist[i] = (int*)malloc(sizeof(int) * (size_t)(2 * n_f + 2)); hsize[i] = 0; }
  for (int i = 0; i < n_f; ++i) assignment[i] = -1;
  const float factor = 1.0f / HISTO_LENGTH;
  int a = 0, b = 0;
  while (a < kf_nodes && b < f_nodes) {
    if (kf_node_id[a] == f_node_id[b]) {
      for (int x = kf_node_off[a]; x < kf_node_off[a + 1]; ++x) {
        const int iKF = kf_node_feat[x];
        if (!kf_has_mp[iKF]) continue;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256, bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
        for (int y = f_node_off[b]; y < f_node_off[b + 1]; ++y) {
          const int iF = f_node_feat[y];
          if (assignment[iF] >= 0) continue;
          const int dist = oracle_descriptor_distance(kf_desc + 32 * (size_t)iKF, f_desc + 32 * (size_t)iF);
          if (n_left_f < 0 || iF < n_left_f) {
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = iF; }
            else if (dist < bestDist2) bestDist2 = dist;
          } else {
            if (dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = iF; }
            else if (dist < bestDist2R) bestDist2R = dist;
          }
        }
        if (bestDist1 <= th_low) {
          if ((float)bestDist1 < nn_ratio * (float)bestDist2) {
            assignment[bestIdxF] = iKF;
            if (check_orientation) {
              float rot = kf_angle[iKF] - f_angle[bestIdxF];
              if (rot < 0.0) rot += 360.0f;
              int bin = (int)roundf(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              hist[bin][hsize[bin]++] = bestIdxF;
            }
            nmatches++;
          }
          if (bestDist1R <= th_low) {
            assignment[bestIdxFR] = iKF;
            if (check_orientation) {
              float rot = kf_angle[iKF] - f_angle[bestIdxFR];
              if (rot < 0.0) rot += 360.0f;
              int bin = (int)roundf(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              hist[bin][hsize[bin]++] = bestIdxFR;
            }
            nmatches++;
          }
        }
      }
      ++a; ++b;
    } else if (kf_node_id[a] < f_node_id[b]) {
      while (a < kf_nodes && kf_node_id[a] < f_node_id[b]) ++a;   /* vFeatVecKF.lower_bound(Fit->first) */
    } else {
      while (b < f_nodes && f_node_id[b] < kf_node_id[a]) ++b;
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(hsize, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int j = 0; j < hsize[i]; j++) { assignment[hist[i][j]] = -1; nmatches--; }
    }
  }
  for (int i = 0; i < HISTO_LENGTH; ++i) free(hist[i]);
  return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12), src/ORBmatcher.cc:765-905 (loop closing,
 * map merging).  Like the keyframe / frame variant, but: a candidate of keyframe 2 must hold a good map point and must not have
 * been matched yet (vbMatched2, :822-826); features with index >= lim (mvKeysUn.size() of a fisheye rig keyframe, :801,817) are
 * skipped on both sides (lim < 0: no limit); accept bestDist1 < TH_LOW (strict, :841) and the float ratio test; the histogram
 * holds keyframe-1 indices and pruning a match leaves vbMatched2 set.  match12[n1] (out): feature of keyframe 2 or -1. */
int oracle_orb_search_by_bow_kf(int n1, int n2, int lim1, int lim2, const uint8_t* desc1, const uint8_t* desc2,
                                const uint8_t* has_mp1, const uint8_t* has_mp2,
                                int nodes1, const int32_t* node_id1, const int32_t* node_off1, const int32_t* node_feat1,
                                int nodes2, const int32_t* node_id2, const int32_t* node_off2, const int32_t* node_feat2,
                                const float* angle1, const float* angle2, float nn_ratio, int th_low, int check_orientation,
                                int32_t* match12) {
  enum { HISTO_LENGTH = 30 };
  int nmatches = 0;
  int* hist[HISTO_LENGTH];
  int hsize[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; ++i) { hist[i] = (int*)malloc(sizeof(int) * (size_t)(n1 + 1)); hsize[i] = 0; }
  uint8_t* matched2 = (uint8_t*)calloc((size_t)(n2 ? n2 : 1), 1);
  for (int i = 0; i < n1; ++i) match12[i] = -1;
  const float factor = 1.0f / HISTO_LENGTH;
  int a = 0, b = 0;
  while (a < nodes1 && b < nodes2) {
    if (node_id1[a] == node_id2[b]) {
      for (int x = node_off1[a]; x < node_off1[a + 1]; ++x) {
        const int idx1 = node_feat1[x];
        if (lim1 >= 0 && idx1 >= lim1) continue;
        if (!has_mp1[idx1]) continue;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int y = node_off2[b]; y < node_off2[b + 1]; ++y) {
          const int idx2 = node_feat2[y];
          if (lim2 >= 0 && idx2 >= lim2) continue;
          if (matched2[idx2] || !has_mp2[idx2]) continue;
          const int dist = oracle_descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 < th_low && (float)bestDist1 < nn_ratio * (float)bestDist2) {
          match12[idx1] = bestIdx2;
          matched2[bestIdx2] = 1;
          if (check_orientation) {
            float rot = angle1[idx1] - angle2[bestIdx2];
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            hist[bin][hsize[bin]++] = idx1;
          }
          nmatches++;
        }
      }
      ++a; ++b;
    } else if (node_id1[a] < node_id2[b]) {
      while (a < nodes1 && node_id1[a] < node_id2[b]) ++a;
    } else {
      while (b < nodes2 && node_id2[b] < node_id1[a]) ++b;
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(hsize, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int j = 0; j < hsize[i]; j++) { match12[hist[i][j]] = -1; nmatches--; }
    }
  }
  for (int i = 0; i < HISTO_LENGTH; ++i) free(hist[i]);
  free(matched2);
  return nmatches;
}


/* ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight) -- src/ORBmatcher.cc:1148-1338, the part after the projection
 * gates: candidate point q (in vpMapPoints order; skip[q]: null, or fails a gate of :1198-1253) has the feature list
 * cand_idx[cand_off[q] .. cand_off[q+1]) (GetFeaturesInArea order, level and chi2 gates of :1263-1300 applied).  Best = first
 * strictly smaller distance (:1307-1311); bestDist <= th_low: the slot's resident is replaced by / replaces the candidate
 * according to Observations() (:1316-1326, MapPoint::Replace src/MapPoint.cc:248-297), or the candidate is added (:1328-1332).
 * State: slot[k] = id of the keyframe's match at feature k (-1 none); points by id: candidates [0, n_q), residents 100000 + r.
 * nobs / bad / replaced / in_kf are indexed by "point index" = id for candidates, n_q + r for residents.  Every observation of a
 * point outside this keyframe is in a monocular keyframe no other point is seen in (so Replace moves each of them: +1), an
 * observation in this keyframe counts 2 when stereo[k] (mvuRight[k] >= 0, src/MapPoint.cc:140-165) else 1.  Returns nFused. */
int oracle_orb_fuse(int n_q, int n_res, int n_feat, const uint8_t* q_desc, const uint8_t* feat_desc, const uint8_t* skip,
                    const int32_t* cand_off, const int32_t* cand_idx, const uint8_t* stereo, int th_low,
                    int32_t* slot, int32_t* nobs, uint8_t* bad, int32_t* replaced, uint8_t* in_kf) {
  (void)n_feat;
  int n_fused = 0;
#define PIDX(id) ((id) >= 100000 ? n_q + ((id) - 100000) : (id))
  for (int q = 0; q < n_q; ++q) {
    if (skip[q] == 2) continue;                    /* null entry */
    if (bad[q]) continue;
    if (in_kf[q]) continue;
    if (skip[q]) continue;
    int best_dist = 256, best_idx = -1;
    for (int c = cand_off[q]; c < cand_off[q + 1]; ++c) {
      const int d = oracle_descriptor_distance(q_desc + 32 * (size_t)q, feat_desc + 32 * (size_t)cand_idx[c]);
      if (d < best_dist) { best_dist = d; best_idx = cand_idx[c]; }
    }
    if (best_dist > th_low) continue;
    const int w = stereo[best_idx] ? 2 : 1;
    const int res_id = slot[best_idx];
    if (res_id >= 0) {
      const int r = PIDX(res_id);
      if (!bad[r]) {
        if (nobs[r] > nobs[q]) {
          /* pMP->Replace(pMPinKF): the candidate's observations (none in this keyframe) move to the resident */
          nobs[r] += nobs[q]; nobs[q] = nobs[q];   /* Observations() of a replaced point keeps its count (nObs is not reset) */
          bad[q] = 1; replaced[q] = res_id;
        } else {
          /* pMPinKF->Replace(pMP): the resident's observations move to the candidate; the one in this keyframe re-points the slot */
          nobs[q] += (nobs[r] - w) + w;
          bad[r] = 1; replaced[r] = q; in_kf[r] = 0;
          slot[best_idx] = q; in_kf[q] = 1;
        }
      }
    } else {
      nobs[q] += w; in_kf[q] = 1; slot[best_idx] = q;
    }
    ++n_fused;
  }
#undef PIDX
  return n_fused;
}


/* ORBmatcher::Fuse(KeyFrame*, Sim3f&, const vector<MapPoint*>&, th, vpReplacePoint) -- src/ORBmatcher.cc:1340-1455 after its gates
 * (skip[q]: bad, already found at entry, or fails a projection gate; lists: GetFeaturesInArea order with the level gate).  A best
 * feature within th_low that holds a (non-bad: slot_bad[k] == 0) map point reports it in replace[q]; an empty one takes the
 * candidate (AddObservation + AddMapPoint: nobs[q] += 2 when stereo[k] else 1, and later candidates find it there).
 * slot[k]: id at feature k (-1 none; candidates are ids [0, n_q)).  Returns nFused. */
int oracle_orb_fuse_sim3(int n_q, const uint8_t* q_desc, const uint8_t* feat_desc, const uint8_t* skip, const int32_t* cand_off,
                         const int32_t* cand_idx, const uint8_t* stereo, const uint8_t* slot_bad, int th_low, int32_t* slot, int32_t* nobs,
                         int32_t* replace) {
  int n_fused = 0;
  for (int q = 0; q < n_q; ++q) {
    replace[q] = -1;
    if (skip[q]) continue;
    int best_dist = 0x7fffffff, best_idx = -1;
    for (int c = cand_off[q]; c < cand_off[q + 1]; ++c) {
      const int d = oracle_descriptor_distance(q_desc + 32 * (size_t)q, feat_desc + 32 * (size_t)cand_idx[c]);
      if (d < best_dist) { best_dist = d; best_idx = cand_idx[c]; }
    }
    if (best_dist > th_low) continue;
    if (slot[best_idx] >= 0) {
      if (!(slot[best_idx] >= 100000 && slot_bad[best_idx])) replace[q] = slot[best_idx];
    } else {
      nobs[q] += stereo[best_idx] ? 2 : 1;
      slot[best_idx] = q;
    }
    ++n_fused;
  }
  return n_fused;
}


/* ORBmatcher::SearchBySim3 -- src/ORBmatcher.cc:1457-1674, after the projection gates (the caller lists, per keypoint slot of one
 * keyframe that holds a usable map point, the candidates KeyFrame::GetFeaturesInArea returns in the OTHER keyframe, in its order,
 * already restricted to the levels L-1 .. L; skipK[i] = the slot has no query: no point, bad point, already matched, or a gate failed).
 * Per direction the first candidate at the smallest distance wins (`dist < bestDist`, :1549,1629) if that distance is <= TH_HIGH;
 * a pair is a match when each side chose the other (:1659-1673).  match12[i1] = slot of keyframe 2 or -1; returns nFound. */
int oracle_orb_search_by_sim3(int n1, int n2, const uint8_t* desc_mp1, const uint8_t* desc_mp2, const uint8_t* desc_kf1, const uint8_t* desc_kf2,
                              const uint8_t* skip1, const int32_t* off1, const int32_t* idx1, const uint8_t* skip2, const int32_t* off2,
                              const int32_t* idx2, int th_high, int32_t* match12) {
  int32_t* m1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n1 + 1));
  int32_t* m2 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
  for (int i = 0; i < n1; ++i) {
    m1[i] = -1;
    if (skip1[i]) continue;
    int best = 0x7fffffff, bi = -1;
    for (int c = off1[i]; c < off1[i + 1]; ++c) {
      const int d = oracle_descriptor_distance(desc_mp1 + 32 * (size_t)i, desc_kf2 + 32 * (size_t)idx1[c]);
      if (d < best) { best = d; bi = idx1[c]; }
    }
    if (best <= th_high) m1[i] = bi;
  }
  for (int i = 0; i < n2; ++i) {
    m2[i] = -1;
    if (skip2[i]) continue;
    int best = 0x7fffffff, bi = -1;
    for (int c = off2[i]; c < off2[i + 1]; ++c) {
      const int d = oracle_descriptor_distance(desc_mp2 + 32 * (size_t)i, desc_kf1 + 32 * (size_t)idx2[c]);
      if (d < best) { best = d; bi = idx2[c]; }
    }
    if (best <= th_high) m2[i] = bi;
  }
  int n_found = 0;
  for (int i1 = 0; i1 < n1; ++i1) {
    match12[i1] = -1;
    const int i2 = m1[i1];
    if (i2 >= 0 && m2[i2] == i1) { match12[i1] = i2; ++n_found; }
  }
  free(m1); free(m2);
  return n_found;
}

/* ORBmatcher::SearchForTriangulation -- src/ORBmatcher.cc:907-1146, two pinhole keyframes (no mpCamera2).  kp = x y angle uright per
 * feature (uright < 0: monocular), F12 = K1^-T [t12]x R12 K2^-1 row-major (src/CameraModels/Pinhole.cpp:107-112), ep = the epipole in
 * image 2, level_sigma2_2 / scale_factor_2 by octave of keyframe 2.  Per unmatched feature of keyframe 1, in the order of the
 * vocabulary walk: candidates = unmatched features of keyframe 2 in the same node; `dist > TH_LOW || dist > bestDist` skips, so a
 * later candidate at the same distance replaces the earlier one (:1020); a monocular pair must lie farther than 10 px (scaled) from
 * the epipole (:1033-1044); accepted when coarse or dsqr < 3.84 sigma2 of the second keypoint's level (Pinhole.cpp:114-128). */
int oracle_orb_search_for_triangulation(int n1, int n2, const uint8_t* desc1, const uint8_t* desc2, const uint8_t* has_mp1, const uint8_t* has_mp2,
                                        int nodes1, const int32_t* node_id1, const int32_t* node_off1, const int32_t* node_feat1,
                                        int nodes2, const int32_t* node_id2, const int32_t* node_off2, const int32_t* node_feat2,
                                        const float* kp1, const float* kp2, const int32_t* octave2, const float* F12, const float* ep,
                                        const float* scale_factor_2, const float* level_sigma2_2, int only_stereo, int coarse, int th_low,
                                        int check_orientation, int32_t* match12) {
  enum { HISTO_LENGTH = 30 };
  (void)n2;
  int nmatches = 0;
  int* hist[HISTO_LENGTH];
  int hsize[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; ++i) { hist[i] = (int*)malloc(sizeof(int) * (size_t)(n1 + 1)); hsize[i] = 0; }
  for (int i = 0; i < n1; ++i) match12[i] = -1;
  const float factor = 1.0f / HISTO_LENGTH;
  int a = 0, b = 0;
  while (a < nodes1 && b < nodes2) {
    if (node_id1[a] == node_id2[b]) {
      for (int x = node_off1[a]; x < node_off1[a + 1]; ++x) {
        const int idx1 = node_feat1[x];
        if (has_mp1[idx1]) continue;
        const int stereo1 = kp1[4 * idx1 + 3] >= 0;
        if (only_stereo && !stereo1) continue;
        int bestDist = th_low, bestIdx2 = -1;
        for (int y = node_off2[b]; y < node_off2[b + 1]; ++y) {
          const int idx2 = node_feat2[y];
          if (has_mp2[idx2]) continue;
          const int stereo2 = kp2[4 * idx2 + 3] >= 0;
          if (only_stereo && !stereo2) continue;
          const int dist = oracle_descriptor_distance(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
          if (dist > th_low || dist > bestDist) continue;
          const float x2 = kp2[4 * idx2], y2 = kp2[4 * idx2 + 1];
          if (!stereo1 && !stereo2) {
            const float distex = ep[0] - x2, distey = ep[1] - y2;
            if (distex * distex + distey * distey < 100 * scale_factor_2[octave2[idx2]]) continue;
          }
          int ok = coarse;
          if (!ok) {
            const float x1 = kp1[4 * idx1], y1 = kp1[4 * idx1 + 1];
            const float la = x1 * F12[0] + y1 * F12[3] + F12[6];
            const float lb = x1 * F12[1] + y1 * F12[4] + F12[7];
            const float lc = x1 * F12[2] + y1 * F12[5] + F12[8];
            const float num = la * x2 + lb * y2 + lc;
            const float den = la * la + lb * lb;
            if (den != 0) { const float dsqr = num * num / den; ok = dsqr < 3.84 * level_sigma2_2[octave2[idx2]]; }
          }
          if (ok) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
          match12[idx1] = bestIdx2;
          nmatches++;
          if (check_orientation) {
            float rot = kp1[4 * idx1 + 2] - kp2[4 * bestIdx2 + 2];
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            hist[bin][hsize[bin]++] = idx1;
          }
        }
      }
      ++a; ++b;
    } else if (node_id1[a] < node_id2[b]) {
      while (a < nodes1 && node_id1[a] < node_id2[b]) ++a;
    } else {
      while (b < nodes2 && node_id2[b] < node_id1[a]) ++b;
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(hsize, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int j = 0; j < hsize[i]; j++) { match12[hist[i][j]] = -1; nmatches--; }
    }
  }
  for (int i = 0; i < HISTO_LENGTH; ++i) free(hist[i]);
  return nmatches;
}


/* ORBmatcher::SearchForInitialization -- src/ORBmatcher.cc:648-763 after the candidate generation: keypoint q of frame 1 (skip[q]:
 * level > 0 or an empty window) has the candidate list cand_idx[cand_off[q] .. cand_off[q+1]) (Frame::GetFeaturesInArea order, level
 * 0).  A candidate is passed over while the distance it is currently matched with is <= this one (:680-681); best / second best;
 * bestDist <= th_low and bestDist < nn_ratio * bestDist2 (float): the match replaces an earlier match of the same feature (:698-702).
 * Orientation histogram on angle1 - angle2.  prev_xy (vbPrevMatched) is updated with the matched positions (:755-757). */
int oracle_orb_search_for_initialization(int n1, int n2, const uint8_t* desc1, const uint8_t* desc2, const uint8_t* skip, const int32_t* cand_off,
                                         const int32_t* cand_idx, const float* angle1, const float* angle2, const float* xy2, float nn_ratio,
                                         int th_low, int check_orientation, int32_t* match12, float* prev_xy) {
  enum { HISTO_LENGTH = 30 };
  int nmatches = 0;
  int* hist[HISTO_LENGTH];
  int hsize[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; ++i) { hist[i] = (int*)malloc(sizeof(int) * (size_t)(n1 + 1)); hsize[i] = 0; }
  int* matched_dist = (int*)malloc(sizeof(int) * (size_t)(n2 ? n2 : 1));
  int* match21 = (int*)malloc(sizeof(int) * (size_t)(n2 ? n2 : 1));
  for (int i = 0; i < n2; ++i) { matched_dist[i] = 0x7fffffff; match21[i] = -1; }
  for (int i = 0; i < n1; ++i) match12[i] = -1;
  const float factor = 1.0f / HISTO_LENGTH;
  for (int i1 = 0; i1 < n1; ++i1) {
    if (skip[i1]) continue;
    int bestDist = 0x7fffffff, bestDist2 = 0x7fffffff, bestIdx2 = -1;
    for (int c = cand_off[i1]; c < cand_off[i1 + 1]; ++c) {
      const int i2 = cand_idx[c];
      const int dist = oracle_descriptor_distance(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
      if (matched_dist[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist <= th_low && (float)bestDist < (float)bestDist2 * nn_ratio) {
      if (match21[bestIdx2] >= 0) { match12[match21[bestIdx2]] = -1; nmatches--; }
      match12[i1] = bestIdx2;
      match21[bestIdx2] = i1;
      matched_dist[bestIdx2] = bestDist;
      nmatches++;
      if (check_orientation) {
        float rot = angle1[i1] - angle2[bestIdx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        hist[bin][hsize[bin]++] = i1;
      }
    }
  }
  if (check_orientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    three_maxima(hsize, HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (int j = 0; j < hsize[i]; j++)
        if (match12[hist[i][j]] >= 0) { match12[hist[i][j]] = -1; nmatches--; }
    }
  }
  for (int i1 = 0; i1 < n1; ++i1)
    if (match12[i1] >= 0) { prev_xy[2 * i1] = xy2[2 * match12[i1]]; prev_xy[2 * i1 + 1] = xy2[2 * match12[i1] + 1]; }
  for (int i = 0; i < HISTO_LENGTH; ++i) free(hist[i]);
  free(matched_dist); free(match21);
  return nmatches;
}
