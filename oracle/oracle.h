/*
 * oracle.h -- entry points of the CPU oracle (TEST INFRASTRUCTURE ONLY, see
 * lba_oracle.c / orb_oracle.c headers).  Built into oracle/liboracle.so by
 * oracle/Makefile; loaded with ctypes from oracle/binding.py.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#include "../include/orbslam3_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- local BA (lba_oracle.c) ---- */
int  oracle_lba_solve(const osh_lba_problem* p, osh_lba_result* res);
int  oracle_lba_linearize(const osh_lba_problem* p, double* Hpp, double* bp, double* Hll, double* bl,
                          double* Hpl, double* chi2, double* robust_chi2);
int  oracle_lba_schur_step(const osh_lba_problem* p, double lambda, double* S, double* bs, double* x);
void oracle_pose_oplus(const double update[6], double qt[7]);
void oracle_edge_error(int kind, const double qt[7], const double cam[5], const double X[3],
                       const double obs[3], double err[3]);
void oracle_edge_jacobians(int kind, const double qt[7], const double cam[5], const double X[3],
                           double Jxi[9], double Jxj[18]);
int  oracle_edge_depth_positive(const double qt[7], const double X[3]);
void oracle_huber(double e, double delta, double rho[3]);
void oracle_kb8_project(const double cam[5], const double kb[4], const double Xc[3], double uv[2]);
void oracle_kb8_project_jac(const double cam[5], const double kb[4], const double Xc[3], double J[6]);
void oracle_edge_error_kb8(const double qt[7], const double cam[5], const double kb[4], const double X[3], const double obs[3], double err[3]);
void oracle_edge_jacobians_kb8(const double qt[7], const double cam[5], const double kb[4], const double X[3], double Jxi[9], double Jxj[18]);
void oracle_edge_error_body(const double qt[7], const double cam2[8], const double trl[7], const double X[3], const double obs[3], double err[3]);
void oracle_edge_jacobians_body(const double qt[7], const double cam2[8], const double trl[7], const double X[3], double Jxi[9], double Jxj[18]);
int  oracle_edge_depth_positive_body(const double qt[7], const double trl[7], const double X[3]);
int  oracle_ldlt_solve(int n, double* A, const double* b, double* x, double* tmp);

/* ---- frustum projection of map points into a frame (frustum_oracle.c) ---- */
void oracle_frustum(const osh_frustum_frame* f, const osh_frustum_points* p, osh_frustum_result* out);

/* ---- pose-only optimisation of a frame (pose_oracle.c) ---- */
int  oracle_pose_optimize(const osh_pose_problem* p, osh_pose_result* res);

/* ---- local inertial BA (liba_oracle.c) ---- */
int  oracle_liba_solve(const osh_liba_problem* p, osh_liba_result* res);
int  oracle_liba_linearize(const osh_liba_problem* p, double* H, double* b, double* Hll, double* Hpl, double* chi2);
int  oracle_liba_inertial_edge(const osh_liba_problem* p, int link, double* r9, double* J9x24);
void oracle_exp_so3(const double* w, double* R);
int  oracle_posei_optimize(const osh_posei_problem* p, osh_posei_result* res);
int  oracle_posei_linearize(const osh_posei_problem* p, double* H, double* b);
void oracle_marginalize_previous(const double* H30, double* out15);
void oracle_constraint_pose_imu_H(double* H15);
void oracle_log_so3(const double* R, double* w);

/* ---- ORB matching (orb_oracle.c) ---- */
int  oracle_descriptor_distance(const uint8_t* a, const uint8_t* b);
void oracle_distance_matrix(int n, int m, const uint8_t* a, const uint8_t* b, int32_t* out);
void oracle_orb_search(int n_query, int n_train, const uint8_t* query_desc, const uint8_t* train_desc,
                       const int32_t* train_level, const int32_t* cand_off, const int32_t* cand_idx,
                       const uint8_t* occupied,
                       int32_t* best_idx, int32_t* best_dist, int32_t* second_dist,
                       int32_t* best_level, int32_t* second_level);
int  oracle_orb_match_local_points(int n_query, int n_train, const uint8_t* query_desc,
                                   const uint8_t* train_desc, const int32_t* train_level,
                                   const int32_t* cand_off, const int32_t* cand_idx,
                                   float nn_ratio, int th_high, uint8_t* occupied, int32_t* assignment);
int  oracle_orb_search_by_bow(int n_kf, int n_f, int n_left_f, const uint8_t* kf_desc, const uint8_t* f_desc, const uint8_t* kf_has_mp,
                              int kf_nodes, const int32_t* kf_node_id, const int32_t* kf_node_off, const int32_t* kf_node_feat,
                              int f_nodes, const int32_t* f_node_id, const int32_t* f_node_off, const int32_t* f_node_feat,
                              const float* kf_angle, const float* f_angle, float nn_ratio, int th_low, int check_orientation,
                              int32_t* assignment);
int  oracle_orb_search_by_bow_kf(int n1, int n2, int lim1, int lim2, const uint8_t* desc1, const uint8_t* desc2,
                                 const uint8_t* has_mp1, const uint8_t* has_mp2,
                                 int nodes1, const int32_t* node_id1, const int32_t* node_off1, const int32_t* node_feat1,
                                 int nodes2, const int32_t* node_id2, const int32_t* node_off2, const int32_t* node_feat2,
                                 const float* angle1, const float* angle2, float nn_ratio, int th_low, int check_orientation,
                                 int32_t* match12);
int  oracle_orb_search_for_triangulation(int n1, int n2, const uint8_t* desc1, const uint8_t* desc2, const uint8_t* has_mp1, const uint8_t* has_mp2,
                                         int nodes1, const int32_t* node_id1, const int32_t* node_off1, const int32_t* node_feat1,
                                         int nodes2, const int32_t* node_id2, const int32_t* node_off2, const int32_t* node_feat2,
                                         const float* kp1, const float* kp2, const int32_t* octave2, const float* F12, const float* ep,
                                         const float* scale_factor_2, const float* level_sigma2_2, int only_stereo, int coarse, int th_low,
                                         int check_orientation, int32_t* match12);
int  oracle_orb_search_for_initialization(int n1, int n2, const uint8_t* desc1, const uint8_t* desc2, const uint8_t* skip, const int32_t* cand_off,
                                          const int32_t* cand_idx, const float* angle1, const float* angle2, const float* xy2, float nn_ratio,
                                          int th_low, int check_orientation, int32_t* match12, float* prev_xy);
int  oracle_orb_fuse(int n_q, int n_res, int n_feat, const uint8_t* q_desc, const uint8_t* feat_desc, const uint8_t* skip,
                     const int32_t* cand_off, const int32_t* cand_idx, const uint8_t* stereo, int th_low,
                     int32_t* slot, int32_t* nobs, uint8_t* bad, int32_t* replaced, uint8_t* in_kf);
int  oracle_orb_search_by_sim3(int n1, int n2, const uint8_t* desc_mp1, const uint8_t* desc_mp2, const uint8_t* desc_kf1, const uint8_t* desc_kf2,
                               const uint8_t* skip1, const int32_t* off1, const int32_t* idx1, const uint8_t* skip2, const int32_t* off2,
                               const int32_t* idx2, int th_high, int32_t* match12);
int  oracle_orb_fuse_sim3(int n_q, const uint8_t* q_desc, const uint8_t* feat_desc, const uint8_t* skip, const int32_t* cand_off,
                          const int32_t* cand_idx, const uint8_t* stereo, const uint8_t* slot_bad, int th_low, int32_t* slot, int32_t* nobs,
                          int32_t* replace);
int  oracle_orb_match_local_points_rig(int n_query, int n_left, int n_right, const uint8_t* query_desc, const uint8_t* desc,
                                       const int32_t* level_left, const int32_t* level_right,
                                       const uint8_t* in_l, const int32_t* candl_off, const int32_t* candl_idx,
                                       const uint8_t* in_r, const int32_t* candr_off, const int32_t* candr_idx,
                                       const int32_t* left_to_right, const int32_t* right_to_left,
                                       float nn_ratio, int th_high, uint8_t* occupied, int32_t* assignment);
int  oracle_orb_match_last_frame(int n_query, int n_train, const uint8_t* query_desc,
                                 const uint8_t* train_desc, const int32_t* cand_off, const int32_t* cand_idx,
                                 const float* query_angle, const float* train_angle,
                                 int th_high, int check_orientation, uint8_t* occupied, int32_t* assignment);
int  oracle_orb_match_last_frame_rig(int n_query, int n_left, int n_right, const uint8_t* query_desc, const uint8_t* desc,
                                     const int32_t* candl_off, const int32_t* candl_idx, const int32_t* candr_off, const int32_t* candr_idx,
                                     const float* query_angle, const float* angle_left, const float* angle_right,
                                     int th_high, int check_orientation, uint8_t* occupied, int32_t* assignment);
#ifdef __cplusplus
}
#endif
#endif
