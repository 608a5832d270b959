/*
 * frustum_oracle.c -- CPU restatement of ORB_SLAM3::Frame::isInFrustum for the monocular /
 * rectified-stereo layout (Nleft == -1), src/Frame.cc:513-587, with Pinhole::project(Vector3f)
 * (src/CameraModels/Pinhole.cpp:43-49) and MapPoint::PredictScale(float, Frame*)
 * (src/MapPoint.cc:531-546).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see lba_oracle.c header).
 * PARITY UNPINNED against a reference binary (Eigen/OpenCV absent, the reference cannot be built);
 * pinned by the hand-computed cases in tests/test_oracle_frustum.py.  Float32 like the reference;
 * built with -ffp-contract=off; three-term sums in the order of Eigen's fixed-size reduction,
 * a0 + (a1 + a2) (recalled behaviour, Eigen is not vendored).
 *
 * Paths relative to /root/reference.
 */
#include <math.h>
#include <stddef.h>
#include "oracle.h"

static float sum3(float a0, float a1, float a2) { return a0 + (a1 + a2); }

void oracle_frustum(const osh_frustum_frame* f, const osh_frustum_points* p, osh_frustum_result* out) {
  for (int i = 0; i < p->n; ++i) {
    const float* P = p->pos + 3 * (size_t)i;
    const float* Pn = p->normal + 3 * (size_t)i;
    /* :515-517 */
    out->stage[i] = 0; out->proj_x[i] = -1.f; out->proj_y[i] = -1.f;
    out->proj_xr[i] = 0.f; out->view_cos[i] = 0.f; out->level[i] = -1;
    /* :523-524  Pc = mRcw * P + mtcw */
    float Pc[3];
    for (int r = 0; r < 3; ++r) Pc[r] = sum3(f->Rcw[3 * r] * P[0], f->Rcw[3 * r + 1] * P[1], f->Rcw[3 * r + 2] * P[2]) + f->tcw[r];
    const float pc_dist = sqrtf(sum3(Pc[0] * Pc[0], Pc[1] * Pc[1], Pc[2] * Pc[2]));
    out->depth[i] = pc_dist;
    /* :527-530 */
    const float invz = 1.0f / Pc[2];
    if (Pc[2] < 0.0f) continue;
    /* :532-537 */
    float u = f->fx * Pc[0] / Pc[2] + f->cx;
    float v = f->fy * Pc[1] / Pc[2] + f->cy;
    if (f->fisheye) {
      /* KannalaBrandt8::project(Vector3f), src/CameraModels/KannalaBrandt8.cpp:66-84; atan2f / cosf / sinf as their correctly
       * rounded values (double function rounded once), the libm-independent convention the device reproduces */
      const float x2y2 = Pc[0] * Pc[0] + Pc[1] * Pc[1];
      const float theta = (float)atan2((double)sqrtf(x2y2), (double)Pc[2]);
      const float psi = (float)atan2((double)Pc[1], (double)Pc[0]);
      const float t2 = theta * theta, t3 = theta * t2, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
      const float rr = theta + f->kb8[0] * t3 + f->kb8[1] * t5 + f->kb8[2] * t7 + f->kb8[3] * t9;
      u = f->fx * rr * (float)cos((double)psi) + f->cx;
      v = f->fy * rr * (float)sin((double)psi) + f->cy;
    }
    if (u < f->min_x || u > f->max_x) continue;
    if (v < f->min_y || v > f->max_y) continue;
    /* :539-540 */
    out->stage[i] = 1; out->proj_x[i] = u; out->proj_y[i] = v;
    /* :543-549 */
    const float max_d = 1.2f * p->max_dist[i], min_d = 0.8f * p->min_dist[i];
    const float PO[3] = {P[0] - f->Ow[0], P[1] - f->Ow[1], P[2] - f->Ow[2]};
    const float dist = sqrtf(sum3(PO[0] * PO[0], PO[1] * PO[1], PO[2] * PO[2]));
    if (dist < min_d || dist > max_d) continue;
    /* :552-557 */
    const float view_cos = sum3(PO[0] * Pn[0], PO[1] * Pn[1], PO[2] * Pn[2]) / dist;
    if (view_cos < f->viewing_cos_limit) continue;
    /* :560, MapPoint.cc:531-546 (log of a float under `using namespace std` resolves to the float overload) */
    const float ratio = p->max_dist[i] / dist;
    /* logf taken as its correctly rounded value (double log rounded once) so that the device can reproduce it bit for bit;
     * glibc's logf differs from this in rare arguments by one ulp (only matters exactly on a level boundary) */
    const float lg = (float)log((double)ratio);
    int ns = (int)ceilf(lg / f->log_scale_factor);
    if (ns < 0) ns = 0; else if (ns >= f->n_scale_levels) ns = f->n_scale_levels - 1;
    /* :563-571 */
    out->stage[i] = 2;
    out->proj_xr[i] = u - f->bf * invz;
    out->view_cos[i] = view_cos;
    out->level[i] = ns;
  }
}
