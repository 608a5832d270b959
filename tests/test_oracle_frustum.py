"""CPU restatement of Frame::isInFrustum (oracle/frustum_oracle.c) against hand-computed cases.

PARITY UNPINNED against a reference binary (the reference cannot be built here: Eigen/OpenCV absent); the expected
values below are worked out by hand from src/Frame.cc:513-587, Pinhole::project (src/CameraModels/Pinhole.cpp:43-49)
and MapPoint::PredictScale (src/MapPoint.cc:531-546)."""
import numpy as np

from orb_slam3_study_kr_amd import orb
from oracle import binding as oracle

FX, FY, CX, CY, BF = 400.0, 400.0, 320.0, 240.0, 40.0
BOUNDS = (0.0, 640.0, 0.0, 480.0)
LOG_SF = float(np.log(np.float32(1.2)))


def identity_frame(limit=0.5):
    return orb.frustum_frame(np.eye(3), np.zeros(3), FX, FY, CX, CY, BF, BOUNDS, LOG_SF, 8, limit)


def test_hand_cases():
    f = identity_frame()
    pos = np.array([
        [0.0, 0.0, 4.0],     # on the optical axis, seen head-on: in view
        [0.0, 0.0, -1.0],    # behind the camera
        [10.0, 0.0, 4.0],    # projects to u = 1320: outside the image
        [1.0, 0.5, 4.0],     # in the image but farther than 1.2 * max distance
        [1.0, 0.5, 4.0],     # in range but seen at 90 degrees from its normal
        [1.0, 0.5, 4.0],     # in view
    ], dtype=np.float32)
    normal = np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [1, 0, 0], [0, 0, 1]], dtype=np.float32)
    min_d = np.array([1.0, 1.0, 1.0, 1.0, 1.0, 1.0], dtype=np.float32)
    max_d = np.array([8.0, 8.0, 20.0, 3.0, 8.0, 8.0], dtype=np.float32)
    o = oracle.frustum(f, pos, normal, min_d, max_d)
    assert o["stage"].tolist() == [2, 0, 0, 1, 1, 2]
    # rejected before the projection is stored: mTrackProjX/Y stay -1 (:515-517)
    assert o["proj_x"][1] == -1 and o["proj_y"][1] == -1 and o["proj_x"][2] == -1
    # stored, then rejected (:539-540)
    assert o["proj_x"][3] == np.float32(420.0) and o["proj_y"][3] == np.float32(290.0)
    # point 0: u = cx, v = cy, depth 4, viewCos 1, uR = u - bf / z = 310; ratio = 8 / 4 -> ceil(log 2 / log 1.2) = 4
    assert (o["proj_x"][0], o["proj_y"][0], o["proj_xr"][0], o["depth"][0], o["view_cos"][0], o["level"][0]) == (320.0, 240.0, 310.0, 4.0, 1.0, 4)
    # point 5: u = 400 * 1 / 4 + 320, v = 400 * .5 / 4 + 240; dist = sqrt(17.25); viewCos = 4 / dist
    d = np.sqrt(np.float32(17.25))
    assert o["proj_x"][5] == np.float32(420.0) and o["proj_y"][5] == np.float32(290.0)
    assert o["depth"][5] == d and o["view_cos"][5] == np.float32(4.0) / d
    assert o["level"][5] == int(np.ceil(np.log(np.float32(8.0) / d) / np.float32(LOG_SF)))


def test_level_is_clamped_and_cos_limit_is_inclusive():
    f = identity_frame(limit=1.0)
    pos = np.array([[0, 0, 1.0], [0, 0, 100.0], [0, 0, 4.0]], dtype=np.float32)
    normal = np.tile(np.array([[0, 0, 1]], dtype=np.float32), (3, 1))
    o = oracle.frustum(f, pos, normal, np.array([0.1, 0.1, 0.1], np.float32), np.array([1000.0, 90.0, 8.0], np.float32))
    # ratio 1000 -> level 37 clamps to 7; ratio 0.9 -> negative clamps to 0; viewCos == limit passes (`<` at :556)
    assert o["stage"].tolist() == [2, 2, 2]
    assert o["level"].tolist() == [7, 0, 4]


def test_pose_is_applied_before_projection():
    # camera 2 m to the right of the origin looking down +z: tcw = -R * C
    R = np.eye(3, dtype=np.float32)
    f = orb.frustum_frame(R, np.array([-2.0, 0, 0], np.float32), FX, FY, CX, CY, BF, BOUNDS, LOG_SF, 8)
    o = oracle.frustum(f, np.array([[2.0, 0, 5.0]], np.float32), np.array([[0, 0, 1.0]], np.float32),
                       np.array([1.0], np.float32), np.array([10.0], np.float32))
    assert o["stage"][0] == 2 and o["proj_x"][0] == 320.0 and o["depth"][0] == 5.0 and o["view_cos"][0] == 1.0
