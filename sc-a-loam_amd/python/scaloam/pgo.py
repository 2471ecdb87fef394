"""Host-side glue of the pose-graph node around the hot path (laserPosegraphOptimization.cpp): the keyframe gate in front of the
ScanContext insert (:598-617, with getOdom :312-323 and diffTransformation :325-336) and the g2o text of the saved pose graph
(:147-175, :198-216).  No GPU work here; GTSAM / iSAM2 itself stays with the caller (out of scope)."""
import numpy as np


def rpy_from_quat(q_xyzw):
    """tf::Matrix3x3(tf::Quaternion).getRPY (:318-320), in double"""
    x, y, z, w = [float(v) for v in q_xyzw]
    n = x * x + y * y + z * z + w * w
    s = 2.0 / n
    m = np.array([[1 - s * (y * y + z * z), s * (x * y - w * z), s * (x * z + w * y)],
                  [s * (x * y + w * z), 1 - s * (x * x + z * z), s * (y * z - w * x)],
                  [s * (x * z - w * y), s * (y * z + w * x), 1 - s * (x * x + y * y)]])
    if abs(m[2, 0]) >= 1.0:  # gimbal lock branch of tf's getEulerYPR
        yaw = 0.0
        if m[2, 0] < 0:
            pitch, roll = np.pi / 2, np.arctan2(m[0, 1], m[0, 2])
        else:
            pitch, roll = -np.pi / 2, np.arctan2(-m[0, 1], -m[0, 2])
        return roll, pitch, yaw
    pitch = -np.arcsin(m[2, 0])
    roll = np.arctan2(m[2, 1] / np.cos(pitch), m[2, 2] / np.cos(pitch))
    yaw = np.arctan2(m[1, 0] / np.cos(pitch), m[0, 0] / np.cos(pitch))
    return roll, pitch, yaw


def _affine_f32(x, y, z, roll, pitch, yaw):
    """pcl::getTransformation (f32 Affine3f: R = Rz(yaw) Ry(pitch) Rx(roll))"""
    A, B, C_, D, E, F = (np.float32(np.cos(yaw)), np.float32(np.sin(yaw)), np.float32(np.cos(pitch)), np.float32(np.sin(pitch)),
                         np.float32(np.cos(roll)), np.float32(np.sin(roll)))
    DE, DF = D * E, D * F
    T = np.eye(4, dtype=np.float32)
    T[0, :3] = [A * C_, A * DF - B * E, B * F + A * DE]
    T[1, :3] = [B * C_, A * E + B * DF, B * DE - A * F]
    T[2, :3] = [-D, C_ * F, C_ * E]
    T[:3, 3] = [x, y, z]
    return T


def diff_transformation(p1, p2):
    """diffTransformation (:325-336): |dx|,|dy|,|dz|,|droll|,|dpitch|,|dyaw| of p1^-1 * p2, f32 as in the reference"""
    d = (np.linalg.inv(_affine_f32(*p1)) @ _affine_f32(*p2)).astype(np.float32)
    droll = np.arctan2(d[2, 1], d[2, 2])
    dpitch = np.arcsin(-d[2, 0])
    dyaw = np.arctan2(d[1, 0], d[0, 0])
    return np.abs(np.array([d[0, 3], d[1, 3], d[2, 3], droll, dpitch, dyaw], np.float64))


class KeyframeGate:
    """Early reject by counting local delta movement (:598-617).  The accumulators start at 1000000 (:67-68), so the first pose is
    always a keyframe."""

    def __init__(self, meter_gap=2.0, deg_gap=10.0):
        self.meter_gap = float(meter_gap)
        self.rad_gap = np.deg2rad(float(deg_gap))  # :875-876
        self.tr = 1000000.0
        self.rot = 1000000.0
        self.curr = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)  # odom_pose_curr {0.0, ...} (:75)

    def __call__(self, q_xyzw, t_xyz):
        pose = (float(t_xyz[0]), float(t_xyz[1]), float(t_xyz[2])) + tuple(rpy_from_quat(q_xyzw))
        prev, self.curr = self.curr, pose
        d = diff_transformation(prev, self.curr)
        self.tr += float(np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]))
        self.rot += float(d[3] + d[4] + d[5])
        if self.tr > self.meter_gap or self.rot > self.rad_gap:
            self.tr, self.rot = 0.0, 0.0
            return True
        return False


def _f6(v):
    return "%f" % v  # std::to_string(double)


def quat_from_rpy(roll, pitch, yaw):
    """gtsam::Rot3::RzRyRx(roll, pitch, yaw).toQuaternion() (:193-196), (x, y, z, w)"""
    cr, sr, cp, sp, cy, sy = np.cos(roll / 2), np.sin(roll / 2), np.cos(pitch / 2), np.sin(pitch / 2), np.cos(yaw / 2), np.sin(yaw / 2)
    return (sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy)


def g2o_vertex(idx, t, q_xyzw):
    """getVertexStr (:147-161)"""
    return "VERTEX_SE3:QUAT %d %s %s %s %s %s %s %s" % ((idx,) + tuple(_f6(v) for v in (t[0], t[1], t[2], q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3])))


def g2o_edge(i, j, t, q_xyzw):
    """writeEdge (:163-176)"""
    return "EDGE_SE3:QUAT %d %d %s %s %s %s %s %s %s" % ((i, j) + tuple(_f6(v) for v in (t[0], t[1], t[2], q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3])))


def write_g2o(path, keyframe_poses6d, edges):
    """saveGTSAMgraphG2oFormat (:198-216): one vertex per keyframe pose (x, y, z, roll, pitch, yaw), then the edge lines as collected
    (odometry edges :668-683, loop edges :759-771); edges = [(i, j, t_xyz, q_xyzw), ...]"""
    with open(path, "w") as f:
        for k, p in enumerate(keyframe_poses6d):
            f.write(g2o_vertex(k, p[:3], quat_from_rpy(p[3], p[4], p[5])) + "\n")
        for i, j, t, q in edges:
            f.write(g2o_edge(i, j, t, q) + "\n")


def read_g2o(path):
    """vertices {idx: (t, q_xyzw)} and edges [(i, j, t, q_xyzw)] of a singlesession_posegraph.g2o"""
    V, E = {}, []
    for ln in open(path):
        w = ln.split()
        if not w:
            continue
        if w[0] == "VERTEX_SE3:QUAT":
            v = [float(x) for x in w[2:9]]
            V[int(w[1])] = (np.array(v[:3]), np.array(v[3:]))
        elif w[0] == "EDGE_SE3:QUAT":
            v = [float(x) for x in w[3:10]]
            E.append((int(w[1]), int(w[2]), np.array(v[:3]), np.array(v[3:])))
    return V, E
