"""Seeded synthetic frames for the ESKF hot path.

Scene: a 20 x 20 x 5 m room plus three slanted planes.  A *map* is sampled on the planes
(spacing ~ filter_size_map, 1 cm normal noise); a *scan* of N points on the same planes is expressed
in the LiDAR frame at a true pose T*; the prior is T* perturbed by U(-0.5,0.5) deg / U(-2,2) cm
(magnitudes taken from /root/reference/Log/mat_out.txt).  Neighbours come
from scipy's cKDTree, the stand-in for the host ikd-Tree (KD_TREE::Nearest_Search,
/root/reference/include/ikd-Tree/ikd_Tree.cpp:350-380): 5 nearest map points, ascending distance,
float32, valid iff 5 were found and the 5th squared distance is <= 5
(/root/reference/src/laserMapping.cpp:1549,1567).

The VIO part renders a smooth 640 x 512 u8 texture, places M map points in front of the camera and
builds their 3-level 8x8 reference patches by sampling the image at the true pose with the same
anchor/bilinear formula the update uses (/root/reference/src/lidar_selection.cpp:807-837), plus noise.
"""
from __future__ import annotations

import dataclasses
import numpy as np

SEED = 20241108

# /root/reference/config/avia.yaml:32-35,42-45 ; config/camera_pinhole.yaml
AVIA_T_LI = np.array([0.04165, 0.02326, -0.0284])
AVIA_R_LI = np.eye(3)
AVIA_RCL = np.array([[0.00162756, -0.999991, 0.00390957],
                     [-0.0126748, -0.00392989, -0.999912],
                     [0.999918, 0.00157786, -0.012681]])
AVIA_PCL = np.array([0.0409257, 0.0318424, -0.0927219])
PINHOLE = dict(width=640, height=512, fx=431.795259219, fy=431.550090267,
               cx=310.833037316, cy=266.985989326,
               d=(-0.0944205499243979, 0.0946727677776504, -0.00807970960613932,
                  8.07461209775283e-05, 0.0))
# /root/reference/config/NTU_VIRAL.yaml:32-35,43-46 ; config/camera_NTU_VIRAL.yaml
NTU_T_LI = np.zeros(3)
NTU_RCL = np.array([[0.0218308, 0.99976, -0.00201407],
                    [-0.0131205, 0.00230088, 0.999911],
                    [0.999676, -0.0218025, 0.0131676]])
NTU_PCL = np.array([0.122993, 0.0398643, -0.0577101])
NTU_CAM = dict(width=752, height=480, fx=4.250258563372763e+02, fy=4.267976260903337e+02,
               cx=3.860151866550880e+02, cy=2.419130336743440e+02,
               d=(-0.288105327549552, 0.074578284234601, 7.784489598138802e-04,
                  -2.277853975035461e-04, 0.0))

# /root/reference/config/MARS_LVIG.yaml:3,9,15-16,36-39,47-50 ; config/camera_MARS_LVIG.yaml:3-12 (HKisland / HKairport calibration)
MARS_RCL = np.array([[0.00438814, -0.999807, -0.0191582],
                     [-0.00978695, 0.0191145, -0.999769],
                     [0.999942, 0.00457463, -0.00970118]])
MARS_PCL = np.array([0.016069, 0.0871753, -0.0718021])
MARS_CAM = dict(width=1224, height=1024, fx=722.215831395, fy=722.171768344, cx=588.900539701, cy=521.800513284,
                d=(-0.05729528706141188, 0.1210407244166642, 0.001274128378760289, 0.0004389741530109464, 0.0))
# /root/reference/config/mid360.yaml:3,32-35,43-46 (max_iteration 5; the camera intrinsics are camera_pinhole.yaml's)
MID360_T_LI = np.array([-0.011, -0.02329, 0.04412])
MID360_RCL = np.array([[0.0268125, -0.999465, 0.0187293],
                       [-0.157156, -0.0227175, -0.987312],
                       [0.98721, 0.0235289, -0.157681]])
MID360_PCL = np.array([-0.112954, 0.0328782, -0.308706])

LASER_POINT_COV = 0.001   # avia.yaml:16
IMG_POINT_COV = 100.0     # avia.yaml:15
INIT_COV = 0.001          # common_lib.h:38


def exp_so3(v):
    """Rodrigues, same thresholding as so3_math.h:54-72."""
    v = np.asarray(v, dtype=np.float64)
    n = np.linalg.norm(v)
    if n <= 1e-5:
        return np.eye(3)
    k = v / n
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(n) * K + (1 - np.cos(n)) * K @ K


def quat_from_R(R):
    """Rotation matrix -> quaternion (x, y, z, w), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


@dataclasses.dataclass
class Scene:
    planes: list          # (origin, e1, e2) rectangles; points = origin + a*e1 + b*e2, a,b in [0,1]
    areas: np.ndarray
    map_xyz: np.ndarray   # float32 (K,3)
    tree: object


def _room_planes():
    L, H = 20.0, 5.0
    h = L / 2
    P = []
    P.append((np.array([-h, -h, 0.0]), np.array([L, 0, 0.0]), np.array([0, L, 0.0])))      # floor
    P.append((np.array([-h, -h, H]), np.array([L, 0, 0.0]), np.array([0, L, 0.0])))        # ceiling
    P.append((np.array([-h, -h, 0.0]), np.array([L, 0, 0.0]), np.array([0, 0, H])))        # wall y=-h
    P.append((np.array([-h, h, 0.0]), np.array([L, 0, 0.0]), np.array([0, 0, H])))         # wall y=+h
    P.append((np.array([-h, -h, 0.0]), np.array([0, L, 0.0]), np.array([0, 0, H])))        # wall x=-h
    P.append((np.array([h, -h, 0.0]), np.array([0, L, 0.0]), np.array([0, 0, H])))         # wall x=+h
    # three slanted panels
    P.append((np.array([3.0, -6.0, 0.0]), np.array([4.0, 2.0, 0.0]), np.array([-1.0, 0.5, 3.5])))
    P.append((np.array([-7.0, 2.0, 0.5]), np.array([1.0, 5.0, 0.5]), np.array([1.5, 0.0, 3.0])))
    P.append((np.array([4.0, 4.0, 0.0]), np.array([3.0, -1.0, 1.0]), np.array([0.5, 2.0, 3.0])))
    return P


def _sample_planes(rng, planes, areas, count, noise):
    which = rng.choice(len(planes), size=count, p=areas / areas.sum())
    a = rng.random(count)
    b = rng.random(count)
    O = np.stack([planes[w][0] for w in range(len(planes))])
    E1 = np.stack([planes[w][1] for w in range(len(planes))])
    E2 = np.stack([planes[w][2] for w in range(len(planes))])
    pts = O[which] + a[:, None] * E1[which] + b[:, None] * E2[which]
    nrm = np.cross(E1, E2)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    pts = pts + nrm[which] * rng.normal(0.0, noise, size=(count, 1))
    return pts


def make_scene(seed=SEED, map_spacing=0.15, noise=0.01):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(seed)
    planes = _room_planes()
    areas = np.array([np.linalg.norm(np.cross(e1, e2)) for (_, e1, e2) in planes])
    k = int(areas.sum() / (map_spacing * map_spacing))
    map_xyz = _sample_planes(rng, planes, areas, k, noise).astype(np.float32)
    return Scene(planes=planes, areas=areas, map_xyz=map_xyz, tree=cKDTree(map_xyz))


def knn5(scene: Scene, world_xyz: np.ndarray):
    """5-NN like KD_TREE::Nearest_Search: (n,5,3) float32 ascending + valid (n,) uint8."""
    d, idx = scene.tree.query(np.asarray(world_xyz, dtype=np.float32), k=5)
    nbr = scene.map_xyz[idx].astype(np.float32)
    sq = (d[:, 4].astype(np.float32)) ** 2
    valid = (np.isfinite(d[:, 4]) & (sq <= 5.0)).astype(np.uint8)
    return np.ascontiguousarray(nbr), valid


@dataclasses.dataclass
class LioFrame:
    n: int
    body_xyz: np.ndarray        # float32 (n,3) LiDAR-frame points
    R_LI: np.ndarray
    t_LI: np.ndarray
    R_true: np.ndarray
    p_true: np.ndarray
    R_prior: np.ndarray
    p_prior: np.ndarray
    vel: np.ndarray
    bg: np.ndarray
    ba: np.ndarray
    grav: np.ndarray
    cov18: np.ndarray           # 18x18 prior covariance (Mode-18 ordering rot,pos,vel,bg,ba,grav)
    cov23: np.ndarray           # 23x23 prior covariance (IKFoM ordering)
    scene: Scene
    laser_point_cov: float = LASER_POINT_COV

    def world_at(self, R, p):
        pi = self.body_xyz.astype(np.float64) @ self.R_LI.T + self.t_LI
        return (pi @ R.T + p).astype(np.float32)


def voxel_order(body_xyz, leaf):
    """Permutation that puts scan points into the order pcl::VoxelGrid emits its centroids in: ascending voxel index
    idx = i + j * dx + k * dx * dy over the leaf grid (PCL voxel_grid.hpp, applyFilter: the index vector is sorted by idx). That is what
    `feats_down_body` looks like when the reference's ESKF loop gets it (laserMapping.cpp:1398-1399, leaf = filter_size_surf) -- a
    synthetic scan sampled plane by plane in random order is not."""
    ijk = np.floor(np.asarray(body_xyz, dtype=np.float64) / leaf).astype(np.int64)
    ijk -= ijk.min(axis=0)
    d = ijk.max(axis=0) + 1
    return np.argsort(ijk[:, 0] + ijk[:, 1] * d[0] + ijk[:, 2] * d[0] * d[1], kind="stable")


def in_voxel_order(fr, leaf):
    """the same frame with its scan points in pcl::VoxelGrid's output order (see voxel_order)"""
    return dataclasses.replace(fr, body_xyz=np.ascontiguousarray(fr.body_xyz[voxel_order(fr.body_xyz, leaf)]))


def _spd(rng, n, base, pert):
    A = rng.normal(size=(n, n))
    return base * np.eye(n) + pert * (A @ A.T) / n


def make_lio_frame(n, seed=SEED, scene=None, t_LI=AVIA_T_LI, R_LI=AVIA_R_LI, scan_noise=0.01,
                   rot_pert_deg=0.5, pos_pert=0.02, point_seed=None):
    """point_seed varies only the scan points (shards of one frame share pose, prior and covariance)."""
    rng = np.random.default_rng(seed + 1 if point_seed is None else point_seed)
    scene = scene or make_scene(seed)
    R_true = exp_so3(np.array([0.03, -0.02, 0.4]))
    p_true = np.array([0.8, -0.5, 1.6])
    pts = np.zeros((0, 3))
    while len(pts) < n:
        cand = _sample_planes(rng, scene.planes, scene.areas, int((n - len(pts)) * 1.3) + 16, scan_noise)
        rng_ = np.linalg.norm(cand - p_true, axis=1)
        cand = cand[(rng_ > 1.0) & (rng_ < 30.0)]
        pts = np.concatenate([pts, cand])[:n]
    p_imu = (pts - p_true) @ R_true                      # R^T (p_w - p)
    body = (p_imu - t_LI) @ R_LI                         # R_LI^T (p_i - t_LI)
    body = body.astype(np.float32)
    rng = np.random.default_rng(seed + 1000)      # state / prior stream, independent of the points
    drot = np.deg2rad(rng.uniform(-rot_pert_deg, rot_pert_deg, 3))
    dpos = rng.uniform(-pos_pert, pos_pert, 3)
    R_prior = R_true @ exp_so3(drot)
    p_prior = p_true + dpos
    cov18 = _spd(rng, 18, INIT_COV, 1e-5)
    cov23 = _spd(rng, 23, INIT_COV, 1e-5)
    return LioFrame(n=n, body_xyz=np.ascontiguousarray(body), R_LI=np.array(R_LI, dtype=np.float64),
                    t_LI=np.array(t_LI, dtype=np.float64), R_true=R_true, p_true=p_true,
                    R_prior=R_prior, p_prior=p_prior,
                    vel=rng.normal(0, 0.1, 3), bg=rng.normal(0, 1e-3, 3), ba=rng.normal(0, 1e-2, 3),
                    grav=np.array([-0.27, -0.40, -9.80]), cov18=cov18, cov23=cov23, scene=scene)


def scan_from_pose(scene, R_true, p_true, n, seed, t_LI=AVIA_T_LI, R_LI=AVIA_R_LI, scan_noise=0.01, max_range=30.0):
    """A LiDAR scan (body frame, float32) of the scene taken from the IMU pose (R_true, p_true): for trajectory tests."""
    rng = np.random.default_rng(seed)
    pts = np.zeros((0, 3))
    while len(pts) < n:
        cand = _sample_planes(rng, scene.planes, scene.areas, int((n - len(pts)) * 1.3) + 16, scan_noise)
        rng_ = np.linalg.norm(cand - p_true, axis=1)
        cand = cand[(rng_ > 1.0) & (rng_ < max_range)]
        pts = np.concatenate([pts, cand])[:n]
    p_imu = (pts - p_true) @ R_true
    body = (p_imu - np.asarray(t_LI)) @ np.asarray(R_LI)
    return np.ascontiguousarray(body.astype(np.float32))


# ------------------------------------------------------------------------------------------ VIO
@dataclasses.dataclass
class VioFrame:
    m: int
    img: np.ndarray             # uint8 (H,W)
    ref_patch: np.ndarray       # float32 (m,3,64): [level][8*x + y]
    pos: np.ndarray             # float64 (m,3) world positions
    search_level: np.ndarray    # int32 (m,)
    cam: dict
    Rcl: np.ndarray
    Pcl: np.ndarray
    R_LI: np.ndarray
    t_LI: np.ndarray
    img_point_cov: float = IMG_POINT_COV
    max_iterations: int = 10
    patch_size: int = 8


def make_image(width, height, seed=SEED):
    rng = np.random.default_rng(seed + 2)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    img = np.zeros((height, width))
    for _ in range(64):
        fx, fy = rng.uniform(-0.12, 0.12, 2)
        ph = rng.uniform(0, 2 * np.pi)
        img += rng.uniform(0.3, 1.0) * np.sin(fx * xx + fy * yy + ph)
    noise = rng.normal(size=(height, width))
    k = np.array([1, 4, 6, 4, 1], dtype=np.float64) / 16
    for _ in range(2):
        noise = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, noise)
        noise = np.apply_along_axis(lambda c: np.convolve(c, k, mode="same"), 0, noise)
    img = img / np.abs(img).max() * 100.0 + 128.0 + 6.0 * noise / noise.std()
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def world_texture(p, seed=SEED):
    """A brightness field attached to the WORLD (sum of spatial sinusoids, wavelengths 0.2-0.9 m): what a Lambertian scene looks
    like from any pose. p (...,3) -> float (...)"""
    rng = np.random.default_rng(seed + 77)
    out = np.zeros(p.shape[:-1])
    for _ in range(24):
        k = rng.normal(size=3)
        k *= rng.uniform(7.0, 30.0) / np.linalg.norm(k)
        out += rng.uniform(0.4, 1.0) * np.sin(p @ k + rng.uniform(0, 2 * np.pi))
    return out


def render_image(scene, cam, Rcw, Pcw, seed=SEED, noise=1.0):
    """The scene as the pinhole camera (Rcw, Pcw: camera <- world) sees it: every pixel's ray is intersected with the scene's
    rectangles and takes the world texture at the nearest hit. Consistent across poses, so photometric alignment is meaningful."""
    W, H = cam["width"], cam["height"]
    vv, uu = np.mgrid[0:H, 0:W].astype(np.float64)
    d_c = np.stack([(uu - cam["cx"]) / cam["fx"], (vv - cam["cy"]) / cam["fy"], np.ones_like(uu)], -1).reshape(-1, 3)
    c = -Rcw.T @ Pcw
    d_w = d_c @ Rcw                                   # Rcw^T d per row
    best_t = np.full(len(d_w), np.inf)
    for (O, e1, e2) in scene.planes:
        n = np.cross(e1, e2)
        den = d_w @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((O - c) @ n) / den
        hit = c + t[:, None] * d_w - O
        G = np.array([[e1 @ e1, e1 @ e2], [e1 @ e2, e2 @ e2]])
        ab = np.linalg.solve(G, np.stack([hit @ e1, hit @ e2]))
        ok = (t > 0.05) & (ab[0] >= 0) & (ab[0] <= 1) & (ab[1] >= 0) & (ab[1] <= 1) & np.isfinite(t)
        best_t = np.where(ok & (t < best_t), t, best_t)
    seen = np.isfinite(best_t)
    pts = c + np.where(seen, best_t, 1.0)[:, None] * d_w
    tex = world_texture(pts, seed)
    img = 128.0 + 11.0 * tex
    img[~seen] = 128.0
    rng = np.random.default_rng(seed + 1234 + int(abs(Pcw[0]) * 1e6) % 100000)
    img = img + rng.normal(0, noise, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8).reshape(H, W)


def world2cam(cam, xyz_c):
    """vk::PinholeCamera::world2cam as restated in oracle/orc_vio.c (from memory of rpg_vikit)."""
    u = xyz_c[..., 0] / xyz_c[..., 2]
    v = xyz_c[..., 1] / xyz_c[..., 2]
    d = cam["d"]
    if not abs(d[0]) > 1e-7:
        return np.stack([cam["fx"] * u + cam["cx"], cam["fy"] * v + cam["cy"]], -1)
    r2 = u * u + v * v
    r4 = r2 * r2
    r6 = r4 * r2
    a1, a2, a3 = 2 * u * v, r2 + 2 * u * u, r2 + 2 * v * v
    cd = 1 + d[0] * r2 + d[1] * r4 + d[4] * r6
    xd = u * cd + d[2] * a1 + d[3] * a2
    yd = v * cd + d[3] * a1 + d[2] * a3
    return np.stack([xd * cam["fx"] + cam["cx"], yd * cam["fy"] + cam["cy"]], -1)


def cam_pose(Rcl, Pcl, R_LI, t_LI, R_wi, p_wi):
    """Rcw, Pcw of lidar_selection.cpp:35-52,780-783."""
    Rli = R_LI.T
    Pli = -R_LI.T @ t_LI
    Rci = Rcl @ Rli
    Pci = Rcl @ Pli + Pcl
    Rcw = Rci @ R_wi.T
    Pcw = -Rci @ R_wi.T @ p_wi + Pci
    return Rcw, Pcw


def sample_patches(img, pc, scale, patch=8):
    """Bilinear 8x8 samples around pixel pc (m,2) at pyramid scale, lidar_selection.cpp:807-837."""
    H, W = img.shape
    m = pc.shape[0]
    pcf = pc.astype(np.float32)
    ui = (np.floor((pc[:, 0] / scale).astype(np.float32)) * scale).astype(np.int64)
    vi = (np.floor((pc[:, 1] / scale).astype(np.float32)) * scale).astype(np.int64)
    su = ((pcf[:, 0] - ui.astype(np.float32)) / np.float32(scale)).astype(np.float32)
    sv = ((pcf[:, 1] - vi.astype(np.float32)) / np.float32(scale)).astype(np.float32)
    wtl = ((1.0 - su.astype(np.float64)) * (1.0 - sv.astype(np.float64))).astype(np.float32)
    wtr = (su.astype(np.float64) * (1.0 - sv.astype(np.float64))).astype(np.float32)
    wbl = ((1.0 - su.astype(np.float64)) * sv.astype(np.float64)).astype(np.float32)
    wbr = su * sv
    half = patch // 2
    xs = np.arange(patch)
    rows = vi[:, None] + (xs[None, :] - half) * scale          # (m,8)
    cols = ui[:, None] + (xs[None, :] - half) * scale
    r = rows[:, :, None]
    c = cols[:, None, :]
    f = img.astype(np.float32)
    out = (wtl[:, None, None] * f[r, c] + wtr[:, None, None] * f[r, c + scale]
           + wbl[:, None, None] * f[r + scale, c] + wbr[:, None, None] * f[r + scale, c + scale])
    return out.reshape(m, patch * patch).astype(np.float32)


def make_vio_frame(m, lio: LioFrame, seed=SEED, cam=None, Rcl=AVIA_RCL, Pcl=AVIA_PCL, distortion=False,
                   ref_noise=2.0, img_point_cov=IMG_POINT_COV, max_iterations=10, patch_seed=None):
    """patch_seed varies only the patch set (the image and camera are shared by all shards)."""
    rng = np.random.default_rng(seed + 3 if patch_seed is None else patch_seed)
    cam = dict(cam or PINHOLE)
    if not distortion:
        cam["d"] = (0.0, 0.0, 0.0, 0.0, 0.0)
    img = make_image(cam["width"], cam["height"], seed)
    Rcw, Pcw = cam_pose(Rcl, Pcl, lio.R_LI, lio.t_LI, lio.R_true, lio.p_true)
    border = (8 // 2 + 1) * 8 + 8   # lidar_selection.cpp:445 margin (+8 so the perturbed prior stays inside)
    pos = np.zeros((0, 3))
    while len(pos) < m:
        k = (m - len(pos)) * 2 + 16
        px = np.stack([rng.uniform(border, cam["width"] - border, k), rng.uniform(border, cam["height"] - border, k)], -1)
        depth = rng.uniform(2.0, 20.0, k)
        xyc = np.stack([(px[:, 0] - cam["cx"]) / cam["fx"], (px[:, 1] - cam["cy"]) / cam["fy"], np.ones(k)], -1) * depth[:, None]
        pc = world2cam(cam, xyc)
        ok = (pc[:, 0] > border) & (pc[:, 0] < cam["width"] - border) & (pc[:, 1] > border) & (pc[:, 1] < cam["height"] - border)
        pw = (xyc[ok] - Pcw) @ Rcw                        # Rcw^T (pf - Pcw)
        pos = np.concatenate([pos, pw])[:m]
    pf = pos @ Rcw.T + Pcw
    pc = world2cam(cam, pf)
    ref = np.zeros((m, 3, 64), dtype=np.float32)
    for level in range(3):
        ref[:, level, :] = sample_patches(img, pc, 1 << level) + rng.normal(0, ref_noise, (m, 64)).astype(np.float32)
    return VioFrame(m=m, img=img, ref_patch=np.ascontiguousarray(ref), pos=np.ascontiguousarray(pos),
                    search_level=np.zeros(m, dtype=np.int32), cam=cam, Rcl=np.array(Rcl), Pcl=np.array(Pcl),
                    R_LI=lio.R_LI, t_LI=lio.t_LI, img_point_cov=img_point_cov, max_iterations=max_iterations)


# ------------------------------------------------------------------------------------------------------------
# IMU frame for ImuProcess::UndistortPcl: one 100 ms LiDAR frame with a 200 Hz IMU
@dataclasses.dataclass
class ImuFrame:
    imu: np.ndarray              # (k,7) float64 [t, gyr xyz, acc xyz] = meas.imu
    last_imu: np.ndarray         # (7,) last_imu_
    last_lidar_end_time: float
    pcl_beg_time: float
    pcl_end_time: float
    pts_xyzt: np.ndarray         # float32 (n,4): x, y, z, curvature (offset from pcl_beg_time in ms)
    R_LI: np.ndarray
    t_LI: np.ndarray
    cov_gyr: np.ndarray
    cov_acc: np.ndarray
    cov_bias_gyr: np.ndarray
    cov_bias_acc: np.ndarray
    mean_acc: np.ndarray
    acc_s_last: np.ndarray
    angvel_last: np.ndarray
    lio: LioFrame                # state (prior) and scene


def make_imu_frame(n, n_imu=20, seed=SEED, time_sorted=True, imu_before_frame=True, first_point_late=False, lio=None, quiet=False):
    rng = np.random.default_rng(seed + 77)
    lio = lio if lio is not None else make_lio_frame(max(n, 16), seed=seed)
    t0 = 1000.0                                    # pcl_beg_time
    dur = 0.1
    dt_imu = dur / n_imu
    # last_imu_ a little before the frame; with imu_before_frame the first samples straddle last_lidar_end_time_
    t_last = t0 - 0.6 * dt_imu if imu_before_frame else t0 + 0.1 * dt_imu
    times = t_last + dt_imu * np.arange(1, n_imu + 1)
    def sample(t):
        w = np.array([0.3 * np.sin(7 * t), -0.2 * np.cos(5 * t), 0.25 * np.sin(3 * t + 1)]) + rng.normal(0, 0.01, 3)
        a = np.array([0.02 * np.sin(11 * t), 0.03 * np.cos(13 * t), 1.0 + 0.01 * np.sin(17 * t)]) + rng.normal(0, 0.002, 3)   # Livox: unit g
        return np.concatenate([[t], w, a])
    imu = np.stack([sample(t) for t in times])
    last_imu = sample(t_last)
    mean_acc = np.array([0.01, -0.02, 0.999])
    if quiet:
        # sensor almost at rest: specific force cancels gravity at the prior attitude, tiny rates. The scan then stays on the
        # scene's planes after undistortion (used by the whole-pipeline benchmark, where the LIO update must converge)
        rest = (np.asarray(lio.R_prior, float).T @ (-np.asarray(lio.grav, float))) / 9.81 * np.linalg.norm(mean_acc) + np.asarray(lio.ba, float) * np.linalg.norm(mean_acc) / 9.81
        for row in (last_imu, *imu):
            row[1:4] = row[1:4] * 0.01 + np.asarray(lio.bg, float)
            row[4:7] = rest + rng.normal(0, 1e-5, 3)
    pts = np.empty((n, 4), np.float32)
    src = lio.body_xyz[rng.integers(0, lio.n, n)] if n else np.zeros((0, 3), np.float32)
    pts[:, :3] = src
    tt = rng.uniform(0.0, dur * 1000.0, n).astype(np.float32)
    if time_sorted:
        tt = np.sort(tt)
    if n > 3 and not first_point_late:
        tt[:2] = 0.0                                # the first returns carry offset 0 (never later than IMUpose[0])
    if n > 3 and first_point_late:
        tt = np.maximum(tt, np.float32(3.5 * dt_imu * 1000.0))   # frame tail only: the first point lies in a later IMU interval
    pts[:, 3] = tt
    return ImuFrame(imu=imu, last_imu=last_imu, last_lidar_end_time=t0 - 0.2 * dt_imu, pcl_beg_time=t0,
                    pcl_end_time=t0 + float(tt.max()) / 1000.0 if n else t0 + dur, pts_xyzt=pts, R_LI=lio.R_LI, t_LI=lio.t_LI,
                    cov_gyr=np.array([0.1, 0.1, 0.1]) * 1e-2, cov_acc=np.array([0.1, 0.12, 0.09]) * 96.2, cov_bias_gyr=np.full(3, 1e-4),
                    cov_bias_acc=np.full(3, 1e-4), mean_acc=mean_acc,
                    acc_s_last=np.array([0.05, -0.03, 0.02]) * (0.0 if quiet else 1.0), angvel_last=np.array([0.01, 0.02, -0.015]) * (0.0 if quiet else 1.0), lio=lio)


# ------------------------------------------------------------------------------------------------------------
# Patch-selection frame for LidarSelector::addFromSparseMap: current image + pose, reference keyframes,
# one candidate (map point + chosen reference observation) per "grid cell", and a scan for the depth image
@dataclasses.dataclass
class SelectFrame:
    vio: VioFrame
    Rcw: np.ndarray
    Pcw: np.ndarray
    keyframes: list               # uint8 (H,W) images; keyframe 0 is the current image seen from the current pose
    kf_R: np.ndarray              # (K,3,3) T_f_w rotation of each keyframe
    kf_t: np.ndarray              # (K,3)
    cand_pos: np.ndarray          # (m,3)
    cand_kf: np.ndarray           # (m,) int32
    cand_px: np.ndarray           # (m,2) ref_ftr->px
    cand_f: np.ndarray            # (m,3) ref_ftr->f
    scan_world: np.ndarray        # float32 (n,3)
    lio: LioFrame = None
    outlier_threshold: float = 300.0


def cam2world(cam, px):
    """Bearing of a pixel (vk::PinholeCamera::cam2world): pinhole inverse, or cv::undistortPoints' five sweeps with distortion."""
    d = cam["d"]
    if not abs(d[0]) > 1e-7:
        b = np.array([(px[0] - cam["cx"]) / cam["fx"], (px[1] - cam["cy"]) / cam["fy"], 1.0])
        return b / np.linalg.norm(b)
    x = (float(np.float32(px[0])) - cam["cx"]) / cam["fx"]
    y = (float(np.float32(px[1])) - cam["cy"]) / cam["fy"]
    x0, y0 = x, y
    for _ in range(5):
        r2 = x * x + y * y
        ic = 1.0 / (1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2)
        dx = 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x)
        dy = d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y
        x, y = (x0 - dx) * ic, (y0 - dy) * ic
    b = np.array([float(np.float32(x)), float(np.float32(y)), 1.0])
    return b / np.linalg.norm(b)


def make_select_frame(m, seed=SEED, n_keyframes=3, discont_frac=0.1, lio=None, distortion=False):
    rng = np.random.default_rng(seed + 501)
    lio = lio if lio is not None else make_lio_frame(2000, seed=seed)
    vf = make_vio_frame(m, lio, seed=seed, distortion=distortion)
    cam = vf.cam
    Rcw, Pcw = cam_pose(vf.Rcl, vf.Pcl, lio.R_LI, lio.t_LI, lio.R_true, lio.p_true)
    kfs, kR, kt = [vf.img], [Rcw.copy()], [Pcw.copy()]
    for k in range(1, n_keyframes):
        dR = exp_so3(rng.uniform(-0.02, 0.02, 3))
        Rk = dR @ Rcw
        tk = dR @ Pcw + rng.uniform(-0.15, 0.15, 3)
        img = np.roll(vf.img, (3 * k, -2 * k), axis=(0, 1)) if k % 2 else make_image(cam["width"], cam["height"], seed + 17 * k)
        kfs.append(np.ascontiguousarray(img)); kR.append(Rk); kt.append(tk)
    kR = np.stack(kR); kt = np.stack(kt)
    pos = vf.pos
    ckf = (np.arange(m) % n_keyframes).astype(np.int32)
    cpx = np.zeros((m, 2)); cf = np.zeros((m, 3))
    for i in range(m):
        for attempt in (ckf[i], 0):
            pf = kR[attempt] @ pos[i] + kt[attempt]
            px = world2cam(cam, pf[None])[0]
            if pf[2] > 0.5 and 45 < px[0] < cam["width"] - 45 and 45 < px[1] < cam["height"] - 45:
                ckf[i] = attempt
                break
        cpx[i] = px
        cf[i] = cam2world(cam, px)
    # scan: around every candidate a few returns at (almost) its depth; for some a return 3 m behind inside the 9x9 window
    pf = pos @ Rcw.T + Pcw
    scan_c = []
    for i in range(m):
        for _ in range(3):
            d = pf[i] * (1.0 + rng.normal(0, 0.002))
            d[:2] += rng.normal(0, 0.004, 2) * pf[i, 2]
            scan_c.append(d)
        if rng.uniform() < discont_frac:
            scan_c.append(pf[i] * (1.0 + 3.0 / pf[i, 2]) + np.array([0.002 * pf[i, 2], 0.0, 0.0]))
    scan_c = np.array(scan_c)
    scan_w = (scan_c - Pcw) @ Rcw
    extra = lio.world_at(lio.R_true, lio.p_true)[:1000]
    scan_w = np.concatenate([scan_w, extra]).astype(np.float32)
    scan_w = scan_w[rng.permutation(len(scan_w))]
    return SelectFrame(vio=vf, Rcw=Rcw, Pcw=Pcw, keyframes=kfs, kf_R=kR, kf_t=kt, cand_pos=np.ascontiguousarray(pos), cand_kf=ckf,
                       cand_px=cpx, cand_f=cf, scan_world=np.ascontiguousarray(scan_w), lio=lio)
