"""Minimal stand-ins for the pytorch3d types that cross the reference's model API
(pytorch3d 0.7.x is a third-party dependency of the reference, README.md:73, not installed here):
`PerspectiveCameras` (call sites dataset/shapenet_r2n2.py:86-93, dataset/pix3d.py:152-159,
model/projection_model.py:134-140) and `Pointclouds` (main_blending.py:346).  Only the fields
the sampling path reads are kept.  Row-vector convention: X_view = X_world @ R + T;
ndc.xy = focal * X_view.xy / X_view.z + principal_point (PerspectiveCameras, in_ndc=True)."""
import math

import torch


class PerspectiveCameras:
    """in_ndc=False (screen-space intrinsics + image_size (H, W), as dataset/pix3d.py:152-159 builds them) is converted
    to NDC at construction with pytorch3d's rule (docs/notes/cameras.md, restated; unpinned: pytorch3d is absent):
        s = min(W, H);  focal_ndc = focal_screen * 2 / s;  principal_ndc = -(principal_screen - (W, H) / 2) * 2 / s"""

    def __init__(self, focal_length=1.0, principal_point=((0.0, 0.0),), R=None, T=None, device="cpu", in_ndc=True,
                 image_size=None):
        def as2(v, n):
            v = torch.as_tensor(v, dtype=torch.float32)
            if v.dim() == 0:
                v = v.view(1, 1).expand(n, 2)
            elif v.dim() == 1:
                v = v.view(-1, 1).expand(-1, 2) if v.shape[0] == n and n != 2 else v.view(1, -1).expand(n, -1)
            return v.contiguous()
        R = torch.eye(3)[None] if R is None else torch.as_tensor(R, dtype=torch.float32)
        T = torch.zeros(1, 3) if T is None else torch.as_tensor(T, dtype=torch.float32)
        n = max(R.shape[0], T.shape[0])
        self.R = R.expand(n, 3, 3).contiguous().to(device)
        self.T = T.expand(n, 3).contiguous().to(device)
        focal, pp = as2(focal_length, n), as2(principal_point, n)
        if not in_ndc:
            if image_size is None:
                raise ValueError("screen-space cameras (in_ndc=False) need image_size=(H, W)")
            hw = torch.as_tensor(image_size, dtype=torch.float32).reshape(-1, 2).expand(n, 2)
            wh = hw.flip(-1)
            scale = wh.min(dim=1, keepdim=True).values
            focal = focal * 2.0 / scale
            pp = -(pp - wh / 2.0) * 2.0 / scale
        self.focal_length = focal.contiguous().to(device)
        self.principal_point = pp.contiguous().to(device)

    def __len__(self):
        return self.R.shape[0]

    @property
    def device(self):
        return self.R.device

    def clone(self):
        return PerspectiveCameras(self.focal_length.clone(), self.principal_point.clone(), self.R.clone(), self.T.clone(),
                                  device=self.R.device)

    def to(self, device):
        return PerspectiveCameras(self.focal_length, self.principal_point, self.R, self.T, device=device)

    def packed(self):
        """(n, 16) float32: R row-major, T, focal, principal point -- the layout bdm_rasterize_points takes."""
        return torch.cat([self.R.reshape(-1, 9), self.T, self.focal_length, self.principal_point], dim=1).contiguous()


def join_cameras(cameras):
    """The datasets collate to a python LIST of single cameras (dataset/shapenet_r2n2.py:601-612)."""
    if isinstance(cameras, (list, tuple)):
        return PerspectiveCameras(torch.cat([c.focal_length for c in cameras]), torch.cat([c.principal_point for c in cameras]),
                                  torch.cat([c.R for c in cameras]), torch.cat([c.T for c in cameras]), device=cameras[0].device)
    return cameras


class Pointclouds:
    def __init__(self, points, features=None):
        self._points, self._features = points, features

    def points_padded(self):
        return self._points

    def features_padded(self):
        return self._features

    def points_list(self):
        return list(self._points)


def r2n2_camera(azimuth, elevation, distance, focal=2.1875):
    """R2N2-style camera looking at the origin (dataset/utils.py:40-114 + shapenet_r2n2.py:46-53,86-93 with the
    identity normalisation mean=0, std=1): used for synthetic benchmark inputs."""
    az, el = -math.pi * azimuth / 180.0, -math.pi * elevation / 180.0
    sa, ca, se, ce = math.sin(az), math.cos(az), math.sin(el), math.cos(el)
    R_world2obj = torch.tensor([[ca * ce, sa * ce, -se], [-sa, ca, 0], [ca * se, sa * se, ce]], dtype=torch.float32)
    R_obj2cam = torch.tensor([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [1.0, 0.0, 0.0]])
    R_world2cam = R_obj2cam.mm(R_world2obj)
    T_world2cam = -(R_obj2cam.mm(torch.tensor([[distance, 0, 0]], dtype=torch.float32).t()))
    RT = torch.cat([torch.cat([R_world2cam, T_world2cam], dim=1), torch.tensor([[0.0, 0, 0, 1]])])
    rot = torch.tensor([[1.0, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]])
    RT = RT.mm(rot)
    s2p = torch.tensor([[-1.0, 0, 0, 0], [0, 1.0, 0, 0], [0, 0, -1.0, 0], [0, 0, 0, 1.0]])
    RT = torch.transpose(RT, 0, 1).mm(s2p)
    R, T = RT[:3, :3].clone(), RT[3, :3].clone()
    return PerspectiveCameras(focal_length=torch.tensor([[focal, focal]]), principal_point=torch.tensor([[0.0, 0.0]]),
                              R=R[None], T=T[None])


def pix3d_camera(rot_mat, trans_mat, pts_mean, pts_std, img_size_wh, bbox, focal_length_mm, out_size=224):
    """Camera of one Pix3D sample for the square crop around its bounding box (dataset/pix3d.py:104-159): the object is
    normalised to zero mean / unit std (m, s), so R_norm = R * s, t_norm = t + m R^T; OpenCV -> pytorch3d axes; the crop
    [cx - h, cx + h]^2 (h = half the longer bbox side) is resized to out_size, which composes an affine map with the
    intrinsics K = [[f, 0, w/2], [0, f, h/2]] where f = focal_mm * w / 32 (32 mm sensor width)."""
    import numpy as np
    R = np.asarray(rot_mat, dtype=np.float64).reshape(3, 3)
    t = np.asarray(trans_mat, dtype=np.float64).reshape(3)
    m = np.asarray(pts_mean, dtype=np.float64).reshape(3)
    R_norm, t_norm = R * float(pts_std), t + m @ R.T
    convert = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], dtype=np.float64)
    R_v1 = (R_norm @ convert).T
    w, h = img_size_wh
    x0, y0, x1, y1 = bbox
    cx, cy = (x0 + x1) / 2, (y0 + y1) / 2
    half_w = max(y1 - y0, x1 - x0) / 2
    x0, y0 = cx - half_w, cy - half_w
    f = focal_length_mm * w / 32
    s = out_size / (2 * half_w)
    fx, fy = s * f, s * f
    tx, ty = s * (w / 2) + s * (-x0), s * (h / 2) + s * (-y0)
    return PerspectiveCameras(focal_length=torch.tensor([[fx, fy]], dtype=torch.float32),
                              principal_point=torch.tensor([[tx, ty]], dtype=torch.float32),
                              R=torch.as_tensor(R_v1, dtype=torch.float32)[None], T=torch.as_tensor(t_norm, dtype=torch.float32)[None],
                              in_ndc=False, image_size=(out_size, out_size))
