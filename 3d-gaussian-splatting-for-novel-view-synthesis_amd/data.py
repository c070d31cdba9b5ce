"""On-disk data layout of the reference and Gaussian initialisation from a point cloud ("next" row 4).

Layout (gaussian_splatting/data_loader.py:153-284; written by datasets/prepare_mipnerf360.py:410-431):
    <data_dir>/images/*.{jpg,png,JPG,PNG}   cam_meta.npy (a pickled dict: fx, fy[, cx, cy][, c2w])   poses.npy [K,4,4]
    pointcloud.ply (ASCII, x y z first)
Functions mirror the reference's names and return values: load_image (:15), load_camera_parameters (:27),
load_point_cloud (:50), GaussianDataset (:153), initialize_gaussians_from_pointcloud (:287).  `write_dataset` produces
the same layout (so a scene can be prepared without the reference's download / COLMAP tooling).  Host-side I/O only.
"""
import json
import pickle
from pathlib import Path

import numpy as np
import torch


def load_image(image_path):
    """[H, W, 3] float32 in [0, 1] (data_loader.py:15-25)."""
    from PIL import Image
    return torch.from_numpy(np.array(Image.open(image_path).convert('RGB'), dtype=np.float32) / 255.0)


class _MetaUnpickler(pickle.Unpickler):
    """Unpickler for cam_meta.npy that can only rebuild numpy arrays / scalars / dtypes and plain Python containers:
    a crafted file cannot name any other callable, so loading a dataset directory executes nothing from it."""
    _ALLOWED = {("numpy", "ndarray"), ("numpy", "dtype"),
                ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"cam_meta.npy may only hold numbers, strings, lists, dicts and numpy arrays; it names {module}.{name}")


def load_camera_parameters(cam_meta_path):
    """The camera dict of the reference's layout (data_loader.py:27-47: fx, fy[, cx, cy][, c2w] ...).  The reference stores it
    as a PICKLED dict inside cam_meta.npy and reads it with allow_pickle=True; here the pickle is read by a restricted
    unpickler (numpy arrays and plain containers only), and a cam_meta.json beside it (write_dataset emits one) is
    preferred when present."""
    path = Path(cam_meta_path)
    js = path.with_suffix('.json')
    if js.exists():
        with open(js) as f:
            return json.load(f)
    with open(path, 'rb') as f:
        version = np.lib.format.read_magic(f)
        _, _, dtype = np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f)
        if not dtype.hasobject:
            raise ValueError(f"{path}: expected the reference's pickled dict, found a plain array")
        obj = _MetaUnpickler(f).load()
    return obj.item() if isinstance(obj, np.ndarray) and obj.shape == () else obj


def _filter_points(pts):
    """The sanitising of data_loader.py:106-148: drop NaN/Inf, keep |coord| < 1000 if any such point exists, otherwise
    fall back to the 99th-percentile / 100 x median distance filters; an empty result raises ValueError."""
    pts = pts[torch.isfinite(pts).all(dim=1)]
    if len(pts) > 0:
        near = (pts.abs() < 1000).all(dim=1)
        if near.sum() > 0:
            pts = pts[near]
        else:
            dist = (pts - pts.mean(dim=0)).norm(dim=1)
            if len(dist) > 100:
                keep = dist < torch.quantile(dist, 0.99)
                if keep.sum() > 0:
                    pts = pts[keep]
                else:
                    med = dist.median()
                    if torch.isfinite(med) and med > 0 and (dist < med * 100).sum() > 0:
                        pts = pts[dist < med * 100]
    if len(pts) == 0:
        raise ValueError("Point cloud is empty after filtering invalid values!")
    return pts


def _load_ply(ply_path):
    """ASCII PLY, first three numbers of each vertex line (data_loader.py:78-104)."""
    pts, n, header = [], 0, True
    with open(ply_path, 'r') as f:
        for line in f:
            if header:
                if line.startswith('element vertex'):
                    n = int(line.split()[-1])
                elif line.startswith('end_header'):
                    header = False
            elif len(pts) < n:
                pts.append([float(v) for v in line.split()[:3]])
    return _filter_points(torch.tensor(pts, dtype=torch.float32).reshape(-1, 3))


def load_point_cloud(pcd_path):
    """.ply (ASCII) / .npy / .pt -> [N, 3+] float tensor (data_loader.py:50-76)."""
    ext = Path(pcd_path).suffix.lower()
    if ext == '.pt':
        return torch.load(pcd_path, weights_only=True)
    if ext == '.npy':
        return torch.from_numpy(np.load(pcd_path)).float()
    if ext == '.ply':
        return _load_ply(pcd_path)
    raise ValueError(f"Unsupported point cloud format: {ext}")


class GaussianDataset:
    """Images + intrinsics + camera-to-world poses in the reference layout (data_loader.py:153-284); same constructor
    arguments (scale_factor defaults to 0.5 there too) and the same sample dict."""

    def __init__(self, data_dir, image_dir='images', cam_meta_path=None, scale_factor=0.5):
        self.data_dir = Path(data_dir)
        self.image_dir = self.data_dir / image_dir
        self.scale_factor = scale_factor
        self.cam_params = load_camera_parameters(self.data_dir / 'cam_meta.npy' if cam_meta_path is None else Path(cam_meta_path))
        self.image_files = sorted(f for ext in ('*.jpg', '*.png', '*.JPG', '*.PNG') for f in self.image_dir.glob(ext))
        if not self.image_files:
            raise ValueError(f"No images found in {self.image_dir}")
        self.c2w_matrices = self._load_camera_poses()

    def _load_camera_poses(self):
        pose_file = self.data_dir / 'poses.npy'
        if pose_file.exists():
            return [torch.from_numpy(p).float() for p in np.load(pose_file)]
        c2w = self.cam_params.get('c2w') if isinstance(self.cam_params, dict) else None
        if isinstance(c2w, (list, np.ndarray)):
            return [torch.from_numpy(np.asarray(p)).float() for p in c2w]
        return None

    def __len__(self):
        return len(self.image_files)

    def __getitem__(self, idx):
        image = load_image(self.image_files[idx])
        if self.scale_factor != 1.0:
            h, w = image.shape[:2]
            image = torch.nn.functional.interpolate(image.permute(2, 0, 1).unsqueeze(0),
                                                    size=(int(h * self.scale_factor), int(w * self.scale_factor)),
                                                    mode='bilinear', align_corners=False).squeeze(0).permute(1, 2, 0)
        H, W = image.shape[:2]
        c2w = self.c2w_matrices[idx] if self.c2w_matrices is not None else torch.eye(4, dtype=torch.float32)
        cp, s = self.cam_params, self.scale_factor
        has_c = 'cx' in cp and 'cy' in cp
        return {'image': image, 'c2w': c2w, 'fx': cp['fx'] * s, 'fy': cp['fy'] * s, 'cx': cp['cx'] * s if has_c else W / 2.0,
                'cy': cp['cy'] * s if has_c else H / 2.0, 'H': H, 'W': W, 'idx': idx}


def initialize_gaussians_from_pointcloud(points, num_sh_bands=3):
    """One Gaussian per point: log-scale -2 +- 0.1, identity quaternion (0,0,0,1), opacity_raw 0.1, colours from columns
    3-5 (divided by 255 if > 1) or uniform random, zero higher-order SH (data_loader.py:287-367).  Only num_sh_bands >= 3
    gives the [N, 45] f_rest that evaluate_sh / the HIP path accept (as in the reference)."""
    if not isinstance(points, torch.Tensor):
        points = torch.from_numpy(np.asarray(points)).float()
    n, dev = points.shape[0], points.device
    pos = points[:, :3]
    scale_raw = torch.randn(n, 3, device=dev) * 0.1 - 2.0
    q_raw = torch.zeros(n, 4, device=dev)
    q_raw[:, 3] = 1.0
    opacity_raw = torch.ones(n, device=dev) * 0.1
    if points.shape[1] >= 6:
        colors = points[:, 3:6]
        if colors.max() > 1.0:
            colors = colors / 255.0
    else:
        colors = torch.rand(n, 3, device=dev)
    width = 45 if num_sh_bands >= 3 else (9 if num_sh_bands == 1 else 0)
    return {'pos': pos, 'opacity_raw': opacity_raw, 'f_dc': colors, 'f_rest': torch.zeros(n, width, device=dev),
            'scale_raw': scale_raw, 'q_raw': q_raw}


def write_dataset(data_dir, images, fx, fy, poses, points=None, cx=None, cy=None):
    """Write a scene in the reference layout: images/000000.png ..., cam_meta.npy, poses.npy, pointcloud.ply (ASCII)."""
    from PIL import Image
    d = Path(data_dir)
    (d / 'images').mkdir(parents=True, exist_ok=True)
    for i, im in enumerate(images):
        a = im.detach().cpu().numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
        Image.fromarray((np.clip(a, 0, 1) * 255).round().astype(np.uint8)).save(d / 'images' / f'{i:06d}.png')
    h, w = (images[0].shape[0], images[0].shape[1])
    meta = {'fx': float(fx), 'fy': float(fy), 'height': int(h), 'width': int(w)}
    if cx is not None and cy is not None:
        meta.update(cx=float(cx), cy=float(cy))
    np.save(d / 'cam_meta.npy', meta, allow_pickle=True)        # the reference's format (its loader needs the pickle)
    with open(d / 'cam_meta.json', 'w') as f:                     # same content, read in preference by load_camera_parameters
        json.dump(meta, f)
    np.save(d / 'poses.npy', np.asarray(poses, dtype=np.float32))
    if points is not None:
        p = np.asarray(points, dtype=np.float64)
        with open(d / 'pointcloud.ply', 'w') as f:
            f.write(f"ply\nformat ascii 1.0\nelement vertex {len(p)}\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
            for x, y, z in p[:, :3]:
                f.write(f"{x:.9g} {y:.9g} {z:.9g}\n")
