"""Seeded synthetic inputs for the descriptor path.

Clouds are (N, 4) float32 AoS [x, y, z, intensity] -- the layout every loader of the reference
produces (src/data/kitti_loader.py:113, nclt_loader.py:230-253, helipr_loader.py:133-150).
"""
import numpy as np

__all__ = ["make_cloud", "make_clouds_packed", "make_clouds_device", "make_pose_chain", "randomize_bn_stats"]


def _sph_to_xyz(az, el, r):
    ce = np.cos(el)
    return r * ce * np.cos(az), r * ce * np.sin(az), r * np.sin(el)


def make_cloud(seed, n_points=120000, kind="uniform"):
    """One seeded cloud.  kind:
      uniform   az~U(-pi,pi), el~U(-26,3) deg, r~U(0.5,90) m  (exercises row clamp + range filter)
      safe      like uniform but every point >= 1e-3 rad from every row/column edge
      ring      HDL-64-like: 64 rings x (n/64) azimuth steps, smooth scene range
      sparse    VLP-16-like: 16 rings in +-15 deg (rows of the default FOV stay empty)
      adversarial  uniform with 0.1 % NaN/Inf coordinates
      wide      el~U(-60,60) deg (out-of-FOV clamp to rows 0/15)
    """
    rng = np.random.default_rng(seed)
    n = int(n_points)
    if kind in ("uniform", "adversarial", "safe", "wide"):
        az = rng.uniform(-np.pi, np.pi, n)
        if kind == "wide":
            el = np.deg2rad(rng.uniform(-60.0, 60.0, n))
        else:
            el = np.deg2rad(rng.uniform(-26.0, 3.0, n))
        r = rng.uniform(0.5, 90.0, n)
        if kind == "safe":
            # pull angles to bin centres +- 0.4 bin so no point sits near an edge
            col = np.floor((az + np.pi) / (2 * np.pi) * 360.0)
            az = -np.pi + (col + 0.5 + rng.uniform(-0.4, 0.4, n)) * (2 * np.pi / 360.0)
            lo, hi = np.deg2rad(-24.8), np.deg2rad(2.0)
            row = np.clip(np.floor((el - lo) / (hi - lo) * 16.0), 0, 15)
            el = lo + (row + 0.5 + rng.uniform(-0.4, 0.4, n)) * ((hi - lo) / 16.0)
    elif kind == "ring":
        rings = 64
        steps = max(n // rings, 1)
        n = rings * steps
        ring_el = np.deg2rad(np.linspace(-24.6, 1.9, rings))
        az = np.tile(np.linspace(-np.pi, np.pi, steps, endpoint=False), rings)
        az = az + rng.normal(0.0, 2e-4, n)
        el = np.repeat(ring_el, steps) + rng.normal(0.0, 2e-4, n)
        ph = rng.uniform(0, 2 * np.pi, 4)
        r = (25.0 + 12.0 * np.sin(2 * az + ph[0]) + 6.0 * np.sin(5 * az + ph[1])
             + 3.0 * np.sin(11 * az + ph[2]) + 20.0 * (np.repeat(np.arange(rings), steps) / rings))
        drop = rng.uniform(0, 1, n) < 0.05          # 5 % dropouts -> holes to interpolate
        r = np.where(drop, 0.2, r)                  # < min_range, filtered out
    elif kind == "sparse":
        rings = 16
        steps = max(n // rings, 1)
        n = rings * steps
        ring_el = np.deg2rad(np.linspace(-15.0, 15.0, rings))
        az = np.tile(np.linspace(-np.pi, np.pi, steps, endpoint=False), rings)
        el = np.repeat(ring_el, steps)
        r = 8.0 + 30.0 * rng.uniform(0, 1, n) ** 2
    else:
        raise ValueError(kind)
    x, y, z = _sph_to_xyz(az, el, r)
    pts = np.stack([x, y, z, rng.uniform(0, 1, n)], axis=1).astype(np.float32)
    if kind == "adversarial":
        k = max(n // 1000, 1)
        idx = rng.choice(n, size=3 * k, replace=False)
        pts[idx[:k], 0] = np.nan
        pts[idx[k:2 * k], 1] = np.inf
        pts[idx[2 * k:], 2] = -np.inf
    return pts


def make_clouds_packed(seeds, n_points=120000, kind="uniform"):
    """Pack several clouds: returns (points (sum_n,4) f32, offsets (len+1,) int64)."""
    clouds = [make_cloud(s, n_points, kind) for s in seeds]
    offsets = np.zeros(len(clouds) + 1, np.int64)
    offsets[1:] = np.cumsum([len(c) for c in clouds])
    return (np.concatenate(clouds, 0) if clouds else np.zeros((0, 4), np.float32)), offsets


def make_clouds_device(n_clouds, n_points, device, seed=0, order="uniform"):
    """Batch of clouds generated directly in HBM with torch (bench workload: 1 024 x 120 k points = 1.97 GB never
    crosses PCIe).  Not bit-reproducible against make_cloud(); parity checks on bench data copy a sample of clouds
    back to the host and feed the same bits to the oracle.

    order: 'uniform'        az, el, r i.i.d. uniform, points in random order (the friendly case for the LDS image)
           'azimuth_major'  HDL-64-like sensor order: 64 lasers fire at one azimuth, then the head turns -- 64
                            consecutive points share a column, every 4 of them a pixel (SURVEY section 7's hazard)
           'ring_major'     the same scan sorted by laser: one ring's whole revolution is contiguous -- ~5 consecutive
                            points per pixel, a wave's 64 points fall on ~12 neighbouring pixels of ONE row
    The scans carry a smooth place-dependent range profile and 5 % dropouts (holes for the interpolation)."""
    import math
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    pts = torch.empty((n_clouds * n_points, 4), dtype=torch.float32, device=device)
    chunk = max(1, (1 << 24) // max(n_points, 1))
    if order != "uniform":
        if order not in ("azimuth_major", "ring_major"):
            raise ValueError(order)
        rings = 64
        steps = n_points // rings
        assert steps * rings == n_points, "n_points must be a multiple of 64 for sensor-ordered clouds"
        for c0 in range(0, n_clouds, chunk):
            c1 = min(n_clouds, c0 + chunk)
            nc = c1 - c0
            if order == "azimuth_major":
                st = torch.arange(steps, device=device, dtype=torch.float32).view(1, steps, 1)
                rg = torch.arange(rings, device=device, dtype=torch.float32).view(1, 1, rings)
                shape = (nc, steps, rings)
            else:
                st = torch.arange(steps, device=device, dtype=torch.float32).view(1, 1, steps)
                rg = torch.arange(rings, device=device, dtype=torch.float32).view(1, rings, 1)
                shape = (nc, rings, steps)
            ph = torch.rand((nc, 3), generator=g, device=device, dtype=torch.float32) * (2 * math.pi)
            ph = ph.view(nc, 3, 1, 1)
            noise = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
            az = -math.pi + st * (2 * math.pi / steps) + noise * 2e-4
            noise = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
            el = (-24.6 + rg * (26.5 / (rings - 1))) * (math.pi / 180.0) + noise * 2e-4
            r = (25.0 + 12.0 * torch.sin(2 * az + ph[:, 0]) + 6.0 * torch.sin(5 * az + ph[:, 1])
                 + 3.0 * torch.sin(11 * az + ph[:, 2]) + 20.0 * rg / rings)
            drop = torch.rand(shape, generator=g, device=device, dtype=torch.float32) < 0.05
            r = torch.where(drop, torch.full_like(r, 0.2), r)            # below min_range: filtered out
            ce = torch.cos(el)
            blk = pts[c0 * n_points:c1 * n_points]
            blk[:, 0] = (r * ce * torch.cos(az)).reshape(-1)
            blk[:, 1] = (r * ce * torch.sin(az)).reshape(-1)
            blk[:, 2] = (r * torch.sin(el)).reshape(-1)
            blk[:, 3] = torch.rand(nc * n_points, generator=g, device=device, dtype=torch.float32)
            del az, el, r, ce, noise, drop
        offsets = torch.arange(0, n_clouds + 1, dtype=torch.int64, device=device) * n_points
        return pts, offsets
    for c0 in range(0, n_clouds, chunk):
        c1 = min(n_clouds, c0 + chunk)
        m = (c1 - c0) * n_points
        u = torch.rand((4, m), generator=g, device=device, dtype=torch.float32)
        az = (u[0] * 2.0 - 1.0) * math.pi
        el = (u[1] * 29.0 - 26.0) * (math.pi / 180.0)
        r = u[2] * 89.5 + 0.5
        ce = torch.cos(el)
        blk = pts[c0 * n_points:c1 * n_points]
        blk[:, 0] = r * ce * torch.cos(az)
        blk[:, 1] = r * ce * torch.sin(az)
        blk[:, 2] = r * torch.sin(el)
        blk[:, 3] = u[3]
        del u, az, el, r, ce
    offsets = torch.arange(0, n_clouds + 1, dtype=torch.int64, device=device) * n_points
    return pts, offsets


def make_pose_chain(n, seed=0):
    """Smooth 2-D random-walk SE(3) poses (n,4,4) float64: step 0.8-1.5 m, slow yaw drift."""
    rng = np.random.default_rng(seed)
    yaw = np.cumsum(rng.normal(0.0, 0.03, n))
    step = rng.uniform(0.8, 1.5, n)
    xy = np.cumsum(np.stack([step * np.cos(yaw), step * np.sin(yaw)], 1), 0)
    poses = np.tile(np.eye(4), (n, 1, 1))
    c, s = np.cos(yaw), np.sin(yaw)
    poses[:, 0, 0] = c
    poses[:, 0, 1] = -s
    poses[:, 1, 0] = s
    poses[:, 1, 1] = c
    poses[:, 0, 3] = xy[:, 0]
    poses[:, 1, 3] = xy[:, 1]
    poses[:, 2, 3] = rng.normal(0.0, 0.05, n)
    return poses


def randomize_bn_stats(model, seed=1):
    """Give every BatchNorm of a freshly initialised model non-trivial running stats and affine terms (fresh modules
    have 0 / 1, which would make eval-mode BatchNorm a no-op in benchmarks and parity tests alike)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            with torch.no_grad():
                m.running_mean.copy_((torch.randn(m.num_features, generator=g) * 0.1).to(m.running_mean.device))
                m.running_var.copy_((torch.rand(m.num_features, generator=g) * 1.5 + 0.25).to(m.running_var.device))
                m.weight.copy_((torch.rand(m.num_features, generator=g) + 0.5).to(m.weight.device))
                m.bias.copy_((torch.randn(m.num_features, generator=g) * 0.1).to(m.bias.device))
