"""synthetic.py — seeded synthetic inputs for the LIDF query path.

Pure data generation on the CPU with torch (no dataset or checkpoint exists in this environment):
ClearGrasp-shaped frames (240x320, pinhole intrinsics), a dense ray-major candidate list of N
segments per ray inside the reference's grid bounds (constants.py:15-16 with the half-voxel margin
of models/pipeline.py:170-173, 9x9x9 voxels of 0.25 m), post-ReLU voxel features, a smooth
32-channel feature map, and decoder weights drawn like the reference's initialisation
(models/implicit_net.py:72-79, :118-127) scaled x5 so activations are not vanishingly small.
Used by bench.py and the tests to feed the HIP path and the CPU oracle identical tensors.
"""
import torch
import torch.nn.functional as F

GRID_XMIN = (-1.125, -1.125, -0.125)
PART_SIZE = 0.25
GRID_RES = 9


def init_decoder_params(kind, inp_dim, seed, scale=1.0, gf=64):
    """State dict (reference parameter names) of an IMNET / IEF decoder, deterministic in seed."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    dims = [(inp_dim + (16 if kind == "IEF" else 0), 4 * gf), (4 * gf, 2 * gf), (2 * gf, gf), (gf, 1)]
    if kind == "IEF":
        p["offset_enc.weight"] = torch.randn(16, 1, generator=g) * 0.02 * scale
        p["offset_enc.bias"] = torch.zeros(16)
    for i, (din, dout) in enumerate(dims, 1):
        w = torch.randn(dout, din, generator=g) * 0.02
        if i == 4:
            w = w + 1e-5
        p["linear_%d.weight" % i] = w * scale
        p["linear_%d.bias" % i] = torch.zeros(dout)
    return p


def pixel_rays(B, h, w):
    """Intrinsics fx=fy=0.9w, principal point at the image centre; unit ray per pixel."""
    fx = torch.full((B,), 0.9 * w)
    fy = torch.full((B,), 0.9 * w)
    cx = torch.full((B,), w / 2 - 0.5)
    cy = torch.full((B,), h / 2 - 0.5)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                            torch.arange(w, dtype=torch.float32), indexing="ij")
    xs = xs.unsqueeze(0).expand(B, h, w)
    ys = ys.unsqueeze(0).expand(B, h, w)
    vx = xs - cx.view(B, 1, 1)
    vy = (ys - cy.view(B, 1, 1)) * fx.view(B, 1, 1) / fy.view(B, 1, 1)
    vz = fx.view(B, 1, 1).expand(B, h, w)
    d = torch.stack((vx, vy, vz), -1)
    d = d / torch.sqrt((d * d).sum(-1, keepdim=True))
    pix = torch.stack((xs, ys), -1).reshape(B * h * w, 2).int()
    return d.reshape(B * h * w, 3).contiguous(), pix.contiguous(), torch.stack((fx, fy, cx, cy), 1)


def synthetic_scene(B, h, w, N, seed, ragged=False, weight_scale=5.0, multires=8,
                    multires_views=4):
    g = torch.Generator().manual_seed(seed)
    ray_dir, ray_pix, intr = pixel_rays(B, h, w)
    R = B * h * w
    ray_bid = torch.arange(B).repeat_interleave(h * w).int()
    ray_flat = torch.arange(h * w).repeat(B).int()
    delta = 1.75 / N
    k = torch.arange(N, dtype=torch.float32)
    t_enter = (0.25 + k * delta).unsqueeze(0).expand(R, N)
    t_leave = t_enter + delta
    mid = ray_dir.unsqueeze(1) * ((t_enter + t_leave) * 0.5).unsqueeze(-1)  # [R,N,3]
    xmin = torch.tensor(GRID_XMIN)
    cell = torch.floor((mid - xmin) / PART_SIZE).long().clamp(0, GRID_RES - 1)
    vox = (cell[..., 0] * GRID_RES + cell[..., 1]) * GRID_RES + cell[..., 2]
    vox = vox + (ray_bid.long() * GRID_RES ** 3).unsqueeze(1)
    if ragged:
        cnt = torch.randint(0, N + 1, (R,), generator=g)
    else:
        cnt = torch.full((R,), N, dtype=torch.long)
    keep = k.unsqueeze(0) < cnt.unsqueeze(1)
    pair_ray = torch.arange(R).unsqueeze(1).expand(R, N)[keep].int()
    pair_vox = vox[keep].int()
    pair_t = torch.stack((t_enter[keep], t_leave[keep]), -1).contiguous()
    pair_off = torch.zeros(R + 1, dtype=torch.int32)
    pair_off[1:] = torch.cumsum(cnt, 0).int()
    V = B * GRID_RES ** 3
    vox_feat = torch.relu(torch.randn(V, 128, generator=g))
    coarse = torch.randn(B, 32, max(h // 8, 1), max(w // 8, 1), generator=g)
    feat_grid = F.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=False).contiguous()
    ci = torch.arange(GRID_RES, dtype=torch.float32)
    cxyz = torch.stack(torch.meshgrid(ci, ci, ci, indexing="ij"), -1).reshape(-1, 3)
    vox_center = (xmin + (cxyz + 0.5) * PART_SIZE).repeat(B, 1).contiguous()
    D = 256 + 2 * (3 + 6 * multires) + (3 + 6 * multires_views)
    return {
        "B": B, "h": h, "w": w, "N": N, "R": R, "P": int(pair_ray.shape[0]), "V": V, "D": D,
        "ray_dir": ray_dir, "ray_pix": ray_pix, "ray_bid": ray_bid, "ray_flat": ray_flat,
        "pair_ray": pair_ray, "pair_vox": pair_vox, "pair_t": pair_t, "pair_off": pair_off,
        "vox_feat": vox_feat, "feat_grid": feat_grid, "vox_center": vox_center,
        "prob_p": init_decoder_params("IMNET", D, 7, weight_scale),
        "off_p": init_decoder_params("IEF", D, 8, weight_scale), "intr": intr.contiguous(),
    }


def synthetic_batch(B, h, w, seed, hole_frac=1.0):
    """A geometry-derived ClearGrasp-shaped batch (the keys of the reference's dataset items,
    datasets/cleargrasp_dataset.py:165-180): a slanted table with boxes and a sphere in front of
    it, elliptical holes where transparent objects corrupt the depth (corrupt_mask = 1,
    depth_corrupt = 0), camera-frame points inside the reference's grid bounds
    (utils/constants.py:15-16). Also returns a smooth 32-channel map standing in for the ResNet
    output `full_rgb_feat` (the producer is upstream of the path). Ragged by construction: the
    number of occupied voxels a ray crosses varies from 0 to ~10."""
    g = torch.Generator().manual_seed(seed)
    ray_dir, _, intr = pixel_rays(B, h, w)
    d = ray_dir.reshape(B, h, w, 3)
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) / h,
                            torch.arange(w, dtype=torch.float32) / w, indexing="ij")
    depth = torch.empty(B, h, w)
    hole = torch.zeros(B, 1, h, w)
    for b in range(B):
        z = 1.55 - 0.55 * ys + 0.10 * xs + 0.02 * b
        for k in range(4):   # boxes standing on the table
            x0, y0 = 0.08 + 0.22 * k + 0.03 * b, 0.25 + 0.1 * ((k + b) % 3)
            m = (xs > x0) & (xs < x0 + 0.14) & (ys > y0) & (ys < y0 + 0.3)
            z = torch.where(m, z - (0.15 + 0.06 * k), z)
        cxs, cys, rad = 0.55 + 0.05 * b, 0.62, 0.16   # a sphere
        rr = ((xs - cxs) ** 2 + ((ys - cys) * h / w) ** 2) / rad ** 2
        z = torch.where(rr < 1, z - 0.3 * torch.sqrt((1 - rr).clamp(min=0)), z)
        depth[b] = z
        for k in range(3):   # transparent objects: elliptical holes in the measured depth
            ex, ey = 0.2 + 0.3 * k + 0.02 * b, 0.45 + 0.12 * ((k + b) % 2)
            e = ((xs - ex) / (0.07 * hole_frac)) ** 2 + ((ys - ey) / (0.16 * hole_frac)) ** 2
            hole[b, 0] = torch.maximum(hole[b, 0], (e < 1).float())
    xyz = (d / d[..., 2:3] * depth.unsqueeze(-1)).permute(0, 3, 1, 2).contiguous()
    coarse = torch.randn(B, 3, max(h // 6, 1), max(w // 6, 1), generator=g)
    rgb = F.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=False).contiguous()
    coarse = torch.randn(B, 32, max(h // 8, 1), max(w // 8, 1), generator=g)
    feat = F.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=False).contiguous()
    batch = {
        "rgb": rgb, "xyz": xyz, "xyz_corrupt": xyz * (1 - hole),
        "depth_corrupt": depth.unsqueeze(1) * (1 - hole), "corrupt_mask": hole.clone(),
        "valid_mask": 1 - hole, "fx": intr[:, 0].clone(), "fy": intr[:, 1].clone(),
        "cx": intr[:, 2].clone(), "cy": intr[:, 3].clone(), "item_path": ["synthetic_%d" % b for b in range(B)],
    }
    return batch, feat
