"""lidf_oracle.py — CPU restatement of the LIDF per-point query path.  TEST INFRASTRUCTURE ONLY.

This file is the parity oracle and the `cpu_baseline` of bench.py. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; nothing under
implicit_depth_amd/ does. It is written with plain torch-CPU / numpy ops of the same kind the
reference uses (nn.functional.linear, leaky_relu, cat, sin/cos, gathers), in the reference's
operation order, each function citing the reference lines it follows (paths relative to
/root/reference/src).

Pinning status:
  * embed / IMNet / IEF / get_miss_ray / get_embedding / get_pred: PINNED against the reference
    itself — tests/golden/*.npz are produced by tests/golden/make_golden.py, which imports the
    reference's models/implicit_net.py and models/pipeline.py in the authoring container.
  * ray_aabb / pcl_aabb: pinned by source (the two .cu files are in the reference tree; no nvcc
    here, so they cannot be executed); also restated in C in oracle/aabb_ref.c.
  * roi_align (torchvision 0.7.0) and torch_scatter: sources are NOT in the reference tree and
    neither package is installed -> PARITY UNPINNED for those two; restated from their documented
    semantics and property-tested.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# a1  Embedder / get_embedder — models/implicit_net.py:9-57
# ----------------------------------------------------------------------------------------------
def embed(x, multires):
    """cat[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)] (implicit_net.py:17-39;
    freq_bands = 2**linspace(0, L-1, L) are exact powers of two)."""
    outs = [x]
    freqs = 2.0 ** torch.linspace(0.0, multires - 1, steps=multires) if multires > 0 else []
    for f in freqs:
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


def embed_dim(multires):
    return 3 + 6 * multires


# ----------------------------------------------------------------------------------------------
# a2/a3  IMNet / IEF — models/implicit_net.py:81-98, 129-152
# params: dict with linear_{1..4}.{weight,bias} (+ offset_enc.{weight,bias}) as torch f32 tensors
# ----------------------------------------------------------------------------------------------
def _out_act(y, use_sigmoid):
    if use_sigmoid:
        return torch.sigmoid(y)
    return torch.max(torch.min(y, y * 0.01 + 0.99), y * 0.01)  # implicit_net.py:96


def _mlp4(p, x):
    l1 = F.leaky_relu(F.linear(x, p["linear_1.weight"], p["linear_1.bias"]), 0.02)
    l2 = F.leaky_relu(F.linear(l1, p["linear_2.weight"], p["linear_2.bias"]), 0.02)
    l3 = F.leaky_relu(F.linear(l2, p["linear_3.weight"], p["linear_3.bias"]), 0.02)
    return F.linear(l3, p["linear_4.weight"], p["linear_4.bias"])


def imnet_forward(p, x, use_sigmoid=False):
    return _out_act(_mlp4(p, x), use_sigmoid)


def ief_forward(p, x, n_iter, use_sigmoid=False, init_offset=0.001):
    off = torch.full((x.shape[0], 1), init_offset, dtype=torch.float32)  # implicit_net.py:104,132
    for _ in range(n_iter):
        feat = F.linear(off, p["offset_enc.weight"], p["offset_enc.bias"])
        off = off + _mlp4(p, torch.cat([x, feat], 1))
    return _out_act(off, use_sigmoid)


def decoder_forward(p, x, kind, n_iter=2, use_sigmoid=False):
    return ief_forward(p, x, n_iter, use_sigmoid) if kind == "IEF" else imnet_forward(p, x, use_sigmoid)


def init_decoder(kind, inp_dim, seed, scale=1.0, gf=64):
    """Reference initialisation (implicit_net.py:72-79 / :118-127): N(0,0.02) weights, zero bias,
    linear_4.weight mean 1e-5; optionally scaled (SURVEY §8d uses x5) — deterministic from seed."""
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


def randomize_biases(p, seed, std=0.05):
    """Tests only: non-zero biases so bias handling is exercised."""
    g = torch.Generator().manual_seed(seed)
    for k in p:
        if k.endswith(".bias"):
            p[k] = torch.randn(p[k].shape, generator=g) * std
    return p


# ----------------------------------------------------------------------------------------------
# a4  LIDF.get_miss_ray (dense part) — models/pipeline.py:208-220
# ----------------------------------------------------------------------------------------------
def ray_dirs(fx, fy, cx, cy, h, w):
    """fx..cy: [bs] f32 tensors. Returns ray_dir [bs,h,w,3] and integer pixel grid [bs,h*w,2]."""
    bs = fx.shape[0]
    y_ind, x_ind = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    x_ind = x_ind.unsqueeze(0).repeat(bs, 1, 1).float()
    y_ind = y_ind.unsqueeze(0).repeat(bs, 1, 1).float()
    img_ind_flat = torch.stack((x_ind, y_ind), -1).reshape(bs, h * w, 2).long()
    cam_x = x_ind - cx.reshape(-1, 1, 1)
    cam_y = (y_ind - cy.reshape(-1, 1, 1)) * fx.reshape(-1, 1, 1) / fy.reshape(-1, 1, 1)
    cam_z = fx.reshape(-1, 1, 1).repeat(1, h, w)
    d = torch.stack((cam_x, cam_y, cam_z), -1)
    d = d / torch.norm(d, dim=-1, keepdim=True)
    return d, img_ind_flat


def get_miss_ray(mask, fx, fy, cx, cy):
    """LIDF.get_miss_ray, eval flavour (models/pipeline.py:208-230, :255-269): mask [bs,h,w]
    (data_dict['pred_mask']); the train-only random window (:232-254) is not restated."""
    bs, h, w = mask.shape
    d, img_ind_flat = ray_dirs(fx, fy, cx, cy, h, w)
    ray_dir_flat = d.reshape(bs, -1, 3)
    miss_idx = torch.nonzero(mask.reshape(bs, -1), as_tuple=False)
    miss_bid, miss_flat_img_id = miss_idx[:, 0], miss_idx[:, 1]
    return {"miss_bid": miss_bid, "miss_flat_img_id": miss_flat_img_id,
            "miss_ray_dir": ray_dir_flat[miss_bid, miss_flat_img_id],
            "miss_img_ind": img_ind_flat[miss_bid, miss_flat_img_id],
            "total_miss_sample_num": miss_idx.shape[0]}


def sample_miss_window(miss_idx, bs, miss_sample_num):
    """The train-only sub-sampling of LIDF.get_miss_ray (models/pipeline.py:232-254): per image a
    contiguous window of miss_sample_num entries of miss_idx [R,2] at a random start
    (np.random.choice(start_range), the reference's own RNG call), every entry if the image has no
    more than that. Returns the selected row indices into miss_idx."""
    if miss_sample_num == -1 or bs * miss_sample_num >= miss_idx.shape[0]:
        return torch.arange(miss_idx.shape[0])
    cnt = torch.bincount(miss_idx[:, 0], minlength=bs)
    edges = torch.cat((torch.zeros(1, dtype=torch.long), torch.cumsum(cnt, 0)))
    sel = []
    for i in range(bs):
        sid, eid, c = int(edges[i]), int(edges[i + 1]), int(cnt[i])
        if c > miss_sample_num:
            start = int(np.random.choice(c - miss_sample_num + 1)) + sid
            sel.append(torch.arange(start, start + miss_sample_num))
        else:
            sel.append(torch.arange(sid, eid))
    return torch.cat(sel, 0)


# ----------------------------------------------------------------------------------------------
# a5  ray_aabb — extensions/ray_aabb/ray_aabb_cuda_kernel.cu:24-88 (numpy, same arithmetic)
# ----------------------------------------------------------------------------------------------
def ray_aabb(ray_dir, voxel_bound, ray_bid, voxel_bid):
    """Returns mask [V,R] int32 and dist [V,R,2] f32, zero where no hit (cu:105-106)."""
    d = np.asarray(ray_dir, dtype=np.float32)
    vb = np.asarray(voxel_bound, dtype=np.float32)
    rb = np.asarray(ray_bid).astype(np.int64)
    vbid = np.asarray(voxel_bid).astype(np.int64)
    R, V = d.shape[0], vb.shape[0]
    mask = np.zeros((V, R), np.int32)
    dist = np.zeros((V, R, 2), np.float32)
    if R == 0 or V == 0:
        return mask, dist
    # 1 / (dir + 1e-12): double add and divide, rounded to float (cu:32,48,67)
    inv = (1.0 / (d.astype(np.float64) + 1e-12)).astype(np.float32)  # [R,3]
    pos = inv >= 0
    for v in range(V):
        lo, hi = vb[v, :3], vb[v, 3:]
        near = np.where(pos, lo[None, :], hi[None, :]).astype(np.float32)
        far = np.where(pos, hi[None, :], lo[None, :]).astype(np.float32)
        tmin = near * inv
        tmax = far * inv
        ok = rb == vbid[v]
        tmin_max = tmin[:, 0].copy()
        tmax_min = tmax[:, 0].copy()
        ok &= ~((tmin_max > tmax[:, 1]) | (tmax_min < tmin[:, 1]))
        tmin_max = np.maximum(tmin_max, tmin[:, 1])
        tmax_min = np.minimum(tmax_min, tmax[:, 1])
        ok &= ~((tmin_max > tmax[:, 2]) | (tmax_min < tmin[:, 2]))
        tmin_max = np.maximum(tmin_max, tmin[:, 2])
        tmax_min = np.minimum(tmax_min, tmax[:, 2])
        mask[v, ok] = 1
        dist[v, ok, 0] = tmin_max[ok]
        dist[v, ok, 1] = tmax_min[ok]
    return mask, dist


def pcl_aabb(pcl_pos, voxel_bound, pcl_bid, voxel_bid):
    """extensions/pcl_aabb/pcl_aabb_cuda_kernel.cu:23-44 — inclusive inside test, mask [V,Np]."""
    p = np.asarray(pcl_pos, dtype=np.float32)
    vb = np.asarray(voxel_bound, dtype=np.float32)
    pb = np.asarray(pcl_bid).astype(np.int64)
    vbid = np.asarray(voxel_bid).astype(np.int64)
    V, N = vb.shape[0], p.shape[0]
    mask = np.zeros((V, N), np.int32)
    for v in range(V):
        ok = pb == vbid[v]
        for a in range(3):
            ok &= ~((p[:, a] < vb[v, a]) | (p[:, a] > vb[v, 3 + a]))
        mask[v, ok] = 1
    return mask


# ----------------------------------------------------------------------------------------------
# a7  torchvision.ops.roi_align(output_size=2, spatial_scale=1.0, sampling_ratio=-1, aligned=True)
# torchvision 0.7.0 (pin: reference README.md:45, Dockerfile:29); source not in the reference tree
# -> PARITY UNPINNED. Restated from the published algorithm (RoIAlign forward + bilinear_interpolate).
# ----------------------------------------------------------------------------------------------
def _bilinear(img, y, x):
    """img [C,H,W] f32 numpy; y,x python floats (f32 semantics kept with np.float32)."""
    C, H, W = img.shape
    if y < -1.0 or y > H or x < -1.0 or x > W:
        return np.zeros(C, np.float32)
    y = np.float32(max(y, 0.0))
    x = np.float32(max(x, 0.0))
    y_low, x_low = int(y), int(x)
    if y_low >= H - 1:
        y_high = y_low = H - 1
        y = np.float32(y_low)
    else:
        y_high = y_low + 1
    if x_low >= W - 1:
        x_high = x_low = W - 1
        x = np.float32(x_low)
    else:
        x_high = x_low + 1
    ly = np.float32(y - np.float32(y_low))
    lx = np.float32(x - np.float32(x_low))
    hy = np.float32(1.0) - ly
    hx = np.float32(1.0) - lx
    w1, w2, w3, w4 = hy * hx, hy * lx, ly * hx, ly * lx
    return (w1 * img[:, y_low, x_low] + w2 * img[:, y_low, x_high]
            + w3 * img[:, y_high, x_low] + w4 * img[:, y_high, x_high]).astype(np.float32)


def roi_align(feat, boxes, output_size=2, spatial_scale=1.0, aligned=True):
    """feat [B,C,H,W] torch f32; boxes [K,5] (bid,x1,y1,x2,y2) f32. Returns [K,C,out,out]."""
    f = feat.detach().numpy().astype(np.float32)
    b = boxes.detach().numpy().astype(np.float32)
    K = b.shape[0]
    C = f.shape[1]
    out = np.zeros((K, C, output_size, output_size), np.float32)
    offset = np.float32(0.5 if aligned else 0.0)
    cache = {}
    for k in range(K):
        key = tuple(b[k].tolist())
        if key in cache:
            out[k] = cache[key]
            continue
        bid = int(b[k, 0])
        rsw = np.float32(b[k, 1] * np.float32(spatial_scale) - offset)
        rsh = np.float32(b[k, 2] * np.float32(spatial_scale) - offset)
        rew = np.float32(b[k, 3] * np.float32(spatial_scale) - offset)
        reh = np.float32(b[k, 4] * np.float32(spatial_scale) - offset)
        roi_w = np.float32(rew - rsw)
        roi_h = np.float32(reh - rsh)
        if not aligned:
            roi_w = max(roi_w, np.float32(1.0))
            roi_h = max(roi_h, np.float32(1.0))
        bin_h = np.float32(roi_h / np.float32(output_size))
        bin_w = np.float32(roi_w / np.float32(output_size))
        gh = int(math.ceil(roi_h / output_size))
        gw = int(math.ceil(roi_w / output_size))
        count = np.float32(max(gh * gw, 1))
        for ph in range(output_size):
            for pw in range(output_size):
                acc = np.zeros(C, np.float32)
                for iy in range(gh):
                    y = np.float32(np.float32(rsh + np.float32(ph) * bin_h)
                                   + np.float32(np.float32(iy + 0.5) * bin_h) / np.float32(gh))
                    for ix in range(gw):
                        x = np.float32(np.float32(rsw + np.float32(pw) * bin_w)
                                       + np.float32(np.float32(ix + 0.5) * bin_w) / np.float32(gw))
                        acc = acc + _bilinear(f[bid], y, x)
                out[k, :, ph, pw] = acc / count
        cache[key] = out[k].copy()
    return torch.from_numpy(out)


def _axis_taps(start, binsz, g, size):
    """Per-axis bilinear taps of the RoIAlign sample points: start/binsz [n] f32, g samples per
    bin. Returns (lo, hi) int64 [n,2,g] and (wlo, whi) f32 [n,2,g] following bilinear_interpolate's
    edge rules (outside [-1, size] -> zero weight; clamp at 0; last row/col collapse)."""
    ph = torch.arange(2, dtype=torch.float32).view(1, 2, 1)
    i = (torch.arange(g, dtype=torch.float32) + 0.5).view(1, 1, g)
    c = (start.view(-1, 1, 1) + ph * binsz.view(-1, 1, 1)) + (i * binsz.view(-1, 1, 1)) / float(g)
    dead = (c < -1.0) | (c > float(size))
    c = c.clamp(min=0.0)
    lo = c.floor().long()
    edge = lo >= size - 1
    lo = torch.where(edge, torch.full_like(lo, size - 1), lo)
    hi = torch.where(edge, lo, lo + 1)
    c = torch.where(edge, lo.float(), c)
    l = c - lo.float()
    wlo, whi = 1.0 - l, l
    wlo = torch.where(dead, torch.zeros_like(wlo), wlo)
    whi = torch.where(dead, torch.zeros_like(whi), whi)
    return lo, hi, wlo, whi


def roi_align_fast(feat, boxes, chunk=4096):
    """Vectorised torch-CPU form of roi_align above (output 2x2, aligned=True, scale 1): the
    bilinear taps are separable per axis; rays are grouped by their (grid_h, grid_w). Sums are
    re-associated, so it matches roi_align to float rounding (checked in tests). Used where the
    per-box Python loop would dominate (cpu_baseline, larger tests)."""
    B, C, H, W = feat.shape
    K = boxes.shape[0]
    out = torch.zeros(K, C, 2, 2)
    bid = boxes[:, 0].long()
    rsw, rsh = boxes[:, 1] - 0.5, boxes[:, 2] - 0.5
    roi_w, roi_h = (boxes[:, 3] - 0.5) - rsw, (boxes[:, 4] - 0.5) - rsh
    bw, bh = roi_w / 2.0, roi_h / 2.0
    gw, gh = torch.ceil(roi_w / 2.0).long(), torch.ceil(roi_h / 2.0).long()
    for g_h in torch.unique(gh).tolist():
        for g_w in torch.unique(gw).tolist():
            sel = ((gh == g_h) & (gw == g_w)).nonzero().flatten()
            if sel.numel() == 0 or g_h == 0 or g_w == 0:
                continue
            for s0 in range(0, sel.numel(), chunk):
                ids = sel[s0:s0 + chunk]
                ylo, yhi, wyl, wyh = _axis_taps(rsh[ids], bh[ids], g_h, H)
                xlo, xhi, wxl, wxh = _axis_taps(rsw[ids], bw[ids], g_w, W)
                n = ids.numel()
                bsel = bid[ids].view(n, 1, 1, 1)
                csel = torch.arange(C).view(1, C, 1, 1)
                acc = torch.zeros(n, C, 2, 2)
                for yi, wy in ((ylo, wyl), (yhi, wyh)):
                    for xi, wx in ((xlo, wxl), (xhi, wxh)):
                        v = feat[bsel, csel, yi.reshape(n, 1, 2 * g_h, 1), xi.reshape(n, 1, 1, 2 * g_w)]
                        v = v * wy.reshape(n, 1, 2 * g_h, 1) * wx.reshape(n, 1, 1, 2 * g_w)
                        acc += v.reshape(n, C, 2, g_h, 2, g_w).sum((3, 5))
                out[ids] = acc / float(max(g_h * g_w, 1))
    return out


def roi_boxes(img_ind, bid, h, w, roi_inp_bbox=8):
    """models/pipeline.py:374-383 — pixel +- roi_inp_bbox//2, corners clamped on int64, .float()."""
    ul = img_ind - roi_inp_bbox // 2
    br = img_ind + roi_inp_bbox // 2
    ul = torch.stack((ul[:, 0].clamp(0, w - 1), ul[:, 1].clamp(0, h - 1)), 1)
    br = torch.stack((br[:, 0].clamp(0, w - 1), br[:, 1].clamp(0, h - 1)), 1)
    return torch.cat((bid.unsqueeze(-1), ul, br), -1).float()


# ----------------------------------------------------------------------------------------------
# torch_scatter — source not in the reference tree, version unpinned -> PARITY UNPINNED.
# scatter_softmax: exp(x - segmax) / segsum.  scatter_max: (max, argmax) with empty -> len(src);
# ties -> lowest index (torch_scatter's CPU loop keeps the first maximum).
# ----------------------------------------------------------------------------------------------
def scatter_softmax(src, index, dim_size=None):
    n = int(index.max()) + 1 if dim_size is None and index.numel() else (dim_size or 0)
    mx = torch.full((n,), -float("inf"), dtype=src.dtype)
    mx = mx.scatter_reduce(0, index, src, reduce="amax", include_self=True)
    e = (src - mx[index]).exp()
    s = torch.zeros(n, dtype=src.dtype).index_add_(0, index, e)
    return e / s[index]


def scatter_max(src, index, dim_size):
    out = torch.zeros(dim_size, dtype=src.dtype)
    arg = torch.full((dim_size,), src.shape[0], dtype=torch.long)
    s = src.detach().numpy()   # values only: the arg-max carries no gradient
    idx = index.numpy()
    best = {}
    for i in range(s.shape[0]):
        r = int(idx[i])
        if r not in best or s[i] > best[r][0]:
            best[r] = (s[i], i)
    for r, (v, i) in best.items():
        out[r] = float(v)
        arg[r] = i
    return out, arg


# ----------------------------------------------------------------------------------------------
# a5..a10  the query itself on ray-major pairs
# ----------------------------------------------------------------------------------------------
def pairs_from_dense(mask, dist):
    """models/pipeline.py:283-285 then re-ordered ray-major (stable): within a ray, pairs stay in
    ascending voxel order. Returns pair_ray, pair_vox (int64), pair_t [P,2], pair_off [R+1]."""
    m = torch.as_tensor(mask).long()
    idx = torch.nonzero(m, as_tuple=False)  # voxel-major
    vox, ray = idx[:, 0], idx[:, 1]
    order = torch.argsort(ray, stable=True)
    vox, ray = vox[order], ray[order]
    t = torch.as_tensor(dist)[vox, ray]
    R = m.shape[1]
    cnt = torch.bincount(ray, minlength=R)
    off = torch.zeros(R + 1, dtype=torch.long)
    off[1:] = torch.cumsum(cnt, 0)
    return ray, vox, t.float(), off


def build_inp_embed(ray_dir, ray_pix, ray_bid, pair_ray, pair_vox, pair_t, feat_grid, vox_feat,
                    multires, multires_views, roi_inp_bbox=8, vox_center=None, pos_rel=False):
    """LIDF.get_embedding (models/pipeline.py:343-410) + the concat of get_pred (:431-433).
    Returns inp_embed [P, D], enter_pos [P,3], dir [P,3]."""
    h, w = feat_grid.shape[2], feat_grid.shape[3]
    d = ray_dir[pair_ray]
    enter = d * pair_t[:, 0:1]
    leave = d * pair_t[:, 1:2]
    if pos_rel:
        c = vox_center[pair_vox]
        inp_enter, inp_leave = enter - c, leave - c
    else:
        inp_enter, inp_leave = enter, leave
    e_enter = embed(inp_enter, multires)
    e_leave = embed(inp_leave, multires)
    e_dir = embed(d, multires_views)
    # ROIAlign is keyed by the ray's pixel: compute once per ray, gather per pair
    boxes = roi_boxes(ray_pix.long(), ray_bid.long(), h, w, roi_inp_bbox)
    ray_rgb = roi_align(feat_grid, boxes).reshape(ray_pix.shape[0], -1)
    rgb = ray_rgb[pair_ray]
    vox = vox_feat[pair_vox]
    inp = torch.cat((vox.contiguous(), rgb.contiguous(), e_enter, e_leave, e_dir), -1)
    return inp, enter, d


def query(ray_dir, ray_pix, ray_bid, pair_ray, pair_vox, pair_t, pair_off, feat_grid, vox_feat,
          prob_p, off_p, off_kind="IEF", n_iter=2, use_sigmoid=False, multires=8, multires_views=4,
          roi_inp_bbox=8, offset_range=(0.0, 1.0), part_size=0.25, vox_center=None, pos_rel=False,
          chunk=262144, fast_roi=False, roi_out_bbox=2, max_pair_id=None):
    """get_embedding + get_pred (models/pipeline.py:338-466) + depth z. Pairs are ray-major.
    roi_out_bbox: model.roi_out_bbox (:387), 2 in every shipped config. max_pair_id [R] int64: the selection by
    ground-truth labels of training while epoch < maxpool_label_epo (:444-446) instead of the arg-max."""
    R = ray_dir.shape[0]
    P = pair_ray.shape[0]
    boxes = roi_boxes(ray_pix.long(), ray_bid.long(), feat_grid.shape[2], feat_grid.shape[3],
                      roi_inp_bbox)
    if fast_roi and roi_out_bbox == 2:
        ray_rgb = roi_align_fast(feat_grid, boxes).reshape(R, -1)
    else:
        ray_rgb = roi_align(feat_grid, boxes, output_size=roi_out_bbox).reshape(R, -1)
    e_dir_ray = embed(ray_dir, multires_views)
    pred_offset = torch.empty(P, 1)
    pred_prob = torch.empty(P, 1)
    pair_pred_pos = torch.empty(P, 3)
    for s in range(0, P, chunk):
        sl = slice(s, min(P, s + chunk))
        pr, pv, pt = pair_ray[sl], pair_vox[sl], pair_t[sl]
        d = ray_dir[pr]
        enter = d * pt[:, 0:1]
        leave = d * pt[:, 1:2]
        if pos_rel:
            c = vox_center[pv]
            ie, il = enter - c, leave - c
        else:
            ie, il = enter, leave
        inp = torch.cat((vox_feat[pv], ray_rgb[pr], embed(ie, multires), embed(il, multires),
                         e_dir_ray[pr]), -1)
        po = decoder_forward(off_p, inp, off_kind, n_iter, use_sigmoid)
        pp = imnet_forward(prob_p, inp, use_sigmoid)
        # pipeline.py:437-439
        sc = po * (offset_range[1] - offset_range[0]) + offset_range[0]
        sc = sc * np.sqrt(3) * part_size
        pred_offset[sl], pred_prob[sl] = po, pp
        pair_pred_pos[sl] = enter + sc * d
    if P > 0:
        sm = scatter_softmax(pred_prob[:, 0], pair_ray, dim_size=R)
    else:
        sm = torch.empty(0)
    if max_pair_id is None:
        _, max_pair_id = scatter_max(sm, pair_ray, dim_size=R)
    dummy = torch.cat((pair_pred_pos, torch.zeros(1, 3)), 0)  # pipeline.py:452-454
    pred_pos = dummy[max_pair_id]
    return {
        "pred_offset": pred_offset, "pred_prob_end": pred_prob, "pair_pred_pos": pair_pred_pos,
        "pred_prob_end_softmax": sm, "max_pair_id": max_pair_id, "pred_pos": pred_pos,
        "ray_rgb": ray_rgb,
    }


# ----------------------------------------------------------------------------------------------
# f4  occupied-voxel build — utils/point_utils.py:12-76 batch_get_occupied_idx (overlap=False)
#     + LIDF.get_occ_vox_bound (models/pipeline.py:162-201)
# ----------------------------------------------------------------------------------------------
def occupied_voxels(valid_xyz, valid_bid, xmin=(-1.0, -1.0, 0.0), xmax=(1.0, 1.0, 2.0), res=8):
    lo = torch.tensor(xmin, dtype=torch.float32)
    hi = torch.tensor(xmax, dtype=torch.float32)
    part_size = torch.min(hi - lo).item() / res
    lo = lo - 0.5 * part_size
    hi = hi + 0.5 * part_size
    v = valid_xyz.clone() - lo.unsqueeze(0)
    r = torch.ceil((hi - lo) / part_size).long()
    coord = torch.floor(v / part_size).long()
    center = coord * part_size + 0.5 * part_size
    rel = v - center
    ok = torch.ones(v.shape[0], dtype=torch.bool)
    for i in range(3):
        ok &= (coord[:, i] >= 0) & (coord[:, i] < r[i])
    pid = torch.arange(v.shape[0])[ok]
    key = torch.cat((valid_bid.long().unsqueeze(-1)[ok], coord[ok]), -1)
    occ, revidx = torch.unique(key, dim=0, return_inverse=True)
    bound_min = lo.unsqueeze(0) + occ[:, 1:] * part_size
    return {"part_size": part_size, "xmin": lo, "revidx": revidx, "valid_v_pid": pid,
            "valid_v_rel_coord": rel[ok], "occ_vox_bid": occ[:, 0], "occ_vox_global_coord": occ[:, 1:],
            "voxel_bound": torch.cat((bound_min, bound_min + part_size), 1)}


# ----------------------------------------------------------------------------------------------
# a11  stage-2 refinement: PointNet2Stage (models/pointnet.py:22-38) and
#      RefineNet.get_pred_refine (models/pipeline.py:922-1030), eval flavour
# ----------------------------------------------------------------------------------------------
def _scatter_max_rows(x, idx, n):
    """torch_scatter.scatter(x, idx, dim=0, reduce='max'); rows without points are 0 (x >= 0)."""
    out = torch.zeros(n, x.shape[1], dtype=x.dtype)
    return out.scatter_reduce(0, idx.view(-1, 1).expand_as(x), x, reduce="amax", include_self=True)


def pointnet2stage(p, inp, vox, n_vox):
    lin = lambda x, k: F.linear(x, p[k + ".weight"], p[k + ".bias"])  # noqa: E731
    f1 = F.relu(lin(inp, "point_lin1"))
    f2 = F.relu(lin(f1, "point_lin2"))
    g1 = F.relu(lin(_scatter_max_rows(f2, vox, n_vox), "vox_lin1"))
    f3 = torch.cat((g1[vox], f2), -1)
    f4 = F.relu(lin(f3, "point_lin3"))
    f5 = F.relu(lin(f4, "point_lin4"))
    return F.relu(lin(_scatter_max_rows(f5, vox, n_vox), "vox_lin2"))


def refine_step(pred_pos, ray_dir, ray_pix, ray_bid, ray_flat, max_pair_id, pair_vox, voxel_bound,
                voxel_bid, rgb_img, feat_grid, valid_inp, valid_vox, pnet_p, off_p, off_kind="IEF",
                n_iter=2, multires=8, multires_views=4, roi_inp_bbox=8, offset_range=(-0.2, 0.2),
                pos_rel=False, pnet_pos_rel=True, ray_rgb=None, pnet_select=None):
    """One get_pred_refine call. Returns (pred_pos_refine, end_voxel_id, occ_voxel_feat).
    pnet_select [R] bool: the mask_type 'all' / refine.use_all_pix False branch
    (models/pipeline.py:987-996) — only the selected rays' predicted points join the PointNet."""
    R, P, V = ray_dir.shape[0], pair_vox.shape[0], voxel_bound.shape[0]
    h, w = rgb_img.shape[2], rgb_img.shape[3]
    # end voxel: arg-max pair's voxel (dummy row -> 0), raised to the largest containing voxel
    pv = torch.cat((pair_vox.long(), torch.zeros(1, dtype=torch.long)))
    end_voxel = pv[max_pair_id.clamp(max=P)].clone()
    m = torch.from_numpy(pcl_aabb(pred_pos.detach().numpy(), voxel_bound.numpy(), ray_bid.numpy(),
                                  voxel_bid.numpy())).long()   # (an index: no gradient, as in autograd)
    idx = torch.nonzero(m, as_tuple=False)
    end_voxel.scatter_reduce_(0, idx[:, 1], idx[:, 0], reduce="amax", include_self=True)
    e_dir = embed(ray_dir, multires_views)
    if ray_rgb is None:
        boxes = roi_boxes(ray_pix.long(), ray_bid.long(), h, w, roi_inp_bbox)
        ray_rgb = roi_align(feat_grid, boxes).reshape(R, -1)
    rgb_flat = rgb_img.permute(0, 2, 3, 1).contiguous().reshape(rgb_img.shape[0], -1, 3)
    miss_rgb = rgb_flat[ray_bid.long(), ray_flat.long()]
    eb = voxel_bound[end_voxel]
    center = (eb[:, :3] + eb[:, 3:]) / 2.0
    pred_inp = torch.cat(((pred_pos - center) if pnet_pos_rel else pred_pos, miss_rgb), 1)
    if pnet_select is not None:
        sel = torch.nonzero(pnet_select.reshape(-1), as_tuple=False)[:, 0]
        pn_inp = torch.cat((valid_inp, pred_inp[sel]), 0)
        pn_vox = torch.cat((valid_vox.long(), end_voxel[sel]), 0)
    else:
        pn_inp = torch.cat((valid_inp, pred_inp), 0)
        pn_vox = torch.cat((valid_vox.long(), end_voxel), 0)
    occ_voxel_feat = pointnet2stage(pnet_p, pn_inp, pn_vox, V)
    enter = (pred_pos - center) if pos_rel else pred_pos
    inp = torch.cat((occ_voxel_feat[end_voxel], ray_rgb, embed(enter, multires), e_dir), -1)
    off = decoder_forward(off_p, inp, off_kind, n_iter)
    scaled = off * (offset_range[1] - offset_range[0]) + offset_range[0]
    return pred_pos + scaled * ray_dir, end_voxel, occ_voxel_feat


def lidf_forward(batch, full_rgb_feat, pnet_p, prob_p, off_p, valid_stride=None, multires=8,
                 multires_views=4, offset_range=(0.0, 1.0), fast_roi=True):
    """LIDF.forward, exp_type 'test', mask_type 'all' (models/pipeline.py:652-717) as a chain of the
    restatements above: prepare_data (:91-133), get_valid_points with every valid pixel (:135-160;
    optionally every valid_stride-th), get_occ_vox_bound, get_miss_ray, compute_ray_aabb,
    PointNet2Stage, get_embedding + get_pred, and the depth map of compute_loss (:593-596).
    Returns (success, dict)."""
    rgb = batch["rgb"]
    bs, _, h, w = rgb.shape
    dd = {"bs": bs, "h": h, "w": w}
    xyz_corrupt_flat = batch["xyz_corrupt"].permute(0, 2, 3, 1).contiguous().reshape(bs, -1, 3)
    valid_mask = 1 - (batch["depth_corrupt"] == 0).squeeze(1).float()
    pred_mask = torch.ones_like(batch["corrupt_mask"].squeeze(1))
    valid_idx = torch.nonzero(valid_mask.reshape(bs, -1), as_tuple=False)
    if valid_stride and valid_stride > 1:
        valid_idx = valid_idx[::valid_stride]
    vb_, vf_ = valid_idx[:, 0], valid_idx[:, 1]
    valid_xyz = xyz_corrupt_flat[vb_, vf_]
    valid_rgb = rgb.permute(0, 2, 3, 1).contiguous().reshape(bs, -1, 3)[vb_, vf_]
    occ = occupied_voxels(valid_xyz, vb_)
    dd.update(occ)
    dd.update({"valid_mask": valid_mask, "valid_xyz": valid_xyz, "valid_rgb": valid_rgb, "valid_bid": vb_})
    V = occ["voxel_bound"].shape[0]
    if V == 0:
        return False, dd
    mr = get_miss_ray(pred_mask, batch["fx"].float(), batch["fy"].float(), batch["cx"].float(),
                      batch["cy"].float())
    dd.update(mr)
    if mr["total_miss_sample_num"] == 0:
        return False, dd
    mask, dist = ray_aabb(mr["miss_ray_dir"].numpy(), occ["voxel_bound"].numpy(),
                          mr["miss_bid"].numpy(), occ["occ_vox_bid"].numpy())
    pair_ray, pair_vox, pair_t, pair_off = pairs_from_dense(mask, dist)
    dd.update({"pair_ray": pair_ray, "pair_vox": pair_vox, "pair_t": pair_t, "pair_off": pair_off})
    if pair_ray.shape[0] == 0:
        return False, dd
    pnet_inp = torch.cat((occ["valid_v_rel_coord"], valid_rgb[occ["valid_v_pid"]]), -1)
    dd["pnet_inp"] = pnet_inp
    dd["occ_voxel_feat"] = pointnet2stage(pnet_p, pnet_inp, occ["revidx"], V)
    out = query(mr["miss_ray_dir"], mr["miss_img_ind"], mr["miss_bid"], pair_ray, pair_vox, pair_t,
                pair_off, full_rgb_feat, dd["occ_voxel_feat"], prob_p, off_p, multires=multires,
                multires_views=multires_views, offset_range=offset_range,
                part_size=occ["part_size"], fast_roi=fast_roi)
    dd.update(out)
    pred_xyz = xyz_corrupt_flat.clone()
    pred_xyz[mr["miss_bid"], mr["miss_flat_img_id"]] = out["pred_pos"]
    dd["pred_depth"] = pred_xyz[:, :, 2].reshape(bs, h, w)
    return True, dd


def init_pointnet(seed, scale=1.0):
    """Deterministic PointNet2Stage(6, 128, 32) parameters (nn.Linear-like uniform init)."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, (dout, din) in {"point_lin1": (32, 6), "point_lin2": (64, 32), "vox_lin1": (64, 64),
                              "point_lin3": (128, 128), "point_lin4": (128, 128),
                              "vox_lin2": (128, 128)}.items():
        b = 1.0 / math.sqrt(din)
        p[name + ".weight"] = (torch.rand(dout, din, generator=g) * 2 - 1) * b * scale
        p[name + ".bias"] = (torch.rand(dout, generator=g) * 2 - 1) * b
    return p


# ----------------------------------------------------------------------------------------------
# Synthetic workload of SURVEY.md §8(d): lives in implicit_depth_amd/synthetic.py (plain data
# generation, no compute path) so that bench.py and the tests feed both sides the same tensors.
# ----------------------------------------------------------------------------------------------
def synthetic_scene(*args, **kw):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from implicit_depth_amd.synthetic import synthetic_scene as _impl
    return _impl(*args, **kw)


def resize_nearest_index(src, dst):
    """Source index of every destination index under cv2.resize(..., interpolation=INTER_NEAREST),
    restated from OpenCV's resizeNN: min(floor(dst index * (1 / (dst/src))), src - 1), factor in
    double. cv2 is not installed here and its source is not in the reference tree: parity with cv2
    itself is UNPINNED (tests/golden/make_golden.py's cv2 stub is this function)."""
    import math
    inv = 1.0 / (dst / src)
    return [min(int(math.floor(i * inv)), src - 1) for i in range(dst)]


def depth_metrics(pred_depth, gt_depth, seg_mask=None, out_size=(144, 256)):
    """Eval statistics of LIDF.compute_loss, bs == 1 branch (models/pipeline.py:577-627).
    cv2.resize(img, (256, 144), interpolation=INTER_NEAREST) is resize_nearest_index above (unpinned
    against cv2); the statistics follow the reference's torch f32 expressions literally (safe_log10
    is the natural log there, :611) and are pinned by tests/golden/g8_metrics.npz — the reference's
    own compute_loss run on a one-frame batch."""
    pred = torch.as_tensor(pred_depth, dtype=torch.float32)
    gt = torch.as_tensor(gt_depth, dtype=torch.float32).clone()
    h, w = gt.shape
    dh, dw = (h, w) if out_size is None else out_size
    sy = torch.tensor(resize_nearest_index(h, dh))
    sx = torch.tensor(resize_nearest_index(w, dw))
    gt = gt[sy][:, sx]
    pred = pred[sy][:, sx]
    gt[torch.isnan(gt)] = 0
    gt[torch.isinf(gt)] = 0
    mask = gt > 0
    if seg_mask is not None:
        sm = torch.as_tensor(seg_mask).to(torch.uint8)[sy][:, sx]
        mask = mask & (sm != 0)
    gt, pred = gt[mask], pred[mask]
    safe_log = lambda x: torch.log(torch.clamp(x, 1e-6, 1e6))  # noqa: E731
    thresh = torch.max(gt / pred, pred / gt)
    return {
        "a1": (thresh < 1.05).float().mean(), "a2": (thresh < 1.10).float().mean(),
        "a3": (thresh < 1.25).float().mean(),
        "rmse": ((gt - pred) ** 2).mean().sqrt(),
        "rmse_log": ((safe_log(gt) - safe_log(pred)) ** 2).mean().sqrt(),
        "log10": (safe_log(gt) - safe_log(pred)).abs().mean(),
        "abs_rel": ((gt - pred).abs() / gt).mean(), "mae": (gt - pred).abs().mean(),
        "sq_rel": ((gt - pred) ** 2 / gt).mean(), "count": torch.tensor(float(mask.sum())),
    }
