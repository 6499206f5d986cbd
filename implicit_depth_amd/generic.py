"""generic.py — the reference's modules at widths other than the shipped configuration.

The register-chained kernels of liblidf_hip are built for the widths every shipped config uses
(imnet_gf 64, pnet_gf 32, pnet_out 128, rgb_out 32, roi_out_bbox 2: models/pipeline.py:56-85 with
experiments/implicit_depth/*.yaml). The reference's constructors take other values
(IMNet / IEF gf_dim and out_dim, PointNet2Stage input_channels / gf_dim / output_channels, the
ROI feature's channels and bins); those run here layer by layer: every nn.Linear is one
lidf_linear_f32 launch (the same f32 matrix-instruction kernel, 256 output columns per launch,
fused bias / activation / gathered term / scatter-max), the ROI pooling is lidf_roi_align_f32.
Same results as the reference's modules to the tolerance of the fast path (the summation order
inside a dot product differs from cuBLAS as it does there). The decoders also train at other
widths: every layer is an autograd function whose backward runs lidf_linear_f32 (input gradient)
and lidf_wgrad_f32 (weight and bias gradients), the PointNet's two poolings and its gather are
torch indexing that autograd differentiates itself; the fused query / stage-2 calls train at the
shipped widths only.

Round 6: a decoder of gf_dim 32, 64 or 128 with a 1-wide head runs its inference as ONE register-chained launch
(lidf_decoder_chain_f32, csrc/lidf_chain16.hip: 16-row sub-tiles of the 16 x 16 x 4 matrix instruction, activations
in registers from layer 1 to the output) instead of one launch per layer and pass — the query and decoder_forward
take it; LIDF_CHAIN16=0 keeps the layer-by-layer path (A/B, and the definition the chain is tested against).

Nothing here is used when the widths are the shipped ones.
"""
import os

import torch

from . import _lib


def chain_ok(mod):
    """A decoder lidf_decoder_chain_f32 is built for: gf_dim 32 / 64 / 128, the reference's 4 gf -> 2 gf -> gf -> 1."""
    if os.environ.get("LIDF_CHAIN16", "1") == "0":
        return False
    gf = int(mod.gf_dim)
    return (gf in (32, 64, 128) and mod.linear_1.out_features == 4 * gf and mod.linear_2.out_features == 2 * gf
            and mod.linear_3.out_features == gf and mod.linear_4.out_features == 1)


def _bias_row(mod):
    """b1 (+ the IEF's constant W1[:, enc columns] benc) as the one-row voxpart table of a chain launch."""
    from .decoders import IEF
    b = mod.linear_1.bias.detach().float()
    if isinstance(mod, IEF):
        d = mod.inp_dim
        b = b + mod.linear_1.weight.detach()[:, d:d + 16].float() @ mod.offset_enc.bias.detach().float()
    return b


def decoder_chain(mod, x, k, w1_col0=0, voxpart=None, vox_idx=None, raypart=None, ray_idx=None, out=None,
                  padded=False, packed=None):
    """The whole decoder on per-row operand columns x[:, :k] (they multiply W1[:, w1_col0 : w1_col0 + k]) plus two
    gathered table rows — lidf_decoder_chain_f32. voxpart must carry b1 (+ the IEF constant: _bias_row). padded: x's
    buffer is readable up to column 16 ceil(k / 16) of its last row (otherwise the trailing rows go through a
    padded copy). packed: a dict the caller keeps over the slabs of one query — the first call leaves its workspace
    (the packed weight stream) there, the later ones skip the pack launch. Returns [n, 1]."""
    from .decoders import _decoder_struct
    import ctypes as C
    n = x.shape[0]
    dev = x.device
    if out is None:
        out = torch.empty((n, 1), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    if x.stride(1) != 1:
        x = x.contiguous()
    ldx = x.stride(0) if n > 1 else x.shape[1]
    kq16 = (k + 15) // 16 * 16
    L = _lib.lib()
    gf = int(mod.gf_dim)
    wsb = L.lidf_decoder_chain_workspace_bytes(gf, k)
    state = packed if packed is not None else {}
    ws = state.get("ws")
    if ws is None:
        ws = state["ws"] = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=dev)
        state["ready"] = 0
    keep = []
    dec = _decoder_struct(mod, keep)

    def run(xs, ld, rows, o, vi, ri, rp):
        with torch.cuda.device(dev):
            _lib.check(L.lidf_decoder_chain_f32(
                C.byref(dec), gf, int(mod.inp_dim), _lib.ptr(xs), ld, k, w1_col0, rows, _lib.ptr(voxpart), _lib.ptr(vi),
                _lib.ptr(rp), _lib.ptr(ri), _lib.ptr(o), state["ready"], _lib.ptr(ws), wsb,
                _lib.current_stream(dev)))
        state["ready"] = 1

    # rows whose 16-column groups would read past the end of x's buffer: through a padded copy
    tail = 0 if (padded or kq16 <= k) else min(n, (kq16 - k + max(ldx, 1) - 1) // max(ldx, 1))
    head = n - tail
    if head:
        run(x, ldx, head, out, vox_idx, ray_idx, raypart)
    if tail:
        xt = torch.zeros((tail, kq16), dtype=torch.float32, device=dev)
        xt[:, :k] = x[head:, :k]
        # (the tail's rows of a ray table without an index array are rows head.. of it)
        rp_t = raypart[head:] if (raypart is not None and ray_idx is None) else raypart
        run(xt, kq16, tail, out[head:], vox_idx[head:] if vox_idx is not None else None,
            ray_idx[head:] if ray_idx is not None else None, rp_t)
    return out


def linear_hip(x, weight, bias=None, act=0, slope=0.0, addrows=None, addidx=None, out=None,
               pool=None, poolidx=None, w_col0=0, k=None, addrows2=None, addidx2=None):
    """out = act(x @ weight[:, w_col0:w_col0+k].T + bias (+ addrows[addidx]) (+ addrows2[addidx2])) through
    lidf_linear_f32 / lidf_linear_gather2_f32.
    x [n, >=k] f32 (row stride free), weight [nout, ldw] f32; act 0 none / 1 max(v, slope*v).
    pool [V, nout] (zero-initialised) receives the scatter-max over poolidx instead of / beside out."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2:
        raise RuntimeError("linear_hip: x must be a CUDA float32 matrix (no CPU path)")
    w = weight.detach()
    if w.dtype != torch.float32 or w.dim() != 2 or (w.shape[1] > 1 and w.stride(1) != 1):
        raise RuntimeError("linear_hip: weight must be float32 [nout, k] with unit column stride")
    n = x.shape[0]
    k = int(k if k is not None else w.shape[1] - w_col0)
    nout = int(w.shape[0])
    if x.shape[1] < k:
        raise RuntimeError("linear_hip: x has %d columns, the layer contracts over %d" % (x.shape[1], k))
    if x.stride(1) != 1:
        x = x.contiguous()
    ldx = x.stride(0) if n > 1 else x.shape[1]   # (a single row: its real width, never a fabricated stride)
    b = bias.detach().contiguous() if bias is not None else None
    if out is None and pool is None:
        out = torch.empty((n, nout), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    wsb = L.lidf_linear_workspace_bytes(k)
    ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=x.device)
    if n == 0:
        return out
    if addrows2 is not None:
        if addrows is None or pool is not None:
            raise RuntimeError("linear_hip: a second gathered term needs the first one and no pooling")
        with torch.cuda.device(x.device):
            _lib.check(L.lidf_linear_gather2_f32(
                _lib.ptr(x), ldx, n, k, w.data_ptr() + 4 * w_col0, max(w.stride(0), w.shape[1]) if nout > 1 else w.shape[1],
                _lib.ptr(b), nout, act, float(slope),
                _lib.ptr(addrows), _lib.ptr(addidx), addrows.stride(0),
                _lib.ptr(addrows2), _lib.ptr(addidx2), addrows2.stride(0),
                _lib.ptr(out), out.stride(0), _lib.ptr(ws), wsb, _lib.current_stream(x.device)))
        return out
    with torch.cuda.device(x.device):
        _lib.check(L.lidf_linear_f32(
            _lib.ptr(x), ldx, n, k, w.data_ptr() + 4 * w_col0, max(w.stride(0), w.shape[1]) if nout > 1 else w.shape[1],
            _lib.ptr(b), nout, act, float(slope),
            _lib.ptr(addrows), _lib.ptr(addidx), addrows.stride(0) if addrows is not None else 0,
            _lib.ptr(out), out.stride(0) if out is not None else 0,
            _lib.ptr(pool), _lib.ptr(poolidx), pool.stride(0) if pool is not None else 0,
            _lib.ptr(ws), wsb, _lib.current_stream(x.device)))
    return out


def _out_act(mod, y):
    # implicit_net.py:93-96 / :148-151 on the [n, out_dim] output
    if mod.use_sigmoid:
        return torch.sigmoid(y)
    return torch.max(torch.min(y, y * 0.01 + 0.99), y * 0.01)


def _decoder_from_layer1(mod, n, dev, layer1):
    """The decoder behind its first layer. layer1(act) -> [n, 4 gf]: W1[:, :inp_dim] x + b1, activated (act True: the
    IMNet's layer 1 in one launch) or not (the IEF's pass-independent part, to which every pass adds the 16
    offset-encoding columns)."""
    from .decoders import IEF
    l1, l2, l3, l4 = mod.linear_1, mod.linear_2, mod.linear_3, mod.linear_4
    if not isinstance(mod, IEF):
        h = layer1(True)
        h = linear_hip(h, l2.weight, l2.bias, act=1, slope=0.02)
        h = linear_hip(h, l3.weight, l3.bias, act=1, slope=0.02)
        return _out_act(mod, linear_hip(h, l4.weight, l4.bias))
    if l4.out_features != 1:
        raise RuntimeError("IEF feeds its output back through offset_enc = Linear(1, 16): out_dim must be 1")
    # layer 1 = W1[:, :d] x + b1 (the same in every pass) + W1[:, d:] enc(off)
    d = mod.inp_dim
    base = layer1(False)
    rows = torch.arange(n, dtype=torch.int32, device=dev)
    from .decoders import _init_offset_value
    off = torch.full((n, 1), _init_offset_value(mod), dtype=torch.float32, device=dev)
    for _ in range(int(mod.n_iter)):
        enc = linear_hip(off, mod.offset_enc.weight, mod.offset_enc.bias)
        h = linear_hip(enc, l1.weight, None, act=1, slope=0.02, addrows=base, addidx=rows, w_col0=d, k=16)
        h = linear_hip(h, l2.weight, l2.bias, act=1, slope=0.02)
        h = linear_hip(h, l3.weight, l3.bias, act=1, slope=0.02)
        off = off + linear_hip(h, l4.weight, l4.bias)
    return _out_act(mod, off)


def decoder_forward(mod, x):
    """IMNet.forward / IEF.forward (models/implicit_net.py:81-98 / :131-152) at any gf_dim / out_dim
    on [n, inp_dim] rows."""
    x = x.detach()
    n, d = x.shape
    if d != mod.inp_dim:
        raise RuntimeError("decoder inp_dim %d != input width %d" % (mod.inp_dim, d))
    l1 = mod.linear_1
    if chain_ok(mod):   # one launch: every input column is a per-row operand, the bias a one-row table
        return decoder_chain(mod, x, d, voxpart=_bias_row(mod).reshape(1, -1).contiguous())
    return _decoder_from_layer1(
        mod, n, x.device,
        lambda act: linear_hip(x, l1.weight, l1.bias, act=1 if act else 0, slope=0.02 if act else 0.0, k=d))


def pointnet_forward(mod, inp_feat, vox2point_idx, n_vox):
    """PointNet2Stage.forward (models/pointnet.py:22-38) at any input_channels / gf_dim /
    output_channels: torch_scatter's max becomes the scatter-max epilogue of the producing layer
    (values are post-ReLU, a voxel without points keeps 0)."""
    x = inp_feat.detach().contiguous()
    idx = vox2point_idx.detach().to(torch.int32).contiguous()
    dev = x.device
    half, outc = mod.point_lin2.out_features, mod.point_lin4.out_features
    if half % 32 or outc % 32:
        raise RuntimeError("PointNet2Stage: output_channels must be a multiple of 64 on this path "
                           "(the scatter-max epilogue raises whole 32-column tiles)")
    f1 = linear_hip(x, mod.point_lin1.weight, mod.point_lin1.bias, act=1)
    pool1 = torch.zeros((max(n_vox, 1), half), dtype=torch.float32, device=dev)
    f2 = linear_hip(f1, mod.point_lin2.weight, mod.point_lin2.bias, act=1, pool=pool1, poolidx=idx,
                    out=torch.empty((x.shape[0], half), dtype=torch.float32, device=dev))
    g1 = linear_hip(pool1[:n_vox], mod.vox_lin1.weight, mod.vox_lin1.bias, act=1)
    # point_lin3 on cat(g1[idx], f2): the voxel half once per voxel, gathered into the point half
    gpart = linear_hip(g1, mod.point_lin3.weight, mod.point_lin3.bias, k=half)
    f4 = linear_hip(f2, mod.point_lin3.weight, None, act=1, addrows=gpart, addidx=idx, w_col0=half, k=half)
    pool2 = torch.zeros((max(n_vox, 1), outc), dtype=torch.float32, device=dev)
    linear_hip(f4, mod.point_lin4.weight, mod.point_lin4.bias, act=1, pool=pool2, poolidx=idx)
    return linear_hip(pool2[:n_vox], mod.vox_lin2.weight, mod.vox_lin2.bias, act=1)


def roi_align_rays(feat_grid, ray_pix, ray_bid, roi_inp_bbox=8, roi_out_bbox=2):
    """The per-ray ROI feature (models/pipeline.py:374-391) at any channel count / output size:
    [R, C * roi_out_bbox^2] through lidf_roi_align_f32."""
    B, Cn, h, w = feat_grid.shape
    R = ray_pix.shape[0]
    out = torch.empty((R, Cn * roi_out_bbox * roi_out_bbox), dtype=torch.float32, device=feat_grid.device)
    fg = feat_grid.detach().contiguous()
    with torch.cuda.device(fg.device):
        _lib.check(_lib.lib().lidf_roi_align_f32(_lib.ptr(fg), B, Cn, h, w, _lib.ptr(ray_pix), _lib.ptr(ray_bid), R,
                                                 int(roi_inp_bbox), int(roi_out_bbox), _lib.ptr(out), out.shape[1],
                                                 _lib.current_stream(fg.device)))
    return out


QUERY_SLAB = 614400   # pairs per slab of the layer-by-layer query (rows [slab, D]: 0.95 GB at D = 385)
CHAIN_SLAB_FACTOR = 16


def query(ray_dir, ray_pix, ray_bid, pair_off, pair_ray, pair_vox, pair_t, feat_grid, vox_feat, prob_dec,
          offset_dec, multires, multires_views, roi_inp_bbox, roi_out_bbox, offset_range, part_size,
          vox_center, pos_rel, ray_flat, depth, want_rayfeat):
    """get_embedding + get_pred (models/pipeline.py:338-466) at widths other than the shipped ones:
    the decoder input rows are materialised as the reference does ([P, pnet_out + rgb_out * roi^2 +
    2E + Ed]) and the decoders run layer by layer. The pieces: lidf_roi_align_f32, lidf_embed_f32,
    lidf_pe_rows_f32, lidf_linear_f32 per layer, lidf_query_tail_f32 (pair positions, per-ray softmax /
    arg-max, select); gathers and the concat are torch indexing on the device."""
    from .decoders import get_embedder
    dev = ray_dir.device
    R, P = ray_dir.shape[0], pair_ray.shape[0]
    E = 3 + 6 * multires
    L = _lib.lib()
    roi = roi_align_rays(feat_grid, ray_pix, ray_bid, roi_inp_bbox, roi_out_bbox)
    edir = get_embedder(multires_views)[0](ray_dir.detach().contiguous())
    D = vox_feat.shape[1] + roi.shape[1] + 2 * E + edir.shape[1]
    if D != prob_dec.inp_dim or D != offset_dec.inp_dim:
        raise RuntimeError("decoder inp_dim must be %d for this configuration" % D)
    # The reference materialises cat(...) [P, D] and every activation [P, 4 gf] at once (7.6 GB at the
    # configs[1] shape for D = 385 alone). Here the pairs go through in slabs of QUERY_SLAB rows (the slab the
    # CPU baseline uses): rows, positional encodings and activations of one slab are live at a time — about
    # 1 GB at D = 385, gf_dim = 64 —, the per-pair outputs [P, 1] are the only full-length arrays.
    # lidf_query_tail_f32 reads both arrays as [P]: the reference's own use (pred_prob_end[:, 0] and a 1-wide
    # offset, pipeline.py:437-442). A wider head would be read interleaved, so it is refused here.
    if prob_dec.linear_4.out_features != 1 or offset_dec.linear_4.out_features != 1:
        raise RuntimeError("the query takes decoders with out_dim == 1 (got prob_dec %d, offset_dec %d)"
                           % (prob_dec.linear_4.out_features, offset_dec.linear_4.out_features))
    vf = vox_feat.detach().contiguous()
    pred_prob = torch.empty((P, prob_dec.linear_4.out_features), dtype=torch.float32, device=dev)
    pred_offset = torch.empty((P, offset_dec.linear_4.out_features), dtype=torch.float32, device=dev)
    # Layer 1 in its factorised form (the algebra of the fixed-width kernels): of the D
    # columns of a row only the 2E position-embedding columns depend on the pair — the voxel columns give one
    # [V, 4 gf] table per decoder (with the bias), the ROI and direction columns one [R, 4 gf] table, and per pair
    # layer 1 is W1[:, pair columns] PE(p) + the two gathered table rows (lidf_linear_gather2_f32): 2E = 102 of the
    # 385 columns are multiplied per pair and the [P, D] rows of the reference are never formed.
    Cv, Cr, Ed = vf.shape[1], roi.shape[1], edir.shape[1]
    ray_rows = torch.arange(R, dtype=torch.int32, device=dev)
    tables = []
    chained = chain_ok(prob_dec) and chain_ok(offset_dec)
    packs = ({}, {})   # the chain launches' packed weight streams, one pack per decoder and query
    for dec in (prob_dec, offset_dec):
        l1 = dec.linear_1
        # (the chain launch takes the IEF's constant W1[:, enc] benc inside the per-voxel table, as the fixed-width
        # kernels do; the layer-by-layer path adds the encoded offset explicitly in every pass)
        voxpart = linear_hip(vf, l1.weight, _bias_row(dec) if chained else l1.bias, k=Cv)
        rp = linear_hip(roi, l1.weight, None, w_col0=Cv, k=Cr)
        raypart = linear_hip(edir, l1.weight, None, w_col0=Cv + Cr + 2 * E, k=Ed, addrows=rp, addidx=ray_rows)
        tables.append((voxpart, raypart))
    # (the chain launch keeps nothing per pair but the 2E position-embedding columns — 408 B at L = 8: the whole
    # configs[1] frame is 2 GB. At gf 128 longer slabs pay — one pack and one ramp-up per decoder and frame: 97.8 ->
    # 99.2 Mpoints/s —, at gf 32 they do not: 588 -> 578, a 614,400-pair slab of embedding rows is 250 MB and is
    # read back twice out of the 256 MB last-level cache instead of HBM)
    slab = QUERY_SLAB * (CHAIN_SLAB_FACTOR if (chained and prob_dec.gf_dim >= 128) else 1)
    for p0 in range(0, P, slab):
        p1 = min(P, p0 + slab)
        n = p1 - p0
        pr_s, pv_s, pt_s = pair_ray[p0:p1], pair_vox[p0:p1], pair_t[p0:p1]
        # (16 floats of slack behind the rows: the chain launch reads whole 16-column groups of the last row)
        pe = torch.empty((n * 2 * E + 16,), dtype=torch.float32, device=dev)[:n * 2 * E].view(n, 2 * E)
        with torch.cuda.device(dev):
            _lib.check(L.lidf_pe_rows_f32(_lib.ptr(pr_s), _lib.ptr(pv_s), _lib.ptr(pt_s), _lib.ptr(ray_dir),
                                          _lib.ptr(vox_center), 1 if pos_rel else 0, multires, n, _lib.ptr(pe),
                                          _lib.current_stream(dev)))
        pr_i, pv_i = pr_s.contiguous(), pv_s.contiguous()
        for dec, (voxpart, raypart), dst, pk in ((prob_dec, tables[0], pred_prob, packs[0]),
                                                  (offset_dec, tables[1], pred_offset, packs[1])):
            w1 = dec.linear_1.weight
            if chained:   # the whole decoder in one launch, straight into its slab of the output
                decoder_chain(dec, pe, 2 * E, w1_col0=Cv + Cr, voxpart=voxpart, vox_idx=pv_i, raypart=raypart,
                              ray_idx=pr_i, out=dst[p0:p1], padded=True, packed=pk)
                continue

            def layer1(act, w1=w1, voxpart=voxpart, raypart=raypart):
                return linear_hip(pe, w1, None, act=1 if act else 0, slope=0.02 if act else 0.0, w_col0=Cv + Cr,
                                  k=2 * E, addrows=voxpart, addidx=pv_i, addrows2=raypart, addidx2=pr_i)
            dst[p0:p1] = _decoder_from_layer1(dec, n, dev, layer1)
        del pe
    f32 = dict(dtype=torch.float32, device=dev)
    pos, sm = torch.empty((P, 3), **f32), torch.empty((P,), **f32)
    mid, pred = torch.empty((R,), dtype=torch.int64, device=dev), torch.empty((R, 3), **f32)
    off1, prob1 = pred_offset.reshape(-1).contiguous(), pred_prob.reshape(-1).contiguous()
    with torch.cuda.device(dev):
        _lib.check(L.lidf_query_tail_f32(_lib.ptr(off1), _lib.ptr(prob1), _lib.ptr(pair_off), _lib.ptr(pair_ray),
                                         _lib.ptr(pair_t), _lib.ptr(ray_dir), R, P, float(offset_range[0]),
                                         float(offset_range[1]), float(part_size), None, _lib.ptr(pos), _lib.ptr(sm),
                                         _lib.ptr(mid), _lib.ptr(pred), _lib.current_stream(dev)))
    if depth is not None:   # pipeline.py:593-596: the queried pixels take the predicted z
        hw = depth.shape[1] * depth.shape[2]
        depth.view(-1)[ray_bid.long() * hw + ray_flat.long()] = pred[:, 2]
    out = {"pred_offset": pred_offset, "pred_prob_end": pred_prob, "pair_pred_pos": pos,
           "pred_prob_end_softmax": sm, "max_pair_id": mid, "pred_pos": pred, "workspace": None}
    if want_rayfeat:
        out["rayfeat"] = torch.cat((roi, edir), 1)
    return out


def refine(ray_dir, ray_pix, ray_bid, ray_flat, pred_pos, max_pair_id, pair_vox, voxel_bound, voxel_bid, rgb_img,
           feat_grid, valid_inp, valid_vox, pnet_model, offset_dec, forward_times, multires, multires_views,
           roi_inp_bbox, roi_out_bbox, offset_range, pos_rel, pnet_pos_rel, ray_rgb, pnet_select):
    """forward_times x RefineNet.get_pred_refine (models/pipeline.py:922-1030, eval flavour) at widths other
    than the shipped ones, step by step as the reference writes it: the end voxel of every ray
    (lidf_pcl_aabb_last_f32 over the arg-max pair's voxel), the PointNet over valid + predicted points,
    the decoder rows [R, pnet_out + rgb_out * roi^2 + E + Ed], the decoder, the offset along the ray. The
    modules run through their own forward (layer by layer where a width differs)."""
    from .decoders import get_embedder
    from .extensions import pcl_aabb
    dev = ray_dir.device
    R, P, V = ray_dir.shape[0], pair_vox.shape[0], voxel_bound.shape[0]
    hw = rgb_img.shape[2] * rgb_img.shape[3]
    if ray_rgb is None:
        ray_rgb = roi_align_rays(feat_grid, ray_pix, ray_bid, roi_inp_bbox, roi_out_bbox)
    edir = get_embedder(multires_views)[0](ray_dir.detach().contiguous())
    embed = get_embedder(multires)[0]
    miss_rgb = rgb_img.permute(0, 2, 3, 1).reshape(-1, 3)[ray_bid.long() * hw + ray_flat.long()]
    pv = torch.cat((pair_vox.long(), torch.zeros(1, dtype=torch.long, device=dev)))
    first = pv[max_pair_id.clamp(max=P)]                       # the dummy row's voxel is 0 (pipeline.py:933)
    sel = None
    if pnet_select is not None:
        sel = torch.nonzero(pnet_select.reshape(-1) != 0, as_tuple=False)[:, 0]
    cur = pred_pos.detach().contiguous()
    end_voxel = first.int()
    for _ in range(int(forward_times)):
        last = pcl_aabb.last_voxel(cur, voxel_bound, ray_bid, voxel_bid).long()     # -1: in no voxel
        end = torch.maximum(first, last)                       # (:939-944: scatter of the containing voxels)
        end_voxel = end.int()
        eb = voxel_bound[end]
        center = (eb[:, :3] + eb[:, 3:]) / 2.0
        pred_inp = torch.cat(((cur - center) if pnet_pos_rel else cur, miss_rgb), 1)
        if sel is not None:
            pn_inp, pn_vox = torch.cat((valid_inp, pred_inp[sel]), 0), torch.cat((valid_vox.long(), end[sel]), 0)
        else:
            pn_inp, pn_vox = torch.cat((valid_inp, pred_inp), 0), torch.cat((valid_vox.long(), end), 0)
        occ = pnet_model(pn_inp.contiguous(), pn_vox.int(), n_vox=V)
        enter = (cur - center) if pos_rel else cur
        rows = torch.cat((occ[end], ray_rgb, embed(enter.contiguous()), edir), 1)
        off = offset_dec(rows)
        cur = (cur + (off * (offset_range[1] - offset_range[0]) + offset_range[0]) * ray_dir).contiguous()
    return cur, end_voxel


class _LinearFn(torch.autograd.Function):
    """act(x W^T + b) through lidf_linear_f32 with a backward through the library: d x = (g * act') W
    (lidf_linear_f32 on W^T), d W = (g * act')^T x and d b = its column sums (lidf_wgrad_f32: fixed
    summation order for layers of >= 32 outputs and >= 4 inputs; the 1-wide output layer and offset_enc =
    Linear(1, 16) add their row slices with float atomics — correct to rounding, not bit-reproducible run
    to run, lidf_hip.h). act' is read off the output (leaky ReLU keeps the sign)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        out = linear_hip(x.detach(), weight, bias, act=act, slope=slope)
        ctx.cfg = (act, slope, bias is not None)
        ctx.save_for_backward(x, weight, out if act else None)   # (version counters catch in-place updates)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w, out = ctx.saved_tensors
        x, w = x.detach(), w.detach()
        act, slope, has_bias = ctx.cfg
        g = g.detach().contiguous().float()
        if act:
            g = g * torch.where(out > 0, torch.ones((), device=g.device), torch.full((), slope, device=g.device))
        n, nout, k = x.shape[0], w.shape[0], w.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = linear_hip(g, w.t().contiguous())
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            L = _lib.lib()
            dw = torch.zeros((nout, k), dtype=torch.float32, device=g.device)
            db = torch.zeros((nout,), dtype=torch.float32, device=g.device) if has_bias else None
            xc = x if x.stride(1) == 1 else x.contiguous()
            wsb = L.lidf_wgrad_workspace_bytes()
            ws = torch.empty((wsb,), dtype=torch.uint8, device=g.device)
            if n:
                with torch.cuda.device(g.device):
                    _lib.check(L.lidf_wgrad_f32(_lib.ptr(g), g.stride(0), nout, _lib.ptr(xc), xc.stride(0) if n > 1 else k,
                                                k, n, _lib.ptr(dw), k, _lib.ptr(db), _lib.ptr(ws), wsb,
                                                _lib.current_stream(g.device)))
        return dx, dw, db if has_bias else None, None, None


def _lin(x, layer, act=0, slope=0.0):
    return _LinearFn.apply(x, layer.weight, layer.bias, act, slope)


def decoder_forward_train(mod, x):
    """IMNet / IEF forward under autograd at any gf_dim / out_dim: the reference's own sequence of
    operations (implicit_net.py:81-98 / :131-152) with every nn.Linear as a _LinearFn; the concat, the
    running offset and the output activation are torch ops that autograd differentiates itself."""
    from .decoders import IEF, _init_offset_value
    if x.shape[1] != mod.inp_dim:
        raise RuntimeError("decoder inp_dim %d != input width %d" % (mod.inp_dim, x.shape[1]))

    def trunk(h):
        for layer in (mod.linear_1, mod.linear_2, mod.linear_3):
            h = _lin(h, layer, 1, 0.02)
        return _lin(h, mod.linear_4)
    if not isinstance(mod, IEF):
        return _out_act(mod, trunk(x))
    if mod.linear_4.out_features != 1:
        raise RuntimeError("IEF feeds its output back through offset_enc = Linear(1, 16): out_dim must be 1")
    off = torch.full((x.shape[0], 1), _init_offset_value(mod), dtype=torch.float32, device=x.device)
    for _ in range(int(mod.n_iter)):
        off = off + trunk(torch.cat((x, _lin(off, mod.offset_enc)), 1))
    return _out_act(mod, off)


def pointnet_forward_train(mod, inp_feat, vox2point_idx, n_vox):
    """PointNet2Stage.forward under autograd at any width: the reference's sequence (models/pointnet.py:
    22-38) with every nn.Linear as a _LinearFn. The scatter-max and the gather are device-side torch
    indexing (scatter_reduce 'amax' / index): their backward routes a voxel's gradient to its maximal
    point as torch_scatter does (exactly equal positive maxima — ties — share it instead)."""
    from .pointnet import _segment_max
    idx = vox2point_idx.long()
    f2 = _lin(_lin(inp_feat, mod.point_lin1, 1), mod.point_lin2, 1)
    g1 = _lin(_segment_max(f2, idx, n_vox), mod.vox_lin1, 1)
    f5 = _lin(_lin(torch.cat((g1[idx], f2), -1), mod.point_lin3, 1), mod.point_lin4, 1)
    return _lin(_segment_max(f5, idx, n_vox), mod.vox_lin2, 1)
