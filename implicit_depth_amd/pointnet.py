"""pointnet.py — host-side mirror of the reference's models/pointnet.py PointNet2Stage.

Same constructor signature, forward signature `forward(inp_feat, vox2point_idx)` and parameter
names (`point_lin1..4`, `vox_lin1..2`) as the reference (models/pointnet.py:7-38), so the
checkpoints' `pnet_model` / `pnet_model_refine` entries load unchanged
(trainers/train_lidf.py:74-75, train_refine.py:364-366). On CUDA f32 inputs without autograd the
forward runs liblidf_hip.so's lidf_pointnet_f32 (MFMA linear layers with the two scatter-max
poolings fused as atomic-max epilogues); with autograd it runs the library's training path
(lidf_pointnet_forward_train_f32 / lidf_pointnet_backward_f32); `forward_composite` is the same
function in torch ops, kept as the definition the gradient tests compare with; CPU tensors are
refused.
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


def _segment_max(x, idx, n):
    """torch_scatter.scatter(x, idx, dim=0, reduce='max') for x >= 0: rows without points are 0."""
    out = torch.zeros((n, x.shape[1]), dtype=x.dtype, device=x.device)
    return out.scatter_reduce(0, idx.view(-1, 1).expand_as(x), x, reduce="amax", include_self=True)


def pointnet_struct(mod, keep):
    def p(t):
        t = t.detach()
        if t.dtype != torch.float32:
            raise RuntimeError("lidf_hip: float32 parameters required")
        t = t.contiguous()
        keep.append(t)
        return t.data_ptr()

    s = _lib.LidfPointNet()
    s.w_p1, s.b_p1 = p(mod.point_lin1.weight), p(mod.point_lin1.bias)
    s.w_p2, s.b_p2 = p(mod.point_lin2.weight), p(mod.point_lin2.bias)
    s.w_v1, s.b_v1 = p(mod.vox_lin1.weight), p(mod.vox_lin1.bias)
    s.w_p3, s.b_p3 = p(mod.point_lin3.weight), p(mod.point_lin3.bias)
    s.w_p4, s.b_p4 = p(mod.point_lin4.weight), p(mod.point_lin4.bias)
    s.w_v2, s.b_v2 = p(mod.vox_lin2.weight), p(mod.vox_lin2.bias)
    return s


def packed_pointnet(mod, s, dev):
    """The module's weight streams, kept per module (_lib.PACK_CACHE) and re-validated on the device
    by every call (lidf_pointnet_pack_guarded_f32: fingerprint of the parameter buffers, pack
    kernels only when it changed — see query._packed_weights); sets s.packed and returns the blob
    (keep it alive for the call)."""
    L = _lib.lib()
    e = _lib.packed_entry(_lib.PACK_CACHE, mod, (str(dev),), L.lidf_pointnet_pack_bytes(), dev)
    frozen = mod in _lib.FROZEN
    if not (frozen and e.frozen_ready):
        s.packed = None
        with torch.cuda.device(dev):
            _lib.check(L.lidf_pointnet_pack_guarded_f32(C.byref(s), _lib.ptr(e.blob), e.blob.numel(),
                                                        _lib.ptr(e.guard), _lib.current_stream(dev)))
        e.frozen_ready = frozen
    s.packed = e.blob.data_ptr()
    return e.blob


def is_shipped(mod):
    """The widths the register-chained kernels are built for (every shipped config)."""
    return (mod.input_channels, mod.gf_dim, mod.point_lin4.out_features) == (6, 32, 128)


def check_pointnet(mod, what="this entry"):
    if not is_shipped(mod):
        raise RuntimeError("lidf_hip: %s is built for PointNet2Stage(input_channels=6, gf_dim=32, "
                           "output_channels=128) (every shipped config); inference at other widths runs "
                           "layer by layer (PointNet2Stage.forward without autograd)" % what)


_PN_ORDER = ("point_lin1", "point_lin2", "vox_lin1", "point_lin3", "point_lin4", "vox_lin2")
_PN_FIELDS = ("p1", "p2", "v1", "p3", "p4", "v2")


def _pn_struct_from(tensors, keep):
    """LidfPointNet from a list [w, b] x (point_lin1, point_lin2, vox_lin1, point_lin3, point_lin4,
    vox_lin2)."""
    s = _lib.LidfPointNet()
    for i, f in enumerate(_PN_FIELDS):
        for j, pre in enumerate(("w_", "b_")):
            t = tensors[2 * i + j].detach()
            if t.dtype != torch.float32:
                raise RuntimeError("lidf_hip: float32 parameters required")
            t = t.contiguous()
            keep.append(t)
            setattr(s, pre + f, t.data_ptr())
    return s


class _PointNetTrainFn(torch.autograd.Function):
    """PointNet2Stage.forward under autograd (models/pointnet.py:22-38): HIP forward that keeps the
    activations (lidf_pointnet_forward_train_f32) and HIP backward (lidf_pointnet_backward_f32)."""

    @staticmethod
    def forward(ctx, inp, idx, n_vox, *params):
        x = inp.detach().contiguous()
        n = x.shape[0]
        keep = []
        s = _pn_struct_from(params, keep)
        L = _lib.lib()
        f32 = dict(dtype=torch.float32, device=x.device)
        act = torch.empty((max(L.lidf_pointnet_train_act_floats(n, n_vox), 1),), **f32)
        wsb = L.lidf_pointnet_train_workspace_bytes(n, n_vox)
        ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=x.device)
        out = torch.empty((n_vox, 128), **f32)
        with torch.cuda.device(x.device):
            _lib.check(L.lidf_pointnet_forward_train_f32(C.byref(s), _lib.ptr(x), _lib.ptr(idx), n, n_vox,
                                                         _lib.ptr(out), _lib.ptr(act), _lib.ptr(ws), wsb,
                                                         _lib.current_stream(x.device)))
        ctx.n_vox, ctx.ws, ctx.wsb = n_vox, ws, wsb
        ctx.save_for_backward(x, idx, act, *params)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, idx, act = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        n, n_vox = x.shape[0], ctx.n_vox
        keep = []
        s = _pn_struct_from(params, keep)
        f32 = dict(dtype=torch.float32, device=x.device)
        g = g_out.detach().contiguous().float()
        grads = [torch.empty_like(p, **f32).contiguous() for p in params]
        gs = _lib.LidfPointNetGrads()
        for i, f in enumerate(_PN_FIELDS):
            setattr(gs, "w_" + f, grads[2 * i].data_ptr())
            setattr(gs, "b_" + f, grads[2 * i + 1].data_ptr())
        d_inp = torch.empty((n, 6), **f32) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().lidf_pointnet_backward_f32(
                C.byref(s), _lib.ptr(x), _lib.ptr(idx), n, n_vox, _lib.ptr(act), _lib.ptr(g),
                _lib.ptr(d_inp), C.byref(gs), _lib.ptr(ctx.ws), ctx.wsb, _lib.current_stream(x.device)))
        return (d_inp, None, None) + tuple(gr if ctx.needs_input_grad[3 + i] else None
                                           for i, gr in enumerate(grads))


class PointNet2Stage(nn.Module):
    def __init__(self, input_channels=6, output_channels=256, gf_dim=64):
        super(PointNet2Stage, self).__init__()
        self.input_channels = input_channels
        self.gf_dim = gf_dim
        half = output_channels // 2
        self.point_lin1 = nn.Linear(self.input_channels, self.gf_dim, bias=True)
        self.point_lin2 = nn.Linear(self.gf_dim, half, bias=True)
        self.vox_lin1 = nn.Linear(half, half, bias=True)
        self.point_lin3 = nn.Linear(output_channels, output_channels, bias=True)
        self.point_lin4 = nn.Linear(output_channels, output_channels, bias=True)
        self.vox_lin2 = nn.Linear(output_channels, output_channels, bias=True)

    def forward(self, inp_feat, vox2point_idx, n_vox=None):
        if n_vox is None:
            n_vox = int(vox2point_idx.max().item()) + 1 if vox2point_idx.numel() else 0
        if not inp_feat.is_cuda:
            from .decoders import _CPU_HINT, _cpu_composite_allowed
            if _cpu_composite_allowed():   # CPU tensors only, opt-in (decoders.py): the torch-op definition
                return self.forward_composite(inp_feat, vox2point_idx, n_vox)
            raise RuntimeError("PointNet2Stage.forward: CUDA tensor required (no CPU path; " + _CPU_HINT + ")")
        needs_grad = torch.is_grad_enabled() and (
            inp_feat.requires_grad or any(p.requires_grad for p in self.parameters()))
        if (inp_feat.dtype != torch.float32 or inp_feat.dim() != 2
                or inp_feat.shape[1] != self.input_channels):
            raise RuntimeError("inp_feat must be float32 [N,%d]" % self.input_channels)
        if tuple(vox2point_idx.shape) != (inp_feat.shape[0],) or vox2point_idx.device != inp_feat.device:
            raise RuntimeError("vox2point_idx must be [N] on the device of inp_feat")
        if not is_shipped(self):   # other widths: layer by layer (generic.py), inference only
            from . import generic
            if needs_grad:   # every layer its own autograd function
                return generic.pointnet_forward_train(self, inp_feat, vox2point_idx, int(n_vox))
            return generic.pointnet_forward(self, inp_feat, vox2point_idx, int(n_vox))
        if needs_grad:   # the library's training path: forward that keeps activations + backward
            params = [t for name in _PN_ORDER for t in (getattr(self, name).weight, getattr(self, name).bias)]
            return _PointNetTrainFn.apply(inp_feat, vox2point_idx.detach().to(torch.int32).contiguous(),
                                          int(n_vox), *params)
        x = inp_feat.detach()
        x = x.contiguous()
        idx = vox2point_idx.detach().to(torch.int32).contiguous()
        n = x.shape[0]
        out = torch.empty((n_vox, 128), dtype=torch.float32, device=x.device)
        keep = []
        s = pointnet_struct(self, keep)
        keep.append(packed_pointnet(self, s, x.device))
        L = _lib.lib()
        wsb = L.lidf_pointnet_workspace_bytes(n, n_vox)
        ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.lidf_pointnet_f32(C.byref(s), _lib.ptr(x), _lib.ptr(idx), n, n_vox,
                                           _lib.ptr(out), _lib.ptr(ws), wsb,
                                           _lib.current_stream(x.device)))
        return out

    def forward_composite(self, inp_feat, vox2point_idx, n_vox):
        """The same function in differentiable torch ops (the definition the tests compare with)."""
        idx = vox2point_idx.long()
        f2 = F.relu(self.point_lin2(F.relu(self.point_lin1(inp_feat))))
        g1 = F.relu(self.vox_lin1(_segment_max(f2, idx, n_vox)))
        f5 = F.relu(self.point_lin4(F.relu(self.point_lin3(torch.cat((g1[idx], f2), -1)))))
        return F.relu(self.vox_lin2(_segment_max(f5, idx, n_vox)))
