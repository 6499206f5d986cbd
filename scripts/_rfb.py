import torch, time
from implicit_depth_amd import _lib
import implicit_depth_amd.query as Q
dev = torch.device("cuda:0")
B, h, w = 1, 240, 320
ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
pix_all = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1).int()
inner = (pix_all[:, 0] >= 4) & (pix_all[:, 0] <= w - 5) & (pix_all[:, 1] >= 4) & (pix_all[:, 1] <= h - 5)
for name, pix in (("all", pix_all), ("interior", pix_all[inner]), ("border", pix_all[~inner])):
    pix = pix.contiguous().to(dev)
    R = pix.shape[0]
    bid = torch.zeros(R, dtype=torch.int32, device=dev)
    g = torch.randn(R, 128 + 27, device=dev)
    d_feat = torch.empty(B, 32, h, w, device=dev)
    wsb = B * 129 * h * w * 4 + (R + 1) * 4
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    L = _lib.lib()
    def run():
        _lib.check(L.lidf_ray_features_backward_f32(_lib.ptr(g), _lib.ptr(pix), _lib.ptr(bid), R, B, h, w, 8, 4,
                                                    _lib.ptr(d_feat), _lib.ptr(ws), wsb, _lib.current_stream(dev)))
    for _ in range(3): run()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20): run()
    torch.cuda.synchronize()
    print(name, R, "%.1f us" % ((time.perf_counter() - t) / 20 * 1e6))
