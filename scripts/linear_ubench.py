"""lidf_linear_f32 alone at the shape of the rows backward's input gradient (n x 256 -> nout), over row strides of the
operand / the output: is the kernel bound by the operand's access pattern (rows a power of two apart) or by its
matrix instructions?  usage (GPU box): python scripts/linear_ubench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from implicit_depth_amd import generic

dev = torch.device("cuda:0")
n = 614400
torch.manual_seed(0)


def run(k, nout, ldx, ldo, reps=10, bias=False):
    xb = torch.randn((n, ldx), device=dev)
    x = xb[:, :k]
    w = torch.randn((nout, k), device=dev) * 0.05
    b = torch.randn((nout,), device=dev) if bias else None
    ob = torch.empty((n, ldo), device=dev)
    o = ob[:, :nout]
    for _ in range(3):
        generic.linear_hip(x, w, b, out=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        generic.linear_hip(x, w, b, out=o)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tiles = (nout + 31) // 32
    kq = (k + 2 + 7) // 8
    tf = 2.0 * n * tiles * 32 * kq * 8 / ms / 1e9
    print("k %4d nout %4d ldx %4d ldo %4d : %.3f ms  %.1f TFLOP/s issued (%.3f of 157.3)" % (k, nout, ldx, ldo, ms, tf, tf / 157.3))


# (clocks: the first configurations of a cold process run 10-20 % slower than the same ones later — warm up first,
# and the 256-float stride is measured again at the end)
run(256, 256, 256, 256, reps=60)
print("---- warm")
for k, nout, ldx, ldo in ((256, 224, 256, 224), (256, 224, 260, 224), (256, 224, 264, 224), (256, 224, 288, 224),
                          (256, 224, 256, 385), (256, 224, 260, 388), (256, 256, 256, 256), (256, 256, 260, 256),
                          (256, 128, 256, 128), (256, 128, 260, 128), (256, 64, 256, 64), (256, 64, 260, 64),
                          (512, 256, 512, 256), (512, 256, 516, 256), (385, 256, 385, 256), (385, 256, 388, 256),
                          (128, 256, 128, 256), (128, 256, 132, 256), (256, 224, 256, 224), (256, 224, 288, 224)):
    run(k, nout, ldx, ldo)
