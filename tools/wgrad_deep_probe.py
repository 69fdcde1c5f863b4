import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
import torch
from gvcnn_tf_amd import _lib
lib = _lib.load(); dev = "cuda:0"; st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
def probe(name, nb, ih, iw, cin, cout, pad, k=3):
    oh, ow = ih + 2 * pad - (k - 1), iw + 2 * pad - (k - 1)
    x = torch.randn(nb, ih, iw, cin, device=dev).bfloat16(); dz = torch.randn(nb, oh, ow, cout, device=dev).bfloat16()
    dw = torch.zeros(k, k, cin, cout, device=dev)
    n = lib.gv_conv2d_wgrad_num_cfgs(_lib.GV_BF16)
    res = []
    for cfg in range(1, n + 1):
        d = _lib.ConvDesc(nb, ih, iw, cin, cin, k, k, 1, pad, pad, oh, ow, cout, cout, 0, 0, 0, _lib.GV_BF16, 0, cfg, 0, 0)
        call = lambda: lib.gv_conv2d_wgrad_ws(C.byref(d), x.data_ptr(), dz.data_ptr(), cout, dw.data_ptr(), ws.data_ptr(), ws.numel(), st)
        if call() != 0: continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call()
        e1.record(); e1.synchronize()
        res.append((e0.elapsed_time(e1) / 5, cfg))
    res.sort()
    fl = 2.0 * nb * oh * ow * cout * k * k * cin
    deep = [(t, c) for t, c in res if c >= 92]
    other = [(t, c) for t, c in res if c < 92]
    print("%s: best other cfg %d %.4f ms (%.0f TF/s) | deep strips: %s" % (name, other[0][1], other[0][0], fl / other[0][0] / 1e9,
          ", ".join("cfg %d %.4f ms (%.0f TF/s)" % (c, t, fl / t / 1e9) for t, c in sorted(deep, key=lambda v: v[1]))), flush=True)
probe("Conv2d_2a (c3 train)", 384, 111, 111, 32, 32, 0)
probe("Conv2d_2b (c3 train)", 384, 109, 109, 32, 64, 1)
probe("Conv2d_4a (c3 train)", 384, 54, 54, 80, 192, 0)
probe("Mixed_5b 3x3 64->96", 384, 25, 25, 64, 96, 1)
probe("Mixed_5b 3x3 96->96", 384, 25, 25, 96, 96, 1)
probe("Mixed_5b 5x5 48->64", 384, 25, 25, 48, 64, 2, k=5)
