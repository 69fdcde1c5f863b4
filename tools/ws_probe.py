#!/usr/bin/env python3
"""Wave-specialised kernel (csrc/conv_ws.hip) against the best of every other 16-bit tile, per layer shape, warm repeats on
one box in one process: TFLOP/s with the product epilogue (dbg 0) and without any epilogue (dbg 4).
    python tools/ws_probe.py [bf16|f16] [c3|c5|both]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gvcnn_tf_amd import _lib  # noqa: E402

lib = _lib.load()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
ty = sys.argv[1] if len(sys.argv) > 1 else "bf16"
which = sys.argv[2] if len(sys.argv) > 2 else "both"
code, td = {"bf16": (_lib.GV_BF16, torch.bfloat16), "f16": (_lib.GV_F16, torch.float16)}[ty]
NWS = 11
WS_NAMES = ["256x192", "256x128", "512x96", "512x64", "256x64", "k64:256x192", "256x128", "256x64", "512x64", "2wg:256x96", "256x64"]


def probe(name, nb, h, w, cin, cout, kh, kw, dbgs=(0, 4), iters=20):
    x = torch.randn(nb, h, w, cin, device=dev).to(td)
    n = lib.gv_packed_filter_bytes(kh, kw, cin, cout, code, 0) // 4
    wf = torch.randn(kh, kw, cin, cout, device=dev) * 0.05
    wp = torch.empty(n, device=dev)
    lib.gv_pack_filter_hwio(wf.data_ptr(), kh, kw, cin, cout, wp.data_ptr(), code, 0, st)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(nb, h, w, cout, device=dev, dtype=td)
    d = _lib.ConvDesc(nb, h, w, cin, cin, kh, kw, 1, kh // 2, kw // 2, h, w, cout, cout, 0, 0, 1, code, 0, 0, 0, 0)
    fl = 2.0 * nb * h * w * cout * kh * kw * cin
    ncfg = lib.gv_conv2d_num_tile_cfgs(-1)
    print("%-28s M=%7d N=%4d K=%5d (%dx%d, cin %d)" % (name, nb * h * w, cout, kh * kw * cin, kh, kw, cin))
    for dbg in dbgs:
        lib.gv_conv2d_set_debug(dbg)
        res = []
        for t in range(ncfg):
            lib.gv_conv2d_set_tile_override(t)
            best = 0.0
            for _ in range(2):
                ms = C.c_float(0)
                rc = lib.gv_conv2d_time(C.byref(d), x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), iters,
                                        C.byref(ms), st)
                if rc == 0:
                    best = max(best, fl / ms.value / 1e9)
            res.append(best)
        lib.gv_conv2d_set_tile_override(-1)
        lib.gv_conv2d_set_debug(0)
        old = res[:ncfg - NWS]
        ws = res[ncfg - NWS:]
        bo = max(range(len(old)), key=lambda i: old[i])
        print("   dbg %d: best other %4.0f (cfg %2d, %.1f us) | ws: %s | ws/other %.2f"
              % (dbg, old[bo], bo, fl / old[bo] / 1e6 if old[bo] else 0, " ".join("%d:%.0f" % (i, r) for i, r in enumerate(ws) if r > 0),
                 max(ws) / old[bo] if old[bo] else 0), flush=True)


if __name__ == "__main__":
    for tag, nb, s5, s6, s7 in (("c3", 384, 25, 12, 5), ("c5", 640, 35, 17, 8)):
        if which not in (tag, "both"):
            continue
        print("==== %s (%d views) %s" % (tag, nb, ty))
        probe("Mixed_6b 1x7 128", nb, s6, s6, 128, 128, 1, 7)
        probe("Mixed_6c 7x1 160", nb, s6, s6, 160, 160, 7, 1)
        probe("Mixed_6c 1x7 160->192", nb, s6, s6, 160, 192, 1, 7)
        probe("Mixed_6e 1x7 192", nb, s6, s6, 192, 192, 1, 7)
        probe("Mixed_6e 7x1 192", nb, s6, s6, 192, 192, 7, 1)
        probe("Mixed_6 siblings 1x1", nb, s6, s6, 768, 704, 1, 1)
        probe("Mixed_5 3x3 64->96", nb, s5, s5, 64, 96, 3, 3)
        probe("Mixed_5 3x3 96->96", nb, s5, s5, 96, 96, 3, 3)
        probe("Mixed_5 siblings 1x1 288", nb, s5, s5, 288, 240, 1, 1)
        probe("Mixed_7 1x3 384", nb, s7, s7, 384, 384, 1, 3)
        probe("Mixed_7 3x3 448->384", nb, s7, s7, 448, 384, 3, 3)
        probe("Mixed_7 siblings 1x1 1280", nb, s7, s7, 1280, 1344, 1, 1)
