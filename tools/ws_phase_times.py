#!/usr/bin/env python3
"""Where do the waves of the wave-specialised convolution (csrc/conv_ws.hip) spend their time?  Profiling build
    GV_PHASE_TIMES=1 python gvcnn-tf_amd/build.py
then on the GPU:  python tools/ws_phase_times.py [bf16|f16]
Per wave the kernel records s_memtime at kernel entry, [consumer] before / after the first barrier, at the end of the
k-loop, at the end of the epilogue; [loader] after the prologue's issue and at the end of its loop; plus the clocks it spent
inside (counted vmcnt wait +) s_barrier of the k-loop.  Printed per layer and tile: mean clocks per phase and role, the
k-loop's clocks per k-step against the MFMA pipe time of a k-step, and the share of the loop each role spends waiting."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GVCNN_HIP_LIB"] = os.path.join(ROOT, "gvcnn-tf_amd", "libgvcnn_hip_pt.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gvcnn_tf_amd import _lib  # noqa: E402

lib = _lib.load()
lib.gv_conv2d_set_phase_buffer.restype = None
lib.gv_conv2d_set_phase_buffer.argtypes = [C.c_void_p]
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
ty = sys.argv[1] if len(sys.argv) > 1 else "bf16"
code, td = {"bf16": (_lib.GV_BF16, torch.bfloat16), "f16": (_lib.GV_F16, torch.float16)}[ty]
NCFG = lib.gv_conv2d_num_tile_cfgs(-1)
WS0 = NCFG - 11
#            BM   BN  KT
TILES = [(256, 192, 32), (256, 128, 32), (512, 96, 32), (512, 64, 32), (256, 64, 32), (256, 192, 64), (256, 128, 64), (256, 64, 64), (512, 64, 64),
         (256, 96, 32), (256, 64, 32)]
NCONS = [8] * 9 + [4] * 2


def run(name, nb, h, w, cin, cout, kh, kw, cfgs):
    x = torch.randn(nb, h, w, cin, device=dev).to(td)
    n = lib.gv_packed_filter_bytes(kh, kw, cin, cout, code, 0) // 4
    wf = torch.randn(kh, kw, cin, cout, device=dev) * 0.05
    wp = torch.empty(n, device=dev)
    lib.gv_pack_filter_hwio(wf.data_ptr(), kh, kw, cin, cout, wp.data_ptr(), code, 0, st)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(nb, h, w, cout, device=dev, dtype=td)
    M = nb * h * w
    print("== %s: M=%d N=%d K=%d (%dx%d, cin %d)" % (name, M, cout, kh * kw * cin, kh, kw, cin))
    for c in cfgs:
        bm, bn, kt = TILES[c]
        if cin % kt:
            continue
        nwg = -(-M // bm) * -(-cout // bn)
        buf = torch.zeros(nwg * 16 * 8, dtype=torch.int64, device=dev)
        d = _lib.ConvDesc(nb, h, w, cin, cin, kh, kw, 1, kh // 2, kw // 2, h, w, cout, cout, 0, 0, 1, code, 0, WS0 + c + 1, 0, 0)
        args = (C.byref(d), x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, y.data_ptr(), None, None, None, st)
        lib.gv_conv2d_set_debug(int(os.environ.get('WS_DBG', '0')))
        rc = lib.gv_conv2d_fwd(*args)
        if rc != 0:
            print("   cfg %d %dx%d k%d: rc %d" % (c, bm, bn, kt, rc))
            continue
        for _ in range(3):
            lib.gv_conv2d_fwd(*args)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.gv_conv2d_fwd(*args)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 100
        lib.gv_conv2d_set_phase_buffer(buf.data_ptr())
        _lib.check(lib.gv_conv2d_fwd(*args), "conv (stamped)")
        torch.cuda.synchronize()
        lib.gv_conv2d_set_phase_buffer(None)
        t = buf.cpu().numpy().reshape(nwg, 16, 8).astype(np.float64)
        cons, load = t[:, :NCONS[c], :], t[:, NCONS[c]:NCONS[c] + (2 if NCONS[c] == 4 else 4), :]
        nk = (kh * kw) * (cin // kt)
        nw = NCONS[c] + (2 if NCONS[c] == 4 else 4)
        life = (t[:, :nw, 4].max(1) - t[:, :nw, 0].min(1)).mean()
        c_setup = (cons[:, :, 1] - cons[:, :, 0]).mean()
        c_first = (cons[:, :, 2] - cons[:, :, 1]).mean()
        c_loop = (cons[:, :, 3] - cons[:, :, 2]).mean()
        c_epi = (cons[:, :, 4] - cons[:, :, 3]).mean()
        c_wait = cons[:, :, 5].mean()
        l_pro = (load[:, :, 1] - load[:, :, 0]).mean()
        l_loop = (load[:, :, 2] - load[:, :, 1]).mean()
        l_wait = load[:, :, 5].mean()
        span = t[:, :nw, 4].max() - t[:, :nw, 0].min()
        # MFMA pipe clocks of one k-step per SIMD: two consumers x (TM*TN MFMAs x kt/16) x 32 clocks... in s_memtime ticks
        # (100 MHz) the comparison needs the shader clock; print the raw ticks and let the reader scale
        print("   cfg %2d %3dx%3d k%d: %6.1f us (%4.0f TF/s), %d WGs, nk %d, WG life %6.0f clk"
              % (c, bm, bn, kt, us, 2.0 * M * cout * kh * kw * cin / us / 1e6, nwg, nk, life))
        print("        cons: setup %5.0f first %5.0f loop %6.0f (%6.1f/k-step, barrier %4.1f%%) epi %5.0f | load: pro %5.0f loop %6.0f (wait %4.1f%%)"
              % (c_setup, c_first, c_loop, c_loop / max(nk - 1, 1), 100 * c_wait / max(c_loop, 1), c_epi, l_pro, l_loop, 100 * l_wait / max(l_loop, 1)))


if __name__ == "__main__":
    for tag, nb, s5, s6, s7 in (("c3", 384, 25, 12, 5), ("c5", 640, 35, 17, 8)):
        print("######## %s %s" % (tag, ty))
        run("Mixed_6e 1x7 192", nb, s6, s6, 192, 192, 1, 7, (0, 2, 5, 9))
        run("Mixed_6b 1x7 128", nb, s6, s6, 128, 128, 1, 7, (1, 6, 8))
        run("Mixed_5 3x3 64->96", nb, s5, s5, 64, 96, 3, 3, (2, 9))
        run("Mixed_6 siblings 1x1", nb, s6, s6, 768, 704, 1, 1, (0, 6))
