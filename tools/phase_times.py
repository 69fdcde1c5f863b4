#!/usr/bin/env python3
"""Where does a workgroup of the LDS-DMA convolution (csrc/conv_dma.hip) spend its life?  Needs the profiling build
    GV_PHASE_TIMES=1 python gvcnn-tf_amd/build.py        (libgvcnn_hip_pt.so: s_memtime stamps at the phase boundaries)
then, on the GPU,
    python tools/phase_times.py [bf16|f16]
Every wave records t0 (kernel entry), t1 (first k-tile in LDS), t2 (main loop done), t3 (epilogue's last store issued),
t4 (stores acknowledged: s_waitcnt vmcnt(0)) plus HW_ID / XCC_ID.  Printed per layer and tile configuration: the mean
length of each phase, the share of the workgroup's life, and — from the stamps of consecutive workgroups on the same CU —
the dispatch gap between one workgroup's end and its successor's start.  (s_memtime ticks at a constant 100 MHz.)"""
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
DMA0 = lib.gv_conv2d_special_tile_cfg(-1) + 1          # first LDS-DMA configuration
NAMES = ["128x128", "256x128/8", "128x256/8", "256x256/8", "128x192", "256x192/8", "128x64", "64x128", "128x96", "256x128/4",
         "128x256/4", "192x128", "192x128 s2", "128x128 s2", "128x192 s2", "256x128/4 s2"]
WAVES = [4, 8, 8, 8, 4, 8, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4]
BM = [128, 256, 128, 256, 128, 256, 128, 64, 128, 256, 128, 192, 192, 128, 128, 256]
BN = [128, 128, 256, 256, 192, 192, 64, 128, 96, 128, 256, 128, 128, 128, 192, 128]


def run(nb, h, w, cin, cout, kh, kw, cfgs, dbg=0):
    x = torch.randn(nb, h, w, cin, device=dev).to(td)
    n = lib.gv_packed_filter_bytes(kh, kw, cin, cout, code, 0) // 4
    wf = torch.randn(kh, kw, cin, cout, device=dev) * 0.05
    wp = torch.empty(n, device=dev)
    lib.gv_pack_filter_hwio(wf.data_ptr(), kh, kw, cin, cout, wp.data_ptr(), code, 0, st)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(nb, h, w, cout, device=dev, dtype=td)
    M = nb * h * w
    fl = 2.0 * M * cout * kh * kw * cin
    print("== M=%d N=%d K=%d (%dx%d, cin %d)%s" % (M, cout, kh * kw * cin, kh, kw, cin, " dbg %d" % dbg if dbg else ""))
    for c in cfgs:
        nwg = -(-M // BM[c]) * -(-cout // BN[c])
        buf = torch.zeros(nwg * WAVES[c] * 8, dtype=torch.int64, device=dev)
        d = _lib.ConvDesc(nb, h, w, cin, cin, kh, kw, 1, kh // 2, kw // 2, h, w, cout, cout, 0, 0, 1, code, 0, DMA0 + c + 1, 0, 0)
        lib.gv_conv2d_set_debug(dbg)
        args = (C.byref(d), x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, y.data_ptr(), None, None, None, st)
        for _ in range(3):
            _lib.check(lib.gv_conv2d_fwd(*args), "conv")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.gv_conv2d_fwd(*args)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        lib.gv_conv2d_set_phase_buffer(buf.data_ptr())
        _lib.check(lib.gv_conv2d_fwd(*args), "conv (stamped)")
        torch.cuda.synchronize()
        lib.gv_conv2d_set_phase_buffer(None)
        lib.gv_conv2d_set_debug(0)
        t = buf.cpu().numpy().reshape(nwg, WAVES[c], 8).astype(np.int64)
        ts = t[:, :, :5].astype(np.float64) * 0.01                    # us
        t0 = ts[:, :, 0].min(1)
        end = ts[:, :, 4].max(1)
        life = end - t0
        ph = [(ts[:, :, i + 1] - ts[:, :, i]).mean() for i in range(4)]
        setup, issue = (t[:, :, 7] >> 32).mean() * 0.01, (t[:, :, 7] & 0xffffffff).mean() * 0.01
        span = end.max() - t0.min()
        # dispatch gap: per (xcc, se, cu) slot order workgroups by start; gap = start - the end that freed the slot.
        hw = t[:, 0, 5]
        cu = ((hw >> 32) & 0xf) * 256 + ((hw >> 8) & 0xff)           # XCC_ID, then HW_ID's SE_ID | SH_ID | CU_ID bits
        per_tile = [((t[:, :, 6] >> sh) & 0xffff).mean() * 0.01 for sh in (48, 32, 16, 0)]
        gaps = []
        per_cu = []
        for k in np.unique(cu):
            idx = np.where(cu == k)[0]
            o = idx[np.argsort(t0[idx])]
            ends = []
            per_cu.append(len(o))
            for j in o:
                if len(ends) >= 2:                                     # the slot that freed last before this start
                    freed = max(e for e in ends if e <= t0[j] + 1e-9) if any(e <= t0[j] + 1e-9 for e in ends) else None
                    if freed is not None:
                        gaps.append(t0[j] - freed)
                ends.append(end[j])
        if os.environ.get("GV_PT_DIST"):                               # distributions: cold (first on their CU) vs later workgroups
            first = np.zeros(nwg, bool)
            for k in np.unique(cu):
                idx = np.where(cu == k)[0]
                first[idx[np.argsort(t0[idx])][:2]] = True
            iss = (t[:, :, 7] & 0xffffffff).mean(1) * 0.01
            stp = (t[:, :, 7] >> 32).mean(1) * 0.01
            epi = (ts[:, :, 3] - ts[:, :, 2]).mean(1)
            mainl = (ts[:, :, 2] - ts[:, :, 1]).mean(1)
            print("      issue per prologue tile: " + "  ".join("%.2f" % v for v in per_tile))
            for nm, v in (("setup", stp), ("issue", iss), ("main", mainl), ("epilogue", epi)):
                print("      %-9s first-on-CU mean %7.2f | later: mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f"
                      % (nm, v[first].mean(), v[~first].mean() if (~first).any() else float("nan"),
                         *(np.percentile(v[~first], q) if (~first).any() else float("nan") for q in (10, 50, 90))))
        print("  %-13s %7.1f us  %5.0f TF/s | wg %5d on %3d CUs (max %2d per CU) life %6.2f: prologue %5.2f (setup %5.2f, issue %5.2f)  main %6.2f  epilogue %5.2f  "
              "store drain %5.2f (x100 clocks) | gap after a slot frees: median %.2f"
              % (NAMES[c], ms * 1e3, fl / ms / 1e9, nwg, len(np.unique(cu)), max(per_cu), life.mean(), ph[0], setup, issue, ph[1], ph[2], ph[3],
                 float(np.median(gaps)) if gaps else float("nan")))


if __name__ == "__main__":
    if os.environ.get("GV_PT_DIST"):
        run(384, 52, 52, 80, 192, 3, 3, [4, 0])
        run(384, 12, 12, 768, 704, 1, 1, [11])
        sys.exit(0)
    run(384, 52, 52, 80, 192, 3, 3, [4, 11, 14, 0, 13])
    run(384, 12, 12, 160, 160, 7, 1, [4, 14, 11, 0])
    run(384, 12, 12, 768, 704, 1, 1, [3, 11, 12, 0])
    run(384, 25, 25, 64, 96, 3, 3, [0, 13, 8])
    run(384, 52, 52, 80, 192, 3, 3, [4, 14], dbg=4)
