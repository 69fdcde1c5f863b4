#!/usr/bin/env python3
"""One launch of one wave-specialised tile (debugging): python tools/ws_one.py TILE KH KW CIN COUT H W NB [DBG]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gvcnn_tf_amd import _lib  # noqa: E402

lib = _lib.load()
t, kh, kw, cin, cout, h, w, nb = [int(v) for v in sys.argv[1:9]]
dbg = int(sys.argv[9]) if len(sys.argv) > 9 else 0
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 1       # TILE < 0: absolute tile configuration -TILE-1 (any kernel)
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
code, td = _lib.GV_BF16, torch.bfloat16
x = torch.randn(nb, h, w, cin, device=dev).to(td)
n = lib.gv_packed_filter_bytes(kh, kw, cin, cout, code, 0) // 4
wf = torch.randn(kh, kw, cin, cout, device=dev) * 0.05
wp = torch.empty(n, device=dev)
lib.gv_pack_filter_hwio(wf.data_ptr(), kh, kw, cin, cout, wp.data_ptr(), code, 0, st)
sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
y = torch.empty(nb, h, w, cout, device=dev, dtype=td)
ws0 = lib.gv_conv2d_num_tile_cfgs(-1) - 11
d = _lib.ConvDesc(nb, h, w, cin, cin, kh, kw, 1, kh // 2, kw // 2, h, w, cout, cout, 0, 0, 1, code, 0, (ws0 + t + 1) if t >= 0 else -t, 0, 0)
lib.gv_conv2d_set_debug(dbg)
for _ in range(reps):
    rc = lib.gv_conv2d_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, y.data_ptr(), None, None, None, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    lib.gv_conv2d_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, y.data_ptr(), None, None, None, st)
e1.record()
e1.synchronize()
us = e0.elapsed_time(e1) * 50
lib.gv_conv2d_set_debug(dbg & ~16388)
lib.gv_conv2d_fwd(C.byref(d), x.data_ptr(), wp.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, y.data_ptr(), None, None, None, st)
torch.cuda.synchronize()
ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wf.to(td).float().permute(3, 2, 0, 1), padding=(kh // 2, kw // 2)).permute(0, 2, 3, 1)
err = (torch.relu(ref) - y.float()).abs().max().item() if rc == 0 else -1
print("tile %d dbg %5d rc %d max err %.4f | %.1f us %.0f TF/s" % (t, dbg, rc, err, us, 2.0 * nb * h * w * cout * kh * kw * cin / us / 1e6), flush=True)
