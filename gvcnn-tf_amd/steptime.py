"""In-sequence launch timing of a training step: a proxy around the ctypes library that brackets every entry-point
call with a hipEvent pair on the launch stream while a real step runs, so each launch is timed where it actually
sits (cold caches, its true predecessor), not as a warm repeat of itself.

Used by bench.py --train (the `roofline` object of the training line) and tools/step_times.py (per-layer tables).
The event pairs add ~2 us of host work per launch; the step is GPU-bound (99 % busy), so the kernels still run
back to back and a pair's elapsed time is the launch's duration plus at most the gap in front of it.
"""
import ctypes as C

import torch

# entry point -> family of the training step
FAMILY = {
    "gv_conv2d_fwd": "conv", "gv_conv2d_fwd_xpre": "conv", "gv_conv2d_fwd_bnstats": "conv", "gv_conv2d_wgrad": "wgrad",
    "gv_conv2d_wgrad_ws": "wgrad",
    "gv_conv2d_dgrad_s2": "conv",
    "gv_bn_sums_grouped_t": "bn", "gv_bn_finalize_apply_grouped_t": "bn", "gv_bn_relu_bwd_sums_grouped_t": "bn",
    "gv_bn_relu_bwd_apply_grouped_t": "bn", "gv_bn_finalize_t": "bn", "gv_bn_bwd_finalize_t": "bn",
    "gv_pool2d_fwd": "pool", "gv_pool2d_fwd_argmax": "pool", "gv_pool2d_bwd": "pool", "gv_pool2d_bwd_argmax": "pool",
    "gv_bn_bwd_coeffs_t": "bn", "gv_pool2d_bwd_argmax_bn": "bn",
    "gv_accumulate_t": "elementwise", "gv_bias_grad_t": "elementwise",
}


class TimedLib:
    """Wraps the loaded library: every gv_* call made through it is timed with an event pair on the current stream.
    records: list of dicts {fn, tag, phase, e0, e1}; `tag` / `phase` are whatever the driver set before the call."""

    def __init__(self, lib, device):
        self._lib = lib
        self._device = device
        self.records = []
        self.tag = None
        self.phase = None
        self.enabled = True

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith("gv_") or fn.restype is not C.c_int:
            return fn

        def call(*args):
            if not self.enabled:
                return fn(*args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s = torch.cuda.current_stream(self._device)
            e0.record(s)
            rc = fn(*args)
            e1.record(s)
            self.records.append(dict(fn=name, tag=self.tag, phase=self.phase, e0=e0, e1=e1))
            return rc
        return call

    def collect(self):
        """-> [(fn, tag, phase, ms)] in launch order (synchronises)."""
        torch.cuda.synchronize(self._device)
        out = [(r["fn"], r["tag"], r["phase"], r["e0"].elapsed_time(r["e1"])) for r in self.records]
        self.records = []
        return out


def timed_step(eng, views, labels, steps=1):
    """Run `steps` whole training steps (forward, loss, backward; no update) with every launch timed in sequence.
    -> list of per-step lists [(fn, op name, phase, kind, ms)]."""
    lib0 = eng.lib
    tl = TimedLib(lib0, eng.device)
    fwd0, bwd0 = eng._forward_op, eng._backward_op

    def fwd(op, zeroed=False, part="all"):
        tl.tag, tl.phase = (op["name"], op["kind"]), "fwd"
        return fwd0(op, zeroed, part)

    def bwd(op, zeroed=False, part="all"):
        tl.tag, tl.phase = (op["name"], op["kind"]), "bwd"
        return bwd0(op, zeroed, part)

    if eng._packed_dirty:
        eng.repack()                                      # (not part of a step's launch list)
    eng.lib, eng._forward_op, eng._backward_op = tl, fwd, bwd
    out = []
    try:
        for _ in range(steps):
            tl.tag, tl.phase = ("head", "head"), "fwd"
            eng.forward(views, labels, check=False)
            tl.tag, tl.phase = ("head", "head"), "bwd"
            eng.backward()
            out.append([(fn, tag[0], ph, tag[1], ms) for fn, tag, ph, ms in tl.collect()])
    finally:
        eng.lib, eng._forward_op, eng._backward_op = lib0, fwd0, bwd0
    return out


def by_family(step):
    """Sum one step's records: {family: (ms, launches)}; conv data gradients (gv_conv2d_fwd in the backward phase)
    are their own family."""
    tot = {}
    for fn, name, phase, kind, ms in step:
        fam = FAMILY.get(fn, "head" if kind == "head" else "other")
        if fam == "conv" and phase == "bwd":
            fam = "dgrad"
        t, n = tot.get(fam, (0.0, 0))
        tot[fam] = (t + ms, n + 1)
    return tot
