// plan.hip — native launch sequence for the per-view backbone ("plan") + ABI housekeeping.
//
// The reference builds V unrolled copies of the backbone graph, one per view at batch N
// (nets/model.py:129-141), and lets the TF executor schedule them.  Here the whole view batch
// is folded into one image axis and the backbone is an ordered list of kernel launches held in
// native code: building it is host-only work done once; running it only enqueues kernels on
// the caller's stream (no allocation, no synchronisation), so a run can be captured into a
// hipGraph by the caller.
#include <new>
#include <vector>

#include "conv_common.h"
#include "gv_common.h"

namespace {

enum OpKind { OP_CONV = 0, OP_POOL = 1, OP_SSA = 2, OP_CHAIN = 3, OP_UNIT = 4 };

struct Ref {
    int32_t slot;
    int64_t off;   // element offset (fp32 elements for scale/shift, `dtype` elements otherwise)
};

struct Op {
    OpKind kind;
    gv_conv_desc conv;
    gv_pool_desc pool;
    gv_chain_desc chain;        // OP_CHAIN: x, w (conv3), scale / shift (conv3), res, y as for a conv; z = y2; the pre-activation's
    Ref w1, scale1, shift1;     //   scale / shift in scale2 / shift2; the next conv1's filter / scale / shift here
    gv_unit_desc unit;          // OP_UNIT: the chain's operands + conv2's filter / scale / shift
    Ref w0, scale0, shift0;
    // scale_shift_act
    int64_t npix;
    int32_t c, x_ld, y_ld, relu, dtype;
    Ref x, w, scale, shift, res, y, y2, scale2, shift2;
    Ref xscale{-1, 0}, xshift{-1, 0};      // gv_plan_set_conv_xpre: pre-activation applied by this conv's loader
    // schedule: launch lane (0 = the caller's stream) and the earlier ops on OTHER lanes this op
    // must wait for (data or buffer-reuse hazards); same-lane order is stream order.
    int32_t lane = 0;
    std::vector<int32_t> deps;
    bool signal = false;        // some later op on another lane waits for this one
};

inline size_t elem_size(int dtype) { return dtype == GV_F32 ? 4 : 2; }

}  // namespace

struct gv_plan {
    std::vector<Op> ops;
    int32_t max_slot = -1;
    int32_t num_lanes = 1;
    // lazily created on the first multi-lane run (the plan is then not re-entrant across streams)
    std::vector<hipStream_t> lane_streams;      // [0] unused (caller's stream)
    std::vector<hipEvent_t> op_events;          // one per op with `signal`
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    ~gv_plan() {
        for (size_t i = 1; i < lane_streams.size(); ++i) if (lane_streams[i]) (void)hipStreamDestroy(lane_streams[i]);
        for (hipEvent_t e : op_events) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_join) if (e) (void)hipEventDestroy(e);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
    }
};

namespace {

inline void note(gv_plan* p, const Ref& r) {
    if (r.slot > p->max_slot) p->max_slot = r.slot;
}

inline void* at(void* const* bufs, const Ref& r, size_t esz) {
    if (r.slot < 0) return nullptr;
    return (void*)((char*)bufs[r.slot] + (size_t)r.off * esz);
}

int run_op(const Op& o, void* const* bufs, void* stream) {
    switch (o.kind) {
        case OP_CONV: {
            const size_t es = elem_size(o.conv.dtype);
            // fp32 network input; three-plane operands are 6 bytes per value
            const size_t xes = (o.conv.flags & GV_CONV_X_F32) ? 4 : ((o.conv.flags & GV_CONV_X_P3) ? 6 : es);
            const size_t yes = (o.conv.flags & GV_CONV_Y_P3) ? 6 : es;
            const size_t y2es = (o.conv.flags & GV_CONV_Y2_P3) ? 6 : es;
            if (o.xscale.slot >= 0)
                return gv_conv2d_fwd_xpre(&o.conv, at(bufs, o.x, xes), (const float*)at(bufs, o.xscale, 4),
                                          (const float*)at(bufs, o.xshift, 4), at(bufs, o.w, es),
                                          (const float*)at(bufs, o.scale, 4), (const float*)at(bufs, o.shift, 4),
                                          at(bufs, o.res, es), at(bufs, o.y, yes), at(bufs, o.y2, y2es),
                                          (const float*)at(bufs, o.scale2, 4), (const float*)at(bufs, o.shift2, 4), stream);
            return gv_conv2d_fwd(&o.conv, at(bufs, o.x, xes), at(bufs, o.w, es),
                                 (const float*)at(bufs, o.scale, 4), (const float*)at(bufs, o.shift, 4),
                                 at(bufs, o.res, es), at(bufs, o.y, yes), at(bufs, o.y2, y2es),
                                 (const float*)at(bufs, o.scale2, 4),
                                 (const float*)at(bufs, o.shift2, 4), stream);
        }
        case OP_POOL: {
            const size_t es = elem_size(o.pool.dtype);
            const size_t xes = (o.pool.mode & GV_POOL_X_P3) ? 6 : es;
            const size_t yes = (o.pool.mode & GV_POOL_Y_P3) ? 6 : es;
            return gv_pool2d_fwd(&o.pool, at(bufs, o.x, xes), at(bufs, o.y, yes), stream);
        }
        case OP_CHAIN: {
            const size_t es = elem_size(o.chain.dtype);
            return gv_bottleneck_chain_fwd(&o.chain, at(bufs, o.x, es), at(bufs, o.w, es), (const float*)at(bufs, o.scale, 4),
                                           (const float*)at(bufs, o.shift, 4), at(bufs, o.res, es), at(bufs, o.y, es),
                                           (const float*)at(bufs, o.scale2, 4), (const float*)at(bufs, o.shift2, 4),
                                           at(bufs, o.w1, es), (const float*)at(bufs, o.scale1, 4),
                                           (const float*)at(bufs, o.shift1, 4), at(bufs, o.y2, es), stream);
        }
        case OP_UNIT: {
            const size_t es = elem_size(o.unit.dtype);
            return gv_bottleneck_unit_fwd(&o.unit, at(bufs, o.x, es), at(bufs, o.w0, es), (const float*)at(bufs, o.scale0, 4),
                                          (const float*)at(bufs, o.shift0, 4), at(bufs, o.w, es), (const float*)at(bufs, o.scale, 4),
                                          (const float*)at(bufs, o.shift, 4), at(bufs, o.res, es), at(bufs, o.y, es),
                                          (const float*)at(bufs, o.scale2, 4), (const float*)at(bufs, o.shift2, 4),
                                          at(bufs, o.w1, es), (const float*)at(bufs, o.scale1, 4),
                                          (const float*)at(bufs, o.shift1, 4), at(bufs, o.y2, es), stream);
        }
        case OP_SSA: {
            const size_t es = elem_size(o.dtype);
            return gv_scale_shift_act(at(bufs, o.x, es), o.npix, o.c, o.x_ld,
                                      (const float*)at(bufs, o.scale, 4),
                                      (const float*)at(bufs, o.shift, 4), o.relu, at(bufs, o.y, es),
                                      o.y_ld, o.dtype, stream);
        }
    }
    return GV_E_PLAN;
}

}  // namespace

extern "C" int gv_abi_version(void) { return GV_ABI_VERSION; }

extern "C" const char* gv_error_string(int code) {
    switch (code) {
        case GV_OK: return "ok";
        case GV_E_BADARG: return "gvcnn: bad argument (null pointer, non-positive size or inconsistent descriptor)";
        case GV_E_UNSUPPORTED: return "gvcnn: unsupported request (dtype or size cap)";
        case GV_E_ALIGN: return "gvcnn: pointer or stride not aligned for the vector path";
        case GV_E_PLAN: return "gvcnn: plan misuse (bad slot index or null plan)";
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "gvcnn: unknown error code";
}

extern "C" int gv_plan_create(gv_plan** out) {
    if (!out) return GV_E_BADARG;
    *out = new (std::nothrow) gv_plan();
    return *out ? GV_OK : GV_E_PLAN;
}

extern "C" void gv_plan_destroy(gv_plan* p) { delete p; }

extern "C" int gv_plan_num_ops(const gv_plan* p) { return p ? (int)p->ops.size() : GV_E_PLAN; }

extern "C" int gv_plan_add_conv(gv_plan* p, const gv_conv_desc* d, int32_t x_slot, int64_t x_off,
                                int32_t w_slot, int64_t w_off, int32_t ss_slot, int64_t scale_off,
                                int64_t shift_off, int32_t res_slot, int64_t res_off, int32_t y_slot,
                                int64_t y_off, int32_t y2_slot, int64_t y2_off, int64_t scale2_off,
                                int64_t shift2_off) {
    if (!p) return GV_E_PLAN;
    if (!d || x_slot < 0 || w_slot < 0 || ss_slot < 0 || y_slot < 0) return GV_E_BADARG;
    Op o{};
    o.kind = OP_CONV;
    o.conv = *d;
    o.x = {x_slot, x_off};
    o.w = {w_slot, w_off};
    o.scale = {ss_slot, scale_off};
    o.shift = {ss_slot, shift_off};
    o.res = {res_slot, res_off};
    o.y = {y_slot, y_off};
    o.y2 = {y2_slot, y2_off};
    const bool second = y2_slot >= 0 || (d->flags & GV_CONV_POOL_ACT2) != 0;   // (a second activation of the pooled tensor too)
    o.scale2 = {second ? ss_slot : -1, scale2_off};
    o.shift2 = {second ? ss_slot : -1, shift2_off};
    for (const Ref* r : {&o.x, &o.w, &o.scale, &o.res, &o.y, &o.y2}) note(p, *r);
    p->ops.push_back(o);
    return GV_OK;
}

extern "C" int gv_plan_add_chain(gv_plan* p, const gv_chain_desc* d, int32_t x_slot, int64_t x_off, int32_t w_slot,
                                 int64_t w3_off, int64_t w1_off, int32_t ss_slot, int64_t scale3_off, int64_t shift3_off,
                                 int64_t pre_scale_off, int64_t pre_shift_off, int64_t scale1_off, int64_t shift1_off,
                                 int32_t res_slot, int64_t res_off, int32_t y_slot, int64_t y_off, int32_t z_slot,
                                 int64_t z_off) {
    if (!p) return GV_E_PLAN;
    const bool proj = d && (d->flags & GV_CHAIN_PROJ) != 0;   // (no shortcut operand: res_slot = -1)
    if (!d || x_slot < 0 || w_slot < 0 || ss_slot < 0 || (res_slot < 0) != proj || y_slot < 0 || z_slot < 0) return GV_E_BADARG;
    Op o{};
    o.kind = OP_CHAIN;
    o.chain = *d;
    o.x = {x_slot, x_off};
    o.w = {w_slot, w3_off};
    o.w1 = {w_slot, w1_off};
    o.scale = {ss_slot, scale3_off};
    o.shift = {ss_slot, shift3_off};
    o.scale2 = {ss_slot, pre_scale_off};
    o.shift2 = {ss_slot, pre_shift_off};
    o.scale1 = {ss_slot, scale1_off};
    o.shift1 = {ss_slot, shift1_off};
    o.res = {res_slot, res_off};
    o.y = {y_slot, y_off};
    o.y2 = {z_slot, z_off};
    for (const Ref* r : {&o.x, &o.w, &o.scale, &o.res, &o.y, &o.y2}) note(p, *r);
    p->ops.push_back(o);
    return GV_OK;
}

extern "C" int gv_plan_add_unit(gv_plan* p, const gv_unit_desc* d, int32_t x_slot, int64_t x_off, int32_t w_slot, int64_t w2_off,
                                int64_t w3_off, int64_t w1_off, int32_t ss_slot, int64_t scale2_off, int64_t shift2_off,
                                int64_t scale3_off, int64_t shift3_off, int64_t pre_scale_off, int64_t pre_shift_off,
                                int64_t scale1_off, int64_t shift1_off, int32_t res_slot, int64_t res_off, int32_t y_slot,
                                int64_t y_off, int32_t z_slot, int64_t z_off) {
    if (!p) return GV_E_PLAN;
    if (!d || x_slot < 0 || w_slot < 0 || ss_slot < 0 || res_slot < 0 || y_slot < 0 || z_slot < 0) return GV_E_BADARG;
    Op o{};
    o.kind = OP_UNIT;
    o.unit = *d;
    o.x = {x_slot, x_off};
    o.w0 = {w_slot, w2_off};
    o.w = {w_slot, w3_off};
    o.w1 = {w_slot, w1_off};
    o.scale0 = {ss_slot, scale2_off};
    o.shift0 = {ss_slot, shift2_off};
    o.scale = {ss_slot, scale3_off};
    o.shift = {ss_slot, shift3_off};
    o.scale2 = {ss_slot, pre_scale_off};
    o.shift2 = {ss_slot, pre_shift_off};
    o.scale1 = {ss_slot, scale1_off};
    o.shift1 = {ss_slot, shift1_off};
    o.res = {res_slot, res_off};
    o.y = {y_slot, y_off};
    o.y2 = {z_slot, z_off};
    for (const Ref* r : {&o.x, &o.w, &o.scale, &o.res, &o.y, &o.y2}) note(p, *r);
    p->ops.push_back(o);
    return GV_OK;
}

extern "C" int gv_plan_set_conv_tile(gv_plan* p, int32_t op_index, int32_t tile_cfg) {
    if (!p) return GV_E_PLAN;
    if (op_index < 0 || (size_t)op_index >= p->ops.size() || tile_cfg < 0) return GV_E_BADARG;
    Op& o = p->ops[(size_t)op_index];
    if (o.kind != OP_CONV) return GV_E_BADARG;
    o.conv.tile_cfg = tile_cfg;
    return GV_OK;
}

extern "C" int gv_plan_set_conv_xpre(gv_plan* p, int32_t op_index, int64_t xscale_off, int64_t xshift_off) {
    if (!p) return GV_E_PLAN;
    if (op_index < 0 || (size_t)op_index >= p->ops.size() || xscale_off < 0 || xshift_off < 0) return GV_E_BADARG;
    Op& o = p->ops[(size_t)op_index];
    if (o.kind != OP_CONV) return GV_E_BADARG;
    o.xscale = {o.scale.slot, xscale_off};            // the table that holds this op's own scale / shift
    o.xshift = {o.scale.slot, xshift_off};
    return GV_OK;
}

extern "C" int gv_plan_set_schedule(gv_plan* p, int32_t op_index, int32_t lane, const int32_t* deps,
                                    int32_t num_deps) {
    if (!p) return GV_E_PLAN;
    if (op_index < 0 || (size_t)op_index >= p->ops.size() || lane < 0 || lane >= 8 || num_deps < 0 ||
        (num_deps > 0 && !deps))
        return GV_E_BADARG;
    if (!p->lane_streams.empty()) return GV_E_PLAN;          // schedule is frozen after the first run
    Op& o = p->ops[(size_t)op_index];
    o.lane = lane;
    o.deps.clear();
    for (int32_t i = 0; i < num_deps; ++i) {
        if (deps[i] < 0 || deps[i] >= op_index) return GV_E_BADARG;   // only earlier ops
        o.deps.push_back(deps[i]);
    }
    if (lane + 1 > p->num_lanes) p->num_lanes = lane + 1;
    return GV_OK;
}

extern "C" int gv_plan_add_pool(gv_plan* p, const gv_pool_desc* d, int32_t x_slot, int64_t x_off,
                                int32_t y_slot, int64_t y_off) {
    if (!p) return GV_E_PLAN;
    if (!d || x_slot < 0 || y_slot < 0) return GV_E_BADARG;
    Op o{};
    o.kind = OP_POOL;
    o.pool = *d;
    o.x = {x_slot, x_off};
    o.y = {y_slot, y_off};
    o.w = o.scale = o.shift = o.res = o.y2 = o.scale2 = o.shift2 = {-1, 0};
    note(p, o.x);
    note(p, o.y);
    p->ops.push_back(o);
    return GV_OK;
}

extern "C" int gv_plan_add_scale_shift_act(gv_plan* p, int64_t npix, int32_t c, int32_t x_ld,
                                           int32_t y_ld, int32_t relu, int32_t dtype, int32_t x_slot,
                                           int64_t x_off, int32_t ss_slot, int64_t scale_off,
                                           int64_t shift_off, int32_t y_slot, int64_t y_off) {
    if (!p) return GV_E_PLAN;
    if (x_slot < 0 || ss_slot < 0 || y_slot < 0) return GV_E_BADARG;
    Op o{};
    o.kind = OP_SSA;
    o.npix = npix; o.c = c; o.x_ld = x_ld; o.y_ld = y_ld; o.relu = relu; o.dtype = dtype;
    o.x = {x_slot, x_off};
    o.scale = {ss_slot, scale_off};
    o.shift = {ss_slot, shift_off};
    o.y = {y_slot, y_off};
    o.w = o.res = o.y2 = o.scale2 = o.shift2 = {-1, 0};
    note(p, o.x); note(p, o.scale); note(p, o.y);
    p->ops.push_back(o);
    return GV_OK;
}

// Fork/join over the plan's lanes: lane 0 is the caller's stream, the others are plan-owned streams
// that wait for the caller's stream at the start and are joined back at the end, so from outside
// the run is ordered on `stream` like a single-stream run (and can be captured into a hipGraph).
static int run_lanes(gv_plan* p, void* const* bufs, hipStream_t main) {
    const int nl = p->num_lanes;
    if (p->lane_streams.empty()) {
        p->lane_streams.assign((size_t)nl, nullptr);
        p->ev_join.assign((size_t)nl, nullptr);
        for (int l = 1; l < nl; ++l) {
            GV_HIP_CHECK(hipStreamCreateWithFlags(&p->lane_streams[(size_t)l], hipStreamNonBlocking));
            GV_HIP_CHECK(hipEventCreateWithFlags(&p->ev_join[(size_t)l], hipEventDisableTiming));
        }
        GV_HIP_CHECK(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
        p->op_events.assign(p->ops.size(), nullptr);
        for (size_t i = 0; i < p->ops.size(); ++i)
            for (int32_t d : p->ops[i].deps)
                if (p->ops[(size_t)d].lane != p->ops[i].lane) p->ops[(size_t)d].signal = true;
        for (size_t i = 0; i < p->ops.size(); ++i)
            if (p->ops[i].signal) GV_HIP_CHECK(hipEventCreateWithFlags(&p->op_events[i], hipEventDisableTiming));
    }
    GV_HIP_CHECK(hipEventRecord(p->ev_fork, main));
    for (int l = 1; l < nl; ++l) GV_HIP_CHECK(hipStreamWaitEvent(p->lane_streams[(size_t)l], p->ev_fork, 0));
    // a failing op must not leave the fork open: the lane streams are ALWAYS joined back into the caller's stream
    // (an un-joined fork would end a stream capture with hipErrorStreamCaptureUnjoined and let later work on `main`
    // overtake what the lanes already enqueued)
    int rc = GV_OK;
    for (size_t i = 0; i < p->ops.size() && rc == GV_OK; ++i) {
        const Op& o = p->ops[i];
        hipStream_t st = o.lane == 0 ? main : p->lane_streams[(size_t)o.lane];
        for (int32_t d : o.deps)
            if (p->ops[(size_t)d].lane != o.lane && rc == GV_OK) {
                const hipError_t e = hipStreamWaitEvent(st, p->op_events[(size_t)d], 0);
                if (e != hipSuccess) rc = (int)e;
            }
        if (rc == GV_OK) rc = run_op(o, bufs, (void*)st);
        if (rc == GV_OK && o.signal) {
            const hipError_t e = hipEventRecord(p->op_events[i], st);
            if (e != hipSuccess) rc = (int)e;
        }
    }
    for (int l = 1; l < nl; ++l) {
        hipError_t e = hipEventRecord(p->ev_join[(size_t)l], p->lane_streams[(size_t)l]);
        if (e == hipSuccess) e = hipStreamWaitEvent(main, p->ev_join[(size_t)l], 0);
        if (e != hipSuccess && rc == GV_OK) rc = (int)e;
    }
    return rc;
}

extern "C" int gv_plan_run_range(const gv_plan* p, int32_t first, int32_t count,
                                 void* const* buffers_host, int32_t num_slots, void* stream) {
    if (!p) return GV_E_PLAN;
    if (!buffers_host || first < 0 || count < 0 || (size_t)first + (size_t)count > p->ops.size())
        return GV_E_BADARG;
    if (num_slots <= p->max_slot) return GV_E_PLAN;
    for (int32_t i = 0; i <= p->max_slot; ++i)
        if (!buffers_host[i]) return GV_E_BADARG;
    if (p->num_lanes > 1 && first == 0 && (size_t)count == p->ops.size())
        return run_lanes(const_cast<gv_plan*>(p), buffers_host, (hipStream_t)stream);
    for (int32_t i = first; i < first + count; ++i) {
        const int rc = run_op(p->ops[(size_t)i], buffers_host, stream);
        if (rc != GV_OK) return rc;
    }
    return GV_OK;
}

extern "C" int gv_plan_run(const gv_plan* p, void* const* buffers_host, int32_t num_slots,
                           void* stream) {
    if (!p) return GV_E_PLAN;
    return gv_plan_run_range(p, 0, (int32_t)p->ops.size(), buffers_host, num_slots, stream);
}

extern "C" int gv_plan_time(const gv_plan* p, int32_t first, int32_t count, void* const* buffers_host,
                            int32_t num_slots, int32_t iters, float* ms_avg_host, void* stream) {
    if (!p) return GV_E_PLAN;
    if (!ms_avg_host || iters <= 0) return GV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    GV_HIP_CHECK(hipEventCreate(&e0));
    GV_HIP_CHECK(hipEventCreate(&e1));
    int rc = gv_plan_run_range(p, first, count, buffers_host, num_slots, stream);   // warm
    if (rc == GV_OK) {
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < iters && rc == GV_OK; ++i)
            rc = gv_plan_run_range(p, first, count, buffers_host, num_slots, stream);
        (void)hipEventRecord(e1, st);
        const hipError_t e = hipEventSynchronize(e1);
        if (rc == GV_OK && e != hipSuccess) rc = (int)e;
        float ms = 0.f;
        if (rc == GV_OK) { (void)hipEventElapsedTime(&ms, e0, e1); *ms_avg_host = ms / (float)iters; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

// Every op of the plan timed IN SEQUENCE: `iters` whole passes on the caller's stream (single launch lane), an event in
// front of the first op and one behind every op, so a launch is timed where it sits in the step — behind its real
// predecessor, on whatever that left in the caches — not as a warm repeat of itself.  ms_per_op_host[i] = the average
// over the passes of (event behind op i) - (event behind op i-1).
extern "C" int gv_plan_time_each(const gv_plan* p, void* const* buffers_host, int32_t num_slots, int32_t iters,
                                 float* ms_per_op_host, void* stream) {
    if (!p) return GV_E_PLAN;
    if (!buffers_host || !ms_per_op_host || iters <= 0) return GV_E_BADARG;
    if (num_slots <= p->max_slot) return GV_E_PLAN;
    for (int32_t i = 0; i <= p->max_slot; ++i)
        if (!buffers_host[i]) return GV_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t n = p->ops.size();
    std::vector<hipEvent_t> ev(n + 1, nullptr);
    int rc = GV_OK;
    for (size_t i = 0; i <= n && rc == GV_OK; ++i) {
        const hipError_t e = hipEventCreate(&ev[i]);
        if (e != hipSuccess) rc = (int)e;
    }
    for (size_t i = 0; i < n; ++i) ms_per_op_host[i] = 0.f;
    for (size_t i = 0; i < n && rc == GV_OK; ++i) rc = run_op(p->ops[i], buffers_host, stream);   // warm pass
    for (int it = 0; it < iters && rc == GV_OK; ++it) {
        (void)hipEventRecord(ev[0], st);
        for (size_t i = 0; i < n && rc == GV_OK; ++i) {
            rc = run_op(p->ops[i], buffers_host, stream);
            (void)hipEventRecord(ev[i + 1], st);
        }
        if (rc == GV_OK) {
            const hipError_t e = hipEventSynchronize(ev[n]);
            if (e != hipSuccess) rc = (int)e;
        }
        for (size_t i = 0; i < n && rc == GV_OK; ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            ms_per_op_host[i] += ms / (float)iters;
        }
    }
    for (size_t i = 0; i <= n; ++i)
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    return rc;
}

// ---- hipGraph capture of whatever the caller enqueues on a stream (gvcnn_hip.h) --------------------------------
struct gv_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

extern "C" int gv_capture_begin(void* stream) {
    if (!stream) return GV_E_BADARG;                       // the legacy default stream cannot be captured
    // the LDS-DMA kernels' zero page is allocated (hipMalloc + hipMemset) on first use: do that BEFORE the capture
    // starts, or the first such launch under capture would invalidate it
    if (!gvconv::dma_zero_page()) return GV_E_UNSUPPORTED;
    GV_HIP_CHECK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
    return GV_OK;
}

extern "C" int gv_capture_end(void* stream, gv_graph** out) {
    if (!stream || !out) return GV_E_BADARG;
    gv_graph* g = new (std::nothrow) gv_graph();
    if (!g) return GV_E_PLAN;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g->graph);
    if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        if (g->graph) (void)hipGraphDestroy(g->graph);
        delete g;
        return (int)e;
    }
    *out = g;
    return GV_OK;
}

extern "C" int gv_graph_launch(const gv_graph* g, void* stream) {
    if (!g || !g->exec) return GV_E_BADARG;
    GV_HIP_CHECK(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return GV_OK;
}

extern "C" void gv_graph_destroy(gv_graph* g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}
