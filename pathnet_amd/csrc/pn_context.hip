// pn_context.hip -- the caller-owned context (second stream, fork/join events, per-stage timing records), the device
// query and the clock probe.  The library keeps no other state between calls ("no global state except
// an opaque handle").
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <tuple>
#include <vector>

#include "pn_internal.h"

struct pn_context {
    int device = 0;
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int prof_mode = 0, prof_stage = -1;
    struct Rec {
        int stage;
        hipEvent_t a, b;
    };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> free_events;
    std::map<const void *, int> lds_attr;      // kernel -> dynamic LDS bytes already granted on this device
    std::map<std::tuple<const void *, size_t, int>, int> slots;    // (kernel, dynamic LDS bytes, threads) -> resident workgroups per CU
    int cus = 0;
    pn::Knobs knobs;
};

namespace {

__global__ void warm_kernel() {}

__global__ void step_state_advance_kernel(pn_step_state *s) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    s->epoch += 1;
    s->adam_step += 1;
    uint64_t z = (s->seed += 0x9E3779B97F4A7C15ull);       // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    s->seed = z ^ (z >> 31);
}

// spins for ~`spin_ticks` ticks of the constant 100 MHz wall clock and reports shader cycles against wall ticks
__global__ void clock_probe_kernel(long long spin_ticks, long long *out) {
    const long long w0 = wall_clock64();
    const long long c0 = (long long)__builtin_readcyclecounter();
    long long w1 = w0;
    while (w1 - w0 < spin_ticks) {
        __builtin_amdgcn_s_sleep(8);
        w1 = wall_clock64();
    }
    const long long c1 = (long long)__builtin_readcyclecounter();
    out[0] = c1 - c0;
    out[1] = w1 - w0;
}

hipEvent_t take_event(pn_context *ctx) {
    if (!ctx->free_events.empty()) {
        hipEvent_t e = ctx->free_events.back();
        ctx->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    // (timing brackets: the device-scope form is the one meant for timing the commands between two events)
    (void)hipEventCreateWithFlags(&e, ctx->knobs.event_dev ? hipEventDisableSystemFence : hipEventDefault);
    return e;
}

}  // namespace

namespace {

struct KnobName {
    const char *name;
    int pn::Knobs::*field;
};
const KnobName kKnobs[] = {{"PN_NODE_GEMM3", &pn::Knobs::node_gemm3}, {"PN_EVAL_ZW", &pn::Knobs::eval_zw},
                           {"PN_POOL_BWD_WG", &pn::Knobs::pool_bwd_wg}, {"PN_POOL_STEP", &pn::Knobs::pool_step}, {"PN_ZERO_EARLY", &pn::Knobs::zero_early}, {"PN_SMALL_SIDE", &pn::Knobs::small_side}, {"PN_EVENT_DEVICE_SCOPE", &pn::Knobs::event_dev}, {"PN_NODE_RGRAD", &pn::Knobs::node_rgrad},
                           {"PN_SAMPLER_STAGE", &pn::Knobs::sampler_stage}, {"PN_SEQ4", &pn::Knobs::seq4},
                           {"PN_SEQH_TAIL", &pn::Knobs::seqh_tail}};

}  // namespace

namespace pn {

const Knobs &knobs_of(const pn_context *ctx) {
    static const Knobs defaults;
    return ctx ? ctx->knobs : defaults;
}

bool profiling_every_stage(const pn_context *ctx) { return ctx && ctx->prof_mode == 1; }

int resident_slots(pn_context *ctx, const void *kernel, int threads, size_t lds_bytes, int *slots, int *cus) {
    if (ctx) {
        auto it = ctx->slots.find(std::make_tuple(kernel, lds_bytes, threads));
        if (it != ctx->slots.end() && ctx->cus > 0) {
            *slots = it->second * ctx->cus;
            *cus = ctx->cus;
            return PN_OK;
        }
    }
    int dev = 0, n_cu = 0, per_cu = 0;
    PN_CHECK_HIP(hipGetDevice(&dev));
    PN_CHECK_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    PN_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds_bytes));
    if (ctx) {
        ctx->slots[std::make_tuple(kernel, lds_bytes, threads)] = per_cu;
        ctx->cus = n_cu;
    }
    *slots = per_cu * n_cu;
    *cus = n_cu;
    return PN_OK;
}

int context_check_device(const pn_context *ctx) {
    if (!ctx) return PN_OK;
    int dev = -1;
    PN_CHECK_HIP(hipGetDevice(&dev));
    if (dev != ctx->device)
        PN_FAIL(PN_ERR_ARG, "pn_context belongs to device %d, the current device is %d", ctx->device, dev);
    return PN_OK;
}

int ensure_dynamic_lds(pn_context *ctx, const void *kernel, int bytes) {
    if (ctx) {
        auto it = ctx->lds_attr.find(kernel);
        if (it != ctx->lds_attr.end() && it->second >= bytes) return PN_OK;
    }
    PN_CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (ctx) ctx->lds_attr[kernel] = bytes;
    return PN_OK;
}

void *context_side_stream(pn_context *ctx) { return ctx ? (void *)ctx->side : nullptr; }
void *context_fork(pn_context *ctx, void *stream) {
    if (!ctx || !ctx->side) return nullptr;
    if (hipEventRecord(ctx->fork, (hipStream_t)stream) != hipSuccess) return nullptr;
    if (hipStreamWaitEvent(ctx->side, ctx->fork, 0) != hipSuccess) return nullptr;
    return ctx->side;
}
int context_record_join(pn_context *ctx) {
    PN_CHECK_HIP(hipEventRecord(ctx->join, ctx->side));
    return PN_OK;
}
int context_join(pn_context *ctx, void *stream) {
    PN_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, ctx->join, 0));
    return PN_OK;
}

StageTimer::StageTimer(pn_context *ctx_, int stage, void *stream_) : ctx(ctx_), slot(-1), stream(stream_) {
    if (!ctx || ctx->prof_mode == 0 || (ctx->prof_mode == 2 && stage != ctx->prof_stage)) return;
    pn_context::Rec r{stage, take_event(ctx), take_event(ctx)};
    (void)hipEventRecord(r.a, (hipStream_t)stream);
    slot = (int)ctx->recs.size();
    ctx->recs.push_back(r);
}
StageTimer::~StageTimer() {
    if (slot >= 0) (void)hipEventRecord(ctx->recs[(size_t)slot].b, (hipStream_t)stream);
}

}  // namespace pn

extern "C" {

int pn_context_create(pn_context **out) try {
    if (!out) PN_FAIL(PN_ERR_ARG, "pn_context_create: null");
    *out = nullptr;
    pn_context *c = new (std::nothrow) pn_context();
    if (!c) PN_FAIL(PN_ERR_NOMEM, "pn_context_create: out of memory");
    auto fail = [&](hipError_t e, const char *what) {
        pn::set_error("%s failed: %s", what, hipGetErrorString(e));
        (void)pn_context_destroy(c);
        return PN_ERR_HIP;
    };
    for (const KnobName &k : kKnobs)
        if (const char *v = std::getenv(k.name)) c->knobs.*(k.field) = std::atoi(v);
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) return fail(e, "hipGetDevice");
    if ((e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
    // (both events order kernels of this device only: no host ever inspects them, so the system-scope release a recorded event
    //  performs by default buys nothing -- the kernels' own agent-scope release at their end is what the other queue needs)
    const unsigned ev_flags = hipEventDisableTiming | (c->knobs.event_dev ? hipEventDisableSystemFence : 0u);
    if ((e = hipEventCreateWithFlags(&c->fork, ev_flags)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->join, ev_flags)) != hipSuccess) return fail(e, "hipEventCreate");
    // the runtime builds a stream's hardware queue at its first launch (milliseconds): pay that here
    hipLaunchKernelGGL(warm_kernel, dim3(1), dim3(64), 0, c->side);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e, "first launch on the context's stream");
    *out = c;
    return PN_OK;
} catch (...) {
    PN_FAIL(PN_ERR_NOMEM, "pn_context_create: exception");
}

int pn_context_destroy(pn_context *c) {
    if (!c) return PN_OK;
    int rc = PN_OK;
    if (c->side) {
        if (hipStreamSynchronize(c->side) != hipSuccess) rc = PN_ERR_HIP;
        (void)hipStreamDestroy(c->side);
    }
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->join) (void)hipEventDestroy(c->join);
    for (auto &r : c->recs) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
    delete c;
    if (rc != PN_OK) pn::set_error("pn_context_destroy: the context's stream reported an error");
    return rc;
}

int pn_context_set_knob(pn_context *ctx, const char *name, int32_t value) {
    if (!ctx || !name) PN_FAIL(PN_ERR_ARG, "pn_context_set_knob: null");
    for (const KnobName &k : kKnobs)
        if (std::strcmp(k.name, name) == 0) {
            const bool events_change = k.field == &pn::Knobs::event_dev && (ctx->knobs.event_dev != 0) != (value != 0);
            ctx->knobs.*(k.field) = value;
            if (events_change) {        // the fork / join pair and the pooled timing events carry the flag: made again (idle device first)
                PN_CHECK_HIP(hipDeviceSynchronize());
                const unsigned ev_flags = hipEventDisableTiming | (value ? hipEventDisableSystemFence : 0u);
                hipEvent_t fork = nullptr, join = nullptr;
                PN_CHECK_HIP(hipEventCreateWithFlags(&fork, ev_flags));
                PN_CHECK_HIP(hipEventCreateWithFlags(&join, ev_flags));
                (void)hipEventDestroy(ctx->fork);
                (void)hipEventDestroy(ctx->join);
                ctx->fork = fork;
                ctx->join = join;
                for (hipEvent_t e : ctx->free_events) (void)hipEventDestroy(e);
                ctx->free_events.clear();
            }
            return PN_OK;
        }
    PN_FAIL(PN_ERR_ARG, "pn_context_set_knob: no knob named %s", name);
}
int pn_context_get_knob(const pn_context *ctx, const char *name, int32_t *value) {
    if (!name || !value) PN_FAIL(PN_ERR_ARG, "pn_context_get_knob: null");
    for (const KnobName &k : kKnobs)
        if (std::strcmp(k.name, name) == 0) {
            *value = pn::knobs_of(ctx).*(k.field);
            return PN_OK;
        }
    PN_FAIL(PN_ERR_ARG, "pn_context_get_knob: no knob named %s", name);
}

int pn_profile_configure(pn_context *ctx, int32_t mode, int32_t stage) {
    if (!ctx) PN_FAIL(PN_ERR_ARG, "pn_profile_configure: null context");
    if (mode < 0 || mode > 2) PN_FAIL(PN_ERR_ARG, "profile mode %d", mode);
    ctx->prof_mode = mode;
    ctx->prof_stage = stage;
    return PN_OK;
}
int pn_profile_stage_count(void) { return pn::ST_COUNT; }
const char *pn_profile_stage_name(int32_t stage) {
    static const char *const names[pn::ST_COUNT] = {"sampler_glibc_fill", "sampler_walk", "gather",    "fc0",      "bank",
                                                    "plan_pack",          "seq_fwd",      "pool_fwd",  "fc2_grad", "pool_bwd",
                                                    "seq_bwd",            "wgrad",        "bias_grad", "bank_bwd", "fc0_bwd", "zero_fill"};
    return (stage >= 0 && stage < pn::ST_COUNT) ? names[stage] : "?";
}
int pn_profile_read(pn_context *ctx, double *ms_sum, int64_t *count) try {
    if (!ctx || !ms_sum || !count) PN_FAIL(PN_ERR_ARG, "pn_profile_read: null");
    for (auto &r : ctx->recs) {
        PN_CHECK_HIP(hipEventSynchronize(r.b));
        float ms = 0.0f;
        PN_CHECK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        ms_sum[r.stage] += ms;
        count[r.stage] += 1;
        ctx->free_events.push_back(r.a);
        ctx->free_events.push_back(r.b);
    }
    ctx->recs.clear();
    return PN_OK;
} catch (...) {
    PN_FAIL(PN_ERR_NOMEM, "pn_profile_read: exception");
}

int pn_device_query(pn_device_info *out) {
    if (!out) PN_FAIL(PN_ERR_ARG, "pn_device_query: null");
    int dev = 0;
    PN_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PN_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    std::snprintf(out->name, sizeof out->name, "%s", prop.name);
    std::snprintf(out->arch, sizeof out->arch, "%s", prop.gcnArchName);
    out->compute_units = prop.multiProcessorCount;
    out->lds_bytes_per_block = (int32_t)prop.sharedMemPerBlock;
    out->hbm_bytes = (int64_t)prop.totalGlobalMem;
    out->clock_khz = prop.clockRate;
    return PN_OK;
}

int pn_step_state_advance(pn_step_state *dev_state, void *stream_) {
    if (!dev_state) PN_FAIL(PN_ERR_ARG, "pn_step_state_advance: null");
    hipLaunchKernelGGL(step_state_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, dev_state);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

int pn_clock_probe(double *mhz, void *stream_) {
    if (!mhz) PN_FAIL(PN_ERR_ARG, "pn_clock_probe: null");
    hipStream_t stream = (hipStream_t)stream_;
    long long *buf = nullptr;
    PN_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&buf), 2 * sizeof(long long), hipHostMallocDefault));
    buf[0] = buf[1] = 0;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, 200000LL /* 2 ms of the 100 MHz clock */, buf);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    const long long cyc = buf[0], ticks = buf[1];
    (void)hipHostFree(buf);
    if (e != hipSuccess) PN_FAIL(PN_ERR_HIP, "clock probe failed: %s", hipGetErrorString(e));
    if (ticks <= 0) PN_FAIL(PN_ERR_HIP, "clock probe: the wall clock did not advance");
    *mhz = (double)cyc / (double)ticks * 100.0;
    return PN_OK;
}

}  // extern "C"
