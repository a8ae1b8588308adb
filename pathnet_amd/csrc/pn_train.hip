// pn_train.hip -- the two element-wise pieces of the reference's training step that sit between the aggregator's
// forward and backward: the loss and the optimizer update.
//   loss      torch.nn.CrossEntropyLoss()      /root/reference/PathNet_run.py:297, :346
//   update    torch.optim.Adam(lr, weight_decay)  PathNet_run.py:295-296, :352
// On the bench workload torch spends ~100 us per step on them (a single-block nll reduction, softmax forward and
// backward launches, a multi_tensor_apply Adam); both are one small launch here.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/pathnet_hip.h"
#include "pn_internal.h"

namespace {

// ---- softmax cross entropy, mean over rows, and d loss / d logits -----------------------------------------------
// one wavefront per 64 rows is plenty for C <= a few hundred classes: a thread walks its row three times
__global__ __launch_bounds__(1024) void cross_entropy_kernel(const float *__restrict__ logits,
                                                             const int64_t *__restrict__ target, int rows, int classes,
                                                             float inv_rows, float *__restrict__ loss,
                                                             float *__restrict__ g_logits, int store) {
    __shared__ float part[16];
    float mine = 0.0f;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
        const float *x = logits + (int64_t)r * classes;
        float m = x[0];
        for (int c = 1; c < classes; c++) m = fmaxf(m, x[c]);
        float s = 0.0f;
        for (int c = 0; c < classes; c++) s += expf(x[c] - m);
        const float lse = m + logf(s);
        const int t = (int)target[r];
        mine += lse - x[t];
        if (g_logits) {
            float *g = g_logits + (int64_t)r * classes;
            for (int c = 0; c < classes; c++) g[c] = (expf(x[c] - lse) - (c == t ? 1.0f : 0.0f)) * inv_rows;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) s += part[w];
        if (store)
            *loss = s * inv_rows;           // one workgroup computes the whole loss: no zero-fill launch ahead of it
        else
            atomicAdd(loss, s * inv_rows);  // (several workgroups, or a loss that accumulates over micro-batches)
    }
}

// ---- Adam over a list of tensors in one launch -------------------------------------------------------------------
constexpr int ADAM_CHUNK = 2048;      // elements per block
struct AdamList {
    pn_adam_tensor t[PN_ADAM_MAX_TENSORS];
    int block_begin[PN_ADAM_MAX_TENSORS + 1];
    int n;
    float lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, weight_decay;
    float lr;
    const pn_step_state *dyn;     // step count in device memory (hipGraph replay): the bias corrections are formed here
    int advance;                  // pn_adam_step_advance, last launch: the workgroup that finishes last moves *dyn to the next step
};

// epoch += 1, adam_step += 1, seed = splitmix64(seed)   (= step_state_advance_kernel, pn_context.hip)
__device__ __forceinline__ void advance_step_state(pn_step_state *s) {
    s->epoch += 1;
    s->adam_step += 1;
    uint64_t z = (s->seed += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    s->seed = z ^ (z >> 31);
}
__global__ void adam_advance_only_kernel(pn_step_state *s) {
    if (threadIdx.x == 0 && blockIdx.x == 0) advance_step_state(s);
}

__global__ __launch_bounds__(256) void adam_kernel(AdamList a) {
    float lr_over_bc1 = a.lr_over_bc1, inv_sqrt_bc2 = a.inv_sqrt_bc2;
    if (a.dyn) {      // block-uniform: beta^step by squaring (exact enough in double; step < 2^31)
        long long n = a.dyn->adam_step;
        double p1 = 1.0, p2 = 1.0, b1 = (double)a.beta1, b2 = (double)a.beta2;
        while (n > 0) {
            if (n & 1) {
                p1 *= b1;
                p2 *= b2;
            }
            b1 *= b1;
            b2 *= b2;
            n >>= 1;
        }
        lr_over_bc1 = (float)((double)a.lr / (1.0 - p1));
        inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - p2));
    }
    int k = 0;
    while (k + 1 < a.n && (int)blockIdx.x >= a.block_begin[k + 1]) k++;      // block-uniform
    const pn_adam_tensor t = a.t[k];
    const int64_t base = (int64_t)((int)blockIdx.x - a.block_begin[k]) * ADAM_CHUNK;
    for (int i = threadIdx.x; i < ADAM_CHUNK; i += 256) {
        const int64_t e = base + i;
        if (e >= t.count) break;
        // torch/optim/adam.py (_single_tensor_adam): L2 weight decay folded into the gradient, bias-corrected
        float g = t.grad[e];
        const float p = t.param[e];
        g += a.weight_decay * p;
        const float m = a.beta1 * t.exp_avg[e] + (1.0f - a.beta1) * g;
        const float v = a.beta2 * t.exp_avg_sq[e] + (1.0f - a.beta2) * g * g;
        t.exp_avg[e] = m;
        t.exp_avg_sq[e] = v;
        t.param[e] = p - lr_over_bc1 * (m / (sqrtf(v) * inv_sqrt_bc2 + a.eps));
    }
    if (a.dyn && a.advance) {       // (block-uniform) every workgroup has read adam_step by the time it takes its ticket
        __syncthreads();
        if (threadIdx.x == 0) {
            pn_step_state *s = const_cast<pn_step_state *>(a.dyn);
            unsigned int *ticket = reinterpret_cast<unsigned int *>(&s->reserved);
            if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
                atomicExch(ticket, 0u);
                advance_step_state(s);
            }
        }
    }
}

}  // namespace

namespace pn {
// loss[0] += scale * sum over rows of (logsumexp - logit[target]);  g_logits = (softmax - onehot) * scale
int launch_cross_entropy(const float *logits, const int64_t *target, int rows, int classes, float scale, float *loss,
                         float *g_logits, void *stream_, bool overwrite) {
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (rows < 1) return PN_OK;
    // Up to CE_ONE_BLOCK_ROWS rows one workgroup of 1024 threads walks them all (a few microseconds) and the loss is a
    // fixed-order sum: the same bits on every run.  Beyond that the workgroups add their parts with an atomic -- the
    // reported loss may then differ in its last bit from run to run; g_logits never does (row-wise).
    constexpr int CE_ONE_BLOCK_ROWS = 32768;
    const int threads = rows <= CE_ONE_BLOCK_ROWS ? 1024 : 256;
    const int blocks = rows <= CE_ONE_BLOCK_ROWS ? 1 : ((rows + 255) / 256 < 256 ? (rows + 255) / 256 : 256);
    if (overwrite && blocks > 1) PN_CHECK_HIP(hipMemsetAsync(loss, 0, sizeof(float), stream));
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(blocks), dim3(threads), 0, stream, logits, target, rows, classes, scale,
                       loss, g_logits, overwrite && blocks == 1 ? 1 : 0);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}
}  // namespace pn

extern "C" {

int pn_cross_entropy(const float *logits, const int64_t *target, int32_t rows, int32_t classes, float *loss,
                     float *g_logits, void *stream_) {
    if (!logits || !target || !loss) PN_FAIL(PN_ERR_ARG, "pn_cross_entropy: null argument");
    if (rows < 1 || classes < 1) PN_FAIL(PN_ERR_ARG, "pn_cross_entropy: rows=%d classes=%d", rows, classes);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    return pn::launch_cross_entropy(logits, target, rows, classes, 1.0f / (float)rows, loss, g_logits, stream, true);
}

static int adam_step_impl(const pn_adam_tensor *tensors, int32_t n_tensors, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int64_t step, const pn_step_state *step_state, bool advance, void *stream_) {
    if (n_tensors < 0 || (n_tensors > 0 && !tensors)) PN_FAIL(PN_ERR_ARG, "pn_adam_step: bad tensor list");
    if (advance && !step_state) PN_FAIL(PN_ERR_ARG, "pn_adam_step_advance: no step state to advance");
    bool advanced = false;
    if (step_state) step = 1;       // (ignored: the kernel reads step_state->adam_step)
    if (step < 1) PN_FAIL(PN_ERR_ARG, "pn_adam_step: step counts from 1 (got %lld)", (long long)step);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step), bc2 = 1.0 - std::pow((double)beta2, (double)step);
    for (int32_t at = 0; at < n_tensors; at += PN_ADAM_MAX_TENSORS) {
        AdamList a{};
        a.n = n_tensors - at < PN_ADAM_MAX_TENSORS ? n_tensors - at : PN_ADAM_MAX_TENSORS;
        int blocks = 0;
        for (int i = 0; i < a.n; i++) {
            const pn_adam_tensor &t = tensors[at + i];
            if (t.count < 0 || (t.count > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq)))
                PN_FAIL(PN_ERR_ARG, "pn_adam_step: tensor %d has a null pointer", at + i);
            a.t[i] = t;
            a.block_begin[i] = blocks;
            blocks += (int)((t.count + ADAM_CHUNK - 1) / ADAM_CHUNK);
        }
        a.block_begin[a.n] = blocks;
        a.lr_over_bc1 = (float)((double)lr / bc1);
        a.inv_sqrt_bc2 = (float)(1.0 / std::sqrt(bc2));
        a.beta1 = beta1;
        a.beta2 = beta2;
        a.eps = eps;
        a.weight_decay = weight_decay;
        a.lr = lr;
        a.dyn = step_state;
        a.advance = (advance && at + PN_ADAM_MAX_TENSORS >= n_tensors) ? 1 : 0;      // the step's last launch
        if (blocks == 0) continue;
        hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, stream, a);
        PN_CHECK_HIP(hipGetLastError());
        advanced = advanced || a.advance != 0;
    }
    if (advance && !advanced) {     // nothing to update in the last launch: the state still moves on
        hipLaunchKernelGGL(adam_advance_only_kernel, dim3(1), dim3(64), 0, stream, const_cast<pn_step_state *>(step_state));
        PN_CHECK_HIP(hipGetLastError());
    }
    return PN_OK;
}

int pn_adam_step(const pn_adam_tensor *tensors, int32_t n_tensors, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int64_t step, const pn_step_state *step_state, void *stream_) {
    return adam_step_impl(tensors, n_tensors, lr, beta1, beta2, eps, weight_decay, step, step_state, false, stream_);
}

int pn_adam_step_advance(const pn_adam_tensor *tensors, int32_t n_tensors, float lr, float beta1, float beta2, float eps,
                         float weight_decay, pn_step_state *step_state, void *stream_) {
    return adam_step_impl(tensors, n_tensors, lr, beta1, beta2, eps, weight_decay, 1, step_state, true, stream_);
}

}  // extern "C"
