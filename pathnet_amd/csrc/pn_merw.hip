// pn_merw.hip -- the MERW transition probabilities, i.e. what produces the sampler's edge_input/<name>.in
// (SURVEY.md §8 f-2).  Replaces
//   /root/reference/preprocess/compute_merw.py:107-121  compute_merw(A): dominant eigenpair (lambda, psi) of the symmetric
//       adjacency matrix by scipy's eigsh, then P[i, j] = A[i, j] * psi[j] / (lambda * psi[i]) on the non-zeros,
//   called by /root/reference/preprocess/init_rw.py:76.
// Here: power iteration on A + I in fp64 on the device (the shift makes the Perron pair the unique dominant one also on
// bipartite graphs), CSR SpMV with one thread per row (citation-graph rows hold a handful of entries), fixed-order
// two-stage reductions (deterministic), convergence on the residual |A x - lambda x| checked every 32 iterations.
// HBM-bound and tiny: Pubmed's 88 k non-zeros are 1.4 MB per SpMV.  gfx950 only.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/pathnet_hip.h"
#include "pn_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 1024;

// y = (A + I) x ; per-block partial sums of y.y and x.y
__global__ __launch_bounds__(kThreads) void spmv_kernel(int n, const int64_t *__restrict__ row_off,
                                                         const int32_t *__restrict__ col, const double *__restrict__ val,
                                                         const double *__restrict__ x, double *__restrict__ y,
                                                         double *__restrict__ part) {
    __shared__ double s_yy[kThreads / 64], s_xy[kThreads / 64];
    double yy = 0.0, xy = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        double acc = x[i];
        for (int64_t k = row_off[i]; k < row_off[i + 1]; k++) acc += (val ? val[k] : 1.0) * x[col[k]];
        y[i] = acc;
        yy += acc * acc;
        xy += acc * x[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        yy += __shfl_xor(yy, o, 64);
        xy += __shfl_xor(xy, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        s_yy[threadIdx.x >> 6] = yy;
        s_xy[threadIdx.x >> 6] = xy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (s_yy[0] + s_yy[1]) + (s_yy[2] + s_yy[3]);
        part[2 * blockIdx.x + 1] = (s_xy[0] + s_xy[1]) + (s_xy[2] + s_xy[3]);
    }
}

// scal[0] = |y|, scal[1] = x.y (= lambda + 1 for unit x), then x = y / |y| ; residual partials for the next check
__global__ void finish_kernel(const double *__restrict__ part, int nblocks, double *__restrict__ scal) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double yy = 0.0, xy = 0.0;
    for (int b = 0; b < nblocks; b++) {
        yy += part[2 * b];
        xy += part[2 * b + 1];
    }
    scal[0] = sqrt(yy);
    scal[1] = xy;
}

// res partial = sum (y - (lambda + 1) x)^2 with the OLD x; then x <- y / |y|
__global__ __launch_bounds__(kThreads) void update_kernel(int n, const double *__restrict__ y, double *__restrict__ x,
                                                           const double *__restrict__ scal, double *__restrict__ part) {
    __shared__ double s_r[kThreads / 64];
    const double inv = 1.0 / scal[0], lam1 = scal[1];
    double rr = 0.0;
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const double yi = y[i], r = yi - lam1 * x[i];
        rr += r * r;
        x[i] = yi * inv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rr += __shfl_xor(rr, o, 64);
    if ((threadIdx.x & 63) == 0) s_r[threadIdx.x >> 6] = rr;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (s_r[0] + s_r[1]) + (s_r[2] + s_r[3]);
}

__global__ void residual_kernel(const double *__restrict__ part, int nblocks, double *__restrict__ scal) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double rr = 0.0;
    for (int b = 0; b < nblocks; b++) rr += part[b];
    scal[2] = sqrt(rr);
}

__global__ __launch_bounds__(kThreads) void init_kernel(int n, double *__restrict__ x) {
    const double v = 1.0 / sqrt((double)n);
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) x[i] = v;
}

// p[k] = A[i, j] psi[j] / (lambda psi[i]) for the stored entry k = (i, j)      compute_merw.py:116-120
__global__ __launch_bounds__(kThreads) void prob_kernel(int n, const int64_t *__restrict__ row_off,
                                                         const int32_t *__restrict__ col, const double *__restrict__ val,
                                                         const double *__restrict__ psi, double lambda,
                                                         double *__restrict__ p) {
    for (int i = blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads) {
        const double denom = lambda * psi[i];
        for (int64_t k = row_off[i]; k < row_off[i + 1]; k++) p[k] = (val ? val[k] : 1.0) * psi[col[k]] / denom;
    }
}

int blocks_for(int n) {
    int b = (n + kThreads - 1) / kThreads;
    return b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b);
}

}  // namespace

extern "C" {

int pn_merw_workspace_bytes(int32_t n, int64_t *bytes) {
    if (!bytes || n < 0) PN_FAIL(PN_ERR_ARG, "pn_merw_workspace_bytes: bad argument");
    *bytes = ((int64_t)n + 2 * kMaxBlocks + 8) * (int64_t)sizeof(double);
    return PN_OK;
}

int pn_merw_probabilities(int32_t n, int64_t nnz, const int64_t *row_off, const int32_t *col, const double *val,
                          double *p, double *psi, double *lambda, int32_t max_iter, double tol, int32_t *iters,
                          void *workspace, int64_t workspace_bytes, void *stream_) {
    if (n < 1 || nnz < 0 || !row_off || (nnz > 0 && !col) || !p || !psi || !lambda)
        PN_FAIL(PN_ERR_ARG, "pn_merw_probabilities: bad argument");
    int64_t need = 0;
    pn_merw_workspace_bytes(n, &need);
    if (!workspace || workspace_bytes < need)
        PN_FAIL(PN_ERR_CAPACITY, "MERW workspace holds %lld bytes, need %lld", (long long)workspace_bytes, (long long)need);
    if (max_iter < 1) max_iter = 100000;
    if (!(tol > 0.0)) tol = 1e-13;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double *y = reinterpret_cast<double *>(workspace);
    double *part = y + n;
    double *scal = part + 2 * kMaxBlocks;       // [0] |y|, [1] x.y, [2] residual
    const int nb = blocks_for(n);
    hipLaunchKernelGGL(init_kernel, dim3(nb), dim3(kThreads), 0, stream, n, psi);
    double host[3] = {0.0, 0.0, 0.0};
    int it = 0;
    bool converged = false;
    while (it < max_iter && !converged) {
        const int burst = max_iter - it < 32 ? max_iter - it : 32;
        for (int k = 0; k < burst; k++) {
            hipLaunchKernelGGL(spmv_kernel, dim3(nb), dim3(kThreads), 0, stream, n, row_off, col, val, psi, y, part);
            hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(64), 0, stream, part, nb, scal);
            hipLaunchKernelGGL(update_kernel, dim3(nb), dim3(kThreads), 0, stream, n, y, psi, scal, part);
        }
        it += burst;
        hipLaunchKernelGGL(residual_kernel, dim3(1), dim3(64), 0, stream, part, nb, scal);
        PN_CHECK_HIP(hipMemcpyAsync(host, scal, sizeof host, hipMemcpyDeviceToHost, stream));
        PN_CHECK_HIP(hipStreamSynchronize(stream));
        if (!(host[0] > 0.0) || !std::isfinite(host[1]))
            PN_FAIL(PN_ERR_FORMAT, "MERW power iteration broke down (|y| = %g): empty or non-finite adjacency", host[0]);
        converged = host[2] <= tol * std::fabs(host[1]);      // |A x - lambda x| <= tol (lambda + 1), x the previous iterate
    }
    const double lam = host[1] - 1.0;
    *lambda = lam;
    if (iters) *iters = it;
    if (!converged) PN_FAIL(PN_ERR_ARG, "MERW power iteration: residual %g after %d iterations (tol %g): disconnected graph "
                                        "or tiny spectral gap", host[2], it, tol);
    hipLaunchKernelGGL(prob_kernel, dim3(nb), dim3(kThreads), 0, stream, n, row_off, col, val, psi, lam, p);
    PN_CHECK_HIP(hipGetLastError());
    return PN_OK;
}

}  // extern "C"
