// How long a workgroup waits for 20 KB it asks for at once, by the width of the loads (the first phase of the fused pooling step:
// 1 299 workgroups of 256 threads, each reading its node's 40 x 128 floats, six workgroups per CU):
//   b32   twenty 4-byte loads per lane (lane = column, as pool_step2_kernel did until round 6's last change)
//   b128  five 16-byte loads per lane  (half a wave per row)
// cold: 1 GB written in between (the rows come from HBM); hot: the launch repeated.
//   hipcc -O2 --offload-arch=gfx950 tools/load_width_probe.hip -o tools/_bin/load_width_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);    \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

__global__ __launch_bounds__(256, 6) void rows_b32(const float *__restrict__ hn, float *__restrict__ out, long long *stamp) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long t0 = wall_clock64();
    const float *g = hn + (size_t)blockIdx.x * 40 * 128;
    float v[10][2];
#pragma unroll
    for (int k = 0; k < 10; k++)
#pragma unroll
        for (int i = 0; i < 2; i++) v[k][i] = g[(wave + 4 * k) * 128 + lane + 64 * i];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 10; k++) s += v[k][0] * v[k][1];
    out[(size_t)blockIdx.x * 256 + tid] = s;
    __syncthreads();
    if (tid == 0) stamp[blockIdx.x] = wall_clock64() - t0;
}

__global__ __launch_bounds__(256, 6) void rows_b128(const float *__restrict__ hn, float *__restrict__ out, long long *stamp) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long t0 = wall_clock64();
    const float4 *g = reinterpret_cast<const float4 *>(hn + (size_t)blockIdx.x * 40 * 128);
    float4 v[5];
#pragma unroll
    for (int k = 0; k < 5; k++) v[k] = g[(wave + 4 * (2 * k + (lane >> 5))) * 32 + (lane & 31)];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; k++) s += v[k].x * v[k].y + v[k].z * v[k].w;
    out[(size_t)blockIdx.x * 256 + tid] = s;
    __syncthreads();
    if (tid == 0) stamp[blockIdx.x] = wall_clock64() - t0;
}

__global__ void fill_kernel(float4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main() {
    for (int S : {1299, 9464}) {
        float *hn, *out, *junk;
        long long *stamp;
        CK(hipMalloc(&hn, (size_t)S * 40 * 128 * 4));
        CK(hipMalloc(&out, (size_t)S * 256 * 4));
        CK(hipMalloc(&stamp, (size_t)S * 8));
        CK(hipMalloc(&junk, (size_t)1 << 30));
        hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (float4 *)hn, (size_t)S * 40 * 32);
        std::vector<long long> h(S);
        for (int width = 0; width < 2; width++)
            for (int hot = 0; hot < 2; hot++) {
                std::vector<double> med, span;
                for (int rep = 0; rep < 7; rep++) {
                    if (!hot) hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, (float4 *)junk, (size_t)1 << 26);
                    else hipLaunchKernelGGL(width ? rows_b128 : rows_b32, dim3(S), dim3(256), 0, 0, hn, out, stamp);
                    hipEvent_t a, b;
                    CK(hipEventCreate(&a));
                    CK(hipEventCreate(&b));
                    CK(hipEventRecord(a, 0));
                    hipLaunchKernelGGL(width ? rows_b128 : rows_b32, dim3(S), dim3(256), 0, 0, hn, out, stamp);
                    CK(hipEventRecord(b, 0));
                    CK(hipDeviceSynchronize());
                    float ms;
                    CK(hipEventElapsedTime(&ms, a, b));
                    CK(hipMemcpy(h.data(), stamp, (size_t)S * 8, hipMemcpyDeviceToHost));
                    std::sort(h.begin(), h.end());
                    med.push_back(h[S / 2] / 100.0);
                    span.push_back(ms * 1000.0);
                    CK(hipEventDestroy(a));
                    CK(hipEventDestroy(b));
                }
                std::sort(med.begin(), med.end());
                std::sort(span.begin(), span.end());
                printf("S=%d %s %s: a workgroup waits %.2f us (median of medians), launch %.1f us\n", S, width ? "b128" : "b32 ", hot ? "hot " : "cold",
                       med[3], span[3]);
            }
        CK(hipFree(hn)); CK(hipFree(out)); CK(hipFree(stamp)); CK(hipFree(junk));
    }
    return 0;
}
