// What a fork (main -> side) and a join (side -> main) cost between kernels on two HIP streams, per mechanism:
//   0  hipEventRecord(main) + hipStreamWaitEvent(side)                         (what pn_context does)
//   1  the event bound to the kernel itself: hipExtLaunchKernelGGL(..., stopEvent) + hipStreamWaitEvent(side)
//   2  hipStreamWriteValue32(main) + hipStreamWaitValue32(side) on signal memory
//   3  as 0 with events created with hipEventDisableSystemFence (no system-scope release when the event is recorded)
// Modes 0 / 1 / 3 are run twice: with kernels that only spin, and with kernels that leave 64 MB of fresh stores behind.
// Every kernel stamps wall_clock64() (100 MHz) at its start and end; printed: the gaps K1 end -> K2 start (same stream, across the
// fork), K1 end -> K3 start (other stream), and the same for the join K3 end -> K4 start on main.
//   hipcc -O2 --offload-arch=gfx950 tools/event_hop_probe.hip -o /tmp/event_hop_probe && /tmp/event_hop_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void spin_kernel(long long *stamps, int slot, int ticks, float *dirty = nullptr, int per_thread = 0) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot] = t0;
    if (dirty)
        for (int i = 0; i < per_thread; i++) dirty[((size_t)i * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x] = (float)t0;
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2 * slot + 1] = wall_clock64();
}

int main() {
    hipStream_t mainq, side;
    CK(hipStreamCreateWithFlags(&mainq, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    hipEvent_t fork, join, fork_sys, join_sys, fork_dev, join_dev;
    CK(hipEventCreateWithFlags(&fork_sys, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join_sys, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&fork_dev, hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipEventCreateWithFlags(&join_dev, hipEventDisableTiming | hipEventDisableSystemFence));
    float *dirty_buf;
    CK(hipMalloc(&dirty_buf, (size_t)128 << 20));
    long long *stamps;
    const int REPS = 200;
    CK(hipMalloc(&stamps, sizeof(long long) * 2 * 5 * REPS));
    uint32_t *sig = nullptr;
    bool have_sig = hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory) == hipSuccess;
    if (have_sig) CK(hipMemset(sig, 0, 8));
    std::vector<long long> h(2 * 5 * REPS);
    for (int run = 0; run < 8; run++) {
        const int mode_in = run % 4, writes = run / 4;
        if (mode_in == 2 && (!have_sig || writes)) { if (!writes) printf("mode 2: no signal memory\n"); continue; }
        const int mode = mode_in == 3 ? 0 : mode_in;
        fork = mode_in == 3 ? fork_dev : fork_sys;
        join = mode_in == 3 ? join_dev : join_sys;
        float *d1 = writes ? dirty_buf : nullptr, *d3 = writes ? dirty_buf + (16 << 20) : nullptr;
        const int pt1 = writes ? 256 : 0, pt3 = writes ? 2048 : 0;       // 64 MB each: 256 x 256 threads x 256, 32 x 256 threads x 2048
        for (int pass = 0; pass < 2; pass++) {      // pass 0 warms up
            for (int i = 0; i < REPS; i++) {
                long long *st = stamps + 10 * i;
                // K0 main (so that K1 is not the first of the step) ; K1 main ; fork ; K2 main, K3 side ; join ; K4 main
                hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, mainq, st, 0, 2000, (float *)nullptr, 0);
                if (mode == 1)
                    hipExtLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, mainq, nullptr, fork, 0, st, 1, 2000, d1, pt1);
                else
                    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, mainq, st, 1, 2000, d1, pt1);
                if (mode == 0) CK(hipEventRecord(fork, mainq));
                if (mode == 2) {
                    CK(hipStreamWriteValue32(mainq, sig, 2 * (pass * REPS + i) + 1, 0));
                    CK(hipStreamWaitValue32(side, sig, 2 * (pass * REPS + i) + 1, hipStreamWaitValueGte, 0xffffffffu));
                } else {
                    CK(hipStreamWaitEvent(side, fork, 0));
                }
                hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(256), 0, mainq, st, 2, 6000, (float *)nullptr, 0);       // 60 us on main
                if (mode == 1)
                    hipExtLaunchKernelGGL(spin_kernel, dim3(32), dim3(256), 0, side, nullptr, join, 0, st, 3, 3000, d3, pt3);   // 30 us on side
                else
                    hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(256), 0, side, st, 3, 3000, d3, pt3);
                if (mode == 0) CK(hipEventRecord(join, side));
                if (mode == 2) {
                    CK(hipStreamWriteValue32(side, sig + 1, 2 * (pass * REPS + i) + 1, 0));
                    CK(hipStreamWaitValue32(mainq, sig + 1, 2 * (pass * REPS + i) + 1, hipStreamWaitValueGte, 0xffffffffu));
                } else {
                    CK(hipStreamWaitEvent(mainq, join, 0));
                }
                hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, mainq, st, 4, 2000, (float *)nullptr, 0);
            }
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), stamps, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
        std::vector<double> g01, g12, g13, g24, dur;
        for (int i = 0; i < REPS; i++) {
            const long long *s = h.data() + 10 * i;
            g01.push_back((s[2] - s[1]) / 100.0);       // plain same-stream gap
            g12.push_back((s[4] - s[3]) / 100.0);       // across the fork on main
            g13.push_back((s[6] - s[3]) / 100.0);       // main -> side
            g24.push_back((s[8] - s[5]) / 100.0);       // across the join on main (the side kernel ended 30 us earlier)
            dur.push_back((s[9] - s[0]) / 100.0);
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("mode %d%s: plain gap %.2f | fork on main %.2f | fork to side %.2f | join on main %.2f | chain %.2f us (ideal 120)\n", mode_in, writes ? " +64 MB stores" : "",
               med(g01), med(g12), med(g13), med(g24), med(dur));
    }
    return 0;
}
