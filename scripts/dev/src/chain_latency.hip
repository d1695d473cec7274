// Dev: what a step of a serial decode chain costs when the bits it needs come (a) from a dependent GLOBAL load of 16 bytes at a
// cursor that moves a few bytes per step (what the sequence chains do), (b) from LDS, (c) from registers only.  One wave per workgroup,
// few workgroups: latency, not throughput.   build: hipcc -O3 --offload-arch=gfx950 -o /tmp/chain_latency scripts/dev/src/chain_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
template <int MODE>
__global__ __launch_bounds__(64) void chain(const u32* __restrict__ stream, u32 nsteps, u32* out) {
    __shared__ u32 tab[2048];
    __shared__ u32 ring[64 * 64];
    for (u32 i = threadIdx.x; i < 2048; i += 64) tab[i] = (i * 2654435761u) >> 7;
    for (u32 i = threadIdx.x; i < 64 * 64; i += 64) ring[i] = stream[i];
    __syncthreads();
    const u32* w = stream + (size_t)(blockIdx.x * 64 + threadIdx.x) * 65536 + 60000;
    u32 s = threadIdx.x, c = 60000 * 32, acc = 0;
    uint4 x = *(const uint4*)(w - 3);
    for (u32 k = 0; k < nsteps; k++) {
        const u32 e = tab[s & 2047];                       // the state's table entry (LDS, dependent on s)
        const u32 T = 8 + (e & 31);                        // bits this step consumes
        const u32 bit = c & 31;
        const u32 u = (u32)((((unsigned long long)x.w << 32) | x.z) >> bit) & 0x1FF;   // state bits out of the window
        s = (e >> 5) + u;
        c -= T;
        if (MODE == 0) { x = *(const uint4*)(stream + (size_t)(blockIdx.x * 64 + threadIdx.x) * 65536 + (c >> 5) - 3); }          // dependent global load
        else if (MODE == 1) { const u32 at = ((c >> 5) & 63) + 64 * (threadIdx.x & 63); x = make_uint4(ring[(at) & 4095], ring[(at + 1) & 4095], ring[(at + 2) & 4095], ring[(at + 3) & 4095]); }  // LDS ring
        else { x.z ^= e; x.w += T; }
        acc += u;
        if (c < 4096) c = 60000 * 32;
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc + s + x.x;
}
int main() {
    const size_t n = (size_t)64 * 64 * 65536;
    u32* d; u32* o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 64 * 64 * 4);
    std::vector<u32> h(1 << 20);
    for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(i * 2246822519u) ^ (u32)(i >> 3);
    for (size_t off = 0; off < n; off += h.size()) hipMemcpy(d + off, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const u32 steps = 20000;
    for (int wgs : {1, 64}) for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(wgs), dim3(64), 0, 0, d, steps, o);
            else if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(wgs), dim3(64), 0, 0, d, steps, o);
            else hipLaunchKernelGGL(chain<2>, dim3(wgs), dim3(64), 0, 0, d, steps, o);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("workgroups %2d  %s: %.1f ns per step\n", wgs, mode == 0 ? "global window" : mode == 1 ? "LDS window   " : "registers    ", ms * 1e6 / steps);
        }
    }
    return 0;
}
