// Issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950 (round-5 question: is the packed fp32 form full rate?).
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/pk_rate tools/micro/pk_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void k(float* out, int iters, float a, float b) {
    v2f x0 = {threadIdx.x * 1e-3f, 1.0f}, x1 = {2.0f, 3.0f}, x2 = {4.0f, 5.0f}, x3 = {6.0f, 7.0f};
    v2f x4 = {8.0f, 9.0f}, x5 = {1.5f, 2.5f}, x6 = {3.5f, 4.5f}, x7 = {5.5f, 6.5f};
    const v2f va = {a, a}, vb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 16 scalar fmas
            x0.x = __builtin_fmaf(x0.x, a, b); x0.y = __builtin_fmaf(x0.y, a, b); x1.x = __builtin_fmaf(x1.x, a, b); x1.y = __builtin_fmaf(x1.y, a, b);
            x2.x = __builtin_fmaf(x2.x, a, b); x2.y = __builtin_fmaf(x2.y, a, b); x3.x = __builtin_fmaf(x3.x, a, b); x3.y = __builtin_fmaf(x3.y, a, b);
            x4.x = __builtin_fmaf(x4.x, a, b); x4.y = __builtin_fmaf(x4.y, a, b); x5.x = __builtin_fmaf(x5.x, a, b); x5.y = __builtin_fmaf(x5.y, a, b);
            x6.x = __builtin_fmaf(x6.x, a, b); x6.y = __builtin_fmaf(x6.y, a, b); x7.x = __builtin_fmaf(x7.x, a, b); x7.y = __builtin_fmaf(x7.y, a, b);
        } else {                  // 8 packed fmas (the same 16 flop pairs)
            x0 = __builtin_elementwise_fma(x0, va, vb); x1 = __builtin_elementwise_fma(x1, va, vb);
            x2 = __builtin_elementwise_fma(x2, va, vb); x3 = __builtin_elementwise_fma(x3, va, vb);
            x4 = __builtin_elementwise_fma(x4, va, vb); x5 = __builtin_elementwise_fma(x5, va, vb);
            x6 = __builtin_elementwise_fma(x6, va, vb); x7 = __builtin_elementwise_fma(x7, va, vb);
        }
    }
    const v2f s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main() {
    float* d;
    hipMalloc(&d, sizeof(float) * 256 * 2048);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
            else hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, d, iters, 0.999f, 0.001f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fma = 16.0 * iters * 256.0 * 2048.0;
            if (rep) printf("{\"mode\": \"%s\", \"ms\": %.3f, \"TFLOPs\": %.1f, \"vector_instructions_per_iteration\": %d}\n", mode ? "v_pk_fma_f32" : "v_fma_f32", ms, 2.0 * fma / ms * 1e-9, mode ? 8 : 16);
        }
    }
    return 0;
}
