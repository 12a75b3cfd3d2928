// Measured MFMA ceilings of this GPU for the two instructions the engine's GEMM-shaped kernels use
// (v_mfma_f64_16x16x4_f64, v_mfma_f32_32x32x2_f32): register-resident operands, 8 independent accumulator chains per
// wavefront, every SIMD of every CU busy -- nothing but matrix-pipe issue.  The guide quotes the fp32 peak (157.3
// TFLOP/s); the fp64 figure is not in it, so bench.py quotes the number measured here (profiles/r02_mfma_peak.json).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/mfma_peak.hip -o scripts/micro/mfma_peak && scripts/micro/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k_f64(double *out, int iters, double a, double b)
{
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    const double x = a + threadIdx.x * 1e-9, y = b;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c3, 0, 0, 0);
        c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c4, 0, 0, 0);
        c5 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c5, 0, 0, 0);
        c6 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c6, 0, 0, 0);
        c7 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c7, 0, 0, 0);
    }
    d4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ void __launch_bounds__(256) k_f32(float *out, int iters, float a, float b)
{
    f16v c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    const float x = a + threadIdx.x * 1e-6f, y = b;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 8; // 2 workgroups of 4 wavefronts per SIMD... 8 per CU
    double *d64;
    float *d32;
    hipMalloc(&d64, (size_t)blocks * 256 * 8);
    hipMalloc(&d32, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    double best64 = 0, best32 = 0;
    for (int rep = 0; rep < 4; ++rep) {
        float ms;
        hipEventRecord(e0);
        k_f64<<<blocks, 256>>>(d64, iters, 1e-3, 1e-3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double tf64 = (double)blocks * 4 * iters * 8 * (16.0 * 16 * 4 * 2) / (ms * 1e-3) / 1e12;
        hipEventRecord(e0);
        k_f32<<<blocks, 256>>>(d32, iters, 1e-3f, 1e-3f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double tf32 = (double)blocks * 4 * iters * 4 * (32.0 * 32 * 2 * 2) / (ms * 1e-3) / 1e12;
        if (rep > 0) { best64 = tf64 > best64 ? tf64 : best64; best32 = tf32 > best32 ? tf32 : best32; }
    }
    std::printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d, \"mfma_f64_16x16x4_tflops\": %.2f, "
                "\"mfma_f32_32x32x2_tflops\": %.2f, \"method\": \"register-resident operands, %d iterations x 8 (f64) / 4 (f32) "
                "independent accumulator chains per wavefront, %d workgroups of 256 threads, best of 3 timed launches (HIP events)\"}\n",
                p.name, cus, p.clockRate / 1000, best64, best32, iters, blocks);
    return 0;
}
