// Calibration micro-kernels for the bound model of ds_jaccard_topk_kernel:
//   issue rates   VALU (v_fma_f32), SALU (s_add_u32) and LDS atomics (ds_add_u32 / ds_and_rtn_b32 on random words of a
//                 14,352-word tile, the kernel's scatter / collect pattern) in cycles per wave-instruction, at 1, 2 and
//                 4 waves per SIMD -- settles the SIMD width question (2 cycles per VALU wave-instruction = SIMD-32);
//   FETCH_SIZE    known-byte-count global reads in the kernel's own access shapes (8 B per lane posting quads, 16 B per
//                 lane sums, 4 B per lane, random 16 B and 2 B gathers): run under
//                 `rocprofv3 --kernel-trace --pmc FETCH_SIZE` to get the counter's factor per shape.
// Build (cross-compiles without a GPU): hipcc --offload-arch=gfx950 -O3 -o scripts/micro/bin/calibrate scripts/micro/calibrate.hip
// Run on the GPU box: scripts/micro/bin/calibrate [issue|fetch]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(call)                                                                              \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));                    \
            exit(1);                                                                             \
        }                                                                                        \
    } while (0)

// ---- issue rates -------------------------------------------------------------------------------------------------------
constexpr int kIssueRounds = 2000;

__global__ void valu_kernel(float *out, unsigned long long *cycles, float x, float y)
{
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < kIssueRounds; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // 32 independent-enough fused multiply-adds per round (8 chains)
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a4) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a5) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a6) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a7) : "v"(x), "v"(y));
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void salu_kernel(float *out, unsigned long long *cycles)
{
    uint32_t s0 = 0, s1 = 1, s2 = 2, s3 = 3, s4 = 4, s5 = 5, s6 = 6, s7 = 7;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < kIssueRounds; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s1) : : "scc");
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s2) : : "scc");
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s3) : : "scc");
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s4) : : "scc");
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s5) : : "scc");
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s6) : : "scc");
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s7) : : "scc");
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<float>(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7);
}

// a VALU stream and a SALU stream in the same wave: do the two pipes issue side by side?
__global__ void mixed_kernel(float *out, unsigned long long *cycles, float x, float y)
{
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f;
    uint32_t s0 = 0, s1 = 1, s2 = 2, s3 = 3;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < kIssueRounds; ++r) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // 16 VALU + 16 SALU per round
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s0) : : "scc");
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s1) : : "scc");
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s2) : : "scc");
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(x), "v"(y));
            asm volatile("s_add_u32 %0, %0, 3" : "+s"(s3) : : "scc");
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + static_cast<float>(s0 + s1 + s2 + s3);
}

constexpr int kTileWords = 28672 / 2 + 16;  // the kernel's packed score tile
constexpr int kLdsRounds = 200, kLdsPerThread = 32;

template <int MODE>
__global__ void lds_kernel(const uint16_t *rows, float *out, unsigned long long *cycles)
{
    __shared__ uint32_t tile[kTileWords];
    for (int i = threadIdx.x; i < kTileWords; i += blockDim.x) tile[i] = 0u;
    uint32_t local[kLdsPerThread];
    for (int i = 0; i < kLdsPerThread; ++i) local[i] = rows[i * blockDim.x + threadIdx.x];
    uint32_t sum = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < kLdsRounds; ++r) {
#pragma unroll
        for (int i = 0; i < kLdsPerThread; ++i) {
            const uint32_t row = local[i];
            if (MODE == 0) atomicAdd(&tile[row >> 1], 3u << ((row & 1u) << 4));                        // ds_add_u32
            if (MODE == 1) sum += atomicAnd(&tile[row >> 1], ~(0xffffu << ((row & 1u) << 4)));          // ds_and_rtn_b32
            if (MODE == 2) sum += tile[row >> 1];                                                       // ds_read_b32
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<float>(sum + tile[threadIdx.x]);
}

static double mean_cycles(unsigned long long *d_cycles, int blocks)
{
    std::vector<unsigned long long> cycles(blocks);
    CHECK(hipMemcpy(cycles.data(), d_cycles, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double mean = 0;
    for (auto c : cycles) mean += static_cast<double>(c);
    return mean / blocks;
}

static void issue_rates()
{
    const int blocks = 256;  // one workgroup per CU
    float *d_out;
    unsigned long long *d_cycles;
    uint16_t *d_rows;
    CHECK(hipMalloc(&d_out, sizeof(float) * blocks * 1024));
    CHECK(hipMalloc(&d_cycles, sizeof(unsigned long long) * blocks));
    std::vector<uint16_t> rows(static_cast<size_t>(kLdsPerThread) * 1024);
    srand(7);
    for (auto &r : rows) r = static_cast<uint16_t>(rand() % 28672);
    CHECK(hipMalloc(&d_rows, rows.size() * sizeof(uint16_t)));
    CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    printf("issue rates, one workgroup per CU; cycles = shader clock (s_memtime); per SIMD = cycles / (wave-instructions "
           "of one wave x waves per SIMD)\n");
    for (int threads : {256, 512, 1024}) {
        const int waves_per_simd = threads / 256;
        for (int repeat = 0; repeat < 2; ++repeat) {  // the second pass is the warm one
            hipLaunchKernelGGL(valu_kernel, dim3(blocks), dim3(threads), 0, 0, d_out, d_cycles, 1.0001f, 0.5f);
            CHECK(hipDeviceSynchronize());
        }
        const double valu = mean_cycles(d_cycles, blocks) / (kIssueRounds * 32.0) / waves_per_simd;
        hipLaunchKernelGGL(salu_kernel, dim3(blocks), dim3(threads), 0, 0, d_out, d_cycles);
        CHECK(hipDeviceSynchronize());
        const double salu_cu = mean_cycles(d_cycles, blocks) / (kIssueRounds * 32.0) / (threads / 64);
        hipLaunchKernelGGL(mixed_kernel, dim3(blocks), dim3(threads), 0, 0, d_out, d_cycles, 1.0001f, 0.5f);
        CHECK(hipDeviceSynchronize());
        const double mixed = mean_cycles(d_cycles, blocks) / (kIssueRounds * 16.0) / waves_per_simd;
        printf("waves/SIMD=%d  v_fma_f32 %.2f cycles per wave-instruction per SIMD | s_add_u32 %.2f cycles per "
               "instruction per CU (%.2f per SIMD) | 1 VALU + 1 SALU pair %.2f cycles per SIMD\n",
               waves_per_simd, valu, salu_cu, salu_cu * 4, mixed);
        const char *names[3] = {"ds_add_u32 (scatter)", "ds_and_rtn_b32 (collect)", "ds_read_b32 (gather)"};
        for (int mode = 0; mode < 3; ++mode) {
            if (mode == 0) hipLaunchKernelGGL(lds_kernel<0>, dim3(blocks), dim3(threads), 0, 0, d_rows, d_out, d_cycles);
            if (mode == 1) hipLaunchKernelGGL(lds_kernel<1>, dim3(blocks), dim3(threads), 0, 0, d_rows, d_out, d_cycles);
            if (mode == 2) hipLaunchKernelGGL(lds_kernel<2>, dim3(blocks), dim3(threads), 0, 0, d_rows, d_out, d_cycles);
            CHECK(hipDeviceSynchronize());
            const double per_cu = mean_cycles(d_cycles, blocks) / (static_cast<double>(kLdsRounds) * kLdsPerThread) / (threads / 64);
            printf("    %-26s random rows of a 28672-row packed tile: %.2f cycles per wave-instruction per CU\n",
                   names[mode], per_cu);
        }
    }
    // clock: shader cycles of a long VALU kernel against its wall time
    hipEvent_t start, stop;
    CHECK(hipEventCreate(&start));
    CHECK(hipEventCreate(&stop));
    CHECK(hipEventRecord(start, 0));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(valu_kernel, dim3(blocks), dim3(1024), 0, 0, d_out, d_cycles, 1.0001f, 0.5f);
    CHECK(hipEventRecord(stop, 0));
    CHECK(hipEventSynchronize(stop));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, start, stop));
    printf("clock: %.0f shader cycles per VALU launch, %.3f ms per launch -> %.2f GHz while every SIMD issues VALU\n",
           mean_cycles(d_cycles, blocks), ms / 20, mean_cycles(d_cycles, blocks) / (ms / 20 * 1e-3) / 1e9);
}

// ---- FETCH_SIZE calibration ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void stream_kernel(const T *data, size_t count, uint32_t *out)
{
    uint32_t sum = 0;
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < count;
         i += static_cast<size_t>(gridDim.x) * blockDim.x) {
        const T value = data[i];
        const uint32_t *words = reinterpret_cast<const uint32_t *>(&value);
        for (int w = 0; w < static_cast<int>(sizeof(T) / 4); ++w) sum += words[w];
    }
    if (sum == 0x12345678u) out[0] = sum;
}

template <typename T>
__global__ void gather_kernel(const T *data, size_t count, size_t gathers_per_thread, uint32_t *out)
{
    uint64_t state = (blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x) * 0x9e3779b97f4a7c15ull + 1;
    uint32_t sum = 0;
    for (size_t g = 0; g < gathers_per_thread; ++g) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        const T value = data[(state >> 20) % count];
        sum += *reinterpret_cast<const uint16_t *>(&value);
    }
    if (sum == 0x12345678u) out[0] = sum;
}

static void fetch_calibration()
{
    const size_t bytes = size_t(4) << 30;  // 4 GiB: far beyond the 256 MiB Infinity Cache
    uint8_t *d_data;
    uint32_t *d_out;
    CHECK(hipMalloc(&d_data, bytes));
    CHECK(hipMalloc(&d_out, 64));
    CHECK(hipMemset(d_data, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const int blocks = 256 * 8, threads = 256;
    hipLaunchKernelGGL(stream_kernel<uint32_t>, dim3(blocks), dim3(threads), 0, 0, reinterpret_cast<uint32_t *>(d_data), bytes / 4, d_out);
    hipLaunchKernelGGL(stream_kernel<uint2>, dim3(blocks), dim3(threads), 0, 0, reinterpret_cast<uint2 *>(d_data), bytes / 8, d_out);
    hipLaunchKernelGGL(stream_kernel<uint4>, dim3(blocks), dim3(threads), 0, 0, reinterpret_cast<uint4 *>(d_data), bytes / 16, d_out);
    const size_t gathers = 256;  // per thread: 2048 * 256 * 256 = 134M gathers per kernel
    hipLaunchKernelGGL(gather_kernel<uint4>, dim3(blocks), dim3(threads), 0, 0, reinterpret_cast<uint4 *>(d_data), bytes / 16, gathers, d_out);
    hipLaunchKernelGGL(gather_kernel<uint16_t>, dim3(blocks), dim3(threads), 0, 0, reinterpret_cast<uint16_t *>(d_data), bytes / 2, gathers, d_out);
    CHECK(hipDeviceSynchronize());
    const double n_gathers = static_cast<double>(blocks) * threads * gathers;
    printf("known bytes per kernel: stream_kernel<unsigned int> %.0f | stream_kernel<uint2> %.0f | stream_kernel<uint4> %.0f | "
           "gather_kernel<uint4> requested %.0f (%.0f gathers; %.0f if every gather fetches a 64 B line, %.0f for 128 B) | "
           "gather_kernel<unsigned short> requested %.0f (%.0f / %.0f)\n",
           (double)bytes, (double)bytes, (double)bytes, n_gathers * 16, n_gathers, n_gathers * 64, n_gathers * 128,
           n_gathers * 2, n_gathers * 64, n_gathers * 128);
}

int main(int argc, char **argv)
{
    const char *what = argc > 1 ? argv[1] : "issue";
    if (!strcmp(what, "issue")) issue_rates();
    else if (!strcmp(what, "fetch")) fetch_calibration();
    else { fprintf(stderr, "usage: calibrate [issue|fetch]\n"); return 2; }
    return 0;
}
