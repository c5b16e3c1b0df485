// Micro-benchmark: throughput of LDS atomics / plain RMW on gfx950 for the scatter pattern of ds_jaccard.hip.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_atomics scripts/micro/lds_atomics.hip && /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kTile = 32768;

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void kernel(const uint16_t *rows, int per_thread, int rounds, float *out,
                                                  unsigned long long *cycles)
{
    extern __shared__ float scores[];
    for (int i = threadIdx.x; i < kTile + 64; i += THREADS) scores[i] = 0.f;
    __syncthreads();
    unsigned int *iscores = reinterpret_cast<unsigned int *>(scores);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < per_thread; ++i) {
            const int row = rows[(size_t)(i * THREADS + threadIdx.x)];
            if (MODE == 0) atomicAdd(&scores[row], 1.5f);                       // ds_add_f32
            if (MODE == 1) atomicAdd(&iscores[row], 3u);                        // ds_add_u32
            if (MODE == 2) scores[row] = scores[row] + 1.5f;                    // racy read-modify-write
            if (MODE == 3) scores[row] = 1.5f;                                  // plain store
            if (MODE == 4) out[threadIdx.x] += atomicExch(&scores[row], 0.f);   // ds_wrxchg_rtn_b32
            if (MODE == 5) out[threadIdx.x] += atomicAdd(&scores[row], 1.5f);   // ds_add_rtn_f32
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * THREADS + threadIdx.x] += scores[threadIdx.x];
}

template <int MODE, int THREADS>
void run(const char *name, const uint16_t *d_rows, int per_thread, float *d_out, unsigned long long *d_cycles)
{
    const int rounds = 20, blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kernel<MODE, THREADS>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (kTile + 64) * 4);
    hipLaunchKernelGGL((kernel<MODE, THREADS>), dim3(blocks), dim3(THREADS), (kTile + 64) * 4, 0, d_rows, per_thread,
                       rounds, d_out, d_cycles);
    hipDeviceSynchronize();
    std::vector<unsigned long long> cycles(blocks);
    hipMemcpy(cycles.data(), d_cycles, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto c : cycles) mean += c;
    mean /= blocks;
    const double ops = (double)rounds * per_thread * THREADS;
    printf("%-28s threads=%4d  %8.0f cycles  %.3f lane-ops/cycle/CU  (%.1f cycles per wave-instruction)\n", name,
           THREADS, mean, ops / mean, mean / (ops / 64.0) * (THREADS / 64) / (THREADS / 64));
}

int main()
{
    const int per_thread = 64, max_threads = 1024;
    std::vector<uint16_t> random_rows((size_t)per_thread * max_threads), sorted_rows(random_rows.size());
    srand(1);
    for (auto &r : random_rows) r = rand() % kTile;
    // "posting-like": ascending rows with random gaps (density ~0.27), lanes take consecutive postings
    int row = 0;
    for (size_t i = 0; i < sorted_rows.size(); ++i) { row = (row + 1 + rand() % 6) % kTile; sorted_rows[i] = row; }
    uint16_t *d_random, *d_sorted;
    float *d_out;
    unsigned long long *d_cycles;
    hipMalloc(&d_random, random_rows.size() * 2);
    hipMalloc(&d_sorted, sorted_rows.size() * 2);
    hipMalloc(&d_out, 256 * max_threads * 4);
    hipMalloc(&d_cycles, 256 * 8);
    hipMemcpy(d_random, random_rows.data(), random_rows.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(d_sorted, sorted_rows.data(), sorted_rows.size() * 2, hipMemcpyHostToDevice);
    hipMemset(d_out, 0, 256 * max_threads * 4);
    for (int pass = 0; pass < 2; ++pass) {
        const uint16_t *rows = pass == 0 ? d_random : d_sorted;
        printf("---- %s rows\n", pass == 0 ? "uniform random" : "posting-like ascending");
        run<0, 1024>("ds_add_f32", rows, per_thread, d_out, d_cycles);
        run<0, 512>("ds_add_f32", rows, per_thread, d_out, d_cycles);
        run<0, 256>("ds_add_f32", rows, per_thread, d_out, d_cycles);
        run<1, 1024>("ds_add_u32", rows, per_thread, d_out, d_cycles);
        run<1, 512>("ds_add_u32", rows, per_thread, d_out, d_cycles);
        run<2, 1024>("read+add+write (racy)", rows, per_thread, d_out, d_cycles);
        run<3, 1024>("ds_write_b32", rows, per_thread, d_out, d_cycles);
        run<4, 1024>("ds_wrxchg_rtn_b32", rows, per_thread, d_out, d_cycles);
        run<4, 512>("ds_wrxchg_rtn_b32", rows, per_thread, d_out, d_cycles);
        run<5, 512>("ds_add_rtn_f32", rows, per_thread, d_out, d_cycles);
    }
    return 0;
}
