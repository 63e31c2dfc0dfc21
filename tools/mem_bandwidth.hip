// Calibration: read bandwidth of this MI355X by working-set size - what the vector-memory path delivers when the data sit in L2
// (<= 4 MB per XCD), in the 256 MB memory-side cache, or in HBM - with the access shapes of the kernels in this repo: 16 bytes per
// lane, (a) wave-contiguous 1 KB per instruction, (b) 512-byte rows at random row positions (the gathers).
// build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/mem_bandwidth.hip -o /tmp/mem_bandwidth && /tmp/mem_bandwidth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// every workgroup streams the whole working set `reps` times, starting at its own offset (all CUs hit the same lines: L2-resident sets
// stay resident; sets larger than the caches stream from HBM once per pass and workgroup)
__global__ __launch_bounds__(256) void k_read_shared(const f32x4* __restrict__ buf, size_t n_vec, int reps, float* out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t start = ((size_t)blockIdx.x * 9973 * 256) % n_vec;
    for (int r = 0; r < reps; ++r)
        for (size_t i = threadIdx.x; i < n_vec; i += 256 * 4) {
            size_t j = start + i;
            j = j >= n_vec ? j - n_vec : j;
            const f32x4 a = buf[j], b = buf[(j + 256) % n_vec], c = buf[(j + 512) % n_vec], d = buf[(j + 768) % n_vec];
            acc += a + b + c + d;
        }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// the chip streams the working set once per pass, every workgroup its own slice (the shape of a copy kernel's read side)
__global__ __launch_bounds__(256) void k_read_split(const f32x4* __restrict__ buf, size_t n_vec, int reps, float* out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t per = n_vec / gridDim.x, lo = per * blockIdx.x;
    for (int r = 0; r < reps; ++r)
        for (size_t i = threadIdx.x; i + 768 < per; i += 1024) {
            acc += buf[lo + i] + buf[lo + i + 256] + buf[lo + i + 512] + buf[lo + i + 768];
        }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// 512-byte rows at pseudo-random positions of the working set: half a wave per row (the aggregation's / the gate's gathers)
__global__ __launch_bounds__(256) void k_read_rows(const f32x4* __restrict__ buf, size_t n_rows, int rows_per_wave, float* out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned long long s = 0x9E3779B97F4A7C15ull * (wave + 1);   // (wave-uniform: the row index costs scalar arithmetic only)
    for (int r = 0; r < rows_per_wave; r += 8) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // 4 instructions x 2 rows in flight
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const size_t row = ((s >> 20) + (lane >> 5) * 7919) & (n_rows - 1);   // n_rows is a power of two
            v[u] = buf[row * 32 + (lane & 31)];
        }
        acc += v[0] + v[1] + v[2] + v[3];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// write side (round 4): the set split over the workgroups and written once with 16-byte stores, wave-contiguous 1 KB per instruction;
// `read_every` > 0 also reads one 16-byte piece per `read_every` written (read_every = 5: the node projection's 1 : 5 mix of h rows in, P rows out)
__global__ __launch_bounds__(256) void k_write_split(f32x4* __restrict__ buf, const f32x4* __restrict__ src, size_t n_vec, int read_every) {
    const size_t per = (n_vec + gridDim.x - 1) / gridDim.x;
    const size_t lo = per * blockIdx.x, hi = lo + per < n_vec ? lo + per : n_vec;
    f32x4 v = {1.f, 2.f, 3.f, 4.f};
    size_t k = 0;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256, ++k) {
        if (read_every > 0 && k % read_every == 0) v += src[(lo / read_every + k / read_every * 256 + threadIdx.x) % ((1ull << 30) / 16)];   // (src is 1 GB)
        buf[i] = v;
    }
}

int main() {
    const size_t max_bytes = 2ull << 30;
    f32x4* buf;
    float* out;
    hipMalloc(&buf, max_bytes);
    hipMalloc(&out, 4);
    hipMemset(buf, 0, max_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int grid = cus * 8;
    printf("CUs %d, grid %d workgroups of 256 threads\n", cus, grid);
    const size_t sizes[] = {1ull << 20, 2ull << 20, 4ull << 20, 16ull << 20, 32ull << 20, 64ull << 20, 128ull << 20, 256ull << 20, 512ull << 20, 2ull << 30};
    for (size_t bytes : sizes) {
        const size_t n_vec = bytes / 16;
        float ms = 0;
        // (1) every workgroup reads the whole set (only for sets that can be cache-resident)
        if (bytes <= (64ull << 20)) {
            const int reps = (int)((256ull << 20) / bytes) > 1 ? (int)((256ull << 20) / bytes) : 1;
            hipLaunchKernelGGL(k_read_shared, dim3(grid), dim3(256), 0, 0, buf, n_vec, 1, out);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_read_shared, dim3(grid), dim3(256), 0, 0, buf, n_vec, reps, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            printf("set %6zu MB  all workgroups read all of it : %8.2f TB/s through L1\n", bytes >> 20, (double)bytes * reps * grid / (ms * 1e-3) / 1e12);
        }
        // (2) the set split over the workgroups, streamed `reps` times
        {
            const int reps = (int)((4ull << 30) / bytes) > 1 ? (int)((4ull << 30) / bytes) : 1;
            hipLaunchKernelGGL(k_read_split, dim3(grid), dim3(256), 0, 0, buf, n_vec, 1, out);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_read_split, dim3(grid), dim3(256), 0, 0, buf, n_vec, reps, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            printf("set %6zu MB  split over the workgroups, %4d passes: %8.2f TB/s\n", bytes >> 20, reps, (double)bytes * reps / (ms * 1e-3) / 1e12);
        }
        // (3) random 512-byte rows of the set
        {
            const int rows_per_wave = 2048;
            const size_t n_rows = bytes / 512;
            hipLaunchKernelGGL(k_read_rows, dim3(grid), dim3(256), 0, 0, buf, n_rows, 64, out);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_read_rows, dim3(grid), dim3(256), 0, 0, buf, n_rows, rows_per_wave, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            const double moved = (double)grid * 4 * rows_per_wave * 512.0;   // rows_per_wave rows per wave (2 per instruction)
            printf("set %6zu MB  random 512-byte rows               : %8.2f TB/s\n", bytes >> 20, moved / (ms * 1e-3) / 1e12);
        }
    }
    // write side: 256 MB (the projection's output at configs[1]), 512 MB (the gate's), 2 GB
    f32x4* src;
    hipMalloc(&src, 1ull << 30);
    hipMemset(src, 0, 1ull << 30);
    for (size_t bytes : {256ull << 20, 512ull << 20, 2ull << 30}) {
        for (int read_every : {0, 5, 1}) {
            const size_t n_vec = bytes / 16;
            float ms = 0, best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(a);
                hipLaunchKernelGGL(k_write_split, dim3(grid), dim3(256), 0, 0, buf, src, n_vec, read_every);
                hipEventRecord(b);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
                if (rep > 0 && ms < best) best = ms;
            }
            const double moved = (double)bytes * (read_every ? 1.0 + 1.0 / read_every : 1.0);
            printf("write %5zu MB once, 16 B per lane, %s: %8.2f TB/s (%.4f ms)\n", bytes >> 20,
                   read_every == 0 ? "stores only           " : read_every == 5 ? "1 read per 5 stores   " : "1 read per store (copy)", moved / (best * 1e-3) / 1e12, best);
        }
        float ms = 0, best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a);
            hipMemsetAsync(buf, 0, bytes, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("write %5zu MB once, hipMemsetAsync                        : %8.2f TB/s (%.4f ms)\n", bytes >> 20, (double)bytes / (best * 1e-3) / 1e12, best);
    }
    return 0;
}
