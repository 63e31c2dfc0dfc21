// In what order does v_mfma_f32_32x32x2_f32 (and 16x16x4) add its k terms?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_order_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// For every output element the result is compared bit for bit with the candidate orders of an fma chain from C.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A[32][2], B[2][32], C/D[32][32]; lane l: a = A[l%32][l/32], b = B[l/32][l%32]; D element r of lane l: row (r%4) + 8*(r/4) + 4*(l/32), col l%32
__global__ void k32(const float* A, const float* B, const float* C, float* D) {
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
// A[16][4], B[4][16]: lane l: a = A[l%16][l/16], b = B[l/16][l%16]; D element r of lane l: row 4*(l/16) + r, col l%16
__global__ void k16(const float* A, const float* B, const float* C, float* D) {
    const int l = threadIdx.x;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(4 * (l >> 4) + r) * 16 + (l & 15)];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}

static float rnd() { return (float)((rand() / (double)RAND_MAX * 2 - 1) * std::ldexp(1.0, rand() % 12 - 6)); }

int main() {
    srand(1);
    int trials = 200;
    long n_asc = 0, n_desc = 0, n_unfused = 0, n_tot = 0, n16[24] = {0}, n16_tot = 0;
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 4096), hipMalloc(&dB, 4096), hipMalloc(&dC, 4096), hipMalloc(&dD, 4096);
    for (int t = 0; t < trials; ++t) {
        std::vector<float> A(64), B(64), C(1024), D(1024);
        for (auto& v : A) v = rnd();
        for (auto& v : B) v = rnd();
        for (auto& v : C) v = (t % 2) ? rnd() : 0.f;
        hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice), hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                const float c = C[i * 32 + j], a0 = A[i * 2], a1 = A[i * 2 + 1], b0 = B[j], b1 = B[32 + j], d = D[i * 32 + j];
                const float asc = std::fmaf(a1, b1, std::fmaf(a0, b0, c)), desc = std::fmaf(a0, b0, std::fmaf(a1, b1, c));
                const float unf = (c + a0 * b0) + a1 * b1;
                n_asc += !memcmp(&d, &asc, 4), n_desc += !memcmp(&d, &desc, 4), n_unfused += !memcmp(&d, &unf, 4), ++n_tot;
            }
        // 16x16x4: all 24 orders of the four terms as an fma chain from c
        std::vector<float> A4(64), B4(64), C4(256), D4(256);
        for (auto& v : A4) v = rnd();
        for (auto& v : B4) v = rnd();
        for (auto& v : C4) v = (t % 2) ? rnd() : 0.f;
        hipMemcpy(dA, A4.data(), 256, hipMemcpyHostToDevice), hipMemcpy(dB, B4.data(), 256, hipMemcpyHostToDevice), hipMemcpy(dC, C4.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D4.data(), dD, 1024, hipMemcpyDeviceToHost);
        int perm[4] = {0, 1, 2, 3}, pi = 0;
        do {
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    float s = C4[i * 16 + j];
                    for (int q = 0; q < 4; ++q) s = std::fmaf(A4[i * 4 + perm[q]], B4[perm[q] * 16 + j], s);
                    n16[pi] += !memcmp(&s, &D4[i * 16 + j], 4);
                }
            ++pi;
        } while (std::next_permutation(perm, perm + 4));
        n16_tot += 256;
    }
    printf("32x32x2: of %ld outputs  fma k0-then-k1 %ld   fma k1-then-k0 %ld   unfused asc %ld\n", n_tot, n_asc, n_desc, n_unfused);
    int perm[4] = {0, 1, 2, 3}, pi = 0;
    do {
        if (n16[pi] > n16_tot * 9 / 10) printf("16x16x4: order %d%d%d%d matches %ld of %ld\n", perm[0], perm[1], perm[2], perm[3], (long)n16[pi], n16_tot);
        ++pi;
    } while (std::next_permutation(perm, perm + 4));
    printf("16x16x4: best ascending(0123) %ld of %ld\n", (long)n16[0], n16_tot);
    return 0;
}
