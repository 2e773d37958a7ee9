// Probe of the v_mfma_f64_16x16x4_f64 operand / result layout on gfx950 (prints which (i, j) every lane's 4 result
// registers hold when A[i][k] is taken from lane i + 16 k and B[k][j] from lane j + 16 k).
// hipcc --offload-arch=gfx950 -O2 tools/experiments/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D) {
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[l * 4 + v] = acc[v];
}
int main() {
    double hA[64], hB[64], ref[256], hD[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1.0 + i * 0.37; hB[i] = 2.0 - i * 0.11 + (i % 7); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    double *A, *B, *D; hipMalloc(&A, 512); hipMalloc(&B, 512); hipMalloc(&D, 2048);
    hipMemcpy(A, hA, 512, hipMemcpyHostToDevice); hipMemcpy(B, hB, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(A, B, D); hipMemcpy(hD, D, 2048, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5) for (int v = 0; v < 4; ++v) {
        int fi = -1, fj = -1;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (fabs(ref[i * 16 + j] - hD[l * 4 + v]) < 1e-9 * fabs(ref[i * 16 + j])) { fi = i; fj = j; }
        printf("lane %2d v %d -> (i %2d, j %2d)\n", l, v, fi, fj);
    }
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
        if (fabs(hD[l * 4 + v] - ref[(4 * (l / 16) + v) * 16 + l % 16]) > 1e-9) okA = 0;
        if (fabs(hD[l * 4 + v] - ref[((l / 16) + 4 * v) * 16 + l % 16]) > 1e-9) okB = 0;
    }
    printf("layout i = 4 (lane / 16) + v: %d;  layout i = lane / 16 + 4 v: %d\n", okA, okB);
    return 0;
}
