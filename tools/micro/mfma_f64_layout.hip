// micro-test: operand / accumulator layout of v_mfma_f64_16x16x4_f64 on gfx950, as obca_solver.h assumes it
//   A operand: lane (k = lane >> 4, i = lane & 15) holds A[i][k];  B operand: lane (k, n) holds B[k][n];
//   accumulator register r of lane (g, j) = C[g + 4 r][j]
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_f64_layout mfma_f64_layout.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double *A /*16x4*/, const double *B /*4x16*/, double *C /*16x16*/, double *C2) {
    const int l = threadIdx.x, g = l >> 4, j = l & 15;
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[j * 4 + g], B[g * 16 + j], c, 0, 0, 0);
    for (int r = 0; r < 4; r++) C[(g + 4 * r) * 16 + j] = c[r];
    // chain: C2 = C' * C?  use register kb of c as the B operand of block kb and A = transposed view: C2 = sum_kb A_kb B_kb with A_kb[i][k] = c_kb of lane (k, i) = C[4kb + k][i]
    v4d d = {0, 0, 0, 0};
    for (int kb = 0; kb < 4; kb++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(c[kb], c[kb], d, 0, 0, 0);     // = C' C under the assumed layout
    for (int r = 0; r < 4; r++) C2[(g + 4 * r) * 16 + j] = d[r];
}
int main() {
    double hA[64], hB[64], hC[256], hC2[256], rC[256], rC2[256];
    for (int i = 0; i < 64; i++) { hA[i] = sin(1.0 + 0.37 * i) + 0.01 * i; hB[i] = cos(0.5 + 0.91 * i) - 0.02 * i; }
    for (int i = 0; i < 16; i++) for (int n = 0; n < 16; n++) { double s = 0; for (int q = 0; q < 4; q++) s += hA[i * 4 + q] * hB[q * 16 + n]; rC[i * 16 + n] = s; }
    for (int i = 0; i < 16; i++) for (int n = 0; n < 16; n++) { double s = 0; for (int q = 0; q < 16; q++) s += rC[q * 16 + i] * rC[q * 16 + n]; rC2[i * 16 + n] = s; }
    double *dA, *dB, *dC, *dC2;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dC2, sizeof hC2);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dC2);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost); hipMemcpy(hC2, dC2, sizeof hC2, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 256; i++) { e1 = fmax(e1, fabs(hC[i] - rC[i])); e2 = fmax(e2, fabs(hC2[i] - rC2[i])); }
    printf("mfma_f64 layout: max err product %.3e, chained C'C %.3e  (%s)\n", e1, e2, (e1 < 1e-12 && e2 < 1e-10) ? "layout as assumed" : "LAYOUT DIFFERS");
    if (e1 >= 1e-12) { // try alternatives for the accumulator: row = 4 g + r
        double e3 = 0; for (int g = 0; g < 4; g++) for (int r = 0; r < 4; r++) for (int j = 0; j < 16; j++) e3 = fmax(e3, fabs(hC[(g + 4 * r) * 16 + j] - rC[(4 * g + r) * 16 + j]));
        printf("alternative row = 4 g + r: err %.3e\n", e3);
    }
    return 0;
}
