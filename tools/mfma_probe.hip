// Probe: operand / result layout of v_mfma_i32_32x32x32_i8 on gfx950 (run on the GPU box).
// hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void k(const v4i *a, const v4i *b, v16i *c)
{
    v16i acc = {0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    c[threadIdx.x] = acc;
}
int main()
{
    int8_t A[32][32], B[32][32];    // A[m][k], B[k][n]
    srand(1);
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { A[i][j] = rand() % 5 - 2; B[i][j] = rand() % 7 - 3; }
    int ref[32][32];
    for (int m = 0; m < 32; m++) for (int n = 0; n < 32; n++) { int s = 0; for (int kk = 0; kk < 32; kk++) s += A[m][kk] * B[kk][n]; ref[m][n] = s; }
    for (int variant = 0; variant < 2; variant++) {
        int8_t ha[64][16], hb[64][16];
        for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) {
            int kk = variant == 0 ? 16 * (l / 32) + j : (j < 8 ? 8 * (l / 32) + j : 16 + 8 * (l / 32) + (j - 8));
            ha[l][j] = A[l % 32][kk];
            hb[l][j] = B[kk][l % 32];
        }
        v4i *da, *db; v16i *dc;
        hipMalloc(&da, 64 * 16); hipMalloc(&db, 64 * 16); hipMalloc(&dc, 64 * 64);
        hipMemcpy(da, ha, 64 * 16, hipMemcpyHostToDevice); hipMemcpy(db, hb, 64 * 16, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc);
        int hc[64][16];
        hipMemcpy(hc, dc, 64 * 64, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            if (hc[l][r] != ref[row][col]) bad++;
        }
        printf("variant %d (k = %s): mismatches %d\n", variant, variant == 0 ? "16*(l/32)+j" : "split 8+8", bad);
    }
    return 0;
}
