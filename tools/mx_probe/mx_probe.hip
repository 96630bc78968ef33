// Layout probe for v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 operands, e8m0 block scales) on gfx950: the ISA manual is not in the
// build image, so the operand / scale layout the MX prefill GEMM relies on is established empirically.  Build:
//   hipcc --offload-arch=gfx950 -O2 tools/mx_probe/mx_probe.hip -o tools/mx_probe/mx_probe     (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int OPA, int OPB>
__global__ void k(const i32x8* a, const i32x8* b, f32x4* c, const int* sa, const int* sb) {
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, OPA, sa[l], OPB, sb[l]);
    c[l] = acc;
}

static const float VALS[8] = {0.f, 0.5f, 1.f, 2.f, -0.5f, -1.f, -2.f, 1.5f};
static const uint8_t ENC[8] = {0x00, 0x30, 0x38, 0x40, 0xB0, 0xB8, 0xC0, 0x3C};   // OCP e4m3fn

static int kmap(int hyp, int l, int j) {
    const int g = l >> 4;
    switch (hyp) {
        case 0: return 32 * g + j;                                  // each lane: 32 consecutive k
        case 1: return 16 * g + (j < 16 ? j : 64 + (j - 16));        // two K=64 halves of 16 per lane
        case 2: return 8 * g + (j % 8) + 32 * (j / 8);               // four K=32 quarters of 8 per lane
        default: return 4 * g + (j % 4) + 16 * (j / 4);
    }
}

int main() {
    srand(3);
    std::vector<uint8_t> A(64 * 32), B(64 * 32);
    std::vector<int> ia(64 * 32), ib(64 * 32), sa(64), sb(64);
    for (int i = 0; i < 64 * 32; ++i) { ia[i] = rand() % 8; ib[i] = rand() % 8; A[i] = ENC[ia[i]]; B[i] = ENC[ib[i]]; }
    for (int l = 0; l < 64; ++l) {       // byte 0 = the scale under test (0.25 .. 4), bytes 1..3 = other values (to see which byte opsel picks)
        const int e0 = 125 + rand() % 5, e1 = 125 + rand() % 5;
        sa[l] = e0 | (((e0 + 1) & 255) << 8) | (130 << 16) | (124 << 24);
        sb[l] = e1 | (((e1 + 2) & 255) << 8) | (131 << 16) | (123 << 24);
    }
    void *da, *db, *dc, *dsa, *dsb;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dc, 64 * 16); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
    hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    std::vector<float> C(64 * 4);
    for (int op = 0; op < 2; ++op) {
        if (op == 0) hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, (i32x8*)da, (i32x8*)db, (f32x4*)dc, (int*)dsa, (int*)dsb);
        else hipLaunchKernelGGL((k<1, 1>), dim3(1), dim3(64), 0, 0, (i32x8*)da, (i32x8*)db, (f32x4*)dc, (int*)dsa, (int*)dsb);
        hipDeviceSynchronize();
        hipMemcpy(C.data(), dc, 64 * 16, hipMemcpyDeviceToHost);
        for (int hyp = 0; hyp < 4; ++hyp)
            for (int sbyte = 0; sbyte < 4; ++sbyte)
                for (int smode = 0; smode < 2; ++smode) {     // smode 0: scale of lane (row r, k-group g) applies to that row's k-block g; 1: no scaling
                    // reference: D[i][j] = sum_k A[i][k] * sA(i, k/32) * B[j][k] * sB(j, k/32); operand lane l = (g, row): A row i = l & 15
                    double ref[16][16] = {};
                    static float Am[16][128], Bm[16][128], SA[16][4], SB[16][4];
                    for (int l = 0; l < 64; ++l) {
                        for (int j = 0; j < 32; ++j) {
                            const int kk = kmap(hyp, l, j);
                            Am[l & 15][kk] = VALS[ia[l * 32 + j]];
                            Bm[l & 15][kk] = VALS[ib[l * 32 + j]];
                        }
                        SA[l & 15][l >> 4] = smode ? 1.f : ldexpf(1.f, ((sa[l] >> (8 * sbyte)) & 255) - 127);
                        SB[l & 15][l >> 4] = smode ? 1.f : ldexpf(1.f, ((sb[l] >> (8 * sbyte)) & 255) - 127);
                    }
                    for (int i = 0; i < 16; ++i)
                        for (int j = 0; j < 16; ++j)
                            for (int kk = 0; kk < 128; ++kk) ref[i][j] += (double)Am[i][kk] * SA[i][kk / 32] * Bm[j][kk] * SB[j][kk / 32];
                    // D layout of the 16x16 family: lane l holds column l & 15, rows (l >> 4) * 4 + r
                    int bad = 0, badT = 0;
                    for (int l = 0; l < 64; ++l)
                        for (int r = 0; r < 4; ++r) {
                            const int row = (l >> 4) * 4 + r, col = l & 15;
                            if (fabs(C[l * 4 + r] - ref[row][col]) > 1e-3) ++bad;
                            if (fabs(C[l * 4 + r] - ref[col][row]) > 1e-3) ++badT;
                        }
                    if (bad == 0 || badT == 0)
                        printf("MATCH opsel=%d: k-map hypothesis %d, scale byte %d, %s, D %s\n", op, hyp, sbyte, smode ? "UNSCALED" : "block-scaled",
                               bad == 0 ? "[row=(l>>4)*4+r][col=l&15] with A rows = D rows" : "transposed (A rows = D cols)");
                }
        printf("opsel=%d sample D: %g %g %g %g\n", op, C[0], C[1], C[2], C[3]);
    }
    return 0;
}
