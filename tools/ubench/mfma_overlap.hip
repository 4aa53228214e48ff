// Which register overlaps between vDst and the sources of v_mfma_f32_16x16x4_f32 does gfx950 tolerate?
// clang (ROCm 7.2) treats every overlap as legal for 128-bit results.  Each case computes D = A*B + C with a given
// register assignment and compares with an assignment where all operands are disjoint.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CASE(NAME, DST, SA, SB, SC0, SC1, SC2, SC3, SC)                                                             \
    __global__ void NAME(const float* in, float* out) {                                                            \
        const int l = threadIdx.x;                                                                                 \
        float a = in[l], b = in[64 + l], c0 = in[128 + l], c1 = in[192 + l], c2 = in[256 + l], c3 = in[320 + l];   \
        float d0, d1, d2, d3;                                                                                      \
        asm volatile("v_mov_b32 " SC0 ", %6\n v_mov_b32 " SC1 ", %7\n v_mov_b32 " SC2 ", %8\n v_mov_b32 " SC3 ", %9\n" \
                     "v_mov_b32 " SA ", %4\n v_mov_b32 " SB ", %5\n s_nop 7\n"                                      \
                     "v_mfma_f32_16x16x4_f32 " DST ", " SA ", " SB ", " SC "\n s_nop 15\n s_nop 15\n"               \
                     "v_mov_b32 %0, v40\n v_mov_b32 %1, v41\n v_mov_b32 %2, v42\n v_mov_b32 %3, v43\n"              \
                     : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)                                                      \
                     : "v"(a), "v"(b), "v"(c0), "v"(c1), "v"(c2), "v"(c3)                                          \
                     : "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v50",   \
                       "v51", "v52", "v53", "v54", "v55");                                                         \
        out[l] = d0; out[64 + l] = d1; out[128 + l] = d2; out[192 + l] = d3;                                        \
    }
// vDst is always v[40:43]
CASE(k_ref,        "v[40:43]", "v50", "v51", "v52", "v53", "v54", "v55", "v[52:55]")
CASE(k_tied,       "v[40:43]", "v50", "v51", "v40", "v41", "v42", "v43", "v[40:43]")
CASE(k_c_above,    "v[40:43]", "v50", "v51", "v42", "v43", "v44", "v45", "v[42:45]")   // SrcC = vDst + 2
CASE(k_c_below,    "v[40:43]", "v50", "v51", "v38", "v39", "v40", "v41", "v[38:41]")   // SrcC = vDst - 2
CASE(k_a_first,    "v[40:43]", "v40", "v51", "v52", "v53", "v54", "v55", "v[52:55]")   // SrcA = vDst[0]
CASE(k_a_last,     "v[40:43]", "v43", "v51", "v52", "v53", "v54", "v55", "v[52:55]")   // SrcA = vDst[3]
CASE(k_b_first,    "v[40:43]", "v50", "v40", "v52", "v53", "v54", "v55", "v[52:55]")   // SrcB = vDst[0]
CASE(k_b_last,     "v[40:43]", "v50", "v43", "v52", "v53", "v54", "v55", "v[52:55]")   // SrcB = vDst[3]
// Dependent pair: the second MFMA takes the first one's vDst as SrcC and writes to v[D2:D2+3].  GAP = text between.
#define CHAIN(NAME, D2A, D2B, D2C, D2D, DST2, GAP)                                                                  \
    __global__ void NAME(const float* in, float* out) {                                                            \
        const int l = threadIdx.x;                                                                                 \
        float a = in[l], b = in[64 + l], c0 = in[128 + l], c1 = in[192 + l], c2 = in[256 + l], c3 = in[320 + l];   \
        float d0, d1, d2, d3;                                                                                      \
        asm volatile("v_mov_b32 v52, %6\n v_mov_b32 v53, %7\n v_mov_b32 v54, %8\n v_mov_b32 v55, %9\n"             \
                     "v_mov_b32 v50, %4\n v_mov_b32 v51, %5\n s_nop 7\n"                                          \
                     "v_mfma_f32_16x16x4_f32 v[40:43], v50, v51, v[52:55]\n" GAP                                   \
                     "v_mfma_f32_16x16x4_f32 " DST2 ", v51, v50, v[40:43]\n s_nop 15\n s_nop 15\n"                \
                     "v_mov_b32 %0, " D2A "\n v_mov_b32 %1, " D2B "\n v_mov_b32 %2, " D2C "\n v_mov_b32 %3, " D2D "\n" \
                     : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)                                                      \
                     : "v"(a), "v"(b), "v"(c0), "v"(c1), "v"(c2), "v"(c3)                                          \
                     : "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v50",   \
                       "v51", "v52", "v53", "v54", "v55");                                                         \
        out[l] = d0; out[64 + l] = d1; out[128 + l] = d2; out[192 + l] = d3;                                        \
    }
CHAIN(c_ref,      "v44", "v45", "v46", "v47", "v[44:47]", "s_nop 15\n s_nop 15\n")   // disjoint vDst, long gap
CHAIN(c_tied_0,   "v40", "v41", "v42", "v43", "v[40:43]", "")                       // in place, back to back
CHAIN(c_disj_0,   "v44", "v45", "v46", "v47", "v[44:47]", "")                       // other vDst, back to back
CHAIN(c_disj_1,   "v44", "v45", "v46", "v47", "v[44:47]", "s_nop 1\n")
CHAIN(c_shift_0,  "v38", "v39", "v40", "v41", "v[38:41]", "")                       // vDst = SrcC - 2, back to back
CHAIN(c_shift_1,  "v38", "v39", "v40", "v41", "v[38:41]", "s_nop 1\n")             // ... 2 wait states (what clang leaves)
CHAIN(c_shift_7,  "v38", "v39", "v40", "v41", "v[38:41]", "s_nop 7\n")
CHAIN(c_shift_15, "v38", "v39", "v40", "v41", "v[38:41]", "s_nop 15\n")
CHAIN(c_up_1,     "v42", "v43", "v44", "v45", "v[42:45]", "s_nop 1\n")             // vDst = SrcC + 2
int main() {
    float h[384], r[256], o[256];
    for (int i = 0; i < 384; ++i) h[i] = (float)((i * 37) % 101) * 0.25f - 7.0f;
    float *din, *dout;
    hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    k_ref<<<1, 64>>>(din, dout);
    hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
    // host check of the reference itself: D[i][j] = sum_k A[i][k] B[k][j] + C[i][j]; lane l: A[l&15][l>>4], B[l>>4][l&15], D[4(l>>4)+e][l&15]
    int refbad = 0;
    for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * (l >> 4) + e, j = l & 15;
            float s = h[128 + 64 * e + l];
            for (int k = 0; k < 4; ++k) s += h[16 * k + i] * h[64 + 16 * k + j];
            refbad += (s != r[64 * e + l]);
        }
    printf("reference (all operands disjoint) vs host: %d of 256 differ\n", refbad);
#define RUN(K, WHAT) { K<<<1, 64>>>(din, dout); hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost); int bad = 0; \
        for (int i = 0; i < 256; ++i) bad += o[i] != r[i]; printf("%-44s %3d of 256 results wrong\n", WHAT, bad); }
    RUN(k_tied, "SrcC == vDst (tied)");
    RUN(k_c_above, "SrcC = vDst + 2 registers (partial overlap)");
    RUN(k_c_below, "SrcC = vDst - 2 registers (partial overlap)");
    RUN(k_a_first, "SrcA = first register of vDst");
    RUN(k_a_last, "SrcA = last register of vDst");
    RUN(k_b_first, "SrcB = first register of vDst");
    RUN(k_b_last, "SrcB = last register of vDst");
    c_ref<<<1, 64>>>(din, dout);
    hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
    printf("dependent pairs (2nd MFMA: SrcC = 1st vDst):\n");
    RUN(c_tied_0, "  vDst2 = SrcC, 0 wait states");
    RUN(c_disj_0, "  vDst2 disjoint, 0 wait states");
    RUN(c_disj_1, "  vDst2 disjoint, 2 wait states");
    RUN(c_shift_0, "  vDst2 = SrcC - 2, 0 wait states");
    RUN(c_shift_1, "  vDst2 = SrcC - 2, 2 wait states");
    RUN(c_shift_7, "  vDst2 = SrcC - 2, 8 wait states");
    RUN(c_shift_15, "  vDst2 = SrcC - 2, 16 wait states");
    RUN(c_up_1, "  vDst2 = SrcC + 2, 2 wait states");
    return 0;
}
