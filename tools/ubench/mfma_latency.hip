// How many wait states does gfx950 need between v_mfma_f32_16x16x4_f32 and a VALU read of its vDst, and how many
// does clang (ROCm 7.2) insert?  Part 1 issues the MFMA through inline asm with an explicit s_nop N and reads the
// result with v_mov; part 2 uses the builtin followed immediately by a store (see the disassembly for the nops).
#include <hip/hip_runtime.h>
#include <cstdio>
#define LAT(NAME, GAP)                                                                                             \
    __global__ void NAME(const float* in, float* out) {                                                            \
        const int l = threadIdx.x;                                                                                 \
        float a = in[l], b = in[64 + l], c0 = in[128 + l], c1 = in[192 + l], c2 = in[256 + l], c3 = in[320 + l];   \
        float d0, d1, d2, d3;                                                                                      \
        asm volatile("v_mov_b32 v40, %6\n v_mov_b32 v41, %7\n v_mov_b32 v42, %8\n v_mov_b32 v43, %9\n"             \
                     "v_mov_b32 v50, %4\n v_mov_b32 v51, %5\n s_nop 7\n"                                           \
                     "v_mfma_f32_16x16x4_f32 v[40:43], v50, v51, v[40:43]\n" GAP                                   \
                     "v_mov_b32 %0, v40\n v_mov_b32 %1, v41\n v_mov_b32 %2, v42\n v_mov_b32 %3, v43\n"             \
                     : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)                                                      \
                     : "v"(a), "v"(b), "v"(c0), "v"(c1), "v"(c2), "v"(c3)                                          \
                     : "v40", "v41", "v42", "v43", "v50", "v51");                                                  \
        out[l] = d0; out[64 + l] = d1; out[128 + l] = d2; out[192 + l] = d3;                                        \
    }
LAT(l_ref, "s_nop 15\n s_nop 15\n")
LAT(l_0, "")
LAT(l_1, "s_nop 0\n")
LAT(l_2, "s_nop 1\n")
LAT(l_4, "s_nop 3\n")
LAT(l_6, "s_nop 5\n")
LAT(l_8, "s_nop 7\n")
LAT(l_10, "s_nop 9\n")
LAT(l_11, "s_nop 10\n")
LAT(l_12, "s_nop 11\n")
LAT(l_14, "s_nop 13\n")
LAT(l_16, "s_nop 15\n")
// MFMA result read by a VMEM store (global_store_dwordx4) after GAP wait states
#define LATST(NAME, GAP)                                                                                           \
    __global__ void NAME(const float* in, float* out) {                                                            \
        const int l = threadIdx.x;                                                                                 \
        float a = in[l], b = in[64 + l], c0 = in[128 + l], c1 = in[192 + l], c2 = in[256 + l], c3 = in[320 + l];   \
        float* p = out + 4 * l;                                                                                    \
        asm volatile("v_mov_b32 v40, %2\n v_mov_b32 v41, %3\n v_mov_b32 v42, %4\n v_mov_b32 v43, %5\n"             \
                     "v_mov_b32 v50, %0\n v_mov_b32 v51, %1\n s_nop 7\n"                                           \
                     "v_mfma_f32_16x16x4_f32 v[40:43], v50, v51, v[40:43]\n" GAP                                   \
                     "global_store_dwordx4 %6, v[40:43], off\n s_waitcnt vmcnt(0)\n"                               \
                     :                                                                                             \
                     : "v"(a), "v"(b), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(p)                                  \
                     : "v40", "v41", "v42", "v43", "v50", "v51", "memory");                                        \
    }
LATST(s_ref, "s_nop 15\n s_nop 15\n")
LATST(s_0, "")
LATST(s_2, "s_nop 1\n")
LATST(s_4, "s_nop 3\n")
LATST(s_6, "s_nop 5\n")
LATST(s_8, "s_nop 7\n")
LATST(s_10, "s_nop 9\n")
LATST(s_12, "s_nop 11\n")
LATST(s_16, "s_nop 15\n")
// chains of MFMAs, then a VMEM store of the last result after GAP wait states
#define CHST(NAME, BODY, LASTREG, GAP)                                                                             \
    __global__ void NAME(const float* in, float* out) {                                                            \
        const int l = threadIdx.x;                                                                                 \
        float a = in[l], b = in[64 + l], c0 = in[128 + l], c1 = in[192 + l], c2 = in[256 + l], c3 = in[320 + l];   \
        float* p = out + 4 * l;                                                                                    \
        asm volatile("v_mov_b32 v40, %2\n v_mov_b32 v41, %3\n v_mov_b32 v42, %4\n v_mov_b32 v43, %5\n"             \
                     "v_mov_b32 v44, %2\n v_mov_b32 v45, %3\n v_mov_b32 v46, %4\n v_mov_b32 v47, %5\n"             \
                     "v_mov_b32 v36, %2\n v_mov_b32 v37, %3\n v_mov_b32 v38, %4\n v_mov_b32 v39, %5\n"             \
                     "v_mov_b32 v32, %2\n v_mov_b32 v33, %3\n v_mov_b32 v34, %4\n v_mov_b32 v35, %5\n"             \
                     "v_mov_b32 v50, %0\n v_mov_b32 v51, %1\n s_nop 7\n" BODY GAP                                  \
                     "global_store_dwordx4 %6, " LASTREG ", off\n s_waitcnt vmcnt(0)\n"                            \
                     :                                                                                             \
                     : "v"(a), "v"(b), "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(p)                                  \
                     : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44",  \
                       "v45", "v46", "v47", "v50", "v51", "memory");                                               \
    }
#define DEP4 "v_mfma_f32_16x16x4_f32 v[40:43], v50, v51, v[40:43]\n v_mfma_f32_16x16x4_f32 v[40:43], v51, v50, v[40:43]\n" \
             "v_mfma_f32_16x16x4_f32 v[40:43], v50, v51, v[40:43]\n v_mfma_f32_16x16x4_f32 v[40:43], v51, v50, v[40:43]\n"
#define IND4 "v_mfma_f32_16x16x4_f32 v[32:35], v50, v51, v[32:35]\n v_mfma_f32_16x16x4_f32 v[36:39], v51, v50, v[36:39]\n" \
             "v_mfma_f32_16x16x4_f32 v[44:47], v50, v51, v[44:47]\n v_mfma_f32_16x16x4_f32 v[40:43], v51, v50, v[40:43]\n"
#define SHF4 "v_mfma_f32_16x16x4_f32 v[40:43], v50, v51, v[40:43]\n v_mfma_f32_16x16x4_f32 v[40:43], v51, v50, v[40:43]\n" \
             "v_mfma_f32_16x16x4_f32 v[40:43], v50, v51, v[40:43]\n v_mfma_f32_16x16x4_f32 v[38:41], v51, v50, v[40:43]\n"
CHST(d_ref, DEP4, "v[40:43]", "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n")
CHST(d_8, DEP4, "v[40:43]", "s_nop 7\n")
CHST(d_10, DEP4, "v[40:43]", "s_nop 9\n")
CHST(d_16, DEP4, "v[40:43]", "s_nop 15\n")
CHST(d_32, DEP4, "v[40:43]", "s_nop 15\n s_nop 15\n")
CHST(i_ref, IND4, "v[40:43]", "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n")
CHST(i_8, IND4, "v[40:43]", "s_nop 7\n")
CHST(i_10, IND4, "v[40:43]", "s_nop 9\n")
CHST(i_16, IND4, "v[40:43]", "s_nop 15\n")
CHST(i_32, IND4, "v[40:43]", "s_nop 15\n s_nop 15\n")
CHST(h_ref, SHF4, "v[38:41]", "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n")
CHST(h_8, SHF4, "v[38:41]", "s_nop 7\n")
CHST(h_10, SHF4, "v[38:41]", "s_nop 9\n")
CHST(h_16, SHF4, "v[38:41]", "s_nop 15\n")
CHST(h_32, SHF4, "v[38:41]", "s_nop 15\n s_nop 15\n")
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_builtin(const float* in, float* out) {
    const int l = threadIdx.x;
    f4 c = {in[128 + l], in[192 + l], in[256 + l], in[320 + l]};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(in[l], in[64 + l], c, 0, 0, 0);
    out[l] = c[0]; out[64 + l] = c[1]; out[128 + l] = c[2]; out[192 + l] = c[3];
}
int main() {
    float h[384], r[256], o[256];
    for (int i = 0; i < 384; ++i) h[i] = (float)((i * 37) % 101) * 0.25f - 7.0f;
    float *din, *dout;
    hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    l_ref<<<1, 64>>>(din, dout);
    hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
#define RUN(K, WHAT) { K<<<1, 64>>>(din, dout); hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost); int bad = 0, badhi = 0; \
        for (int i = 0; i < 256; ++i) { bad += o[i] != r[i]; if (i >= 128) badhi += o[i] != r[i]; }                       \
        printf("%-34s %3d of 256 wrong (%d of them in result registers 2,3)\n", WHAT, bad, badhi); }
    RUN(l_0, "MFMA -> v_mov, 0 wait states");
    RUN(l_1, "1 wait state"); RUN(l_2, "2 wait states"); RUN(l_4, "4 wait states"); RUN(l_6, "6 wait states");
    RUN(l_8, "8 wait states"); RUN(l_10, "10 wait states"); RUN(l_11, "11 wait states"); RUN(l_12, "12 wait states");
    RUN(l_14, "14 wait states"); RUN(l_16, "16 wait states");
    RUN(k_builtin, "builtin + immediate store (clang)");
    s_ref<<<1, 64>>>(din, dout);
    hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
#define RUNS(K, WHAT) { hipMemset(dout, 0, sizeof o); K<<<1, 64>>>(din, dout); hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost); int bad = 0, badhi = 0; \
        for (int i = 0; i < 256; ++i) { bad += o[i] != r[i]; if ((i & 3) >= 2) badhi += o[i] != r[i]; }                   \
        printf("%-34s %3d of 256 wrong (%d of them in result registers 2,3)\n", WHAT, bad, badhi); }
    RUNS(s_0, "MFMA -> global_store, 0 wait st."); RUNS(s_2, "2 wait states"); RUNS(s_4, "4 wait states");
    RUNS(s_6, "6 wait states"); RUNS(s_8, "8 wait states"); RUNS(s_10, "10 wait states"); RUNS(s_12, "12 wait states");
    RUNS(s_16, "16 wait states");
    d_ref<<<1, 64>>>(din, dout); hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
    RUNS(d_8, "4 dependent MFMAs -> store,  8 ws"); RUNS(d_10, "  10 wait states"); RUNS(d_16, "  16 wait states"); RUNS(d_32, "  32 wait states");
    i_ref<<<1, 64>>>(din, dout); hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
    RUNS(i_8, "4 independent MFMAs -> store, 8 ws"); RUNS(i_10, "  10 wait states"); RUNS(i_16, "  16 wait states"); RUNS(i_32, "  32 wait states");
    h_ref<<<1, 64>>>(din, dout); hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
    RUNS(h_8, "3 dep + shifted vDst -> store, 8 ws"); RUNS(h_10, "  10 wait states"); RUNS(h_16, "  16 wait states"); RUNS(h_32, "  32 wait states");
    return 0;
}
