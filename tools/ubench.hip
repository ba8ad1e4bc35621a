// Instruction-rate microbenchmark for gfx950 (design input for the disparity
// kernels; not part of the product).  For each instruction form it runs a loop
// of 16 independent copies per iteration at 1/2/4/8 waves per SIMD on every CU
// and reports cycles per wave-instruction per SIMD (s_memtime) and the wall
// clock rate.  Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench tools/ubench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

// 32-bit destination ops: 8 accumulators r0..r7 (%0..%7), sources a (%8), b (%9), scalar s (%10)
#define UB32(NAME, LINE)                                                                   \
    __global__ void NAME(unsigned long long *out, int iters, unsigned sa)                   \
    {                                                                                      \
        unsigned r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4,     \
                 r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;                                    \
        unsigned a = threadIdx.x * 2654435761u, b = a ^ 0x55555555u;                       \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < iters; ++i) {                                                  \
            asm volatile(LINE("%0") LINE("%1") LINE("%2") LINE("%3") LINE("%4") LINE("%5") \
                         LINE("%6") LINE("%7") LINE("%0") LINE("%1") LINE("%2") LINE("%3") \
                         LINE("%4") LINE("%5") LINE("%6") LINE("%7")                       \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),     \
                           "+v"(r6), "+v"(r7)                                              \
                         : "v"(a), "v"(b), "s"(sa));                                       \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        unsigned x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                \
        if (x == 0x12345678u) out[1] = x;                                                  \
        if (threadIdx.x == 0) out[2 + blockIdx.x] = t1 - t0;                               \
    }

// 64-bit destination ops
#define UB64(NAME, LINE)                                                                   \
    __global__ void NAME(unsigned long long *out, int iters, unsigned sa)                   \
    {                                                                                      \
        unsigned long long r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3,        \
                           r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;             \
        unsigned long long a = threadIdx.x * 0x9E3779B97F4A7C15ull;                        \
        unsigned b = (unsigned)a ^ 0x55555555u;                                            \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < iters; ++i) {                                                  \
            asm volatile(LINE("%0") LINE("%1") LINE("%2") LINE("%3") LINE("%4") LINE("%5") \
                         LINE("%6") LINE("%7") LINE("%0") LINE("%1") LINE("%2") LINE("%3") \
                         LINE("%4") LINE("%5") LINE("%6") LINE("%7")                       \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),     \
                           "+v"(r6), "+v"(r7)                                              \
                         : "v"(a), "v"(b), "s"(sa));                                       \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        unsigned long long x = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                      \
        if (x == 0x12345678u) out[1] = x;                                                  \
        if (threadIdx.x == 0) out[2 + blockIdx.x] = t1 - t0;                               \
    }

// LDS reads: address register %8 (per lane), 8 destinations; wait inside
#define UBLDS(NAME, LINE, ADDR_EXPR, DTYPE, CONS)                                           \
    __global__ void NAME(unsigned long long *out, int iters, unsigned sa)                   \
    {                                                                                      \
        __shared__ __attribute__((aligned(16))) unsigned lds[16384];                       \
        for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 7 + sa;         \
        __syncthreads();                                                                   \
        DTYPE r0 = {}, r1 = {}, r2 = {}, r3 = {}, r4 = {}, r5 = {}, r6 = {}, r7 = {};      \
        unsigned lane = threadIdx.x & 63;                                                  \
        unsigned a = (ADDR_EXPR);                                                          \
        unsigned b = a;                                                                    \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                              \
        for (int i = 0; i < iters; ++i) {                                                  \
            asm volatile(LINE("%0", "0") LINE("%1", "256") LINE("%2", "512") LINE("%3", "768")    \
                         LINE("%4", "1024") LINE("%5", "1280") LINE("%6", "1536")          \
                         LINE("%7", "1792") LINE("%0", "2048") LINE("%1", "2304")          \
                         LINE("%2", "2560") LINE("%3", "2816") LINE("%4", "3072")          \
                         LINE("%5", "3328") LINE("%6", "3584") LINE("%7", "3840")          \
                         "s_waitcnt lgkmcnt(0)\n"                                          \
                         : CONS(r0), CONS(r1), CONS(r2), CONS(r3), CONS(r4), CONS(r5),     \
                           CONS(r6), CONS(r7)                                              \
                         : "v"(a), "v"(b), "s"(sa));                                       \
        }                                                                                  \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                              \
        unsigned x = fold(r0) ^ fold(r1) ^ fold(r2) ^ fold(r3) ^ fold(r4) ^ fold(r5) ^     \
                     fold(r6) ^ fold(r7);                                                  \
        if (x == 0x12345678u) out[1] = x;                                                  \
        if (threadIdx.x == 0) out[2 + blockIdx.x] = t1 - t0;                               \
    }
#define CONSV(x) "+v"(x)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned fold(unsigned v) { return v; }
__device__ inline unsigned fold(u32x2 v) { return v.x ^ v.y; }
__device__ inline unsigned fold(u32x4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

#define L_SAD_U32(D) "v_sad_u32 " D ", %8, %9, " D "\n"
#define L_SAD_U32_S(D) "v_sad_u32 " D ", %8, %10, " D "\n"
#define L_MIN_U32(D) "v_min_u32 " D ", %8, " D "\n"
#define L_MIN3_U32(D) "v_min3_u32 " D ", %8, %9, " D "\n"
#define L_SAD_U8(D) "v_sad_u8 " D ", %8, %9, " D "\n"
#define L_MSAD_U8(D) "v_msad_u8 " D ", %8, %9, " D "\n"
#define L_SAD_U16(D) "v_sad_u16 " D ", %8, %9, " D "\n"
#define L_ADD_U32(D) "v_add_u32 " D ", %8, " D "\n"
#define L_ADD3_U32(D) "v_add3_u32 " D ", %8, %9, " D "\n"
#define L_LSHL_OR(D) "v_lshl_or_b32 " D ", %8, 16, " D "\n"
#define L_AND_OR(D) "v_and_or_b32 " D ", %8, %9, " D "\n"
#define L_BFE(D) "v_bfe_u32 " D ", " D ", 8, 8\n"
#define L_BFI(D) "v_bfi_b32 " D ", %8, %9, " D "\n"
#define L_PERM(D) "v_perm_b32 " D ", %8, " D ", %9\n"
#define L_ALIGNBYTE(D) "v_alignbyte_b32 " D ", %8, " D ", 1\n"
#define L_ALIGNBYTE_S(D) "v_alignbyte_b32 " D ", %8, " D ", %10\n"
#define L_MAD_U24(D) "v_mad_u32_u24 " D ", %8, %9, " D "\n"
#define L_MUL_LO(D) "v_mul_lo_u32 " D ", %8, " D "\n"
#define L_PK_ADD_U16(D) "v_pk_add_u16 " D ", %8, " D "\n"
#define L_PK_SUB_U16C(D) "v_pk_sub_u16 " D ", %8, " D " clamp\n"
#define L_PK_MIN_U16(D) "v_pk_min_u16 " D ", %8, " D "\n"
#define L_PK_MAX_I16(D) "v_pk_max_i16 " D ", %8, " D "\n"
#define L_PK_MAD_U16(D) "v_pk_mad_u16 " D ", %8, %9, " D "\n"
#define L_ADD_DPP_ROWSHR(D) "v_add_u32_dpp " D ", %8, " D " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define L_ADD_DPP_WAVESHR(D) "v_add_u32_dpp " D ", %8, " D " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define L_MOV_DPP_WAVESHR(D) "v_mov_b32_dpp " D ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define L_ADD_DPP_BCAST15(D) "v_add_u32_dpp " D ", %8, " D " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define L_SUB_SDWA(D) "v_sub_u32_sdwa " D ", %8, " D " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
#define L_FMA_F32(D) "v_fma_f32 " D ", %8, %9, " D "\n"
#define L_PK_FMA_F32(D) "v_pk_fma_f32 " D ", %8, %8, " D "\n"
#define L_CNDMASK(D) "v_cndmask_b32 " D ", %8, " D ", vcc\n"
#define L_CMP_LT(D) "v_cmp_lt_u32 vcc, %8, " D "\n"
#define L_MAX_U32(D) "v_max_u32 " D ", %8, " D "\n"
#define L_SUB_U32(D) "v_sub_u32 " D ", %8, " D "\n"
#define L_CVT_UBYTE(D) "v_cvt_f32_ubyte1 " D ", " D "\n"
#define L_READLANE(D) "v_readlane_b32 s20, " D ", 3\n"
#define L_WRITELANE(D) "v_writelane_b32 " D ", %10, 0\n"
#define L_QSAD(D) "v_qsad_pk_u16_u8 " D ", %8, %9, " D "\n"
#define L_MQSAD(D) "v_mqsad_pk_u16_u8 " D ", %8, %9, " D "\n"
#define L_PERMLANE32(D) "v_permlane32_swap_b32 " D ", %8\n"

#define L_MIN_F32(D) "v_min_f32 " D ", %8, " D "\n"
#define L_MAX_F32(D) "v_max_f32 " D ", %8, " D "\n"
#define L_MIN3_F32(D) "v_min3_f32 " D ", %8, %9, " D "\n"
#define L_MAX3_F32(D) "v_max3_f32 " D ", %8, %9, " D "\n"
#define L_MED3_F32(D) "v_med3_f32 " D ", %8, %9, " D "\n"
#define L_MUL_F32(D) "v_mul_f32 " D ", %8, " D "\n"
#define L_ADD_F32(D) "v_add_f32 " D ", %8, " D "\n"
#define L_MOV_B32(D) "v_mov_b32 " D ", %8\n"
#define L_AND_B32(D) "v_and_b32 " D ", %8, " D "\n"
#define L_OR_B32(D) "v_or_b32 " D ", %8, " D "\n"
#define L_XOR_B32(D) "v_xor_b32 " D ", %8, " D "\n"
#define L_LSHLREV(D) "v_lshlrev_b32 " D ", 3, " D "\n"
#define L_LSHL_ADD(D) "v_lshl_add_u32 " D ", %8, 16, " D "\n"
#define L_MIN_I32(D) "v_min_i32 " D ", %8, " D "\n"
#define L_SUBREV_U32(D) "v_subrev_u32 " D ", %8, " D "\n"
#define L_MUL_U24(D) "v_mul_u32_u24 " D ", %8, " D "\n"
#define L_DOT4_U8(D) "v_dot4_u32_u8 " D ", %8, %9, " D "\n"
#define L_ADD_CO(D) "v_add_co_u32 " D ", vcc, %8, " D "\n"
#define L_ADDC_CO(D) "v_addc_co_u32 " D ", vcc, " D ", " D ", vcc\n"
#define L_CMP_NLT_F32(D) "v_cmp_nlt_f32 vcc, %8, " D "\n"
#define L_RNDNE(D) "v_rndne_f32 " D ", " D "\n"
#define L_CVT_U32_F32(D) "v_cvt_u32_f32 " D ", " D "\n"
#define L_PK_ADD_F32(D) "v_pk_add_f32 " D ", %8, " D "\n"
#define L_PK_MUL_F32(D) "v_pk_mul_f32 " D ", %8, " D "\n"
UB32(k_min_f32, L_MIN_F32)
UB32(k_max_f32, L_MAX_F32)
UB32(k_min3_f32, L_MIN3_F32)
UB32(k_max3_f32, L_MAX3_F32)
UB32(k_med3_f32, L_MED3_F32)
UB32(k_mul_f32, L_MUL_F32)
UB32(k_add_f32, L_ADD_F32)
UB32(k_mov_b32, L_MOV_B32)
UB32(k_and_b32, L_AND_B32)
UB32(k_or_b32, L_OR_B32)
UB32(k_xor_b32, L_XOR_B32)
UB32(k_lshlrev, L_LSHLREV)
UB32(k_lshl_add, L_LSHL_ADD)
UB32(k_min_i32, L_MIN_I32)
UB32(k_subrev_u32, L_SUBREV_U32)
UB32(k_mul_u24, L_MUL_U24)
UB32(k_dot4_u8, L_DOT4_U8)
UB32(k_add_co, L_ADD_CO)
UB32(k_addc_co, L_ADDC_CO)
UB32(k_cmp_nlt_f32, L_CMP_NLT_F32)
UB32(k_rndne, L_RNDNE)
UB32(k_cvt_u32_f32, L_CVT_U32_F32)
UB64(k_pk_add_f32, L_PK_ADD_F32)
UB64(k_pk_mul_f32, L_PK_MUL_F32)
UB32(k_sad_u32, L_SAD_U32)
UB32(k_sad_u32_s, L_SAD_U32_S)
UB32(k_min_u32, L_MIN_U32)
UB32(k_min3_u32, L_MIN3_U32)
UB32(k_sad_u8, L_SAD_U8)
UB32(k_msad_u8, L_MSAD_U8)
UB32(k_sad_u16, L_SAD_U16)
UB32(k_add_u32, L_ADD_U32)
UB32(k_add3_u32, L_ADD3_U32)
UB32(k_lshl_or, L_LSHL_OR)
UB32(k_and_or, L_AND_OR)
UB32(k_bfe, L_BFE)
UB32(k_bfi, L_BFI)
UB32(k_perm, L_PERM)
UB32(k_alignbyte, L_ALIGNBYTE)
UB32(k_alignbyte_s, L_ALIGNBYTE_S)
UB32(k_mad_u24, L_MAD_U24)
UB32(k_mul_lo, L_MUL_LO)
UB32(k_pk_add_u16, L_PK_ADD_U16)
UB32(k_pk_sub_u16c, L_PK_SUB_U16C)
UB32(k_pk_min_u16, L_PK_MIN_U16)
UB32(k_pk_max_i16, L_PK_MAX_I16)
UB32(k_pk_mad_u16, L_PK_MAD_U16)
UB32(k_add_dpp_rowshr, L_ADD_DPP_ROWSHR)
UB32(k_add_dpp_waveshr, L_ADD_DPP_WAVESHR)
UB32(k_mov_dpp_waveshr, L_MOV_DPP_WAVESHR)
UB32(k_add_dpp_bcast15, L_ADD_DPP_BCAST15)
UB32(k_sub_sdwa, L_SUB_SDWA)
UB32(k_fma_f32, L_FMA_F32)
UB32(k_cndmask, L_CNDMASK)
UB32(k_cmp_lt, L_CMP_LT)
UB32(k_max_u32, L_MAX_U32)
UB32(k_sub_u32, L_SUB_U32)
UB32(k_cvt_ubyte, L_CVT_UBYTE)
UB32(k_writelane, L_WRITELANE)
UB32(k_permlane32, L_PERMLANE32)
UB64(k_qsad, L_QSAD)
UB64(k_mqsad, L_MQSAD)
UB64(k_pk_fma_f32, L_PK_FMA_F32)

#define L_DS_U8(D, O) "ds_read_u8 " D ", %8 offset:" O "\n"
#define L_DS_B32(D, O) "ds_read_b32 " D ", %8 offset:" O "\n"
#define L_DS_B64(D, O) "ds_read_b64 " D ", %8 offset:" O "\n"
#define L_DS_B128(D, O) "ds_read_b128 " D ", %8 offset:" O "\n"
#define L_DS_BPERM(D, O) "ds_bpermute_b32 " D ", %8, %9\n"
#define L_DS_WRITE_B32(D, O) "ds_write_b32 %8, " D " offset:" O "\n"
#define L_DS_WRITE_B8(D, O) "ds_write_b8 %8, " D " offset:" O "\n"

UBLDS(k_ds_u8_consec, L_DS_U8, lane, unsigned, CONSV)               // 64 consecutive bytes
UBLDS(k_ds_u8_bcast, L_DS_U8, 0, unsigned, CONSV)                   // all lanes same byte
UBLDS(k_ds_b32_consec, L_DS_B32, lane * 4, unsigned, CONSV)         // conflict-free dwords
UBLDS(k_ds_b32_bcast, L_DS_B32, 0, unsigned, CONSV)
UBLDS(k_ds_b32_unal1, L_DS_B32, lane * 4 + 1, unsigned, CONSV)      // misaligned by 1, stride 4
UBLDS(k_ds_b32_bytestride, L_DS_B32, lane, unsigned, CONSV)         // stride 1 byte (3/4 misaligned)
UBLDS(k_ds_b32_bytestride_al, L_DS_B32, (lane & ~3u), unsigned, CONSV)  // 4 lanes share a dword
UBLDS(k_ds_b64_consec, L_DS_B64, lane * 8, u32x2, CONSV)
UBLDS(k_ds_b64_bcast, L_DS_B64, 0, u32x2, CONSV)
UBLDS(k_ds_b64_stride4, L_DS_B64, (lane * 4) & ~7u, u32x2, CONSV)
UBLDS(k_ds_b128_consec, L_DS_B128, lane * 16, u32x4, CONSV)
UBLDS(k_ds_b128_bcast, L_DS_B128, 0, u32x4, CONSV)
UBLDS(k_ds_bpermute, L_DS_BPERM, ((lane + 9) & 63) * 4, unsigned, CONSV)
UBLDS(k_ds_write_b32, L_DS_WRITE_B32, lane * 4, unsigned, CONSV)
UBLDS(k_ds_write_b8, L_DS_WRITE_B8, lane, unsigned, CONSV)

typedef void (*kern_t)(unsigned long long *, int, unsigned);
struct Entry { const char *name; kern_t k; };

int main(int argc, char **argv)
{
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    printf("device %s  CUs %d  clockRate %d kHz  lds/block %zu\n", prop.name, cus, prop.clockRate,
           prop.sharedMemPerBlock);
    std::vector<Entry> es = {
#define E(n) {#n, n}
        E(k_min_f32), E(k_max_f32), E(k_min3_f32), E(k_max3_f32), E(k_med3_f32), E(k_mul_f32), E(k_add_f32),
        E(k_mov_b32), E(k_and_b32), E(k_or_b32), E(k_xor_b32), E(k_lshlrev), E(k_lshl_add), E(k_min_i32),
        E(k_subrev_u32), E(k_mul_u24), E(k_dot4_u8), E(k_add_co), E(k_addc_co), E(k_cmp_nlt_f32), E(k_rndne),
        E(k_cvt_u32_f32), E(k_pk_add_f32), E(k_pk_mul_f32),
        E(k_fma_f32), E(k_pk_fma_f32), E(k_add_u32), E(k_sub_u32), E(k_add3_u32), E(k_min_u32), E(k_max_u32),
        E(k_min3_u32), E(k_sad_u32), E(k_sad_u32_s), E(k_sad_u8), E(k_msad_u8), E(k_sad_u16),
        E(k_qsad), E(k_mqsad), E(k_lshl_or), E(k_and_or), E(k_bfe), E(k_bfi), E(k_perm),
        E(k_alignbyte), E(k_alignbyte_s), E(k_mad_u24), E(k_mul_lo), E(k_pk_add_u16),
        E(k_pk_sub_u16c), E(k_pk_min_u16), E(k_pk_max_i16), E(k_pk_mad_u16), E(k_cmp_lt), E(k_cndmask),
        E(k_add_dpp_rowshr), E(k_add_dpp_waveshr), E(k_mov_dpp_waveshr), E(k_add_dpp_bcast15),
        E(k_sub_sdwa), E(k_cvt_ubyte), E(k_writelane), E(k_permlane32),
        E(k_ds_u8_consec), E(k_ds_u8_bcast), E(k_ds_b32_consec), E(k_ds_b32_bcast), E(k_ds_b32_unal1),
        E(k_ds_b32_bytestride), E(k_ds_b32_bytestride_al), E(k_ds_b64_consec), E(k_ds_b64_bcast),
        E(k_ds_b64_stride4), E(k_ds_b128_consec), E(k_ds_b128_bcast), E(k_ds_bpermute),
        E(k_ds_write_b32), E(k_ds_write_b8),
    };
    const char *only = argc > 1 ? argv[1] : nullptr;
    const int iters = 4000;
    unsigned long long *d_out;
    const int max_blocks = cus * 8;
    CHECK(hipMalloc(&d_out, sizeof(unsigned long long) * (max_blocks + 2)));
    std::vector<unsigned long long> h(max_blocks + 2);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%-24s %5s %12s %14s %12s\n", "op", "w/SIMD", "cyc/inst/SIMD", "Ginst/s(chip)", "eff.clk GHz");
    for (auto &e : es) {
        if (only && !strstr(e.name, only)) continue;
        for (int wps : {1, 2, 4, 8}) {
            // blocks of 256 threads = 4 waves = one wave per SIMD; wps blocks per CU
            const int blocks = cus * wps;
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, 200, 3u);  // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, iters, 3u);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * (blocks + 2), hipMemcpyDeviceToHost));
            std::vector<unsigned long long> c(h.begin() + 2, h.begin() + 2 + blocks);
            std::sort(c.begin(), c.end());
            const double cyc = (double)c[c.size() / 2];
            const double inst_per_wave = (double)iters * 16;
            const double cyc_per_inst_simd = cyc / (inst_per_wave * wps);
            const double total_inst = inst_per_wave * 4.0 * blocks;
            const double ginst = total_inst / (ms * 1e-3) / 1e9;
            printf("%-24s %5d %12.3f %14.1f %12.3f\n", e.name, wps, cyc_per_inst_simd, ginst,
                   cyc / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
