// Microbenchmark (tuning aid, not part of the product): issue cost of single integer / float vector opcodes on gfx950,
// one opcode per kernel, 8 waves per SIMD. valu_rate.hip found v_add_u32 at ~2.5 cycles and every class the band kernel
// uses at ~4.1: this one asks opcode by opcode which ones take the fast path.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S "\n" S "\n"
#define K2(NAME, OP)                                                                                         \
  __global__ void __launch_bounds__(256) NAME(unsigned* out, int iters) {                                     \
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11, e = 13 + threadIdx.x, f = 17;           \
    asm volatile("v_cmp_lt_u32 vcc, %0, %1" ::"v"(a), "v"(e) : "vcc");                                        \
    for (int i = 0; i < iters; i++) {                                                                         \
      asm volatile(REP8(REP8(OP " %0, %1, %4\n " OP " %1, %2, %5\n " OP " %2, %3, %4\n " OP " %3, %0, %5"))   \
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");                            \
    }                                                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                                               \
  }
#define K3(NAME, OP)                                                                                         \
  __global__ void __launch_bounds__(256) NAME(unsigned* out, int iters) {                                     \
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11, e = 13 + threadIdx.x, f = 17;           \
    for (int i = 0; i < iters; i++) {                                                                         \
      asm volatile(REP8(REP8(OP " %0, %1, %4, %5\n " OP " %1, %2, %5, %4\n " OP " %2, %3, %4, %5\n " OP " %3, %0, %5, %4")) \
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");                            \
    }                                                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                                               \
  }
#define K1(NAME, OP)                                                                                         \
  __global__ void __launch_bounds__(256) NAME(unsigned* out, int iters) {                                     \
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11;                                         \
    for (int i = 0; i < iters; i++) {                                                                         \
      asm volatile(REP8(REP8(OP " %0, %1\n " OP " %1, %2\n " OP " %2, %3\n " OP " %3, %0"))                   \
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");                                           \
    }                                                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                                               \
  }
K2(k_add, "v_add_u32") K2(k_sub, "v_sub_u32") K2(k_and, "v_and_b32") K2(k_or, "v_or_b32") K2(k_xor, "v_xor_b32")
K2(k_shl, "v_lshlrev_b32") K2(k_shr, "v_lshrrev_b32") K2(k_ashr, "v_ashrrev_i32") K2(k_max, "v_max_i32") K2(k_min, "v_min_u32")
K2(k_cnd, "v_cndmask_b32") K2(k_mul24, "v_mul_u32_u24") K2(k_addf, "v_add_f32") K2(k_mulf, "v_mul_f32") K2(k_addco, "v_add_co_u32")
K2(k_pkadd, "v_pk_add_u16") K2(k_pksub, "v_pk_sub_i16") K2(k_pkmax, "v_pk_max_i16") K2(k_pkshl, "v_pk_lshlrev_b16") K2(k_pkmul, "v_pk_mul_lo_u16")
K2(k_cvtpk, "v_cvt_pk_i16_i32") K2(k_addf16, "v_pk_add_f16") K2(k_max16, "v_max_i16") K2(k_add16, "v_add_u16") K2(k_lshl64, "v_xnor_b32")
K3(k_fma, "v_fma_f32") K3(k_mad24, "v_mad_u32_u24") K3(k_bfe, "v_bfe_u32") K3(k_sad, "v_sad_u8") K3(k_med3, "v_med3_i32") K3(k_add3, "v_add3_u32")
K3(k_andor, "v_and_or_b32") K3(k_or3, "v_or3_b32") K3(k_lshlor, "v_lshl_or_b32") K3(k_pkmad, "v_pk_mad_u16") K3(k_align, "v_alignbit_b32")
K3(k_perm, "v_perm_b32") K3(k_bfi, "v_bfi_b32") K3(k_xad, "v_xad_u32") K3(k_sad16, "v_sad_u16") K3(k_lerp, "v_lerp_u8")
K1(k_mov, "v_mov_b32") K1(k_satpk, "v_sat_pk_u8_i16") K1(k_cvtub, "v_cvt_f32_ubyte1") K1(k_not, "v_not_b32") K1(k_bfrev, "v_bfrev_b32") K1(k_cvtu, "v_cvt_u32_f32")

// ---- second set: encodings, operand kinds, 16-bit VOP2, compares / selects, lane access, fast + slow interleaved ----
#define KRAW(NAME, BODY)                                                                                      \
  __global__ void __launch_bounds__(256) NAME(unsigned* out, int iters) {                                     \
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 11, e = 13 + threadIdx.x, f = 17;           \
    unsigned sg = __builtin_amdgcn_readfirstlane(iters * 5);                                                  \
    for (int i = 0; i < iters; i++) {                                                                         \
      asm volatile(REP8(REP8(BODY))                                                                           \
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "s"(sg) : "vcc", "s20", "s21", "s22", "s23"); \
    }                                                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                                               \
  }
KRAW(r_add_e64, "v_add_u32_e64 %0, %1, %4\n v_add_u32_e64 %1, %2, %5\n v_add_u32_e64 %2, %3, %4\n v_add_u32_e64 %3, %0, %5")
KRAW(r_add_sgpr, "v_add_u32 %0, %6, %1\n v_add_u32 %1, %6, %2\n v_add_u32 %2, %6, %3\n v_add_u32 %3, %6, %0")
KRAW(r_add_lit, "v_add_u32 %0, 0x12345, %1\n v_add_u32 %1, 0x12345, %2\n v_add_u32 %2, 0x12345, %3\n v_add_u32 %3, 0x12345, %0")
KRAW(r_add_inl, "v_add_u32 %0, 8, %1\n v_add_u32 %1, 8, %2\n v_add_u32 %2, 8, %3\n v_add_u32 %3, 8, %0")
KRAW(r_shl_inl, "v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %1, 3, %2\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %3, 3, %0")
KRAW(r_shr_inl, "v_lshrrev_b32 %0, 3, %1\n v_lshrrev_b32 %1, 3, %2\n v_lshrrev_b32 %2, 3, %3\n v_lshrrev_b32 %3, 3, %0")
KRAW(r_cmp_cnd, "v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %0, %1, %4, vcc\n v_cmp_lt_u32 vcc, %3, %5\n v_cndmask_b32 %2, %3, %5, vcc")
KRAW(r_cmp, "v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %5\n v_cmp_lt_u32 vcc, %3, %4\n v_cmp_lt_u32 vcc, %0, %5")
KRAW(r_cmp_e64, "v_cmp_lt_u32 s[20:21], %1, %4\n v_cmp_lt_u32 s[22:23], %2, %5\n v_cmp_lt_u32 s[20:21], %3, %4\n v_cmp_lt_u32 s[22:23], %0, %5")
KRAW(r_cnd, "v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %0, %1, %4, vcc\n v_cndmask_b32 %1, %2, %5, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %0, %5, vcc\n v_cndmask_b32 %0, %1, %4, vcc\n v_cndmask_b32 %1, %2, %5, vcc\n v_cndmask_b32 %2, %3, %4, vcc")
KRAW(r_cnd_e64, "v_cmp_lt_u32 s[20:21], %1, %4\n v_cndmask_b32 %0, %1, %4, s[20:21]\n v_cndmask_b32 %1, %2, %5, s[20:21]\n v_cndmask_b32 %2, %3, %4, s[20:21]\n v_cndmask_b32 %3, %0, %5, s[20:21]\n v_cndmask_b32 %0, %1, %4, s[20:21]\n v_cndmask_b32 %1, %2, %5, s[20:21]\n v_cndmask_b32 %2, %3, %4, s[20:21]")
KRAW(r_sub16, "v_sub_u16 %0, %1, %4\n v_min_i16 %1, %2, %5\n v_max_u16 %2, %3, %4\n v_sub_u16 %3, %0, %5")
KRAW(r_shift16, "v_lshrrev_b16 %0, %1, %4\n v_ashrrev_i16 %1, %2, %5\n v_lshrrev_b16 %2, %3, %4\n v_ashrrev_i16 %3, %0, %5")
KRAW(r_shl16, "v_lshlrev_b16 %0, %1, %4\n v_lshlrev_b16 %1, %2, %5\n v_lshlrev_b16 %2, %3, %4\n v_lshlrev_b16 %3, %0, %5")
KRAW(r_mul16, "v_mul_lo_u16 %0, %1, %4\n v_mul_lo_u16 %1, %2, %5\n v_mul_lo_u16 %2, %3, %4\n v_mul_lo_u16 %3, %0, %5")
KRAW(r_fmac, "v_fmac_f32 %0, %1, %4\n v_fmac_f32 %1, %2, %5\n v_fmac_f32 %2, %3, %4\n v_fmac_f32 %3, %0, %5")
KRAW(r_subrev, "v_subrev_u32 %0, %1, %4\n v_max_u32 %1, %2, %5\n v_subrev_u32 %2, %3, %4\n v_min_i32 %3, %0, %5")
KRAW(r_cvt, "v_cvt_f32_i32 %0, %1\n v_cvt_i32_f32 %1, %2\n v_cvt_f32_u32 %2, %3\n v_cvt_u32_f32 %3, %0")
KRAW(r_rdlane, "v_readlane_b32 s20, %1, 3\n v_readlane_b32 s21, %2, 5\n v_readlane_b32 s22, %3, 7\n v_readlane_b32 s23, %0, 9")
KRAW(r_rfl, "v_readfirstlane_b32 s20, %1\n v_readfirstlane_b32 s21, %2\n v_readfirstlane_b32 s22, %3\n v_readfirstlane_b32 s23, %0")
KRAW(r_wrlane, "v_writelane_b32 %0, %6, 3\n v_writelane_b32 %1, %6, 5\n v_writelane_b32 %2, %6, 7\n v_writelane_b32 %3, %6, 9")
KRAW(r_mix_add_perm, "v_add_u32 %0, %1, %4\n v_perm_b32 %1, %2, %5, %4\n v_add_u32 %2, %3, %4\n v_perm_b32 %3, %0, %5, %4")
KRAW(r_mix_and_add3, "v_and_b32 %0, %1, %4\n v_add3_u32 %1, %2, %5, %4\n v_xor_b32 %2, %3, %4\n v_add3_u32 %3, %0, %5, %4")
KRAW(r_mix_3f1s, "v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_add3_u32 %3, %0, %5, %4")
KRAW(r_mix_salu, "v_perm_b32 %0, %1, %4, %5\n s_add_u32 s20, s20, 3\n v_perm_b32 %1, %2, %5, %4\n s_lshl_b32 s21, s20, 2\n v_perm_b32 %2, %3, %4, %5\n s_and_b32 s22, s21, 7\n v_perm_b32 %3, %0, %5, %4\n s_xor_b32 s23, s22, s20")
KRAW(r_add_dpp, "v_add_u32_dpp %0, %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %2, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %0, %5 row_shr:1 row_mask:0xf bank_mask:0xf")
KRAW(r_and_sdwa, "v_and_b32_sdwa %0, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_or_b32_sdwa %1, %2, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %2, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_or_b32_sdwa %3, %0, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD")
KRAW(r_mov_sdwa, "v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n v_mov_b32_sdwa %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_mov_b32_sdwa %2, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0\n v_mov_b32_sdwa %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3")

// ---- third set: v_cndmask_b32 reading vcc / an SGPR pair, next to its compare or not ----
KRAW(c_cmp_cnd_cnd, "v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %0, %1, %4, vcc\n v_cndmask_b32 %2, %3, %5, vcc\n v_add3_u32 %3, %0, %5, %4")
KRAW(c_cmp_cnd_x_cnd, "v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %0, %1, %4, vcc\n v_add3_u32 %3, %0, %5, %4\n v_cndmask_b32 %2, %3, %5, vcc")
KRAW(c_cnd_far, "v_cndmask_b32 %0, %1, %4, vcc\n v_add3_u32 %1, %2, %5, %4\n v_add3_u32 %2, %3, %4, %5\n v_add3_u32 %3, %0, %5, %4")
KRAW(c_cnd64_far, "v_cndmask_b32 %0, %1, %4, s[20:21]\n v_add3_u32 %1, %2, %5, %4\n v_add3_u32 %2, %3, %4, %5\n v_add3_u32 %3, %0, %5, %4")
KRAW(c_cmp64_cnd_cnd, "v_cmp_lt_u32 s[20:21], %1, %4\n v_cndmask_b32 %0, %1, %4, s[20:21]\n v_cndmask_b32 %2, %3, %5, s[20:21]\n v_add3_u32 %3, %0, %5, %4")
KRAW(c_cnd_const, "v_cndmask_b32 %0, 0, %1, vcc\n v_cndmask_b32 %1, 0, %2, vcc\n v_cndmask_b32 %2, 0, %3, vcc\n v_cndmask_b32 %3, 0, %0, vcc")
KRAW(c_cnd_same, "v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc")
KRAW(c_addco_chain, "v_add_co_u32 %0, vcc, %1, %4\n v_addc_co_u32 %1, vcc, %2, %5, vcc\n v_add_co_u32 %2, vcc, %3, %4\n v_addc_co_u32 %3, vcc, %0, %5, vcc")

// ---- fourth set: runs of nf fast-class opcodes followed by ns slow-class ones (time is reported per 4 instructions x (nf+ns)/4) ----
KRAW(f2s2, "v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4")
KRAW(f4s4, "v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8")
KRAW(f8s8, "v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8")
KRAW(f16s16, "v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8")
KRAW(f6s2, "v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4")
KRAW(f12s4, "v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8")
KRAW(f24s8, "v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8")
KRAW(f16s0, "v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5\n v_and_b32 %0, %1, %4\n v_add_u32 %1, %2, %5\n v_xor_b32 %2, %3, %4\n v_sub_u32 %3, %0, %5")
KRAW(f0s16, "v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8\n v_add3_u32 %0, %1, %4, %5\n v_perm_b32 %1, %2, %5, %4\n v_lshl_add_u32 %2, %3, 1, %4\n v_bfe_u32 %3, %0, 3, 8")
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  unsigned* d; hipMalloc(&d, (size_t)cus * 8 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 1000;
  struct { const char* n; void (*f)(unsigned*, int); } ks[] = {
#define E(x) {#x, x}
    E(k_add), E(k_sub), E(k_and), E(k_or), E(k_xor), E(k_shl), E(k_shr), E(k_ashr), E(k_max), E(k_min), E(k_cnd), E(k_mul24), E(k_addf), E(k_mulf),
    E(k_addco), E(k_pkadd), E(k_pksub), E(k_pkmax), E(k_pkshl), E(k_pkmul), E(k_cvtpk), E(k_addf16), E(k_max16), E(k_add16), E(k_lshl64),
    E(k_fma), E(k_mad24), E(k_bfe), E(k_sad), E(k_med3), E(k_add3), E(k_andor), E(k_or3), E(k_lshlor), E(k_pkmad), E(k_align), E(k_perm), E(k_bfi),
    E(k_xad), E(k_sad16), E(k_lerp), E(k_mov), E(k_satpk), E(k_cvtub), E(k_not), E(k_bfrev), E(k_cvtu),
    E(r_add_e64), E(r_add_sgpr), E(r_add_lit), E(r_add_inl), E(r_shl_inl), E(r_shr_inl), E(r_cmp_cnd), E(r_cmp), E(r_cmp_e64), E(r_cnd), E(r_cnd_e64),
    E(r_sub16), E(r_shift16), E(r_shl16), E(r_mul16), E(r_fmac), E(r_subrev), E(r_cvt), E(r_rdlane), E(r_rfl), E(r_wrlane), E(r_mix_add_perm),
    E(r_mix_and_add3), E(r_mix_3f1s), E(r_mix_salu), E(r_add_dpp), E(r_and_sdwa), E(r_mov_sdwa),
    E(c_cmp_cnd_cnd), E(c_cmp_cnd_x_cnd), E(c_cnd_far), E(c_cnd64_far), E(c_cmp64_cnd_cnd), E(c_cnd_const), E(c_cnd_same), E(c_addco_chain),
    E(f2s2), E(f4s4), E(f8s8), E(f16s16), E(f6s2), E(f12s4), E(f24s8), E(f16s0), E(f0s16)};
  for (auto& k : ks) {
    for (int wps = 1; wps <= 8; wps *= 8) {
      const int grid = cus * wps;
      k.f<<<grid, 256>>>(d, iters); hipDeviceSynchronize();
      hipEventRecord(e0); k.f<<<grid, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-10s waves/SIMD %d: %.2f ns per wave-instruction per SIMD\n", k.n + 2, wps, ms * 1e6 / ((double)iters * 256 * wps));
    }
  }
  return 0;
}
