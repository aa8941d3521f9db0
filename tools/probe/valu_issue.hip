// valu_issue.hip -- how many instructions per cycle does ONE SIMD of gfx950 issue, by instruction class and by the number
// of waves resident on it?  (development aid, GPU box only; DESIGN.md section 5 quotes its output, profiles/r06_valu_issue.txt)
//
//   hipcc --offload-arch=gfx950 -O2 tools/probe/valu_issue.hip -o tools/probe/valu_issue && tools/probe/valu_issue
//
// Every class is a stream of 64 independent instructions (8 chains of 8) per loop trip, written in inline asm so that the
// compiler neither merges nor reorders them.  A launch puts w = 1, 2, 4, 8 waves on every SIMD of every CU (workgroups of
// 4w waves, one -- for w = 8 two -- per CU); every wave stamps s_memtime before and after its loop and records HW_ID.  Per
// SIMD: instructions issued by its waves / (last end - first start) = instructions per shader cycle per SIMD; the table
// prints the median over the SIMDs that held exactly w waves, and the same rate from the launch's wall time (hipEvents; it
// needs the clock, printed as measured by a s_memtime/wall pair).  "cycles per wave-instruction" = 1 / that rate.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

struct Rec {
	unsigned hw_id, xcc_id;
	unsigned long long t0, t1;
};

#define R8(X) X X X X X X X X
#define PROLOGUE \
	unsigned r0 = threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 ^ 0x55, r5 = r0 + 9, r6 = r0 * 11, r7 = r0 + 77; \
	unsigned k = threadIdx.x | 1; \
	const unsigned a4 = (threadIdx.x * 4) & 4095, a8 = (threadIdx.x * 8) & 8191, ag = ((threadIdx.x * 2654435761u) >> 18) & ~7u; \
	__shared__ unsigned lds[4096]; \
	lds[threadIdx.x] = r1; lds[threadIdx.x + 1024] = r2; lds[threadIdx.x + 2048] = r3; lds[threadIdx.x + 3072] = r4; \
	__syncthreads(); \
	unsigned hw, xcc; \
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); \
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); \
	unsigned long long t0, t1; \
	asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#define EPILOGUE \
	asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
	if ((threadIdx.x & 63) == 0) { \
		Rec r = { hw, xcc, t0, t1 }; \
		out[(size_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r; \
	} \
	if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 0x12345678u) sink[threadIdx.x] = r0;

// 8 independent 32-bit chains: OP is a string with %0..%7 the chains and %8 a per-lane operand
#define DEFK32(name, ...) DEFK32_(name, __VA_ARGS__)
#define DEFK32_(name, I0, I1, I2, I3, I4, I5, I6, I7) \
	__global__ __launch_bounds__(1024) void k_##name(Rec *out, unsigned *sink, int iters) { \
		PROLOGUE \
		for (int i = 0; i < iters; i++) \
			asm volatile(R8(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n") \
				: "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(k), "v"(a4) : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "scc"); \
		EPILOGUE \
	}
#define SAME8(F) F("%0"), F("%1"), F("%2"), F("%3"), F("%4"), F("%5"), F("%6"), F("%7")

#define F_ADD(r) "v_add_u32 " r ", " r ", %8"
DEFK32(v_add_u32, SAME8(F_ADD))
#define F_LSHL_ADD(r) "v_lshl_add_u32 " r ", " r ", 1, %8"
DEFK32(v_lshl_add_u32, SAME8(F_LSHL_ADD))
#define F_AND_OR(r) "v_and_or_b32 " r ", " r ", %8, %8"
DEFK32(v_and_or_b32, SAME8(F_AND_OR))
#define F_PERM(r) "v_perm_b32 " r ", " r ", %8, %8"
DEFK32(v_perm_b32, SAME8(F_PERM))
#define F_ALIGNBIT(r) "v_alignbit_b32 " r ", " r ", %8, 7"
DEFK32(v_alignbit_b32, SAME8(F_ALIGNBIT))
#define F_BFE(r) "v_bfe_u32 " r ", " r ", 3, 9"
DEFK32(v_bfe_u32, SAME8(F_BFE))
#define F_MAD24(r) "v_mad_u32_u24 " r ", " r ", %8, %8"
DEFK32(v_mad_u32_u24, SAME8(F_MAD24))
#define F_MUL_LO(r) "v_mul_lo_u32 " r ", " r ", %8"
DEFK32(v_mul_lo_u32, SAME8(F_MUL_LO))
#define F_MIN(r) "v_min_u32 " r ", " r ", %8"
DEFK32(v_min_u32, SAME8(F_MIN))
#define F_DPP(r) "v_mov_b32_dpp " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf"
DEFK32(v_mov_dpp_row_shr, SAME8(F_DPP))
#define F_DPPADD(r) "v_add_u32_dpp " r ", " r ", %8 row_shr:1 row_mask:0xf bank_mask:0xf"
DEFK32(v_add_dpp_row_shr, SAME8(F_DPPADD))
#define F_WSHR(r) "v_mov_b32_dpp " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf"
DEFK32(v_mov_dpp_wave_shr, SAME8(F_WSHR))
#define F_BCAST(r) "v_mov_b32_dpp " r ", " r " row_bcast:15 row_mask:0xa bank_mask:0xf"
DEFK32(v_mov_dpp_row_bcast, SAME8(F_BCAST))
#define F_SDWA(r) "v_add_u32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"
DEFK32(v_add_sdwa, SAME8(F_SDWA))
// a compare into an SGPR pair (what a ballot is) -- eight different destinations
DEFK32(v_cmp_to_sgpr, "v_cmp_lt_u32 s[40:41], %0, %8", "v_cmp_lt_u32 s[42:43], %1, %8", "v_cmp_lt_u32 s[44:45], %2, %8", "v_cmp_lt_u32 s[46:47], %3, %8",
	"v_cmp_lt_u32 s[48:49], %4, %8", "v_cmp_lt_u32 s[50:51], %5, %8", "v_cmp_lt_u32 s[52:53], %6, %8", "v_cmp_lt_u32 s[54:55], %7, %8")
// compare into VCC and the select that reads it (a dependent pair; 4 pairs per 8)
DEFK32(v_cmp_vcc_cndmask, "v_cmp_lt_u32 vcc, %0, %8", "v_cndmask_b32 %0, %0, %8, vcc", "v_cmp_lt_u32 vcc, %1, %8", "v_cndmask_b32 %1, %1, %8, vcc",
	"v_cmp_lt_u32 vcc, %2, %8", "v_cndmask_b32 %2, %2, %8, vcc", "v_cmp_lt_u32 vcc, %3, %8", "v_cndmask_b32 %3, %3, %8, vcc")
#define F_CNDS(r) "v_cndmask_b32 " r ", " r ", %8, s[40:41]"
DEFK32(v_cndmask_sgpr, SAME8(F_CNDS))
#define F_RDLANE(n, r) "v_readlane_b32 s" n ", " r ", 5"
DEFK32(v_readlane, F_RDLANE("40", "%0"), F_RDLANE("41", "%1"), F_RDLANE("42", "%2"), F_RDLANE("43", "%3"), F_RDLANE("44", "%4"), F_RDLANE("45", "%5"), F_RDLANE("46", "%6"), F_RDLANE("47", "%7"))
#define F_RDFIRST(n, r) "v_readfirstlane_b32 s" n ", " r
DEFK32(v_readfirstlane, F_RDFIRST("40", "%0"), F_RDFIRST("41", "%1"), F_RDFIRST("42", "%2"), F_RDFIRST("43", "%3"), F_RDFIRST("44", "%4"), F_RDFIRST("45", "%5"), F_RDFIRST("46", "%6"), F_RDFIRST("47", "%7"))
#define F_BPERM(r) "ds_bpermute_b32 " r ", %9, " r
DEFK32(ds_bpermute, F_BPERM("%0"), F_BPERM("%1"), F_BPERM("%2"), F_BPERM("%3"), F_BPERM("%4"), F_BPERM("%5"), F_BPERM("%6"), F_BPERM("%7") "\n s_waitcnt lgkmcnt(0)")
// scalar ALU alone, and scalar beside vector one for one in the SAME wave
#define F_SADD(n) "s_add_u32 s" n ", s" n ", 3"
DEFK32(s_add_u32, F_SADD("40"), F_SADD("41"), F_SADD("42"), F_SADD("43"), F_SADD("44"), F_SADD("45"), F_SADD("46"), F_SADD("47"))
DEFK32(v_add_plus_s_add_1to1, F_ADD("%0"), F_SADD("40"), F_ADD("%1"), F_SADD("41"), F_ADD("%2"), F_SADD("42"), F_ADD("%3"), F_SADD("43"))
DEFK32(v_add_x3_plus_s_add, F_ADD("%0"), F_ADD("%1"), F_ADD("%2"), F_SADD("40"), F_ADD("%3"), F_ADD("%4"), F_ADD("%5"), F_SADD("41"))
// a taken scalar branch per four vector instructions (the lean kernel: 59 branches per 585 VALU)
DEFK32(v_add_x4_plus_branch, F_ADD("%0"), F_ADD("%1"), F_ADD("%2"), F_ADD("%3") "\n s_cmp_lg_u32 s40, s41\n s_cbranch_scc0 1f\n s_nop 0\n1:", F_ADD("%4"), F_ADD("%5"), F_ADD("%6"), F_ADD("%7") "\n s_cmp_lg_u32 s40, s41\n s_cbranch_scc0 1f\n s_nop 0\n1:")
// LDS: 8 independent reads per wait, conflict-free (lane-linear dwords), and the 2-byte form the merge uses
#define F_DSR32(r, o) "ds_read_b32 " r ", %9 offset:" o
DEFK32(ds_read_b32, F_DSR32("%0", "0"), F_DSR32("%1", "1024"), F_DSR32("%2", "2048"), F_DSR32("%3", "3072"), F_DSR32("%4", "4096"), F_DSR32("%5", "5120"), F_DSR32("%6", "6144"), F_DSR32("%7", "7168") "\n s_waitcnt lgkmcnt(0)")
#define F_DSR64(r, o) "ds_read_b64 " r ", %9 offset:" o
#define F_DSRU16(r, o) "ds_read_u16 " r ", %9 offset:" o
DEFK32(ds_read_u16, F_DSRU16("%0", "0"), F_DSRU16("%1", "1024"), F_DSRU16("%2", "2048"), F_DSRU16("%3", "3072"), F_DSRU16("%4", "4096"), F_DSRU16("%5", "5120"), F_DSRU16("%6", "6144"), F_DSRU16("%7", "7168") "\n s_waitcnt lgkmcnt(0)")
#define F_DSW32(r, o) "ds_write_b32 %9, " r " offset:" o
DEFK32(ds_write_b32, F_DSW32("%0", "0"), F_DSW32("%1", "1024"), F_DSW32("%2", "2048"), F_DSW32("%3", "3072"), F_DSW32("%4", "4096"), F_DSW32("%5", "5120"), F_DSW32("%6", "6144"), F_DSW32("%7", "7168") "\n s_waitcnt lgkmcnt(0)")
#define F_DSMIN(r, o) "ds_min_rtn_u32 " r ", %9, " r " offset:" o
DEFK32(ds_min_rtn_u32, F_DSMIN("%0", "0"), F_DSMIN("%1", "1024"), F_DSMIN("%2", "2048"), F_DSMIN("%3", "3072"), F_DSMIN("%4", "4096"), F_DSMIN("%5", "5120"), F_DSMIN("%6", "6144"), F_DSMIN("%7", "7168") "\n s_waitcnt lgkmcnt(0)")
// vector beside LDS reads in the same wave, 3 : 1
DEFK32(v_add_x6_plus_ds_read_x2, F_ADD("%0"), F_ADD("%1"), F_ADD("%2"), F_DSR32("%6", "0"), F_ADD("%3"), F_ADD("%4"), F_ADD("%5"), F_DSR32("%7", "1024") "\n s_waitcnt lgkmcnt(0)")

// 64-bit classes: 8 independent f64 chains
#define DEFK64(name, ...) DEFK64_(name, __VA_ARGS__)
#define DEFK64_(name, I0, I1, I2, I3, I4, I5, I6, I7) \
	__global__ __launch_bounds__(1024) void k_##name(Rec *out, unsigned *sink, int iters) { \
		PROLOGUE \
		double d0 = r0, d1 = r1, d2 = r2, d3 = r3, d4 = r4, d5 = r5, d6 = r6, d7 = r7, dk = 1.0 + 1e-9 * k; \
				for (int i = 0; i < iters; i++) \
			asm volatile(R8(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n") \
				: "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(dk), "v"(a8), "v"(ag) : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"); \
		r0 = (unsigned) (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7); \
		EPILOGUE \
	}
#define F_ADDF64(r) "v_add_f64 " r ", " r ", %8"
DEFK64(v_add_f64, SAME8(F_ADDF64))
#define F_MULF64(r) "v_mul_f64 " r ", " r ", %8"
DEFK64(v_mul_f64, SAME8(F_MULF64))
#define F_FMAF64(r) "v_fma_f64 " r ", " r ", %8, %8"
DEFK64(v_fma_f64, SAME8(F_FMAF64))
#define F_CMPF64(n, r) "v_cmp_lt_f64 s[" n "], " r ", %8"
DEFK64(v_cmp_f64_to_sgpr, F_CMPF64("40:41", "%0"), F_CMPF64("42:43", "%1"), F_CMPF64("44:45", "%2"), F_CMPF64("46:47", "%3"), F_CMPF64("48:49", "%4"), F_CMPF64("50:51", "%5"), F_CMPF64("52:53", "%6"), F_CMPF64("54:55", "%7"))
#define F_LSHL64(r) "v_lshlrev_b64 " r ", 1, " r
DEFK64(v_lshlrev_b64, SAME8(F_LSHL64))
#define F_PKADD(r) "v_pk_add_f32 " r ", " r ", %8"
DEFK64(v_pk_add_f32, SAME8(F_PKADD))
DEFK64(ds_read_b64, F_DSR64("%0", "0"), F_DSR64("%1", "1024"), F_DSR64("%2", "2048"), F_DSR64("%3", "3072"), F_DSR64("%4", "4096"), F_DSR64("%5", "5120"), F_DSR64("%6", "6144"), F_DSR64("%7", "7168") "\n s_waitcnt lgkmcnt(0)")
// the merge's table gather: an 8-byte read at a per-lane pseudo-random (8-byte aligned) address within 16 KB
#define F_DSG64(r) "ds_read_b64 " r ", %10"
DEFK64(ds_read_b64_gather, F_DSG64("%0"), F_DSG64("%1"), F_DSG64("%2"), F_DSG64("%3"), F_DSG64("%4"), F_DSG64("%5"), F_DSG64("%6"), F_DSG64("%7") "\n s_waitcnt lgkmcnt(0)")

// more one- and two-operand classes: which of them issue at the doubled rate v_add_u32 does?
#define F_MOV(r) "v_mov_b32 " r ", %8"
DEFK32(v_mov_b32, SAME8(F_MOV))
#define F_AND(r) "v_and_b32 " r ", " r ", %8"
DEFK32(v_and_b32, SAME8(F_AND))
#define F_OR(r) "v_or_b32 " r ", " r ", %8"
DEFK32(v_or_b32, SAME8(F_OR))
#define F_XOR(r) "v_xor_b32 " r ", " r ", %8"
DEFK32(v_xor_b32, SAME8(F_XOR))
#define F_LSHL(r) "v_lshlrev_b32 " r ", 3, " r
DEFK32(v_lshlrev_b32, SAME8(F_LSHL))
#define F_LSHR(r) "v_lshrrev_b32 " r ", 3, " r
DEFK32(v_lshrrev_b32, SAME8(F_LSHR))
#define F_LSHRV(r) "v_lshrrev_b32 " r ", %8, " r
DEFK32(v_lshrrev_b32_by_vgpr, SAME8(F_LSHRV))
#define F_SUB(r) "v_sub_u32 " r ", " r ", %8"
DEFK32(v_sub_u32, SAME8(F_SUB))
#define F_SUBREV(r) "v_subrev_u32 " r ", " r ", %8"
DEFK32(v_subrev_u32, SAME8(F_SUBREV))
#define F_MAX(r) "v_max_u32 " r ", " r ", %8"
DEFK32(v_max_u32, SAME8(F_MAX))
#define F_ADDLIT(r) "v_add_u32 " r ", 0x12345, " r
DEFK32(v_add_u32_literal, SAME8(F_ADDLIT))
#define F_ANDLIT(r) "v_and_b32 " r ", 0x7F7F7F7F, " r
DEFK32(v_and_b32_literal, SAME8(F_ANDLIT))
#define F_ADDINL(r) "v_add_u32 " r ", 5, " r
DEFK32(v_add_u32_inline, SAME8(F_ADDINL))
#define F_ADDS(r) "v_add_u32 " r ", s40, " r
DEFK32(v_add_u32_sgpr, SAME8(F_ADDS))
#define F_ADDE64(r) "v_add_u32_e64 " r ", " r ", %8"
DEFK32(v_add_u32_e64, SAME8(F_ADDE64))
#define F_ADDCO(r) "v_add_co_u32 " r ", vcc, " r ", %8"
DEFK32(v_add_co_u32, SAME8(F_ADDCO))
#define F_MUL24(r) "v_mul_u32_u24 " r ", " r ", %8"
DEFK32(v_mul_u32_u24, SAME8(F_MUL24))
#define F_CNDV(r) "v_cndmask_b32 " r ", " r ", %8, vcc"
DEFK32(v_cndmask_vcc, SAME8(F_CNDV))
#define F_BFI(r) "v_bfi_b32 " r ", " r ", %8, %8"
DEFK32(v_bfi_b32, SAME8(F_BFI))
#define F_ADD3(r) "v_add3_u32 " r ", " r ", %8, %8"
DEFK32(v_add3_u32, SAME8(F_ADD3))
#define F_OR3(r) "v_or3_b32 " r ", " r ", %8, %8"
DEFK32(v_or3_b32, SAME8(F_OR3))
#define F_BFREV(r) "v_bfrev_b32 " r ", " r
DEFK32(v_bfrev_b32, SAME8(F_BFREV))
#define F_BCNT(r) "v_bcnt_u32_b32 " r ", " r ", %8"
DEFK32(v_bcnt_u32_b32, SAME8(F_BCNT))
#define F_FFBH(r) "v_ffbh_u32 " r ", " r
DEFK32(v_ffbh_u32, SAME8(F_FFBH))
#define F_ADDF32(r) "v_add_f32 " r ", " r ", %8"
DEFK32(v_add_f32, SAME8(F_ADDF32))
#define F_MULF32(r) "v_mul_f32 " r ", " r ", %8"
DEFK32(v_mul_f32, SAME8(F_MULF32))
#define F_FMAF32(r) "v_fma_f32 " r ", " r ", %8, %8"
DEFK32(v_fma_f32, SAME8(F_FMAF32))
#define F_FMACF32(r) "v_fmac_f32 " r ", %8, %8"
DEFK32(v_fmac_f32, SAME8(F_FMACF32))
#define F_CVT(r) "v_cvt_f32_u32 " r ", " r
DEFK32(v_cvt_f32_u32, SAME8(F_CVT))
#define F_RCP(r) "v_rcp_f32 " r ", " r
DEFK32(v_rcp_f32, SAME8(F_RCP))
#define F_CMPVCC(r) "v_cmp_lt_u32 vcc, " r ", %8"
DEFK32(v_cmp_vcc_only, SAME8(F_CMPVCC))
#define F_WRLANE(r) "v_writelane_b32 " r ", s40, 5"
DEFK32(v_writelane, SAME8(F_WRLANE))
#define F_MBCNT(r) "v_mbcnt_lo_u32_b32 " r ", %8, " r
DEFK32(v_mbcnt_lo, SAME8(F_MBCNT))
// one fast and one slow class interleaved 1 : 1 -- do they share one issue budget?
DEFK32(v_add_u32_plus_v_perm_1to1, F_ADD("%0"), F_PERM("%1"), F_ADD("%2"), F_PERM("%3"), F_ADD("%4"), F_PERM("%5"), F_ADD("%6"), F_PERM("%7"))
DEFK32(v_add_u32_plus_v_add_f64_like_min, F_ADD("%0"), F_MIN("%1"), F_ADD("%2"), F_MIN("%3"), F_ADD("%4"), F_MIN("%5"), F_ADD("%6"), F_MIN("%7"))
// ... and in runs: do two fast ones side by side get their short slots back?
DEFK32(v_add_x2_plus_v_perm_x2, F_ADD("%0"), F_ADD("%1"), F_PERM("%2"), F_PERM("%3"), F_ADD("%4"), F_ADD("%5"), F_PERM("%6"), F_PERM("%7"))
DEFK32(v_add_x4_plus_v_perm_x4, F_ADD("%0"), F_ADD("%1"), F_ADD("%2"), F_ADD("%3"), F_PERM("%4"), F_PERM("%5"), F_PERM("%6"), F_PERM("%7"))
DEFK32(v_add_x3_plus_v_perm_x1, F_ADD("%0"), F_ADD("%1"), F_ADD("%2"), F_PERM("%3"), F_ADD("%4"), F_ADD("%5"), F_ADD("%6"), F_PERM("%7"))
// a vector instruction of the slow class and a scalar one, 1 : 1 and 3 : 1 -- does the scalar one ride along?
DEFK32(v_perm_plus_s_add_1to1, F_PERM("%0"), F_SADD("40"), F_PERM("%1"), F_SADD("41"), F_PERM("%2"), F_SADD("42"), F_PERM("%3"), F_SADD("43"))
DEFK32(v_perm_x3_plus_s_add, F_PERM("%0"), F_PERM("%1"), F_PERM("%2"), F_SADD("40"), F_PERM("%3"), F_PERM("%4"), F_PERM("%5"), F_SADD("41"))

typedef void (*kern_t)(Rec *, unsigned *, int);
struct Class {
	const char *name;
	kern_t fn;
	int instr_per_trip;	// instructions of the class per loop trip (the fillers -- waits, branches -- are not counted)
};
#define C(name, n) { #name, k_##name, n }
static const Class classes[] = {
	C(v_add_u32, 64), C(v_lshl_add_u32, 64), C(v_and_or_b32, 64), C(v_perm_b32, 64), C(v_alignbit_b32, 64), C(v_bfe_u32, 64), C(v_mad_u32_u24, 64),
	C(v_mul_lo_u32, 64), C(v_min_u32, 64), C(v_mov_dpp_row_shr, 64), C(v_add_dpp_row_shr, 64), C(v_mov_dpp_wave_shr, 64), C(v_mov_dpp_row_bcast, 64),
	C(v_add_sdwa, 64), C(v_cmp_to_sgpr, 64), C(v_cmp_vcc_cndmask, 64), C(v_cndmask_sgpr, 64), C(v_readlane, 64), C(v_readfirstlane, 64), C(ds_bpermute, 64),
	C(s_add_u32, 64), C(v_add_plus_s_add_1to1, 64), C(v_add_x3_plus_s_add, 64), C(v_add_x4_plus_branch, 64),
	C(v_add_f64, 64), C(v_mul_f64, 64), C(v_fma_f64, 64), C(v_cmp_f64_to_sgpr, 64), C(v_lshlrev_b64, 64), C(v_pk_add_f32, 64),
	C(v_mov_b32, 64), C(v_and_b32, 64), C(v_or_b32, 64), C(v_xor_b32, 64), C(v_lshlrev_b32, 64), C(v_lshrrev_b32, 64), C(v_lshrrev_b32_by_vgpr, 64), C(v_sub_u32, 64), C(v_subrev_u32, 64), C(v_max_u32, 64),
	C(v_add_u32_literal, 64), C(v_and_b32_literal, 64), C(v_add_u32_inline, 64), C(v_add_u32_sgpr, 64), C(v_add_u32_e64, 64), C(v_add_co_u32, 64), C(v_mul_u32_u24, 64), C(v_cndmask_vcc, 64), C(v_bfi_b32, 64), C(v_add3_u32, 64),
	C(v_or3_b32, 64), C(v_bfrev_b32, 64), C(v_bcnt_u32_b32, 64), C(v_ffbh_u32, 64), C(v_add_f32, 64), C(v_mul_f32, 64), C(v_fma_f32, 64), C(v_fmac_f32, 64), C(v_cvt_f32_u32, 64), C(v_rcp_f32, 64), C(v_cmp_vcc_only, 64),
	C(v_writelane, 64), C(v_mbcnt_lo, 64), C(v_add_u32_plus_v_perm_1to1, 64), C(v_add_u32_plus_v_add_f64_like_min, 64),
	C(v_add_x2_plus_v_perm_x2, 64), C(v_add_x4_plus_v_perm_x4, 64), C(v_add_x3_plus_v_perm_x1, 64), C(v_perm_plus_s_add_1to1, 64), C(v_perm_x3_plus_s_add, 64),
	C(ds_read_b32, 64), C(ds_read_u16, 64), C(ds_read_b64, 64), C(ds_read_b64_gather, 64), C(ds_write_b32, 64), C(ds_min_rtn_u32, 64), C(v_add_x6_plus_ds_read_x2, 64),
};

static double median(std::vector<double> &v) {
	if (v.empty())
		return 0;
	std::sort(v.begin(), v.end());
	return v[v.size() / 2];
}

int main(int argc, char **argv) {
	const int iters = argc > 1 ? atoi(argv[1]) : 2000;
	const char *only = argc > 2 ? argv[2] : nullptr;
	hipDeviceProp_t prop;
	hipGetDeviceProperties(&prop, 0);
	const int cus = prop.multiProcessorCount;
	Rec *d;
	unsigned *sink;
	hipMalloc(&d, sizeof(Rec) * (size_t) cus * 2 * 16);
	hipMalloc(&sink, 4096 * 4);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	printf("# valu_issue: %s, %d CUs, clockRate %d kHz; %d loop trips of 64 instructions per wave\n", prop.gcnArchName, cus, prop.clockRate, iters);
	printf("# per class and waves per SIMD: instructions per shader cycle per SIMD by s_memtime (median over SIMDs holding exactly that many waves), cycles per\n");
	printf("# wave-instruction (its inverse), the s_memtime ticks per microsecond of wall time (= the clock the run held), SIMDs counted\n");
	printf("%-28s %5s %12s %12s %10s %7s\n", "class", "w/SIMD", "instr/cyc", "cyc/instr", "ticks/us", "SIMDs");
	for (const Class &c : classes) {
		if (only && !strstr(c.name, only))
			continue;
		for (int w : { 1, 2, 4, 8 }) {
			const int per_cu = w == 8 ? 2 : 1;
			const int waves = w == 8 ? 16 : 4 * w;
			const int wgs = cus * per_cu;
			hipMemset(d, 0, sizeof(Rec) * (size_t) wgs * waves);
			hipLaunchKernelGGL(c.fn, dim3(wgs), dim3(waves * 64), 0, 0, d, sink, 10);	// warm the instruction cache
			hipEventRecord(e0);
			hipLaunchKernelGGL(c.fn, dim3(wgs), dim3(waves * 64), 0, 0, d, sink, iters);
			hipEventRecord(e1);
			if (hipDeviceSynchronize() != hipSuccess) {
				printf("%s: launch failed\n", c.name);
				return 1;
			}
			float ms;
			hipEventElapsedTime(&ms, e0, e1);
			std::vector<Rec> h((size_t) wgs * waves);
			hipMemcpy(h.data(), d, sizeof(Rec) * h.size(), hipMemcpyDeviceToHost);
			struct Simd {
				int n = 0;
				unsigned long long first = ~0ull, last = 0;
			};
			std::map<unsigned, Simd> simds;
			unsigned long long lo = ~0ull, hi = 0;
			for (const Rec &r : h) {
				const unsigned cu = (r.hw_id >> 8) & 0xF, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7, xcc = r.xcc_id & 0xF, simd = (r.hw_id >> 4) & 3;
				Simd &s = simds[(xcc << 16) | (se << 12) | (sh << 8) | (cu << 4) | simd];
				s.n++;
				s.first = std::min(s.first, r.t0);
				s.last = std::max(s.last, r.t1);
				lo = std::min(lo, r.t0);
				hi = std::max(hi, r.t1);
			}
			std::vector<double> rates, spans;
			for (auto &kv : simds)
				if (kv.second.n == w) {
					rates.push_back((double) w * iters * c.instr_per_trip / (double) (kv.second.last - kv.second.first));
					spans.push_back((double) (kv.second.last - kv.second.first));
				}
			const double r = median(rates);
			// (the XCDs' counters are not in step with each other: the clock is taken from one SIMD's span -- all SIMDs run the whole launch -- against the launch's event time)
			printf("%-28s %5d %12.4f %12.3f %10.1f %7zu\n", c.name, w, r, r > 0 ? 1.0 / r : 0.0, median(spans) / ((double) ms * 1e3), rates.size());
			fflush(stdout);
		}
	}
	return 0;
}
