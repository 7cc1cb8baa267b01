// tools/ubench/valu_rate.hip -- how many cycles of a SIMD does one wave64 VALU instruction of a given kind take when the SIMD
// has 1 / 2 / 4 wavefronts to pick from (gfx950)?  One workgroup of 4 * w wavefronts on one CU (its wavefronts go round the
// four SIMDs), every wavefront runs the same loop of 64 instructions over eight independent accumulators; cycles by
// s_memtime of the slowest wavefront.  (What the wave extension's "VALU issue" bound is measured against: DESIGN 4.5.)
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY(INSTR)                                                                                  \
  asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                   \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                   \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                   \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                   \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                   \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                   \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                   \
               INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                   \
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])  \
               : "v"(c), "s"(m) : "vcc", "s22", "s23");

#define I_ADD(i)   "v_add_u32 %" #i ", %" #i ", %8\n"
#define I_XOR(i)   "v_xor_b32 %" #i ", %" #i ", %8\n"
#define I_FMA(i)   "v_fma_f32 %" #i ", %" #i ", %8, %8\n"
#define I_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define I_CND(i)   "v_cndmask_b32 %" #i ", %" #i ", %8, %9\n"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %8\n"
#define I_DPP(i)   "v_mov_b32_dpp %" #i ", %" #i " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPPROW(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MAXDPP(i) "v_max_i32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_BCNT(i)  "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define I_CMP(i)   "v_cmp_lt_i32 vcc, %" #i ", %8\n"
#define I_LSHL64(i) "v_lshlrev_b64 %10, 3, %10\n"
#define I_MAD64(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define I_SALU(i)  "s_add_u32 s20, s20, s21\n"
#define I_CNDVCC(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define I_MOV(i)   "v_mov_b32 %" #i ", %8\n"
#define I_CNDVCC64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define I_CNDVCCX(i) "v_cndmask_b32_e32 %" #i ", %8, %" #i ", vcc\n"
#define I_CNDMIX(i) "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n v_add_u32 %" #i ", %" #i ", %8\n"
#define I_OR(i)    "v_or_b32 %" #i ", %" #i ", %8\n"
#define I_SUBB(i)  "v_subrev_u32 %" #i ", %" #i ", %8\n"
#define I_ASHR(i)  "v_ashrrev_i32 %" #i ", 1, %" #i "\n"
#define I_NOT(i)   "v_not_b32 %" #i ", %" #i "\n"
#define I_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define I_LSHL(i)  "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define I_SUB(i)   "v_sub_u32 %" #i ", %" #i ", %8\n"
#define I_MIN(i)   "v_min_i32 %" #i ", %" #i ", %8\n"
#define I_AND(i)   "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define I_ADD3(i)  "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define I_FFBL(i)  "v_ffbl_b32 %" #i ", %" #i "\n"
#define I_CMP64(i) "v_cmp_lt_i32_e64 s[22:23], %" #i ", %8\n"
#define I_SDWA(i)  "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
#define I_RDLANE(i) "v_readlane_b32 s22, %" #i ", 5\n"
#define I_ADDK(i)  "v_add_u32 %" #i ", 0x12345, %" #i "\n"
#define I_ADDS(i)  "v_add_u32 %" #i ", s9, %" #i "\n"
#define I_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i "\n"
#define I_BFE(i)   "v_bfe_u32 %" #i ", %" #i ", 3, 7\n"
#define I_XORE64(i) "v_xor_b32_e64 %" #i ", %" #i ", %8\n"
#define I_ADDC(i)  "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define I_MAX(i)   "v_max_i32 %" #i ", %" #i ", %8\n"
#define I_LSHR64(i) "v_lshrrev_b64 %" #i ", 0, %" #i "\n"

template <int KIND>
__global__ void k(int n, unsigned *out, unsigned long long *cyc)
{ unsigned a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x*(i+3);
  unsigned c = threadIdx.x | 1;
  unsigned long long m = 0x5555aaaa5555aaaaull;
  asm volatile("s_mov_b64 vcc, %0" :: "s"(m) : "vcc");
  __syncthreads();
  unsigned long long t0 = clock64();
  for (int it = 0; it < n; it++)
    { if (KIND == 0) BODY(I_ADD)
      if (KIND == 1) BODY(I_XOR)
      if (KIND == 2) BODY(I_FMA)
      if (KIND == 3) BODY(I_ALIGN)
      if (KIND == 4) BODY(I_CND)
      if (KIND == 5) BODY(I_ANDOR)
      if (KIND == 6) BODY(I_DPP)
      if (KIND == 7) BODY(I_DPPROW)
      if (KIND == 8) BODY(I_MAXDPP)
      if (KIND == 9) BODY(I_BCNT)
      if (KIND == 10) BODY(I_CMP)
      if (KIND == 11) BODY(I_MAD64)
      if (KIND == 13) BODY(I_CNDVCC)
      if (KIND == 14) BODY(I_MOV)
      if (KIND == 15) BODY(I_LSHL)
      if (KIND == 16) BODY(I_SUB)
      if (KIND == 17) BODY(I_MIN)
      if (KIND == 18) BODY(I_AND)
      if (KIND == 19) BODY(I_LSHLADD)
      if (KIND == 20) BODY(I_ADD3)
      if (KIND == 21) BODY(I_FFBL)
      if (KIND == 22) BODY(I_CMP64)
      if (KIND == 23) BODY(I_SDWA)
      if (KIND == 24) BODY(I_RDLANE)
      if (KIND == 25) BODY(I_ADDK)
      if (KIND == 26) BODY(I_ADDS)
      if (KIND == 27) BODY(I_MBCNT)
      if (KIND == 28) BODY(I_BFE)
      if (KIND == 29) BODY(I_XORE64)
      if (KIND == 30) BODY(I_ADDC)
      if (KIND == 31) BODY(I_MAX)
      if (KIND == 32) BODY(I_CNDVCC64)
      if (KIND == 33) BODY(I_CNDVCCX)
      if (KIND == 34) BODY(I_CNDMIX)
      if (KIND == 35) BODY(I_OR)
      if (KIND == 36) BODY(I_ASHR)
      if (KIND == 37) BODY(I_NOT)
      if (KIND == 38) BODY(I_ADDCO)
      if (KIND == 12) { asm volatile(I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0)
                                     I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0)
                                     I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0)
                                     I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0)
                                     I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0)
                                     I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0)
                                     I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0)
                                     I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0) I_SALU(0)
                                     ::: "s20", "s21", "scc"); }
    }
  unsigned long long t1 = clock64();
  unsigned s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[threadIdx.x] = s;
  atomicMax(cyc,t1-t0);
}

template <int KIND> static void run(const char *name, unsigned *out, unsigned long long *cyc)
{ const int n = 4000;
  printf("%-34s",name);
  for (int w = 1; w <= 4; w *= 2)
    { unsigned long long h = 0;
      hipMemset(cyc,0,8);
      hipLaunchKernelGGL(k<KIND>,dim3(1),dim3(256*w),0,0,n,out,cyc);
      hipDeviceSynchronize();
      hipMemcpy(&h,cyc,8,hipMemcpyDeviceToHost);
      printf("  %d/SIMD: %5.2f", w, (double) h / ((double) n*64*w));
    }
  { unsigned long long h = 0;                 // one wavefront alone on the CU
    hipMemset(cyc,0,8);
    hipLaunchKernelGGL(k<KIND>,dim3(1),dim3(64),0,0,n,out,cyc);
    hipDeviceSynchronize();
    hipMemcpy(&h,cyc,8,hipMemcpyDeviceToHost);
    printf("  lone wavefront: %5.2f", (double) h / ((double) n*64));
  }
  printf("   cycles of a SIMD per instruction\n");
}

int main()
{ unsigned *out; unsigned long long *cyc;
  hipMalloc(&out,4096*4); hipMalloc(&cyc,64);
  run<0>("v_add_u32",out,cyc);
  run<1>("v_xor_b32",out,cyc);
  run<2>("v_fma_f32",out,cyc);
  run<3>("v_alignbit_b32",out,cyc);
  run<4>("v_cndmask_b32 (sgpr mask)",out,cyc);
  run<5>("v_and_or_b32",out,cyc);
  run<6>("v_mov_b32_dpp wave_shr:1",out,cyc);
  run<7>("v_mov_b32_dpp row_shr:1",out,cyc);
  run<8>("v_max_i32_dpp row_shr:1",out,cyc);
  run<9>("v_bcnt_u32_b32",out,cyc);
  run<10>("v_cmp_lt_i32 -> vcc",out,cyc);
  run<11>("v_mul_lo_u32",out,cyc);
  run<12>("s_add_u32 (per CU: 4 SIMDs share)",out,cyc);
  run<13>("v_cndmask_b32_e32 (vcc)",out,cyc);
  run<32>("v_cndmask_b32_e64 (vcc)",out,cyc);
  run<33>("v_cndmask_b32_e32 (vcc), dst = src1",out,cyc);
  run<34>("v_cndmask_e32 (vcc) + v_add_u32: per pair",out,cyc);
  run<35>("v_or_b32",out,cyc);
  run<36>("v_ashrrev_i32",out,cyc);
  run<37>("v_not_b32",out,cyc);
  run<38>("v_add_co_u32 -> vcc",out,cyc);
  run<14>("v_mov_b32",out,cyc);
  run<15>("v_lshlrev_b32",out,cyc);
  run<16>("v_sub_u32",out,cyc);
  run<17>("v_min_i32",out,cyc);
  run<31>("v_max_i32",out,cyc);
  run<18>("v_and_b32",out,cyc);
  run<19>("v_lshl_add_u32",out,cyc);
  run<20>("v_add3_u32",out,cyc);
  run<21>("v_ffbl_b32",out,cyc);
  run<22>("v_cmp_lt_i32_e64 -> sgpr pair",out,cyc);
  run<23>("v_add_u32_sdwa",out,cyc);
  run<24>("v_readlane_b32",out,cyc);
  run<25>("v_add_u32 with a 32-bit literal",out,cyc);
  run<26>("v_add_u32 with an sgpr",out,cyc);
  run<27>("v_mbcnt_lo_u32_b32",out,cyc);
  run<28>("v_bfe_u32",out,cyc);
  run<29>("v_xor_b32_e64",out,cyc);
  run<30>("v_addc_co_u32",out,cyc);
  return 0;
}
