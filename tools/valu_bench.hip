// Micro-benchmark: sustained VALU issue rate on gfx950 per instruction class (wave-instructions per second and clocks per
// wave-instruction per SIMD at 2.4 GHz), used to price the VALU-bound kernels.
// Build: hipcc --offload-arch=gfx950 -O3 valu_bench.hip -o valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(x) x x x x x x x x
#define BODY(asmtext)                                                                                    \
  unsigned a = threadIdx.x, b = blockIdx.x + 1, c = 3, d = 5, e = 7, f = 11, g = 13, h = 17;               \
  for (int i = 0; i < iters; i++) {                                                                      \
    REP8(REP8(asm volatile(asmtext : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));)) \
  }                                                                                                      \
  out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;

// each asm statement = 8 independent instructions (one per register chain)
#define OP8(op) op " %0, %0, %1\n" op " %1, %1, %2\n" op " %2, %2, %3\n" op " %3, %3, %4\n" op " %4, %4, %5\n" op " %5, %5, %6\n" op " %6, %6, %7\n" op " %7, %7, %0\n"
#define OP8_3(op) op " %0, %0, %1, %2\n" op " %1, %1, %2, %3\n" op " %2, %2, %3, %4\n" op " %3, %3, %4, %5\n" op " %4, %4, %5, %6\n" op " %5, %5, %6, %7\n" op " %6, %6, %7, %0\n" op " %7, %7, %0, %1\n"

__global__ __launch_bounds__(256) void k_sub(unsigned* out, int iters) { BODY(OP8("v_sub_u32")) }
__global__ __launch_bounds__(256) void k_min(unsigned* out, int iters) { BODY(OP8("v_min_u32")) }
__global__ __launch_bounds__(256) void k_pkmin(unsigned* out, int iters) { BODY(OP8("v_pk_min_u16")) }
__global__ __launch_bounds__(256) void k_pksub(unsigned* out, int iters) { BODY(OP8("v_pk_sub_i16")) }
__global__ __launch_bounds__(256) void k_perm(unsigned* out, int iters) { BODY(OP8_3("v_perm_b32")) }
__global__ __launch_bounds__(256) void k_min3(unsigned* out, int iters) { BODY(OP8_3("v_min3_u32")) }
__global__ __launch_bounds__(256) void k_bitop(unsigned* out, int iters) { BODY(OP8_3("v_and_or_b32")) }
__global__ __launch_bounds__(256) void k_mul(unsigned* out, int iters) { BODY(OP8("v_mul_lo_u32")) }
__global__ __launch_bounds__(256) void k_mul24(unsigned* out, int iters) { BODY(OP8("v_mul_u32_u24")) }
__global__ __launch_bounds__(256) void k_sad(unsigned* out, int iters) { BODY(OP8_3("v_sad_u8")) }
__global__ __launch_bounds__(256) void k_dot4(unsigned* out, int iters) { BODY(OP8_3("v_dot4_u32_u8")) }
__global__ __launch_bounds__(256) void k_add(unsigned* out, int iters) { BODY(OP8("v_add_u32")) }
__global__ __launch_bounds__(256) void k_or(unsigned* out, int iters) { BODY(OP8("v_or_b32")) }
__global__ __launch_bounds__(256) void k_and(unsigned* out, int iters) { BODY(OP8("v_and_b32")) }
__global__ __launch_bounds__(256) void k_xor(unsigned* out, int iters) { BODY(OP8("v_xor_b32")) }
__global__ __launch_bounds__(256) void k_shl(unsigned* out, int iters) { BODY(OP8("v_lshlrev_b32")) }
__global__ __launch_bounds__(256) void k_maxi(unsigned* out, int iters) { BODY(OP8("v_max_i32")) }
__global__ __launch_bounds__(256) void k_add3(unsigned* out, int iters) { BODY(OP8_3("v_add3_u32")) }
__global__ __launch_bounds__(256) void k_lshladd(unsigned* out, int iters) { BODY(OP8_3("v_lshl_add_u32")) }
__global__ __launch_bounds__(256) void k_bitop3(unsigned* out, int iters) { BODY(OP8_3("v_or3_b32")) }
__global__ __launch_bounds__(256) void k_align(unsigned* out, int iters) { BODY(OP8_3("v_alignbyte_b32")) }
__global__ __launch_bounds__(256) void k_bfe(unsigned* out, int iters) { BODY(OP8_3("v_bfe_u32")) }
__global__ __launch_bounds__(256) void k_pkadd(unsigned* out, int iters) { BODY(OP8("v_pk_add_u16")) }
__global__ __launch_bounds__(256) void k_addf(unsigned* out, int iters) { BODY(OP8("v_add_f32")) }
__global__ __launch_bounds__(256) void k_mulf(unsigned* out, int iters) { BODY(OP8("v_mul_f32")) }
__global__ __launch_bounds__(256) void k_fma(unsigned* out, int iters) { BODY(OP8_3("v_fma_f32")) }
__global__ __launch_bounds__(256) void k_fmac(unsigned* out, int iters) { BODY(OP8("v_fmac_f32")) }
__global__ __launch_bounds__(256) void k_mov(unsigned* out, int iters) { BODY("v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\nv_mov_b32 %4, %5\nv_mov_b32 %5, %6\nv_mov_b32 %6, %7\nv_mov_b32 %7, %0\n") }
__global__ __launch_bounds__(256) void k_bcnt(unsigned* out, int iters) { BODY(OP8("v_bcnt_u32_b32")) }
__global__ __launch_bounds__(256) void k_cmp(unsigned* out, int iters) {
  BODY("v_cmp_lt_u32 vcc, %0, %1\nv_cndmask_b32 %0, %0, %1, vcc\nv_cmp_lt_u32 vcc, %2, %3\nv_cndmask_b32 %2, %2, %3, vcc\n"
       "v_cmp_lt_u32 vcc, %4, %5\nv_cndmask_b32 %4, %4, %5, vcc\nv_cmp_lt_u32 vcc, %6, %7\nv_cndmask_b32 %6, %6, %7, vcc\n")
}
__global__ __launch_bounds__(256) void k_minf(unsigned* out, int iters) { BODY(OP8("v_min_f32")) }
__global__ __launch_bounds__(256) void k_maxf(unsigned* out, int iters) { BODY(OP8("v_max_f32")) }
__global__ __launch_bounds__(256) void k_min3f(unsigned* out, int iters) { BODY(OP8_3("v_min3_f32")) }
__global__ __launch_bounds__(256) void k_med3f(unsigned* out, int iters) { BODY(OP8_3("v_med3_f32")) }
__global__ __launch_bounds__(256) void k_subf(unsigned* out, int iters) { BODY(OP8("v_sub_f32")) }
__global__ __launch_bounds__(256) void k_cvtub(unsigned* out, int iters) { BODY("v_cvt_f32_ubyte0 %0, %1\nv_cvt_f32_ubyte1 %1, %2\nv_cvt_f32_ubyte2 %2, %3\nv_cvt_f32_ubyte3 %3, %4\nv_cvt_f32_ubyte0 %4, %5\nv_cvt_f32_ubyte1 %5, %6\nv_cvt_f32_ubyte2 %6, %7\nv_cvt_f32_ubyte3 %7, %0\n") }
__global__ __launch_bounds__(256) void k_cvtfu(unsigned* out, int iters) { BODY("v_cvt_f32_u32 %0, %1\nv_cvt_f32_u32 %1, %2\nv_cvt_f32_u32 %2, %3\nv_cvt_f32_u32 %3, %4\nv_cvt_f32_u32 %4, %5\nv_cvt_f32_u32 %5, %6\nv_cvt_f32_u32 %6, %7\nv_cvt_f32_u32 %7, %0\n") }
__global__ __launch_bounds__(256) void k_cvtuf(unsigned* out, int iters) { BODY("v_cvt_u32_f32 %0, %1\nv_cvt_u32_f32 %1, %2\nv_cvt_u32_f32 %2, %3\nv_cvt_u32_f32 %3, %4\nv_cvt_u32_f32 %4, %5\nv_cvt_u32_f32 %5, %6\nv_cvt_u32_f32 %6, %7\nv_cvt_u32_f32 %7, %0\n") }
__global__ __launch_bounds__(256) void k_minu16(unsigned* out, int iters) { BODY(OP8("v_min_u16")) }
__global__ __launch_bounds__(256) void k_subu16(unsigned* out, int iters) { BODY(OP8("v_sub_u16")) }
__global__ __launch_bounds__(256) void k_ashr(unsigned* out, int iters) { BODY(OP8("v_ashrrev_i32")) }
__global__ __launch_bounds__(256) void k_cmpf(unsigned* out, int iters) {
  BODY("v_cmp_lt_f32 vcc, %0, %1\nv_cndmask_b32 %0, %0, %1, vcc\nv_cmp_lt_f32 vcc, %2, %3\nv_cndmask_b32 %2, %2, %3, vcc\n"
       "v_cmp_lt_f32 vcc, %4, %5\nv_cndmask_b32 %4, %4, %5, vcc\nv_cmp_lt_f32 vcc, %6, %7\nv_cndmask_b32 %6, %6, %7, vcc\n")
}
__global__ __launch_bounds__(256) void k_pkaddf(unsigned* out, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a = {(float)threadIdx.x, 1.f}, b = {2.f, 3.f}, c = {4.f, 5.f}, d = {6.f, 7.f};
  for (int i = 0; i < iters; i++) {
    REP8(REP8(asm volatile("v_pk_add_f32 %0, %0, %1\nv_pk_add_f32 %1, %1, %2\nv_pk_add_f32 %2, %2, %3\nv_pk_add_f32 %3, %3, %0\n"
                           "v_pk_add_f32 %0, %0, %1\nv_pk_add_f32 %1, %1, %2\nv_pk_add_f32 %2, %2, %3\nv_pk_add_f32 %3, %3, %0\n"
                           : "+v"(a), "+v"(b), "+v"(c), "+v"(d));))
  }
  out[blockIdx.x * 256 + threadIdx.x] = (unsigned)(a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y);
}
__global__ __launch_bounds__(256) void k_sdwa(unsigned* out, int iters) {
  BODY("v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
       "v_min_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
       "v_min_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
       "v_min_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
       "v_min_u32_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
       "v_min_u32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
       "v_min_u32_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
       "v_min_u32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n")
}
#define SD(op, sel) op " %0, %0, %1 " sel "\n" op " %1, %1, %2 " sel "\n" op " %2, %2, %3 " sel "\n" op " %3, %3, %4 " sel "\n" op " %4, %4, %5 " sel "\n" op " %5, %5, %6 " sel "\n" op " %6, %6, %7 " sel "\n" op " %7, %7, %0 " sel "\n"
__global__ __launch_bounds__(256) void k_sdwa16(unsigned* out, int iters) {
  BODY(SD("v_min_u16_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2"))
}
__global__ __launch_bounds__(256) void k_sdwa16w(unsigned* out, int iters) {
  BODY(SD("v_max_u16_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_3"))
}
__global__ __launch_bounds__(256) void k_sdwasub(unsigned* out, int iters) {
  BODY(SD("v_sub_u16_sdwa", "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:WORD_0"))
}
__global__ __launch_bounds__(256) void k_cmp16(unsigned* out, int iters) {
  BODY("v_cmp_lt_u16 vcc, %0, %1\nv_cmp_lt_u16 s[10:11], %1, %2\nv_cmp_lt_u16 vcc, %2, %3\nv_cmp_lt_u16 s[10:11], %3, %4\n"
       "v_cmp_lt_u16 vcc, %4, %5\nv_cmp_lt_u16 s[10:11], %5, %6\nv_cmp_lt_u16 vcc, %6, %7\nv_cmp_lt_u16 s[10:11], %7, %0\n")
}
__global__ __launch_bounds__(256) void k_cmp16sdwa(unsigned* out, int iters) {
  BODY("v_cmp_lt_u16_sdwa vcc, %0, %1 src0_sel:BYTE_1 src1_sel:WORD_0\nv_cmp_lt_u16_sdwa vcc, %1, %2 src0_sel:BYTE_1 src1_sel:WORD_0\n"
       "v_cmp_lt_u16_sdwa vcc, %2, %3 src0_sel:BYTE_1 src1_sel:WORD_0\nv_cmp_lt_u16_sdwa vcc, %3, %4 src0_sel:BYTE_1 src1_sel:WORD_0\n"
       "v_cmp_lt_u16_sdwa vcc, %4, %5 src0_sel:BYTE_1 src1_sel:WORD_0\nv_cmp_lt_u16_sdwa vcc, %5, %6 src0_sel:BYTE_1 src1_sel:WORD_0\n"
       "v_cmp_lt_u16_sdwa vcc, %6, %7 src0_sel:BYTE_1 src1_sel:WORD_0\nv_cmp_lt_u16_sdwa vcc, %7, %0 src0_sel:BYTE_1 src1_sel:WORD_0\n")
}
__global__ __launch_bounds__(256) void k_cndmask(unsigned* out, int iters) {
  BODY("v_cndmask_b32 %0, %0, %1, vcc\nv_cndmask_b32 %1, %1, %2, vcc\nv_cndmask_b32 %2, %2, %3, vcc\nv_cndmask_b32 %3, %3, %4, vcc\n"
       "v_cndmask_b32 %4, %4, %5, vcc\nv_cndmask_b32 %5, %5, %6, vcc\nv_cndmask_b32 %6, %6, %7, vcc\nv_cndmask_b32 %7, %7, %0, vcc\n")
}
__global__ __launch_bounds__(256) void k_mbcnt(unsigned* out, int iters) {
  BODY("v_mbcnt_lo_u32_b32 %0, %1, %0\nv_mbcnt_hi_u32_b32 %1, %2, %1\nv_mbcnt_lo_u32_b32 %2, %3, %2\nv_mbcnt_hi_u32_b32 %3, %4, %3\n"
       "v_mbcnt_lo_u32_b32 %4, %5, %4\nv_mbcnt_hi_u32_b32 %5, %6, %5\nv_mbcnt_lo_u32_b32 %6, %7, %6\nv_mbcnt_hi_u32_b32 %7, %0, %7\n")
}
// LDS byte / dword reads: 8 independent loads per statement, addresses from the register chain (kept in range by the mask)
__global__ __launch_bounds__(256) void k_ldsu8(unsigned* out, int iters) {
  __shared__ unsigned char sm[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) sm[i] = (unsigned char)i;
  __syncthreads();
  unsigned base = (unsigned)(size_t)sm + threadIdx.x, acc = 0;
  for (int i = 0; i < iters; i++) {
    unsigned r0, r1, r2, r3, r4, r5, r6, r7;
    REP8(asm volatile("ds_read_u8 %0, %8 offset:0\nds_read_u8 %1, %8 offset:192\nds_read_u8 %2, %8 offset:387\nds_read_u8 %3, %8 offset:579\n"
                      "ds_read_u8 %4, %8 offset:768\nds_read_u8 %5, %8 offset:963\nds_read_u8 %6, %8 offset:1155\nds_read_u8 %7, %8 offset:1344\n"
                      "s_waitcnt lgkmcnt(0)\n"
                      : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
                      : "v"(base));
         acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;)
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_ldsb32(unsigned* out, int iters) {
  __shared__ unsigned sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = i;
  __syncthreads();
  unsigned base = (unsigned)(size_t)sm + threadIdx.x * 4, acc = 0;
  for (int i = 0; i < iters; i++) {
    unsigned r0, r1, r2, r3, r4, r5, r6, r7;
    REP8(asm volatile("ds_read_b32 %0, %8 offset:0\nds_read_b32 %1, %8 offset:192\nds_read_b32 %2, %8 offset:388\nds_read_b32 %3, %8 offset:580\n"
                      "ds_read_b32 %4, %8 offset:768\nds_read_b32 %5, %8 offset:964\nds_read_b32 %6, %8 offset:1156\nds_read_b32 %7, %8 offset:1344\n"
                      "s_waitcnt lgkmcnt(0)\n"
                      : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7)
                      : "v"(base));
         acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;)
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef void (*kern_t)(unsigned*, int);
static void run(const char* name, kern_t k, unsigned* d, double per_stmt) {
  const int blocks = 256 * 8 * 4, iters = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 4);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * iters * 64 * per_stmt;  // 64 statements per iteration
  printf("%-10s %8.3f ms  %.3e wave-instr/s  %.2f clk per wave-instr per SIMD @2.4GHz\n", name, ms, winstr / (ms * 1e-3),
         1024 * 2.4e9 / (winstr / (ms * 1e-3)));
}
int main() {
  unsigned* d;
  hipMalloc(&d, 256 * 8 * 4 * 256 * 4);
  run("v_sub_u32", k_sub, d, 8); run("v_min_u32", k_min, d, 8); run("v_min3_u32", k_min3, d, 8); run("v_and_or", k_bitop, d, 8);
  run("v_add_u32", k_add, d, 8); run("v_or_b32", k_or, d, 8); run("v_and_b32", k_and, d, 8); run("v_xor_b32", k_xor, d, 8);
  run("v_lshlrev", k_shl, d, 8); run("v_max_i32", k_maxi, d, 8); run("v_add3", k_add3, d, 8); run("v_lshl_add", k_lshladd, d, 8);
  run("v_or3", k_bitop3, d, 8); run("alignbyte", k_align, d, 8); run("v_bfe_u32", k_bfe, d, 8); run("pk_add_u16", k_pkadd, d, 8);
  run("v_add_f32", k_addf, d, 8); run("v_mul_f32", k_mulf, d, 8); run("v_fma_f32", k_fma, d, 8); run("v_fmac_f32", k_fmac, d, 8);
  run("v_mov_b32", k_mov, d, 8); run("v_bcnt", k_bcnt, d, 8); run("cmp+cndmask", k_cmp, d, 8);
  run("v_min_f32", k_minf, d, 8); run("v_max_f32", k_maxf, d, 8); run("v_min3_f32", k_min3f, d, 8); run("v_med3_f32", k_med3f, d, 8);
  run("v_sub_f32", k_subf, d, 8); run("cvt_f32_ub", k_cvtub, d, 8); run("cvt_f32_u32", k_cvtfu, d, 8); run("cvt_u32_f32", k_cvtuf, d, 8);
  run("v_min_u16", k_minu16, d, 8); run("v_sub_u16", k_subu16, d, 8); run("v_ashrrev", k_ashr, d, 8); run("cmpf+cndm", k_cmpf, d, 8);
  run("pk_add_f32", k_pkaddf, d, 8);
  run("pk_min_u16", k_pkmin, d, 8); run("pk_sub_i16", k_pksub, d, 8); run("v_perm_b32", k_perm, d, 8);
  run("min_sdwa", k_sdwa, d, 8); run("mul_lo_u32", k_mul, d, 8); run("mul_u24", k_mul24, d, 8); run("v_sad_u8", k_sad, d, 8);
  run("dot4_u8", k_dot4, d, 8);
  run("min_u16_sdwa", k_sdwa16, d, 8); run("max_u16_sdwa_w", k_sdwa16w, d, 8); run("sub_u16_sdwa", k_sdwasub, d, 8);
  run("cmp_lt_u16", k_cmp16, d, 8); run("cmp_u16_sdwa", k_cmp16sdwa, d, 8); run("v_cndmask", k_cndmask, d, 8); run("v_mbcnt", k_mbcnt, d, 8);
  run("ds_read_u8", k_ldsu8, d, 8); run("ds_read_b32", k_ldsb32, d, 8);
  return 0;
}
