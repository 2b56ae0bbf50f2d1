// valu_exec_mask.hip -- gfx950 microbenchmark: does a wave64 vector instruction get cheaper when part of the wave is masked off in EXEC?
// (RDNA's wave64 skips a half whose EXEC bits are all zero; GCN never did.  If gfx950 did, a 64-pixel block could live in 32 lanes as pixel pairs.)
// Same method as valu_ceiling.hip: streams of independent instructions, w waves on every SIMD, wall time -> cycles per wave-instruction per SIMD at 2.4 GHz.
//
// build: hipcc --offload-arch=gfx950 -O2 tools/valu_exec_mask.hip -o gpurun_out/valu_exec_mask   (prints JSON lines)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

enum Op : int { OP_ADD_U32, OP_MAD_I24, OP_PK_ADD_U16, OP_PERM, OP_ADD_DPP, OP_COUNT };
static const char *kOpName[OP_COUNT] = { "v_add_u32", "v_mad_i32_i24", "v_pk_add_u16", "v_perm_b32", "v_add_u32_dpp(row_shr:1)" };

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k_stream(uint32_t *out, int iters, uint32_t seed, unsigned long long mask)
{
  uint32_t a[16];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = seed * (i + 1) + threadIdx.x;
  uint32_t b = seed | 1u, c = threadIdx.x + 3u;
  __syncthreads();
  asm volatile("s_mov_b64 exec, %0" : : "s"(mask));
  for (int it = 0; it < iters; it++)
  {
#define X(i)                                                                                                              \
    if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                     \
    else if (OP == OP_MAD_I24) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                 \
    else if (OP == OP_PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                           \
    else if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                       \
    else if (OP == OP_ADD_DPP) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
    REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
  }
  asm volatile("s_mov_b64 exec, -1");
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) r ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

typedef void (*kern_t)(uint32_t *, int, uint32_t, unsigned long long);

int main(int argc, char **argv)
{
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  kern_t tab[OP_COUNT] = { k_stream<OP_ADD_U32>, k_stream<OP_MAD_I24>, k_stream<OP_PK_ADD_U16>, k_stream<OP_PERM>, k_stream<OP_ADD_DPP> };
  const unsigned long long masks[] = { ~0ull, 0x00000000FFFFFFFFull, 0xFFFFFFFF00000000ull, 0x000000000000FFFFull, 0x0000FFFF0000FFFFull, 0x5555555555555555ull, 1ull };
  const char *maskName[] = { "all 64", "low 32", "high 32", "low 16", "lanes 0-15 + 32-47", "every other lane", "one lane" };
  uint32_t *out;
  CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int op = 0; op < OP_COUNT; op++)
    for (int w : { 2, 5, 8 })
      for (int m = 0; m < 7; m++)
      {
        const int blocks = cus * w;
        hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, out, iters / 4, 12345u, masks[m]);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, out, iters, 12345u, masks[m]);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double rate = (double)iters * 64.0 * blocks * 4.0 / (ms * 1e-3);
        printf("{\"op\": \"%s\", \"exec\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_per_simd\": %.3f, \"chip_wave_instr_per_s\": %.4g}\n", kOpName[op], maskName[m], w,
               (double)cus * 4.0 * 2.4e9 / rate, rate);
        fflush(stdout);
      }
  return 0;
}
