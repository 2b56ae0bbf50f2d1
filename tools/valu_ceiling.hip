// valu_ceiling.hip -- gfx950 microbenchmark: what does one wave64 vector instruction cost a SIMD?
//
// Settles DESIGN.md section 5's question (VERDICT r01 "weak" 3): `/opt/skills/guides/MI355X_MICROARCH.md` says a wave64 VALU
// instruction takes 2 cycles of a SIMD once more than one wave is resident (4 for a wave alone); round 1 assumed 4 throughout.
// For each instruction class the kernel runs a long stream of independent instructions (16 accumulators, no dependences
// inside an unrolled group of 64) at 1, 2, 4, 5 and 8 waves per SIMD and reports
//   * cycles per wave-instruction per SIMD from s_memtime inside the kernel (per-wave delta / instructions / waves-per-SIMD),
//   * wave-instructions per second for the whole chip from the launch's wall time (HIP events).
// It also pins the numerics the pixel-order sums would need from the matrix pipe: that v_mfma_f32_16x16x4_f32 and
// v_mfma_f32_4x4x1_16b_f32 with B == 1.0 give bit for bit the serial sum ((acc + a0) + a1) + ..., in k order, denormals included.
//
// build: hipcc --offload-arch=gfx950 -O2 tools/valu_ceiling.hip -o gpurun_out/valu_ceiling   (run on the GPU box; prints JSON lines)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

enum Op : int { OP_ADD_U32, OP_FMA_F32, OP_ADD_F32, OP_MUL_F32, OP_MAD_I24, OP_PK_ADD_U16, OP_PK_MUL_U16, OP_PK_FMA_F32, OP_PK_ADD_F32, OP_DOT2_U16, OP_PERM, OP_MED3, OP_ADD3,
                OP_MUL_LO_U32, OP_CVT_UBYTE, OP_ADD_DPP, OP_MOV_DPP, OP_MIN_F32, OP_LSHL_OR, OP_BFE, OP_CNDMASK, OP_RCP_F32, OP_PERMLANE32_SWAP, OP_CNDMASK_SGPR, OP_ADD_F32_DEP, OP_MAD_I24_DEP, OP_PK_ADD_DEP, OP_MIX_TRIAL, OP_SUB_F32, OP_MAX_F32, OP_FMAC_F32, OP_AND, OP_OR, OP_XOR, OP_LSHL, OP_LSHR, OP_ASHR, OP_SUB_U32, OP_MIN_U32, OP_MUL_U24, OP_MUL_I24, OP_MOV, OP_CVT_F32_I32, OP_CVT_I32_F32, OP_RNDNE, OP_PK_MUL_F32, OP_CMP_CNDMASK, OP_ADD_U32_SGPR, OP_MAD_I24_SGPR, OP_RSQ_F32, OP_READLANE, OP_MFMA_16x16x4, OP_MFMA_4x4x1, OP_COUNT };
static const char *kOpName[OP_COUNT] = { "v_add_u32", "v_fma_f32", "v_add_f32", "v_mul_f32", "v_mad_i32_i24", "v_pk_add_u16", "v_pk_mul_lo_u16", "v_pk_fma_f32", "v_pk_add_f32", "v_dot2_u32_u16",
                                         "v_perm_b32", "v_med3_i32", "v_add3_u32", "v_mul_lo_u32", "v_cvt_f32_ubyte0", "v_add_f32_dpp(row_shr:1)", "v_mov_b32_dpp(quad_perm)", "v_min_f32", "v_lshl_or_b32",
                                         "v_bfe_u32", "v_cndmask_b32", "v_rcp_f32", "v_permlane32_swap", "v_cndmask_b32(sgpr mask)", "v_add_f32 dependent chain", "v_mad_i32_i24 dependent chain", "v_pk_add_u16 dependent chain", "trial-like mix (mad24,perm,pk_add,add_u32)", "v_sub_f32", "v_max_f32", "v_fmac_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_sub_u32", "v_min_u32", "v_mul_u32_u24", "v_mul_i32_i24", "v_mov_b32", "v_cvt_f32_i32", "v_cvt_i32_f32", "v_rndne_f32", "v_pk_mul_f32", "v_cmp_lt_u32+v_cndmask_b32(vcc) pair (per instruction)", "v_add_u32(sgpr operand)", "v_mad_i32_i24(sgpr operand)", "v_rsq_f32", "v_readlane_b32+v_writelane_b32 pair (per instruction)", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_4x4x1_16b_f32" };

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k_stream(uint32_t *out, unsigned long long *cycles, int iters, uint32_t seed)
{
  uint32_t a[16];
  float2_t p[16];
  float4_t m[4];
#pragma unroll
  for (int i = 0; i < 16; i++) { a[i] = seed * (i + 1) + threadIdx.x; p[i] = float2_t{ (float)i, 1.0f }; }
#pragma unroll
  for (int i = 0; i < 4; i++) m[i] = float4_t{ 0.f, 0.f, 0.f, 0.f };
  uint32_t b = seed | 1u, c = threadIdx.x + 3u;
  const unsigned long long mask = 0x5555AAAA3333CCCCull ^ seed;
  float fb = 1.0000001f, fc = 1e-9f;
  float2_t pb = { 1.0000001f, 0.9999999f }, pc = { 1e-9f, 1e-9f };
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++)
  {
#define X(i)                                                                                                                     \
    if (OP == OP_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                            \
    else if (OP == OP_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(fb), "v"(fc));                          \
    else if (OP == OP_ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fc));                                       \
    else if (OP == OP_MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fb));                                       \
    else if (OP == OP_MAD_I24) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                        \
    else if (OP == OP_PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
    else if (OP == OP_PK_MUL_U16) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
    else if (OP == OP_PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));                    \
    else if (OP == OP_PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));                                 \
    else if (OP == OP_DOT2_U16) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));                      \
    else if (OP == OP_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                              \
    else if (OP == OP_MED3) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                              \
    else if (OP == OP_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                              \
    else if (OP == OP_MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                  \
    else if (OP == OP_CVT_UBYTE) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]));                                            \
    else if (OP == OP_ADD_DPP) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));        \
    else if (OP == OP_MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));  \
    else if (OP == OP_MIN_F32) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fb));                                       \
    else if (OP == OP_LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));                                 \
    else if (OP == OP_BFE) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(a[i]));                                                  \
    else if (OP == OP_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                               \
    else if (OP == OP_RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));                                                     \
    else if (OP == OP_PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[(i + 1) & 15]));              \
    else if (OP == OP_CNDMASK_SGPR) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(mask));                 \
    else if (OP == OP_ADD_F32_DEP) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(fc));                                  \
    else if (OP == OP_MAD_I24_DEP) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));                   \
    else if (OP == OP_PK_ADD_DEP) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[0]) : "v"(b));                                 \
    else if (OP == OP_MIX_TRIAL) { if ((i & 3) == 0) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); else if ((i & 3) == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); else if ((i & 3) == 2) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b)); else asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); } \
    else if (OP == OP_SUB_F32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fc)); \
    else if (OP == OP_MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(fc)); \
    else if (OP == OP_FMAC_F32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(fb), "v"(fc)); \
    else if (OP == OP_AND) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if (OP == OP_OR) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if (OP == OP_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if (OP == OP_LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i])); \
    else if (OP == OP_LSHR) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i])); \
    else if (OP == OP_ASHR) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a[i])); \
    else if (OP == OP_SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if (OP == OP_MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if (OP == OP_MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if (OP == OP_MUL_I24) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b)); \
    else if (OP == OP_CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i])); \
    else if (OP == OP_CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i])); \
    else if (OP == OP_RNDNE) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i])); \
    else if (OP == OP_PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb)); \
    else if (OP == OP_CMP_CNDMASK) { if (i & 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"); else asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc"); } \
    else if (OP == OP_ADD_U32_SGPR) asm volatile("v_add_u32 %0, %1, %0" : "+v"(a[i]) : "s"(seed)); \
    else if (OP == OP_MAD_I24_SGPR) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "s"(seed), "v"(c)); \
    else if (OP == OP_RSQ_F32) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i])); \
    else if (OP == OP_READLANE) { uint32_t t_; asm volatile("v_readlane_b32 %0, %1, 3\n\tv_writelane_b32 %1, %0, 5" : "=&s"(t_), "+v"(a[i])); } \
    else if (OP == OP_MFMA_16x16x4) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[i & 3]) : "v"(fb), "v"(fc));    \
    else if (OP == OP_MFMA_4x4x1) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(m[i & 3]) : "v"(fb), "v"(fc));
    REP16(X) REP16(X) REP16(X) REP16(X)
#undef X
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) r ^= a[i] ^ __float_as_uint(p[i].x) ^ __float_as_uint(p[i].y);
#pragma unroll
  for (int i = 0; i < 4; i++) r ^= __float_as_uint(m[i].x) ^ __float_as_uint(m[i].y) ^ __float_as_uint(m[i].z) ^ __float_as_uint(m[i].w);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0)
  {
    const size_t k = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    uint32_t hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    cycles[k] = t1 - t0; cycles[k + 1] = r1 - r0; cycles[k + 2] = r0; cycles[k + 3] = ((unsigned long long)(xcc & 15u) << 32) | hwid;
  }
}

typedef void (*kern_t)(uint32_t *, unsigned long long *, int, uint32_t);
template <int OP> static kern_t get() { return k_stream<OP>; }
template <int... I> static void fill(kern_t *t, std::integer_sequence<int, I...>) { ((t[I] = get<I>()), ...); }

// ---- MFMA numerics: D = A x ones + C as an ordered sum ----------------------------------------------------------------------
// 16x16x4: lane l holds A[i = l % 16][k = l / 16]; with B == 1.0, D[i][j] = fma-chain over k of A[i][k] added to C[i][j].
// steps dependent MFMAs walk 4 * steps terms per row i.  in: [steps][64] floats; out: [64][4] floats (each lane's 4 D values).
__global__ void k_mfma16_chain(const float *in, float *out, int steps)
{
  float4_t acc = { 0.f, 0.f, 0.f, 0.f };
  const int lane = threadIdx.x;
  for (int t = 0; t < steps; t++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(in[t * 64 + lane], 1.0f, acc, 0, 0, 0);
  out[lane * 4 + 0] = acc.x; out[lane * 4 + 1] = acc.y; out[lane * 4 + 2] = acc.z; out[lane * 4 + 3] = acc.w;
}
// 4x4x1_16b: lane l holds A[block l / 4][i = l % 4]; D[b][i][j] += A[b][i] * B[b][j]
__global__ void k_mfma4_chain(const float *in, float *out, int steps)
{
  float4_t acc = { 0.f, 0.f, 0.f, 0.f };
  const int lane = threadIdx.x;
  for (int t = 0; t < steps; t++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(in[t * 64 + lane], 1.0f, acc, 0, 0, 0);
  out[lane * 4 + 0] = acc.x; out[lane * 4 + 1] = acc.y; out[lane * 4 + 2] = acc.z; out[lane * 4 + 3] = acc.w;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { uint64_t x = (rng_state += 0x9E3779B97F4A7C15ull); x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

static float gen_value(int kind)
{
  union { uint32_t u; float f; } v;
  switch (kind)
  {
  case 0: v.f = ((int)(rnd() % 2001) - 1000) / 1000.0f * 0.9999f; break;                         // unit-vector-like components
  case 1: v.u = (uint32_t)(rnd() & 0x807FFFFFu); break;                                           // denormals, both signs
  case 2: { const int e = 100 + (int)(rnd() % 56); v.u = ((uint32_t)(rnd() & 1) << 31) | ((uint32_t)e << 23) | (uint32_t)(rnd() & 0x7FFFFF); } break; // wide dynamic range
  case 3: v.f = (rnd() & 1) ? 0.0f : -0.0f; break;
  default: v.f = (float)((int)(rnd() % 511) - 255) * 0.57727051f; break;                          // d * rsqrt-like
  }
  return v.f;
}

static int check_mfma(bool wide, int steps, int kind, int *orderOut)
{
  std::vector<float> in((size_t)steps * 64), out(256);
  for (auto &x : in) x = gen_value(kind);
  float *din, *dout;
  CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, 1024));
  CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
  if (wide) hipLaunchKernelGGL(k_mfma16_chain, dim3(1), dim3(64), 0, 0, din, dout, steps);
  else hipLaunchKernelGGL(k_mfma4_chain, dim3(1), dim3(64), 0, 0, din, dout, steps);
  CK(hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost));
  CK(hipFree(din)); CK(hipFree(dout));
  int bad_fwd = 0, bad_rev = 0, bad_layout = 0;
  if (wide)
  { // D[i][j]: lane = j + 16 * (i / 4), register = i % 4   (16x16 f32 accumulator layout)
    for (int i = 0; i < 16; i++)
    {
      volatile float sf = 0.0f, sr = 0.0f;
      for (int t = 0; t < steps; t++)
      {
        for (int k = 0; k < 4; k++) sf = sf + in[(size_t)t * 64 + 16 * k + i];
        for (int k = 3; k >= 0; k--) sr = sr + in[(size_t)t * 64 + 16 * k + i];
      }
      for (int j = 0; j < 16; j++)
      {
        const float got = out[(j + 16 * (i / 4)) * 4 + (i % 4)];
        float f = sf, r = sr;
        if (memcmp(&got, &f, 4) != 0) bad_fwd++;
        if (memcmp(&got, &r, 4) != 0) bad_rev++;
      }
    }
  }
  else
  { // D[b][i][j]: lane = 4 b + j, register = i
    for (int b = 0; b < 16; b++)
      for (int i = 0; i < 4; i++)
      {
        volatile float sf = 0.0f;
        for (int t = 0; t < steps; t++) sf = sf + in[(size_t)t * 64 + 4 * b + i];
        for (int j = 0; j < 4; j++)
        {
          const float got = out[(4 * b + j) * 4 + i];
          float f = sf;
          if (memcmp(&got, &f, 4) != 0) bad_fwd++;
        }
      }
    bad_rev = bad_fwd;
  }
  (void)bad_layout;
  *orderOut = bad_fwd == 0 ? 1 : (bad_rev == 0 ? -1 : 0);
  return bad_fwd;
}

int main(int argc, char **argv)
{
  int iters = argc > 1 ? atoi(argv[1]) : 5000;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, cus, prop.clockRate);

  // ---- numerics of the f32 matrix instructions as ordered sums ----
  for (int wide = 1; wide >= 0; wide--)
    for (int kind = 0; kind < 5; kind++)
    {
      int order = 0, bad = 0;
      for (int rep = 0; rep < 20; rep++) { int o; bad += check_mfma(wide != 0, 16 * (wide ? 1 : 4), kind, &o); if (rep == 0 || o != order) order = (rep == 0) ? o : 0; }
      printf("{\"mfma_sum_check\": \"%s\", \"data\": %d, \"mismatches_vs_serial_k0123\": %d, \"order\": \"%s\"}\n", wide ? "v_mfma_f32_16x16x4_f32" : "v_mfma_f32_4x4x1_16b_f32", kind, bad,
             order == 1 ? "k ascending == serial add" : (order == -1 ? "k descending" : "neither"));
    }

  kern_t tab[OP_COUNT];
  fill(tab, std::make_integer_sequence<int, OP_COUNT>{});
  uint32_t *out; unsigned long long *cyc;
  const int maxBlocks = cus * 8;
  CK(hipMalloc(&out, (size_t)maxBlocks * 256 * 4)); CK(hipMalloc(&cyc, (size_t)maxBlocks * 4 * 32));
  std::vector<unsigned long long> hc((size_t)maxBlocks * 16);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int wavesPerSimd[] = { 1, 2, 4, 5, 8 };
  for (int op = 0; op < OP_COUNT; op++)
    for (int w : wavesPerSimd)
    {
      const int blocks = cus * w; // 256 threads = 4 waves = one per SIMD; w workgroups per CU
      const int it = (op == OP_MFMA_16x16x4) ? iters / 8 : ((op == OP_MFMA_4x4x1 || op == OP_CNDMASK) ? iters / 4 : iters);
      hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, out, cyc, it / 4, 12345u); // warm-up
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, out, cyc, it, 12345u);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(hc.data(), cyc, (size_t)blocks * 4 * 32, hipMemcpyDeviceToHost));
      std::vector<double> tick((size_t)blocks * 4), real((size_t)blocks * 4);
      for (size_t k = 0; k < (size_t)blocks * 4; k++) { tick[k] = (double)hc[4 * k]; real[k] = (double)hc[4 * k + 1]; }
      // where did the waves run?  key = (xcc, se, sh, cu, simd); gfx9 HW_ID: simd 5:4, cu 11:8, sh 12, se 15:13
      std::vector<uint32_t> keys;
      unsigned long long firstStart = ~0ull, lastStart = 0;
      for (size_t k = 0; k < (size_t)blocks * 4; k++)
      {
        const unsigned long long v = hc[4 * k + 3];
        const uint32_t hw = (uint32_t)v, xcc = (uint32_t)(v >> 32);
        keys.push_back((xcc << 16) | (((hw >> 13) & 7u) << 12) | (((hw >> 12) & 1u) << 8) | (((hw >> 8) & 15u) << 4) | ((hw >> 4) & 3u));
        firstStart = std::min(firstStart, hc[4 * k + 2]); lastStart = std::max(lastStart, hc[4 * k + 2]);
      }
      std::sort(keys.begin(), keys.end());
      int simds = 0, maxPer = 0, minPer = 1 << 30;
      for (size_t a0 = 0; a0 < keys.size();) { size_t b0 = a0; while (b0 < keys.size() && keys[b0] == keys[a0]) b0++; simds++; maxPer = std::max(maxPer, (int)(b0 - a0)); minPer = std::min(minPer, (int)(b0 - a0)); a0 = b0; }
      std::sort(tick.begin(), tick.end()); std::sort(real.begin(), real.end());
      const double medTick = tick[tick.size() / 2], medReal = real[real.size() / 2];
      const double instr = (double)it * 64.0;
      const double tickMHz = medTick / medReal * 100.0;                 // s_memtime ticks per microsecond (s_memrealtime runs at 100 MHz)
      const double rate = instr * blocks * 4.0 / (ms * 1e-3);           // wave-instructions per second, whole chip, launch included
      const double cpiWall = (double)cus * 4.0 * 2.4e9 / rate;          // cycles of a 2.4 GHz SIMD per wave-instruction, from the wall time
      const double cpiWave = medReal * 1e-8 * 2.4e9 / instr / w;        // the same from the median wave's own 100 MHz timer (no launch overhead)
      printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_instr_per_simd_wall\": %.3f, \"cycles_per_instr_per_simd_wave_timer\": %.3f, \"chip_wave_instr_per_s\": %.4g, \"ms\": %.4f, \"s_memtime_MHz\": %.1f, \"simds_used\": %d, \"waves_per_simd_min\": %d, \"waves_per_simd_max\": %d, \"wave_ms_min\": %.4f, \"wave_ms_max\": %.4f, \"start_spread_ms\": %.4f}\n",
             kOpName[op], w, cpiWall, cpiWave, rate, ms, tickMHz, simds, minPer, maxPer, real.front() * 1e-5, real.back() * 1e-5, (double)(lastStart - firstStart) * 1e-5);
      fflush(stdout);
    }
  return 0;
}
