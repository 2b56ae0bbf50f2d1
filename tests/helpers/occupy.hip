// TEST HELPER (not part of the product): a "foreign" kernel that holds whole CUs for a given time, for tests/test_gpu_concurrency.py.
// Each workgroup declares 144 KiB of LDS, so it owns its CU's LDS (no workgroup of limg's persistent kernel, 24.3 KiB, fits beside it) and at most one lands
// per CU; every wave leaves as soon as the 100 MHz wall clock says the time is up -- an exit condition every wave reaches, whatever else runs.
// Built by __graft_entry__.build() into tests/helpers/liboccupy.so (hipcc, gfx950).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace
{
  __global__ __launch_bounds__(64) void k_occupy(unsigned long long ticks, uint32_t *sink)
  {
    __shared__ uint32_t lds[144 * 1024 / 4];
    lds[threadIdx.x] = threadIdx.x; // (keeps the allocation)
    const unsigned long long t0 = wall_clock64();
    uint32_t spins = 0;
    while (wall_clock64() - t0 < ticks && spins < (1u << 28)) { __builtin_amdgcn_s_sleep(32); spins++; }
    if (sink && lds[threadIdx.x] == 0xFFFFFFFFu) *sink = spins;
  }
}

extern "C" int occupy_launch(int workgroups, int microseconds, void *stream)
{
  if (workgroups <= 0 || microseconds < 0 || microseconds > 200000) return 1;
  hipLaunchKernelGGL(k_occupy, dim3((uint32_t)workgroups), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * 100ull, (uint32_t *)nullptr);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
