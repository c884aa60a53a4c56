// host_ring_store.hip -- what limits an asynchronous session's served steps per second?  Per served step the step
// server does ONE agent-scope read-modify-write on the ticket word, ONE on the completion count and ONE 8-byte store into a ring in
// pinned, coherent host memory.  This measures each of them alone and together, issued by 1024 one-wave workgroups (lane 0) as fast
// as they can, with a configurable amount of dependent ALU work in between (0 = back to back):
//   rmw1       fetch_add on one device word
//   rmw2       fetch_add on two device words on different cache lines
//   host       8-byte relaxed system-scope store to consecutive slots of a pinned ring (slot = a per-workgroup counter)
//   rmw+host   fetch_add on a device word, then the store to the slot it names (what the server does)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/host_ring_store.hip -o build/host_ring_store && build/host_ring_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(64) k(unsigned* ctl, unsigned long long* ring, unsigned mask, int mode, int iters) {
  if (threadIdx.x != 0) return;
  unsigned own = blockIdx.x * 4096u;
  for (int i = 0; i < iters; ++i) {
    unsigned c = own++;
    if (mode == 0 || mode == 1 || mode == 3) c = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mode == 1) c += __hip_atomic_fetch_add(ctl + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mode == 2 || mode == 3) __hip_atomic_store(ring + (c & mask), ((unsigned long long)(c + 1u) << 32) | blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (mode == 0 || mode == 1) { if (c == 0xFFFFFFFFu) ring[0] = c; }      // (keep the result alive)
  }
}

int main() {
  const int WG = 1024, IT = 4000;
  unsigned* ctl; unsigned long long* ring; const unsigned cap = 1u << 16;
  CHECK(hipMalloc(&ctl, 1024)); CHECK(hipMemset(ctl, 0, 1024));
  CHECK(hipHostMalloc((void**)&ring, sizeof(unsigned long long) * cap, hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable));
  const char* names[4] = {"rmw1", "rmw2", "host", "rmw+host"};
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(k, dim3(WG), dim3(64), 0, 0, ctl, ring, cap - 1u, mode, 100);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(WG), dim3(64), 0, 0, ctl, ring, cap - 1u, mode, IT);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-9s %d workgroups x %d: %.3f ms -> %.1f M operations/s\n", names[mode], WG, IT, ms, (double)WG * IT / ms / 1e3);
  }
  return 0;
}
