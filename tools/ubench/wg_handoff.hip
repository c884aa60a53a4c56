// wg_handoff.hip -- what does it cost one workgroup to hand a piece of work to ANOTHER workgroup and get the answer back?
//
// VERDICT r05 #4 (second half): "try the three evaluation passes (8.9 k of 29 k cycles per lone iteration) on 4 waves only when the
// work counter is exhausted (tail mode)".  The step kernel is one wavefront per environment (64-thread workgroups, seven per CU: the
// register file and the LDS are handed out per environment); a launch ends with a handful of long chains running alone while
// thousands of wave slots idle.  A second, third and fourth wave for those chains can only come from OTHER workgroups -- helpers that
// have run out of environments -- and whatever they compute has to travel through global memory: the bus vectors out (V: 2 x 118
// doubles), the partial mismatch / Jacobian rows back, each way behind an agent-scope release / acquire pair (the helpers sit on
// other CUs, usually other XCDs with their own L2).  This measures exactly that round trip: workgroup A writes a 2 KB payload,
// releases a flag; workgroup B (spinning on it) acquires, reads the payload, writes 2 KB back, releases its flag; A acquires and
// reads.  Printed: shader-clock cycles and wall-clock ns per round trip, for B on the same CU-neighbourhood (adjacent block index)
// and far away (block index + 128: another XCD under the round-robin block placement).
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/wg_handoff.hip -o build/wg_handoff && build/wg_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int PAYLOAD = 256;      // doubles per direction (2 KB)
constexpr int ROUNDS = 2000;

__global__ void __launch_bounds__(64) handoff(double* buf, unsigned* flags, long long* out, int partner_of_0) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b != 0 && b != partner_of_0) return;
  double* ab = buf;                 // A -> B
  double* ba = buf + PAYLOAD;       // B -> A
  unsigned* fa = flags;             // A's "payload ready" counter
  unsigned* fb = flags + 32;        // B's "answer ready" counter (another cache line)
  double acc = 0.0;
  if (b == 0) {
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int r = 1; r <= ROUNDS; ++r) {
      for (int k = lane; k < PAYLOAD; k += 64) ab[k] = (double)(r + k) + acc * 1e-30;
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (lane == 0) __hip_atomic_store(fa, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (;;) {
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_load(fb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= (unsigned)r) break;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int k = lane; k < PAYLOAD; k += 64) acc += ba[k];
    }
    if (lane == 0) { out[0] = clock64() - c0; out[1] = wall_clock64() - w0; out[2] = (long long)acc; }
  } else {
    for (int r = 1; r <= ROUNDS; ++r) {
      for (;;) {
        unsigned v = 0;
        if (lane == 0) v = __hip_atomic_load(fa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= (unsigned)r) break;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      for (int k = lane; k < PAYLOAD; k += 64) ba[k] = ab[k] * 2.0;
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (lane == 0) __hip_atomic_store(fb, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main() {
  double* buf; unsigned* flags; long long* out;
  CHECK(hipMalloc(&buf, 2 * PAYLOAD * sizeof(double))); CHECK(hipMalloc(&flags, 64 * sizeof(unsigned))); CHECK(hipMalloc(&out, 4 * sizeof(long long)));
  for (int partner : {1, 8, 128, 200}) {
    CHECK(hipMemset(flags, 0, 64 * sizeof(unsigned))); CHECK(hipMemset(buf, 0, 2 * PAYLOAD * sizeof(double)));
    hipLaunchKernelGGL(handoff, dim3(256), dim3(64), 0, 0, buf, flags, out, partner);
    CHECK(hipDeviceSynchronize());
    long long h[3];
    CHECK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
    printf("partner block %3d: %7.0f shader cycles, %6.0f ns per round trip (2 KB out, 2 KB back, release/acquire at agent scope each way)\n",
           partner, (double)h[0] / ROUNDS, (double)h[1] * 10.0 / ROUNDS);
  }
  return 0;
}
