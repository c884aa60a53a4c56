// ppn_kernel_tu.hip -- one share of the kernel instances of libppn.so, compiled on its own.
//
// The library is ONE set of sources (ppn_device.h, ppn_solve.inc, ppn_game.inc, ppn_obs.inc, ppn_kernels.inc); every ppn_kernel<W, KIND, NT>
// is a 20-30 k instruction kernel and there are 63 of them, so a single translation unit takes five minutes to compile.  The split build
// (__graft_entry__.build_hip) compiles the host side (ppn_engine.hip with -DPPN_SPLIT_BUILD: every kernel instance is an `extern
// template` there) and six shares of the instances -- this file with -DPPN_TU_W=1|2|4 -DPPN_TU_NT=0|1 -- in parallel and links the
// objects: same kernels, same symbols, about a minute.  No relocatable device code is needed: kernels never call into another
// translation unit, and a kernel's host-side handle is an ordinary symbol.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DPPN_TU_W=2 -DPPN_TU_NT=1 ppn_kernel_tu.hip -o ppn_kernels_w2n1.o
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <hip/hip_runtime.h>

#define PPN_KERNEL_TU 1
#include "../../include/ppn.h"
#include "ppn_device.h"
#include "ppn_solve.inc"
#include "ppn_game.inc"
#include "ppn_obs.inc"
#include "ppn_kernels.inc"

#if !defined(PPN_TU_W) || !defined(PPN_TU_NT)      // (a bare `hipcc -c` of this file compiles the first share)
#undef PPN_TU_W
#undef PPN_TU_NT
#define PPN_TU_W 1
#define PPN_TU_NT 0
#endif

#define PPN_INST(K) template __global__ void ppn_kernel<PPN_TU_W, K, PPN_TU_NT>(const KArgs);
PPN_INST(K_STEP) PPN_INST(K_GAMEOVER) PPN_INST(K_RESET) PPN_INST(K_RUNPF) PPN_INST(K_ROLLOUT) PPN_INST(K_STEP_PERSIST)
PPN_INST(K_POLICY_ROLLOUT) PPN_INST(K_STEP_OBS) PPN_INST(K_SERVE)
#if PPN_TU_NT == 0      // (kernels without a solve exist as NT = 0 only)
PPN_INST(K_VALID) PPN_INST(K_OBS) PPN_INST(K_POLICY)
#endif
#undef PPN_INST
#if PPN_TU_W == 4       // the schedule pre-pass of the engines whose busbars may split
template __global__ void ppn_sched_kernel<4, PPN_TU_NT, 64>(const KArgs);
template __global__ void ppn_sched_kernel<4, PPN_TU_NT, 256>(const KArgs);
#endif
