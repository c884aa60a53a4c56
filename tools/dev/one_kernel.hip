// Developer aid: compiles ONE instance of the step kernel (default: W = 2, K_STEP, Newton flavour) so that register
// allocation and ISA of a change can be inspected in seconds instead of minutes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c -Rpass-analysis=kernel-resource-usage \
//         [-DDEV_W=2 -DDEV_KIND=K_STEP -DDEV_NT=1] tools/dev/one_kernel.hip -o /tmp/one.o
#include <hip/hip_runtime.h>
#include "../../include/ppn.h"
#include "../../pypownet_amd/csrc/ppn_device.h"
#include "../../pypownet_amd/csrc/ppn_solve.inc"
#include "../../pypownet_amd/csrc/ppn_game.inc"
#include "../../pypownet_amd/csrc/ppn_obs.inc"
#include "../../pypownet_amd/csrc/ppn_kernels.inc"
#ifndef DEV_W
#define DEV_W 2
#endif
#ifndef DEV_KIND
#define DEV_KIND K_STEP
#endif
#ifndef DEV_NT
#define DEV_NT 1
#endif
template __global__ void ppn_kernel<DEV_W, DEV_KIND, DEV_NT>(const KArgs);
