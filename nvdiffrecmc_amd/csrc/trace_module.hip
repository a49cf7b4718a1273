// trace_module.hip -- the shadow-ray traversal kernel as a stand-alone gfx950 code object (nvdr_trace_gfx950.hsaco).
//
// Same body as the copies inside libnvdr_hip.so (trace_kernel.h).  The library loads this file with hipModuleLoad -- every
// load places the code in memory of its own -- when it selects the instance of the kernel a process runs
// (env_shade.hip, "Which INSTANCE of the traversal kernel this process launches").  Device code only.
#include "trace_kernel.h"

extern "C" __global__ void __launch_bounds__(NVDR_QUERY_BLOCK, 8) nvdr_trace_module_kernel(TraceLaunch a)
{
    extern __shared__ __attribute__((aligned(16))) int smem[];
    env_trace_body<false, 3>(a, smem);
}
