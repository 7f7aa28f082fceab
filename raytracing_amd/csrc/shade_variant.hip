// One feature set of the pipeline's shading kernels (wf_settle_kernel, wf_shade_kernel; see
// RT_SHADE_VARIANTS in rt_wavefront.h): compiled once per set with -DRT_SHADE_VARIANT=<mask>, so
// that the sets build side by side.  librt_hip.so links them; rt_hip.hip picks one per scene.
#include <hip/hip_runtime.h>

#define RT_SHADE_VARIANT_TU 1
#include "rt_wavefront.h"

#ifndef RT_SHADE_VARIANT
#error "compile with -DRT_SHADE_VARIANT=<feature mask>"
#endif

template void rtd::launch_shade_stage<RT_SHADE_VARIANT>(rtd::wavefront_params const&, uint32_t,
    dim3, dim3, uint32_t, hipStream_t);
