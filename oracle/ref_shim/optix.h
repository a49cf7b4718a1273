// stand-in for <optix.h> (device side): just what envsampling/kernel.cu touches.
#pragma once
#include "cuda_shim.h"

typedef unsigned long long OptixTraversableHandle;
typedef unsigned int OptixVisibilityMask;
enum {
    OPTIX_RAY_FLAG_DISABLE_ANYHIT = 1u << 0,
    OPTIX_RAY_FLAG_TERMINATE_ON_FIRST_HIT = 1u << 2,
    OPTIX_RAY_FLAG_DISABLE_CLOSESTHIT = 1u << 3,
};

static inline uint3 optixGetLaunchIndex() { return ref_launch_index; }
static inline uint3 optixGetLaunchDimensions() { return ref_launch_dim; }

extern thread_local unsigned int *ref_payload0;
static inline void optixSetPayload_0(unsigned int v) { *ref_payload0 = v; }

// any-hit over the driver's triangle soup; calls the reference's own miss program on a miss
void optixTrace(OptixTraversableHandle handle, float3 o, float3 d, float tmin, float tmax, float time,
                OptixVisibilityMask mask, unsigned int flags, unsigned int sbt_offset, unsigned int sbt_stride,
                unsigned int miss_index, unsigned int &p0);
