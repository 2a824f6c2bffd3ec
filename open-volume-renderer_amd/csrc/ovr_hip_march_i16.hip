// ovr_hip_march_i16.hip - the march / shade kernels of one voxel type (explicit instantiation; see ovr_hip_device.h)
#include "ovr_hip_device.h"

namespace ovrhip {
template hipError_t launch_v<VOX_I16>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
}
