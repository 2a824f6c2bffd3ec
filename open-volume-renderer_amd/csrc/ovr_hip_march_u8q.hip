// ovr_hip_march_u8q.hip - the march / shade kernels of one voxel layout (explicit instantiation; see ovr_hip_device.h)
#include "ovr_hip_device.h"

namespace ovrhip {
template hipError_t launch_v<VOX_U8_Q>(const RayMarchParams&, hipStream_t, const hipEvent_t*);
}
