"""Host-side mirror of the reference's renderer interface for the HIP device.

`DeviceHIP` has the same method names, argument meaning, call protocol and error behaviour as
`ovr::MainRenderer` (reference ovr/renderer.h:82-341) as implemented by its GPU device
(ovr/devices/optix7/device.cpp:16-49): queued thread-safe setters, `init(scene, camera)`, then per frame
`commit()`, `render()`, `mapframe()`, `swap()`.  Everything is forwarded to the C ABI in include/ovr_hip.h;
torch is only used to hand device memory across (volumes living in HBM, frames mapped as tensors)."""
import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _lib as L

_NP_TO_TYPE = {
    np.dtype(np.uint8): L.TYPE_UINT8, np.dtype(np.int8): L.TYPE_INT8,
    np.dtype(np.uint16): L.TYPE_UINT16, np.dtype(np.int16): L.TYPE_INT16,
    np.dtype(np.uint32): L.TYPE_UINT32, np.dtype(np.int32): L.TYPE_INT32,
    np.dtype(np.float32): L.TYPE_FLOAT, np.dtype(np.float64): L.TYPE_DOUBLE,
}


@dataclass
class Camera:
    """ovr::scene::Camera (reference ovr/scene.h:201-231); fovy defaults to 60 like PerspectiveCamera."""
    eye: Sequence[float] = (0.0, 0.0, -1000.0)   # `from` in the reference (a Python keyword)
    at: Sequence[float] = (0.0, 0.0, 0.0)
    up: Sequence[float] = (0.0, 1.0, 0.0)
    fovy: float = 60.0


@dataclass
class TransferFunction:
    """ovr::scene::TransferFunction (scene.h:233-237): color = N x 4 float (rgb + unused w), opacity = M float."""
    color: np.ndarray = None
    opacity: np.ndarray = None
    value_range: Sequence[float] = (1.0, -1.0)


@dataclass
class Scene:
    """The subset of ovr::scene::Scene the path consumes: one structured-regular volume + its transfer function
    (what parse_single_volume_scene accepts, scene.h:413-426) and the render settings main_batch forwards."""
    volume: object = None                      # numpy array or torch tensor, shape (nz, ny, nx), x fastest
    grid_origin: Sequence[float] = (0.0, 0.0, 0.0)
    grid_spacing: Sequence[float] = (1.0, 1.0, 1.0)
    transfer_function: TransferFunction = field(default_factory=TransferFunction)
    camera: Camera = field(default_factory=Camera)
    volume_sampling_rate: float = 1.0
    spp: int = 1


class _DevicePtr:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch.as_tensor can view it without a copy."""

    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 3}


class CrossDeviceBuffer:
    """Mirror of the reference's CrossDeviceBuffer (ovr/common/cross_device_buffer.h:19-208): a non-owning view of the
    frame, either on the host or on the device; to_cpu() returns host data."""
    DEVICE_CPU, DEVICE_HIP = 0, 1

    def __init__(self):
        self._data = None
        self.device = self.DEVICE_CPU
        self.nbytes = 0

    def set_data(self, data, nbytes, device):
        self._data, self.nbytes, self.device = data, nbytes, device

    def data(self):
        return self._data

    def to_cpu(self):
        if self.device == self.DEVICE_CPU:
            return self
        out = CrossDeviceBuffer()
        out.set_data(self._data.cpu().numpy(), self.nbytes, self.DEVICE_CPU)
        return out


class FrameBufferData:
    """MainRenderer::FrameBufferData (renderer.h:89-97)."""

    def __init__(self):
        self.rgba = CrossDeviceBuffer()
        self.grad = CrossDeviceBuffer()


def _f3(v):
    a = (C.c_float * 3)(*[float(x) for x in v])
    return a


class DeviceHIP:
    """The "hip" device.  Method-for-method mirror of ovr::MainRenderer + DeviceOptix7."""

    def __init__(self, device_id: int = 0, devices=None):
        """devices: several HIP device ordinals -> one in-process device group behind this object (ovr_hip_create_group: image tiles
        sharded over the devices, gathered on devices[0]); the reference's device knows one GPU (device_impl.cpp:371-372)"""
        self._lib = L.load()
        self._h = C.c_void_p()
        if devices is not None and len(devices) > 0:
            ids = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            L.check(self._lib.ovr_hip_create_group(C.byref(self._h), ids, len(devices)))
            device_id = int(devices[0])
        else:
            L.check(self._lib.ovr_hip_create(C.byref(self._h), int(device_id)))
        self.device_id = int(device_id)
        self.current_scene: Optional[Scene] = None
        self.variance = float("inf")          # renderer.h:287
        self._fbsize = (0, 0)
        self._keep = []                        # keeps ctypes buffers alive across calls

    # ---- lifetime -------------------------------------------------------------------------------------------
    def close(self):
        if self._h:
            self._lib.ovr_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- thread-safe setters (renderer.h:135-248) ------------------------------------------------------------
    def set_fbsize(self, fbsize):
        self._fbsize = (int(fbsize[0]), int(fbsize[1]))
        L.check(self._lib.ovr_hip_set_fbsize(self._h, *self._fbsize))

    def set_camera(self, camera_or_from, at=None, up=None):
        """set_camera(Camera) or set_camera(from, at, up).  The three-vector form builds Camera{from, at, up} whose fovy
        is the default 60 degrees - exactly what renderer.h:149-152 does (it discards a scene's fovy)."""
        cam = camera_or_from if isinstance(camera_or_from, Camera) else Camera(camera_or_from, at, up)
        L.check(self._lib.ovr_hip_set_camera(self._h, _f3(cam.eye), _f3(cam.at), _f3(cam.up), float(cam.fovy)))

    def set_transfer_function(self, c, o, r):
        """c: flat RGB triples, o: flat (position, alpha) pairs, r: (lo, hi) in raw data units (renderer.h:154-161)."""
        c = np.ascontiguousarray(c, dtype=np.float32).ravel()
        o = np.ascontiguousarray(o, dtype=np.float32).ravel()
        if c.size % 3 or o.size % 2:
            raise RuntimeError("transfer function arrays must hold RGB triples and (position, alpha) pairs")
        L.check(self._lib.ovr_hip_set_transfer_function(
            self._h, c.ctypes.data_as(C.POINTER(C.c_float)), c.size // 3, o.ctypes.data_as(C.POINTER(C.c_float)), o.size // 2,
            float(r[0]), float(r[1])))

    def set_focus(self, center, scale, base_noise):
        L.check(self._lib.ovr_hip_set_focus(self._h, float(center[0]), float(center[1]), float(scale), float(base_noise)))

    def set_sample_per_pixel(self, spp):
        L.check(self._lib.ovr_hip_set_sample_per_pixel(self._h, int(spp)))

    def set_sparse_sampling(self, on):
        L.check(self._lib.ovr_hip_set_sparse_sampling(self._h, int(bool(on))))

    def set_frame_accumulation(self, on):
        L.check(self._lib.ovr_hip_set_frame_accumulation(self._h, int(bool(on))))

    def set_volume_sampling_rate(self, rate):
        self._rate_set_by_app = True
        L.check(self._lib.ovr_hip_set_volume_sampling_rate(self._h, float(rate)))

    def set_path_tracing(self, on):
        if on:  # the path tracer is outside this backend's scope
            raise RuntimeError("[hip] path tracing is not part of the ray-marching backend")

    # accepted and ignored exactly as the reference's ray marcher ignores them (device_impl.cpp:113-197 never reads them)
    def set_add_lights(self, v): pass
    def set_photonmapping(self, v): pass
    def set_volume_density_scale(self, v): pass
    def set_mat_ambient(self, v): pass
    def set_mat_diffuse(self, v): pass
    def set_mat_specular(self, v): pass
    def set_mat_shininess(self, v): pass
    def set_light_phi(self, v): pass
    def set_light_theta(self, v): pass
    def set_light_radius(self, v): pass
    def set_light_intensity(self, v): pass

    # ---- extensions of this backend ----------------------------------------------------------------------------
    def set_shading(self, mode):
        L.check(self._lib.ovr_hip_set_shading(self._h, int(mode)))

    def set_shading_pipeline(self, mode):
        """0 auto, 1 in place, 2 pooled (include/ovr_hip.h) - both produce bit-identical frames"""
        L.check(self._lib.ovr_hip_set_shading_pipeline(self._h, int(mode)))

    def set_empty_space_skipping(self, on):
        """skip the voxel fetch of samples in macrocells whose max TF opacity is 0 (frames stay bit-identical)"""
        L.check(self._lib.ovr_hip_set_empty_space_skipping(self._h, int(bool(on))))

    def macrocells(self):
        """(minmax[mz,my,mx,2], majorant[mz,my,mx]) of the reference's 16^3 macrocell grids"""
        dims = (C.c_int32 * 3)()
        L.check(self._lib.ovr_hip_get_macrocells(self._h, dims, None, None, 0))
        n = dims[0] * dims[1] * dims[2]
        mm = np.zeros((dims[2], dims[1], dims[0], 2), np.float32)
        mj = np.zeros((dims[2], dims[1], dims[0]), np.float32)
        L.check(self._lib.ovr_hip_get_macrocells(self._h, dims, mm.ctypes.data_as(C.POINTER(C.c_float)), mj.ctypes.data_as(C.POINTER(C.c_float)), n))
        return mm, mj

    def set_pixel_jitter(self, mode):
        """0 = RandomTEA, applied iff spp > 1 (the reference); 1 = blue-noise tile (set_noise_tile), applied to every sample
        of every frame - progressive accumulation with frame-indexed slices (BASELINE C5)"""
        L.check(self._lib.ovr_hip_set_pixel_jitter(self._h, int(mode)))

    def volume_info(self):
        """dims, bytes resident in HBM, the data range found at load (array.cpp:27-66,297) and the TF range in effect"""
        v = L.VolumeInfo()
        L.check(self._lib.ovr_hip_get_volume_info(self._h, C.byref(v)))
        return v

    def set_lds_staging(self, on):
        """LDS-staged bricks for the unshaded march of float volumes (include/ovr_hip.h); off by default"""
        L.check(self._lib.ovr_hip_set_lds_staging(self._h, int(bool(on))))

    def set_phase_timing(self, on):
        """per-phase device times in stats() (march_ms / shade_ms / composite_ms): on by default; off saves the two events between the
        frame's kernels (~16 us per frame).  kernel_ms is measured either way; frames are identical."""
        L.check(self._lib.ovr_hip_set_phase_timing(self._h, int(bool(on))))

    def set_volume_layouts(self, mode):
        """which layouts of the volume the next init / volume upload keeps in HBM: 0 general only, 1 (default) thin replicas
        when they fit comfortably, 2 always (include/ovr_hip.h)"""
        L.check(self._lib.ovr_hip_set_volume_layouts(self._h, int(mode)))

    def set_layout_choice(self, choice):
        """-1 (default): automatic (camera direction; the quad replica for frames that shade every sample); 0 / 1 / 2 / 3: forced
        (general, thin, thin transposed, quad).  Frames are bit-identical."""
        L.check(self._lib.ovr_hip_set_layout_choice(self._h, int(choice)))

    def set_grid_convention(self, convention):
        L.check(self._lib.ovr_hip_set_grid_convention(self._h, int(convention)))

    def set_noise_tile(self, tile):
        tile = np.ascontiguousarray(tile, dtype=np.float32)
        xy = int(round((tile.size // 64) ** 0.5))
        if xy * xy * 64 != tile.size:
            raise RuntimeError("noise tile must hold xy*xy*64 floats, layout [y][x][t]")
        L.check(self._lib.ovr_hip_set_noise_tile(self._h, tile.ctypes.data_as(C.POINTER(C.c_float)), xy))

    def set_image_shard(self, rank, world, tile_w=64, tile_h=64):
        L.check(self._lib.ovr_hip_set_image_shard(self._h, int(rank), int(world), int(tile_w), int(tile_h)))

    def set_stream(self, stream_ptr):
        L.check(self._lib.ovr_hip_set_stream(self._h, C.c_void_p(int(stream_ptr) if stream_ptr else None)))

    # ---- init / per-frame protocol (renderer.h:107-116,290-341) ----------------------------------------------
    def set_scene(self, scene: Scene):
        """MainRenderer::set_scene (renderer.h:299-341): flatten the scene TF into the app-side format and queue it."""
        tfn = scene.transfer_function
        if tfn is not None and tfn.color is not None and tfn.opacity is not None:
            color = np.asarray(tfn.color, dtype=np.float32).reshape(-1, 4)
            opacity = np.asarray(tfn.opacity, dtype=np.float32).ravel()
            tfn_colors = color[:, :3].ravel()
            pos = np.arange(opacity.size, dtype=np.float32) / np.float32(max(opacity.size - 1, 1))
            tfn_alphas = np.stack([pos, opacity], axis=1).ravel()
            self.set_transfer_function(tfn_colors, tfn_alphas, tfn.value_range)
        self.current_scene = scene

    def init(self, scene: Scene, camera: Camera):
        """MainRenderer::init(argc, argv, scene, camera) (renderer.h:290-297) followed by DeviceOptix7::init ->
        Impl::buildScene (device_impl.cpp:283-302): upload the volume, take the scene's sampling rate, first commit."""
        self.set_scene(scene)
        self.set_camera(camera)
        self._upload_volume(scene)
        # buildScene applies the scene's rate directly; an earlier set_volume_sampling_rate() stays queued and wins at commit
        # (device_impl.cpp:298 then :190-196) - reproduced by not queuing the scene's rate when the app already set one.
        if not getattr(self, "_rate_set_by_app", False):
            L.check(self._lib.ovr_hip_set_volume_sampling_rate(self._h, float(scene.volume_sampling_rate)))
        self.commit()

    def _upload_volume(self, scene: Scene):
        vol = scene.volume
        if vol is None:
            raise RuntimeError("expect only one instance")  # parse_single_volume_scene, scene.h:416
        origin, spacing = _f3(scene.grid_origin), _f3(scene.grid_spacing)
        if isinstance(vol, np.ndarray):
            if vol.ndim != 3:
                raise RuntimeError("volume must have shape (nz, ny, nx)")
            vol = np.ascontiguousarray(vol)
            vt = _NP_TO_TYPE.get(vol.dtype)
            if vt is None:
                raise RuntimeError("[Optix7] unexpected volume type ...")
            dims = (C.c_int32 * 3)(vol.shape[2], vol.shape[1], vol.shape[0])
            L.check(self._lib.ovr_hip_set_volume(self._h, C.c_void_p(vol.ctypes.data), L.MEM_HOST, vt, dims, origin, spacing))
            return
        import torch
        if not isinstance(vol, torch.Tensor) or vol.dim() != 3:
            raise RuntimeError("volume must be a numpy array or torch tensor of shape (nz, ny, nx)")
        vol = vol.contiguous()
        tmap = {torch.uint8: L.TYPE_UINT8, torch.int8: L.TYPE_INT8, torch.int16: L.TYPE_INT16, torch.int32: L.TYPE_INT32,
                torch.float32: L.TYPE_FLOAT, torch.float64: L.TYPE_DOUBLE}
        if hasattr(torch, "uint16"):
            tmap[torch.uint16] = L.TYPE_UINT16
        vt = tmap.get(vol.dtype)
        if vt is None:
            raise RuntimeError("[Optix7] unexpected volume type ...")
        dims = (C.c_int32 * 3)(vol.shape[2], vol.shape[1], vol.shape[0])
        kind = L.MEM_DEVICE if vol.is_cuda else L.MEM_HOST
        if vol.is_cuda:
            torch.cuda.current_stream(vol.device).synchronize()
        L.check(self._lib.ovr_hip_set_volume(self._h, C.c_void_p(vol.data_ptr()), kind, vt, dims, origin, spacing))

    def commit(self):
        L.check(self._lib.ovr_hip_commit(self._h))

    def render(self):
        L.check(self._lib.ovr_hip_render(self._h))
        self.variance = 0.0  # device_impl.cpp:266

    def render_async(self):
        L.check(self._lib.ovr_hip_render_async(self._h))

    def sync(self):
        L.check(self._lib.ovr_hip_sync(self._h))

    def swap(self):
        L.check(self._lib.ovr_hip_swap(self._h))

    def mapframe(self, fb: FrameBufferData, device: bool = False):
        """Impl::mapframe (device_impl.cpp:271-281).  device=True hands out device memory as torch tensors (what the
        reference does, DEVICE_CUDA); device=False returns host arrays (the caller's to_cpu())."""
        rgba, grad = C.c_void_p(), C.c_void_p()
        nb_rgba, nb_grad = C.c_size_t(), C.c_size_t()
        kind = L.MEM_DEVICE if device else L.MEM_HOST
        L.check(self._lib.ovr_hip_mapframe(self._h, kind, C.byref(rgba), C.byref(nb_rgba), C.byref(grad), C.byref(nb_grad)))
        w, h = self._fbsize
        if device:
            import torch
            dev = torch.device("cuda", self.device_id)
            t_rgba = torch.as_tensor(_DevicePtr(rgba.value, (h, w, 4)), device=dev)
            t_grad = torch.as_tensor(_DevicePtr(grad.value, (h, w, 3)), device=dev)
            fb.rgba.set_data(t_rgba, nb_rgba.value, CrossDeviceBuffer.DEVICE_HIP)
            fb.grad.set_data(t_grad, nb_grad.value, CrossDeviceBuffer.DEVICE_HIP)
        else:
            a_rgba = np.ctypeslib.as_array(C.cast(rgba, C.POINTER(C.c_float)), shape=(h, w, 4))
            a_grad = np.ctypeslib.as_array(C.cast(grad, C.POINTER(C.c_float)), shape=(h, w, 3))
            fb.rgba.set_data(a_rgba, nb_rgba.value, CrossDeviceBuffer.DEVICE_CPU)
            fb.grad.set_data(a_grad, nb_grad.value, CrossDeviceBuffer.DEVICE_CPU)
        return fb

    def mapframe_rgba8(self, flip_vertical=True, device=False):
        """the current frame as packed RGBA8, converted on the GPU exactly like the reference's image_to_rgba8
        (imageio.cpp:146-181; renderbatch saves its PNG from this, flipped).  Returns a (H, W, 4) uint8 numpy array (host) or
        torch tensor (device=True); valid until the next call."""
        ptr, nb = C.c_void_p(), C.c_size_t()
        L.check(self._lib.ovr_hip_mapframe_rgba8(self._h, L.MEM_DEVICE if device else L.MEM_HOST, 1 if flip_vertical else 0, C.byref(ptr), C.byref(nb)))
        w, h = self._fbsize
        if device:
            import torch
            return torch.as_tensor(_DevicePtr(ptr.value, (h, w, 4), typestr="|u1"), device=torch.device("cuda", self.device_id))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(h, w, 4))

    def mapframe_rgba16f(self, flip_vertical=True, device=False):
        """the current frame as RGBA half (IEEE binary16 bit patterns, (H, W, 4) uint16), converted on the GPU with the float ->
        half rule of the reference's EXR writer (imageio.cpp:15-83, tinyexr); valid until the next call."""
        ptr, nb = C.c_void_p(), C.c_size_t()
        L.check(self._lib.ovr_hip_mapframe_rgba16f(self._h, L.MEM_DEVICE if device else L.MEM_HOST, 1 if flip_vertical else 0, C.byref(ptr), C.byref(nb)))
        w, h = self._fbsize
        if device:
            import torch
            return torch.as_tensor(_DevicePtr(ptr.value, (h, w, 4), typestr="<i2"), device=torch.device("cuda", self.device_id))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint16)), shape=(h, w, 4))

    def save_image(self, path, flip_vertical=True):
        """ovr::save_image of the reference (imageio.cpp:264-284) for the current frame: the extension picks the format -
        "exr" (RGBA half, rows flipped before the write), "png" / "jpg" (image_to_rgba8, rows flipped on write).  The pixel
        conversion runs on the GPU (mapframe_rgba16f / mapframe_rgba8), imageio.py writes the file."""
        from . import imageio
        ext = path.rsplit(".", 1)[-1].lower()
        if ext == "exr":
            imageio.save_exr(path, np.array(self.mapframe_rgba16f(flip_vertical=flip_vertical), copy=True))
        elif ext in ("jpg", "jpeg"):
            imageio.save_jpg(path, np.array(self.mapframe_rgba8(flip_vertical=flip_vertical), copy=True))
        else:
            imageio.save_png(path, np.array(self.mapframe_rgba8(flip_vertical=flip_vertical), copy=True))

    # ---- getters ------------------------------------------------------------------------------------------------
    @property
    def render_time(self):
        """MainRenderer::render_time (renderer.h:87): accumulated milliseconds inside render()."""
        return float(self._lib.ovr_hip_render_time_ms(self._h))

    def unsafe_get_fbsize(self):
        return self._fbsize

    def unsafe_get_variance(self):
        return self.variance

    def stats(self):
        s = L.Stats()
        L.check(self._lib.ovr_hip_get_stats(self._h, C.byref(s)))
        return s

    def group_info(self):
        """(devices, gather: 0 none / 1 peer copies / 2 RCCL, host milliseconds of the last frame's gather tail)"""
        n, kind, ms = C.c_int32(), C.c_int32(), C.c_double()
        L.check(self._lib.ovr_hip_group_info(self._h, C.byref(n), C.byref(kind), C.byref(ms)))
        return n.value, kind.value, ms.value

    def group_host_times(self):
        """host microseconds of the last frame's steps on the leader's thread: (enqueue, ship, finish, scatter); zeros without a group (ABI v10)"""
        out = (C.c_double * 4)()
        L.check(self._lib.ovr_hip_group_host_times(self._h, out))
        return tuple(out)

    def upload_times(self):
        """milliseconds of the last ovr_hip_set_volume: dict(total, alloc, copy, kernels) (ABI v10)"""
        out = (C.c_double * 4)()
        L.check(self._lib.ovr_hip_get_upload_times(self._h, out))
        return dict(total_ms=out[0], alloc_ms=out[1], copy_ms=out[2], kernels_ms=out[3])

    def member_stats(self, member):
        s = L.Stats()
        L.check(self._lib.ovr_hip_get_member_stats(self._h, int(member), C.byref(s)))
        return s

    # ---- stand-alone pieces for known-answer tests ----------------------------------------------------------------
    def sparse_mask(self, frame_index):
        import torch
        w, h = self._fbsize
        out = torch.empty(w * h * 2, dtype=torch.int32, device=torch.device("cuda", self.device_id))
        n = C.c_int64()
        L.check(self._lib.ovr_hip_sparse_mask(self._h, int(frame_index), C.c_void_p(out.data_ptr()), out.numel() * 4, C.byref(n)))
        return out[: n.value].cpu().numpy()

    def tea_floats(self, v0v1):
        import torch
        dev = torch.device("cuda", self.device_id)
        st = torch.as_tensor(np.ascontiguousarray(v0v1, dtype=np.uint32).view(np.int32)).to(dev)
        out = torch.empty(st.numel(), dtype=torch.float32, device=dev)
        L.check(self._lib.ovr_hip_tea_floats(self._h, C.c_void_p(st.data_ptr()), C.c_void_p(out.data_ptr()), st.numel() // 2))
        return out.cpu().numpy(), st.cpu().numpy().view(np.uint32)


    def pow_floats(self, x, y, which=0):
        """the kernels' `__powf(x, y)` on the device (known-answer entry): which = 0 the pow this library was built with, 1 the deterministic pair"""
        import torch
        dev = torch.device("cuda", self.device_id)
        xd = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
        yd = torch.as_tensor(np.ascontiguousarray(y, dtype=np.float32)).to(dev)
        out = torch.empty(xd.numel(), dtype=torch.float32, device=dev)
        L.check(self._lib.ovr_hip_pow_floats(self._h, C.c_void_p(xd.data_ptr()), C.c_void_p(yd.data_ptr()), C.c_void_p(out.data_ptr()), xd.numel(), int(which)))
        return out.cpu().numpy()


def create_renderer(name: str, device_id: int = 0, devices=None):
    """create_renderer(name) (reference ovr/renderer.cpp:42-61).  Only "hip" exists here; anything else raises the
    same way the reference's factory does for an unknown device.  devices = [ordinals]: an in-process device group."""
    if name == "hip":
        return DeviceHIP(device_id, devices)
    raise RuntimeError(f"OVR ERROR: Could not find device_{name} (only the 'hip' device is built)")
