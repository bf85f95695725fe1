"""Python handles over the two product libraries.

HostScene  = a scene built by the C++ mirror of scene.rs (libvecchio_host.so) and flattened
             into a vk_scene_desc (what the Rust shim's flatten() would hand over).
DeviceScene = that description uploaded through the C ABI (vk_scene_create) to one MI355X;
             render() is one call of the drop-in for main.rs:181-198.
"""
import ctypes as C

import numpy as np

from . import ffi


class HostScene:
    def __init__(self, name, seed=1):
        self._lib = ffi.load_host_lib()
        self._h = self._lib.vkh_scene_build(name.encode(), seed)
        if not self._h:
            raise RuntimeError(self._lib.vkh_last_error().decode())
        self.name = name
        self.desc = self._lib.vkh_scene_desc(self._h)  # POINTER(SceneDesc), owned by the handle
        ar, integ, bg = C.c_float(), C.c_uint32(), C.c_uint32()
        col = ffi.F3()
        self._lib.vkh_scene_defaults(self._h, C.byref(ar), C.byref(integ), C.byref(bg), col)
        self.aspect_ratio = ar.value
        self.integrator = integ.value
        self.background = bg.value
        self.background_color = tuple(col)

    def next_camera(self):
        cam = ffi.Camera()
        if not self._lib.vkh_scene_next_camera(self._h, C.byref(cam)):
            return None
        return cam

    def params(self, width, spp, max_depth, seed=2, height=None, tile_rank=0, tile_world=1, output_format=ffi.VK_OUTPUT_F32):
        """vk_render_params with the scene's integrator/background; height = width/aspect (main.rs:172)."""
        p = ffi.RenderParams()
        p.width = width
        p.height = height if height is not None else int(np.float32(width) / np.float32(self.aspect_ratio))
        p.samples_per_pixel = spp
        p.max_depth = max_depth
        p.seed = seed
        p.integrator = self.integrator
        p.background = self.background
        p.background_color = ffi.F3(*self.background_color)
        p.tile_rank = tile_rank
        p.tile_world = tile_world
        p.output_format = output_format
        return p

    def close(self):
        if self._h:
            self._lib.vkh_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def check(lib, status):
    if status != ffi.VK_OK:
        raise RuntimeError(f"vecchio_amd status {status}: {lib.vk_last_error().decode()}")


class DeviceScene:
    def __init__(self, desc, device=0, devices=None, lib=None):
        """device: one MI355X (vk_scene_create).  devices=[...]: one handle over several (vk_scene_create_multi):
        render() then deals the tiles over them and gathers on devices[0] inside the library.
        lib: another build of the library (ffi.load_debug_lib(): the handle belongs to the library that made it)."""
        self._lib = lib or ffi.load_device_lib()
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            check(self._lib, self._lib.vk_scene_create_multi(desc, arr, len(devices), C.byref(h)))
            device = devices[0]
        else:
            check(self._lib, self._lib.vk_scene_create(desc, device, C.byref(h)))
        self._h = h
        self.device = device

    def info(self):
        inf = ffi.SceneInfo()
        check(self._lib, self._lib.vk_scene_get_info(self._h, C.byref(inf)))
        return inf

    def render(self, cam, params, out=None):
        """Blocking render into a host numpy array (height, width, 3): float32 with y = 0 the bottom row, or
        (params.output_format == VK_OUTPUT_RGB8) uint8 with row 0 the top row."""
        dt = np.uint8 if params.output_format == ffi.VK_OUTPUT_RGB8 else np.float32
        if out is None:
            out = np.zeros((params.height, params.width, 3), dtype=dt)
        assert out.dtype == dt
        stats = ffi.Stats()
        check(self._lib, self._lib.vk_render(self._h, C.byref(cam), C.byref(params), out.ctypes.data_as(C.c_void_p), C.byref(stats)))
        return out, stats

    def render_device(self, cam, params, d_ptr, stream=None):
        """Enqueue a render into device memory at d_ptr on `stream` (no host sync)."""
        stats = ffi.Stats()
        check(self._lib, self._lib.vk_render_device(self._h, C.byref(cam), C.byref(params), C.c_void_p(d_ptr),
                                                    C.c_void_p(stream or 0), C.byref(stats)))
        return stats

    def last_kernel_ms(self):
        """HIP-event time of the launches of the last render (waits for them)."""
        ms = C.c_double()
        check(self._lib, self._lib.vk_scene_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def parts(self):
        """one record per device share of the scene (vk_scene_part_info): device, name, PCI bus id, peer access to devices[0], the kernel
        time of its launches in the last frame (waits for them)"""
        out = []
        pi = ffi.PartInfo()
        check(self._lib, self._lib.vk_scene_part_info(self._h, 0, C.byref(pi)))
        for j in range(pi.n_parts):
            check(self._lib, self._lib.vk_scene_part_info(self._h, j, C.byref(pi)))
            out.append({"part": j, "device_index": pi.device, "device": pi.name.decode(), "pci_bus_id": pi.pci_bus_id.decode(),
                        "can_access_devices0": bool(pi.can_access_landing_device), "kernel_ms": round(pi.kernel_ms, 3)})
        return out

    def last_requeued_samples(self):
        """Exact re-treeing: samples of the last render that went through the second launch, on the tree as handed over (waits)."""
        n = C.c_uint64()
        check(self._lib, self._lib.vk_scene_last_requeued_samples(self._h, C.byref(n)))
        return n.value

    def pack_tiles_device(self, d_fb, width, height, output_format, tile_rank, tile_world, d_slab, stream=None):
        """this rank's tiles of the f32 framebuffer at d_fb -> its slab at d_slab (floats, or bytes through to_color for RGB8)"""
        check(self._lib, self._lib.vk_pack_tiles_device(self._h, C.c_void_p(d_fb), width, height, output_format, tile_rank, tile_world,
                                                        C.c_void_p(d_slab), C.c_void_p(stream or 0)))

    def unpack_tiles_device(self, d_slab, width, height, output_format, tile_rank, tile_world, d_img, stream=None):
        """rank tile_rank's slab at d_slab -> its tiles of the full image at d_img (f32 y up, or RGB8 top row first)"""
        check(self._lib, self._lib.vk_unpack_tiles_device(self._h, C.c_void_p(d_slab), width, height, output_format, tile_rank, tile_world,
                                                          C.c_void_p(d_img), C.c_void_p(stream or 0)))

    def to_color_device(self, d_rgb, width, height, d_rgb8, stream=None):
        check(self._lib, self._lib.vk_to_color_device(self._h, C.c_void_p(d_rgb), width, height, C.c_void_p(d_rgb8), C.c_void_p(stream or 0)))

    def close(self):
        if self._h:
            self._lib.vk_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
