"""ctypes mirror of include/vecchio_amd.h and vecchio_amd/host/host_api.h.

Plumbing only: the product is libvecchio_amd.so (HIP megakernel behind the C ABI) and
libvecchio_host.so (C++ stand-in for the reference's Rust host side).  There is no CPU
fallback here: `load_device_lib()` raises if the HIP library is missing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
ASSETS_DIR = os.path.join(os.path.dirname(_HERE), "tests", "golden", "assets")

VK_ABI_VERSION = 6
VK_OK, VK_ERR_BAD_ARG, VK_ERR_UNSUPPORTED, VK_ERR_HIP, VK_ERR_NO_DEVICE, VK_ERR_OOM = range(6)

(VK_KIND_NONE, VK_KIND_BVH, VK_KIND_SPHERE, VK_KIND_MOVING_SPHERE, VK_KIND_RECT, VK_KIND_LIST,
 VK_KIND_MEDIUM, VK_KIND_TRANSLATE, VK_KIND_ROTATE) = range(9)
VK_REF_FLIP = 0x08000000
VK_REF_INDEX_MASK = 0x07FFFFFF

(VK_MAT_LAMBERTIAN, VK_MAT_METAL, VK_MAT_DIELECTRIC, VK_MAT_DIFFUSE_LIGHT, VK_MAT_ISOTROPIC,
 VK_MAT_SPEC_DIFFUSE) = range(6)
VK_TEX_SOLID, VK_TEX_CHECKER, VK_TEX_IMAGE, VK_TEX_NOISE = range(4)
VK_INTEGRATOR_PDF, VK_INTEGRATOR_SCATTER = 0, 1
VK_BACKGROUND_SOLID, VK_BACKGROUND_SKY = 0, 1
VK_OUTPUT_F32, VK_OUTPUT_RGB8 = 0, 1
VK_SCENE_FAST_ACCEL = 1
VK_SCENE_REFERENCE_TREE = 2
VK_SCENE_EMPIRICAL_TREES = 4
VK_SCENE_RCCL_GATHER = 8


def make_ref(kind, index, flip=False):
    return (kind << 28) | (index & VK_REF_INDEX_MASK) | (VK_REF_FLIP if flip else 0)


F3 = C.c_float * 3


class BvhNode(C.Structure):
    _fields_ = [("bb_min", F3), ("bb_max", F3), ("left", C.c_uint32), ("right", C.c_uint32)]


class Sphere(C.Structure):
    _fields_ = [("center", F3), ("radius", C.c_float), ("material", C.c_uint32)]


class MovingSphere(C.Structure):
    _fields_ = [("center0", F3), ("center1", F3), ("time0", C.c_float), ("time1", C.c_float),
                ("radius", C.c_float), ("material", C.c_uint32)]


class Rect(C.Structure):
    _fields_ = [("c0", C.c_float), ("c1", C.c_float), ("d0", C.c_float), ("d1", C.c_float), ("k", C.c_float),
                ("axis0", C.c_uint8), ("axis1", C.c_uint8), ("axis2", C.c_uint8), ("_pad", C.c_uint8),
                ("material", C.c_uint32)]


class List(C.Structure):
    _fields_ = [("first", C.c_uint32), ("count", C.c_uint32)]


class Medium(C.Structure):
    _fields_ = [("boundary", C.c_uint32), ("neg_inv_density", C.c_float), ("material", C.c_uint32)]


class Translate(C.Structure):
    _fields_ = [("child", C.c_uint32), ("offset", F3)]


class Rotate(C.Structure):
    _fields_ = [("child", C.c_uint32), ("axis", C.c_uint32), ("sin_theta", C.c_float), ("cos_theta", C.c_float)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32), ("param", C.c_float), ("a", C.c_uint32), ("b", C.c_uint32)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", F3), ("a", C.c_uint32), ("b", C.c_uint32), ("scale", C.c_float)]


class Image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgb", C.POINTER(C.c_uint8))]


class Perlin(C.Structure):
    _fields_ = [("ranvec", (C.c_float * 3) * 256), ("perm_x", C.c_uint32 * 256), ("perm_y", C.c_uint32 * 256),
                ("perm_z", C.c_uint32 * 256)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_bvh", C.c_uint32), ("bvh", C.POINTER(BvhNode)),
        ("n_spheres", C.c_uint32), ("spheres", C.POINTER(Sphere)),
        ("n_moving_spheres", C.c_uint32), ("moving_spheres", C.POINTER(MovingSphere)),
        ("n_rects", C.c_uint32), ("rects", C.POINTER(Rect)),
        ("n_lists", C.c_uint32), ("lists", C.POINTER(List)),
        ("n_list_items", C.c_uint32), ("list_items", C.POINTER(C.c_uint32)),
        ("n_media", C.c_uint32), ("media", C.POINTER(Medium)),
        ("n_translates", C.c_uint32), ("translates", C.POINTER(Translate)),
        ("n_rotates", C.c_uint32), ("rotates", C.POINTER(Rotate)),
        ("n_materials", C.c_uint32), ("materials", C.POINTER(Material)),
        ("n_textures", C.c_uint32), ("textures", C.POINTER(Texture)),
        ("n_images", C.c_uint32), ("images", C.POINTER(Image)),
        ("n_perlins", C.c_uint32), ("perlins", C.POINTER(Perlin)),
        ("world", C.c_uint32),
        ("n_lights", C.c_uint32), ("lights", C.POINTER(C.c_uint32)),
        ("flags", C.c_uint32),
    ]


class Camera(C.Structure):
    _fields_ = [("origin", F3), ("lower_left_corner", F3), ("horizontal", F3), ("vertical", F3),
                ("u", F3), ("v", F3), ("w", F3), ("lens_radius", C.c_float), ("time0", C.c_float), ("time1", C.c_float)]


class RenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_pixel", C.c_uint32),
                ("max_depth", C.c_uint32), ("seed", C.c_uint64), ("integrator", C.c_uint32),
                ("background", C.c_uint32), ("background_color", F3), ("tile_rank", C.c_uint32),
                ("tile_world", C.c_uint32), ("output_format", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("seconds", C.c_double), ("kernel_ms", C.c_double),
                ("kernel_launches", C.c_uint32), ("scene_in_lds", C.c_uint32), ("clamped_samples", C.c_uint64)]


class SceneInfo(C.Structure):
    _fields_ = [("n_items", C.c_uint32), ("n_prims", C.c_uint32), ("n_instances", C.c_uint32),
                ("device_bytes", C.c_uint64), ("lds_bytes", C.c_uint32), ("features", C.c_uint32), ("tree", C.c_uint32),
                ("tree_suspended_frames", C.c_uint32), ("gather", C.c_uint32)]


class PartInfo(C.Structure):
    _fields_ = [("n_parts", C.c_uint32), ("device", C.c_int32), ("name", C.c_char * 64), ("pci_bus_id", C.c_char * 32),
                ("can_access_landing_device", C.c_uint32), ("kernel_ms", C.c_double)]


VK_TREE_HANDED_OVER, VK_TREE_REBUILT_PROVEN, VK_TREE_REBUILT_EMPIRICAL, VK_TREE_REBUILT_FAST, VK_TREE_REBUILT_NEAR, VK_TREE_REBUILT_GRID = range(6)
VK_GATHER_NONE, VK_GATHER_PEER_COPY, VK_GATHER_RCCL = range(3)


_host = None
_dev = None


def _build_locked(target, builder):
    """Build `target` (if missing or older than its sources) under an exclusive file lock, so that N ranks
    started together on a fresh checkout run ONE compile and nobody loads a half-written library: the
    builder writes to a temporary name and renames it into place."""
    import fcntl
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            builder()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if not os.path.exists(target):
        raise RuntimeError(f"{target} was not built")


def load_host_lib():
    """libvecchio_host.so: scene builders / BVHNode::new / Camera::new (C++ mirror of scene.rs)."""
    global _host
    if _host is not None:
        return _host
    path = os.path.join(LIB_DIR, "libvecchio_host.so")
    from . import build
    if build.host_is_stale():
        _build_locked(path, build.build_host)   # a fresh or edited checkout: compile (g++), never substitute anything
    lib = C.CDLL(path)
    lib.vkh_scene_build.restype = C.c_void_p
    lib.vkh_scene_build.argtypes = [C.c_char_p, C.c_uint64]
    lib.vkh_scene_free.argtypes = [C.c_void_p]
    lib.vkh_scene_desc.restype = C.POINTER(SceneDesc)
    lib.vkh_scene_desc.argtypes = [C.c_void_p]
    lib.vkh_scene_next_camera.restype = C.c_int
    lib.vkh_scene_next_camera.argtypes = [C.c_void_p, C.POINTER(Camera)]
    lib.vkh_scene_defaults.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), F3]
    lib.vkh_camera_new.argtypes = [F3, F3, F3, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Camera)]
    lib.vkh_last_error.restype = C.c_char_p
    lib.vkh_write_ppm.restype = C.c_int
    lib.vkh_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
    lib.vkh_to_color.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    lib.vkh_set_assets_dir.argtypes = [C.c_char_p]
    # the decoded copies of the reference's assets/*.png (data fixtures, tests/golden/make_assets.py)
    lib.vkh_set_assets_dir((os.environ.get("VECCHIO_ASSETS") or ASSETS_DIR).encode())
    _host = lib
    return lib


DEVICE_SYMBOLS = [
    "vk_abi_version", "vk_device_count", "vk_last_error", "vk_gather_backends", "vk_scene_create", "vk_scene_destroy",
    "vk_render", "vk_render_device", "vk_to_color_device", "vk_scene_get_info", "vk_scene_create_multi",
    "vk_scene_last_kernel_ms", "vk_scene_last_clamped_samples", "vk_scene_last_requeued_samples", "vk_scene_part_info",
    "vk_tile_slab_bytes", "vk_pack_tiles_device", "vk_unpack_tiles_device",
]


def device_lib_path():
    # VK_DEVICE_LIB: an alternate build of the same library (kernel experiments); never a different backend
    return os.environ.get("VK_DEVICE_LIB") or os.path.join(LIB_DIR, "libvecchio_amd.so")


def _bind(lib):
    """restype / argtypes of every entry point of include/vecchio_amd.h"""
    lib.vk_abi_version.restype = C.c_int
    lib.vk_gather_backends.restype = C.c_int
    lib.vk_device_count.restype = C.c_int
    lib.vk_last_error.restype = C.c_char_p
    lib.vk_scene_create.restype = C.c_int
    lib.vk_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
    lib.vk_scene_create_multi.restype = C.c_int
    lib.vk_scene_create_multi.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    lib.vk_scene_destroy.argtypes = [C.c_void_p]
    lib.vk_scene_last_kernel_ms.restype = C.c_int
    lib.vk_scene_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.vk_scene_last_clamped_samples.restype = C.c_int
    lib.vk_scene_last_clamped_samples.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.vk_scene_last_requeued_samples.restype = C.c_int
    lib.vk_scene_last_requeued_samples.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.vk_scene_part_info.restype = C.c_int
    lib.vk_scene_part_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(PartInfo)]
    lib.vk_render.restype = C.c_int
    lib.vk_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]
    lib.vk_render_device.restype = C.c_int
    lib.vk_render_device.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    lib.vk_to_color_device.restype = C.c_int
    lib.vk_to_color_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.vk_tile_slab_bytes.restype = C.c_size_t
    lib.vk_tile_slab_bytes.argtypes = [C.c_uint32] * 5
    for fn in (lib.vk_pack_tiles_device, lib.vk_unpack_tiles_device):
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.vk_scene_get_info.restype = C.c_int
    lib.vk_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]


_dbg = None


def load_debug_lib():
    """libvecchio_amd_debug.so: the same sources built with -DVK_DEBUG_LIB (instrumented kernel builds, device arithmetic probe:
    include/vecchio_amd_debug.h).  TESTS AND DIAGNOSTICS ONLY.  A vk_scene belongs to the library that created it:
    DeviceScene(desc, lib=load_debug_lib())."""
    global _dbg
    if _dbg is not None:
        return _dbg
    path = os.path.join(LIB_DIR, "libvecchio_amd_debug.so")
    from . import build
    if build.debug_is_stale():
        _build_locked(path, build.build_device_debug)
    lib = C.CDLL(path)
    _bind(lib)
    lib.vk_debug_phase_stats.restype = C.c_int
    lib.vk_debug_phase_stats.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.POINTER(C.c_uint64 * 24)]
    lib.vk_debug_math.restype = C.c_int
    lib.vk_debug_math.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    _dbg = lib
    return lib


def load_device_lib():
    """libvecchio_amd.so: the HIP product behind include/vecchio_amd.h.  Fails loudly if absent."""
    global _dev
    if _dev is not None:
        return _dev
    path = device_lib_path()
    from . import build
    if not os.environ.get("VK_DEVICE_LIB") and build.device_is_stale():
        try:                        # a fresh or edited checkout: compile with hipcc; there is no CPU fallback to use instead
            _build_locked(path, build.build_device)
        except Exception as e:
            raise RuntimeError(f"{path} missing or stale and hipcc could not build it ({e}); the HIP extension is required "
                               "(no CPU fallback exists)") from e
    lib = C.CDLL(path)
    _bind(lib)
    _dev = lib
    return lib
