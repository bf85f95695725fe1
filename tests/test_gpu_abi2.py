"""-m gpu: the ABI-2 additions of include/vecchio_amd.h through the C ABI —
multi-device scenes (vk_scene_create_multi: SURVEY §8b "uploads to every participating GPU", §8e gather to
devices[0] + de-interleave + one D2H), the fused output stage (VK_OUTPUT_RGB8 = Vec3::to_color vec3.rs:44-61 +
top-down rows main.rs:209) and the MAX_DEPTH = 0 corner (main.rs:126-128)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_checks as G
from vecchio_amd import DeviceScene, HostScene, ffi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_to_color_kernel_against_numpy_restatement(device, host_scenes):
    """vk_to_color_device against an independent numpy restatement of Vec3::clamp / to_color (vec3.rs:44-61):
    NaN (falls through the clamp, `as u32` -> 0), negatives (sqrt -> NaN -> 0), -0, values >= 1 and +inf (clamped
    to 0.999 -> 255), the 0.999 boundary, denormals, and random values — and the top-down row flip (main.rs:209)."""
    import torch
    hs, cam = host_scenes("cornell_box")
    ds = DeviceScene(hs.desc)
    w, h = 64, 37
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 1.2, (h, w, 3)).astype(np.float32)
    b = np.float32(0.999) ** 2
    special = np.array([np.nan, -1.0, -0.0, 0.0, 1.0, 2.0, np.inf, -np.inf, b, np.nextafter(b, np.float32(0)), np.nextafter(b, np.float32(2)),
                        1e-45, 1e-38, (255.0 / 256) ** 2, (1.0 / 256) ** 2, np.nextafter(np.float32((1.0 / 256) ** 2), np.float32(0)), 0.25, 1e30],
                       np.float32)
    img.reshape(-1)[:len(special)] = special
    img[5, 7] = (np.nan, 0.5, -3.0)
    d_rgb = torch.from_numpy(img).cuda()
    d_out = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
    ds.to_color_device(d_rgb.data_ptr(), w, h, d_out.data_ptr())
    torch.cuda.synchronize()
    with np.errstate(invalid="ignore"):
        want = G.to_color(img)[::-1]                     # rows top-down
    assert np.array_equal(d_out.cpu().numpy(), want)
    ds.close()


def test_rgb8_output_is_to_color_of_the_f32_image(device, host_scenes):
    for name, w, spp in (("cornell_box", 100, 64), ("random_spheres_iow", 150, 70)):
        hs, cam = host_scenes(name)
        ds = DeviceScene(hs.desc)
        p = hs.params(w, spp, 30)
        img, _ = ds.render(cam, p)
        p8 = hs.params(w, spp, 30, output_format=ffi.VK_OUTPUT_RGB8)
        img8, _ = ds.render(cam, p8)
        assert np.array_equal(img8, G.to_color(img)[::-1])
        # tile partitions of an RGB8 image leave the other tiles alone
        acc = np.full_like(img8, 7)
        for r in range(3):
            ds.render(cam, hs.params(w, spp, 30, tile_rank=r, tile_world=3, output_format=ffi.VK_OUTPUT_RGB8), out=acc)
        assert np.array_equal(acc, img8)
        ds.close()


@pytest.mark.parametrize("name,w,spp", [("random_spheres_iow", 203, 130), ("cornell_box", 96, 70), ("final_scene", 64, 8)])
def test_multi_device_scene_is_bit_identical(name, w, spp, device, host_scenes):
    """vk_scene_create_multi with the one visible device listed 1, 2 and 5 times (each entry = one share with its own
    stream, slab and peer copy): vk_render must give the one-device image bit for bit, f32 and RGB8, also when
    the call itself is one rank of an outer tile partition."""
    hs, cam = host_scenes(name)
    single = DeviceScene(hs.desc)
    p = hs.params(w, spp, 50)
    p8 = hs.params(w, spp, 50, output_format=ffi.VK_OUTPUT_RGB8)
    want, st1 = single.render(cam, p)
    want8, _ = single.render(cam, p8)
    for n in (1, 2, 5):
        multi = DeviceScene(hs.desc, devices=[0] * n)
        assert multi.info().n_items == single.info().n_items
        got, st = multi.render(cam, p)
        assert st.samples == st1.samples
        assert np.array_equal(got, want), f"{n} shares: f32 image differs"
        got8, _ = multi.render(cam, p8)
        assert np.array_equal(got8, want8), f"{n} shares: RGB8 image differs"
        again, _ = multi.render(cam, p)                   # frame after frame on the same handle (main.rs:176)
        assert np.array_equal(again, want)
        acc = np.zeros_like(want)
        for r in range(2):                                # the group as one rank of an outer 2-way partition
            multi.render(cam, hs.params(w, spp, 50, tile_rank=r, tile_world=2), out=acc)
        assert np.array_equal(acc, want)
        assert multi.last_kernel_ms() > 0
        multi.close()
    single.close()


def test_rccl_gather_flag(device, host_scenes, capfd):
    """VK_SCENE_RCCL_GATHER on the one-GPU box: over ONE device the communicator initialises (ncclCommInitAll of one rank) and the frame is
    the one-device frame; a device listed twice cannot be two communicator ranks: the scene says so and moves its slabs by peer copies.
    (Slabs really travelling by ncclSend / ncclRecv takes two physical devices: the driver's multi-GPU run.)"""
    hs, cam = host_scenes("cornell_box")
    p = hs.params(96, 32, 50)
    single = DeviceScene(hs.desc)
    want, _ = single.render(cam, p)
    assert single.info().gather == ffi.VK_GATHER_NONE
    single.close()
    old = hs.desc.contents.flags
    try:
        hs.desc.contents.flags = old | ffi.VK_SCENE_RCCL_GATHER
        one = DeviceScene(hs.desc, devices=[0])
        assert one.info().gather == ffi.VK_GATHER_RCCL
        got, _ = one.render(cam, p)
        assert np.array_equal(got, want)
        got, _ = one.render(cam, p)
        assert np.array_equal(got, want)
        one.close()
        capfd.readouterr()
        two = DeviceScene(hs.desc, devices=[0, 0])
        assert two.info().gather == ffi.VK_GATHER_PEER_COPY
        assert "listed more than once" in capfd.readouterr().err
        got, _ = two.render(cam, p)
        assert np.array_equal(got, want)
        two.close()
    finally:
        hs.desc.contents.flags = old
    plain = DeviceScene(hs.desc, devices=[0, 0])
    assert plain.info().gather == ffi.VK_GATHER_PEER_COPY
    plain.close()


def test_rccl_gather_orchestration_with_a_test_double(device):
    """The ORCHESTRATION of VK_SCENE_RCCL_GATHER — sends on the parts' streams behind render and pack, receives on devices[0]'s receive
    stream, ONE event behind the group, part 0 unpacked from its own slab — run on the one-GPU box: the DEBUG build of the library loads
    tests/mock_rccl (VK_RCCL_LIB: ncclSend / ncclRecv as event-ordered device copies, pairs matched inside ncclGroupEnd) and is allowed
    to list the device several times.  Three shares: the frames must be the one-device frames bit for bit, f32 and RGB8, frame after
    frame, whole frame and as one rank of an outer partition, and the double must have moved two slabs per frame.  (Real RCCL over
    two physical devices: the driver's multi-GPU run.)  In a child process: the library reads the switches once."""
    from vecchio_amd import build
    mock = build.build_mock_rccl()
    code = ("import sys, ctypes as C; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from vecchio_amd import DeviceScene, HostScene, ffi\n"
            "dbg = ffi.load_debug_lib(); mock = C.CDLL(%r); mock.mock_rccl_pairs.restype = C.c_ulonglong\n"
            "hs = HostScene('cornell_box', 1); cam = hs.next_camera()\n"
            "p = hs.params(200, 48, 50); p8 = hs.params(200, 48, 50, output_format=ffi.VK_OUTPUT_RGB8)\n"
            "one = DeviceScene(hs.desc, lib=dbg); want = one.render(cam, p)[0]; want8 = one.render(cam, p8)[0]; one.close()\n"
            "hs.desc.contents.flags |= ffi.VK_SCENE_RCCL_GATHER\n"
            "m = DeviceScene(hs.desc, devices=[0, 0, 0], lib=dbg)\n"
            "assert m.info().gather == ffi.VK_GATHER_RCCL, m.info().gather\n"
            "ok = []\n"
            "for k in range(3): ok.append(np.array_equal(m.render(cam, p)[0], want))\n"
            "ok.append(np.array_equal(m.render(cam, p8)[0], want8))\n"
            "acc = np.zeros_like(want)\n"
            "for r in range(2): m.render(cam, hs.params(200, 48, 50, tile_rank=r, tile_world=2), out=acc)\n"
            "ok.append(np.array_equal(acc, want))\n"
            "parts = m.parts(); m.close()\n"
            "print('OK', all(ok), ok, 'PAIRS', mock.mock_rccl_pairs(), 'PARTS', len(parts))\n") % (ROOT, mock)
    env = dict(os.environ, VK_RCCL_LIB=mock, VK_RCCL_ALLOW_DUPLICATE_DEVICES="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "OK True" in r.stdout, r.stdout + r.stderr
    w = r.stdout.split()
    assert int(w[w.index("PAIRS") + 1]) == 2 * 6 and int(w[w.index("PARTS") + 1]) == 3, r.stdout      # six frames, two travelling slabs each


def test_multi_device_render_device_and_errors(device, host_scenes):
    import torch
    hs, cam = host_scenes("cornell_box")
    multi = DeviceScene(hs.desc, devices=[0, 0])
    p = hs.params(80, 64, 20)
    fb = torch.zeros((p.height, p.width, 3), dtype=torch.float32, device="cuda")
    st = multi.render_device(cam, p, fb.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want, _ = multi.render(cam, p)
    assert np.array_equal(fb.cpu().numpy(), want) and st.samples == 80 * p.height * 64
    multi.close()
    lib = device
    h = C.c_void_p()
    assert lib.vk_scene_create_multi(hs.desc, (C.c_int * 2)(0, 99), 2, C.byref(h)) == ffi.VK_ERR_BAD_ARG and not h.value
    assert lib.vk_scene_create_multi(hs.desc, None, 2, C.byref(h)) == ffi.VK_ERR_BAD_ARG
    assert lib.vk_scene_create_multi(hs.desc, (C.c_int * 1)(0), 0, C.byref(h)) == ffi.VK_ERR_BAD_ARG
    ds = DeviceScene(hs.desc)
    bad = hs.params(16, 4, 10, output_format=5)
    out = np.zeros((bad.height, 16, 3), np.float32)
    assert lib.vk_render(ds._h, C.byref(cam), C.byref(bad), out.ctypes.data, None) == ffi.VK_ERR_BAD_ARG
    ds.close()
    # (the diagnostic entry points live in libvecchio_amd_debug.so; a scene belongs to the library that made it)
    dbg = ffi.load_debug_lib()
    dd = DeviceScene(hs.desc, lib=dbg)
    dbg.vk_debug_phase_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert dbg.vk_debug_phase_stats(dd._h, None, None, None) == ffi.VK_ERR_BAD_ARG          # used to dereference params first
    dbg.vk_debug_phase_stats.argtypes = [C.c_void_p, C.POINTER(ffi.Camera), C.POINTER(ffi.RenderParams), C.POINTER(C.c_uint64 * 24)]
    dd.close()
    # (instrumented builds exist for the sphere-only/scatter and the full/PDF variants)
    from vecchio_amd import HostScene
    hs2 = HostScene("random_spheres_iow", 1)
    cam2 = hs2.next_camera()
    dd = DeviceScene(hs2.desc, lib=dbg)
    out = (C.c_uint64 * 24)()
    ps = hs2.params(96, 8, 20)
    assert dbg.vk_debug_phase_stats(dd._h, C.byref(cam2), C.byref(ps), C.byref(out)) == 0, dbg.vk_last_error()
    assert out[0] > 0 and out[12] > 0
    dd.close(); hs2.close()
    assert not hasattr(lib, "vk_debug_phase_stats") and not hasattr(lib, "vk_debug_math")      # the product library: production kernels only


def test_max_depth_zero_and_one(device, oracle, host_scenes):
    """MAX_DEPTH = 0: ray_color returns (0,0,0) before tracing (depth 1 > 0, main.rs:126-128) — a black image, also
    under a sky; MAX_DEPTH = 1 and 2 against the oracle."""
    for name in ("random_spheres_iow", "cornell_box"):
        hs, cam = host_scenes(name)
        ds = DeviceScene(hs.desc)
        for depth in (0, 1, 2):
            p = hs.params(48, 6, depth)
            got = np.full((p.height, 48, 3), 5.0, np.float32)
            ds.render(cam, p, out=got)
            ref, _ = oracle.render(hs.desc, cam, p)
            assert np.abs(got - ref).max() < 1e-4, (name, depth)
            if depth == 0:
                assert ref.max() == 0.0 and got.max() == 0.0
        ds.close()


def test_firefly_samples_saturate_and_are_counted(device):
    """ABI 3: a light of radiance 4e10 seen directly.  Every sample that hits it exceeds the accumulator's per-sample clamp
    (min(1e10, 1.3e11 / spp)): the pixel sums SATURATE (no int64 wrap: the pixel stays huge, where a wrapped sum would come
    out negative -> NaN -> black) and vk_stats.clamped_samples counts them; a frame without such samples reports 0."""
    from descs import Desc, camera, params
    d = Desc()
    hot = d.light(4e10, 4e10, 4e10)
    q = d.xy_rect(-1.0, 1.0, -1.0, 1.0, 0.0, hot)
    grey = d.sphere((0.0, -101.0, 0.0), 100.0, d.lambertian(0.5, 0.5, 0.5))
    desc = d.finish(d.big_box(q, grey), [q])
    cam = camera((0, 0, 6), (0, 0, 0), vfov=30.0)
    ds = DeviceScene(desc)
    for spp in (4, 64):
        p = params(32, 32, spp, max_depth=4, integrator=ffi.VK_INTEGRATOR_SCATTER)
        img, st = ds.render(cam, p)
        clampv = min(1e10, 1.3e11 / spp)
        centre = img[12:20, 12:20]                       # the quad covers the middle of the frame
        assert np.isfinite(img).all() and img.min() >= 0.0
        assert np.allclose(centre, clampv, rtol=1e-6), (spp, centre.min(), centre.max())
        assert st.clamped_samples >= 64 * spp
        n = C.c_uint64()
        assert ds._lib.vk_scene_last_clamped_samples(ds._h, C.byref(n)) == ffi.VK_OK and n.value == st.clamped_samples
    # an ordinary frame: nothing clamped
    d2 = Desc()
    l2 = d2.light(7.0, 7.0, 7.0)
    q2 = d2.xy_rect(-1.0, 1.0, -1.0, 1.0, 0.0, l2)
    desc2 = d2.finish(d2.big_box(q2, d2.sphere((0.0, -101.0, 0.0), 100.0, d2.lambertian(0.5, 0.5, 0.5))), [q2])
    ds2 = DeviceScene(desc2)
    _, st2 = ds2.render(cam, params(32, 32, 16, max_depth=4, integrator=ffi.VK_INTEGRATOR_SCATTER))
    assert st2.clamped_samples == 0
    ds.close(); ds2.close()


def test_unit_count_stays_below_32_bits(device, host_scenes):
    """tiles x sample chunks is a 32-bit counter: 4096 x 4096 at 2^20 spp would need 4.3 G units at 64 spp per unit.  The call is
    accepted (the chunks grow) — checked on a partition small enough to render: every 100 000th tile of that frame."""
    hs, cam = host_scenes("random_spheres_iow")
    ds = DeviceScene(hs.desc)
    p = hs.params(4096, 1 << 20, 2, height=4096, tile_rank=7, tile_world=100000)
    img, st = ds.render(cam, p)
    assert st.samples == 3 * 64 * (1 << 20)             # tiles 7, 100 007, 200 007 of 262 144
    assert np.isfinite(img).all()
    ds.close()


def test_dual_launch_self_check_falls_back_when_the_launches_do_not_overlap(device, tmp_path):
    """Sphere-only LDS scenes run as TWO concurrent launches (1024- and 768-thread workgroups: seven waves per SIMD).  If a runtime
    serialises them the first launch does all the work at 16 waves per CU; each launch counts the units it pulls and after two
    lopsided frames in a row the scene falls back to the single-launch shape.  Forced here with VK_DUAL_SAME_STREAM=1 (both
    launches on one stream); every frame is bit-identical to the normal scene's either way."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["VK_ROOT"])
from vecchio_amd import DeviceScene, HostScene
hs = HostScene("random_spheres_iow", 1); cam = hs.next_camera(); p = hs.params(1920, 64, 50)
ds = DeviceScene(hs.desc)
imgs = [ds.render(cam, p)[0].copy() for _ in range(4)]
assert all(np.array_equal(imgs[0], im) for im in imgs[1:])
np.save(os.environ["VK_OUT"], imgs[0])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for label, extra in (("normal", {"VK_DUAL_DEBUG": "1"}), ("serial", {"VK_DUAL_SAME_STREAM": "1", "VK_DUAL_DEBUG": "1"})):
        env = dict(os.environ, VK_ROOT=root, VK_OUT=str(tmp_path / (label + ".npy")), **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[label] = r.stderr
    assert "single-launch shape" not in outs["normal"] and "dual launch:" in outs["normal"], outs["normal"][-800:]
    assert "single-launch shape" in outs["serial"], outs["serial"][-800:]
    assert outs["serial"].count("dual launch:") == 2          # two lopsided frames, then no dual launch any more
    assert np.array_equal(np.load(tmp_path / "normal.npy"), np.load(tmp_path / "serial.npy"))
