"""Error behaviour of scene upload / render arguments.  The reference panics (unwrap, assert,
index out of bounds); across the C ABI that becomes a status code and a message.  The
lineariser is shared by the HIP library and tests/emu, so it is exercised here on the CPU."""
import ctypes as C

import numpy as np

from descs import Desc, camera, params
from vecchio_amd import ffi


def emu_status(emu, desc, cam, p):
    lib = emu.load()
    img = np.zeros((p.height, p.width, 3), np.float32)
    return lib.emu_render(desc, C.byref(cam), C.byref(p), img.ctypes.data, None, 1, None, None), lib.emu_last_error().decode()


def test_bad_indices_rejected(emu):
    cam = camera((0, 0, -5), (0, 0, 0))
    p = params(8, 8, 1)
    d = Desc()
    d.lambertian(0.5, 0.5, 0.5)
    s = d.sphere((0, 0, 0), 1.0, 7)                       # material index out of range
    st, msg = emu_status(emu, d.finish(s, [s]), cam, p)
    assert st == ffi.VK_ERR_BAD_ARG and "material" in msg
    d = Desc()
    d.lambertian(0.5, 0.5, 0.5)
    st, msg = emu_status(emu, d.finish(ffi.make_ref(ffi.VK_KIND_SPHERE, 3)), cam, p)   # dangling reference
    assert st == ffi.VK_ERR_BAD_ARG
    d = Desc()
    s = d.sphere((0, 0, 0), 1.0, d.lambertian(0.5, 0.5, 0.5))
    desc = d.finish(s, [s])
    d.desc.abi_version = 99
    st, msg = emu_status(emu, desc, cam, p)
    assert st == ffi.VK_ERR_BAD_ARG and "abi" in msg


def test_cyclic_graph_rejected(emu):
    cam = camera((0, 0, -5), (0, 0, 0))
    p = params(8, 8, 1)
    d = Desc()
    s = d.sphere((0, 0, 0), 1.0, d.lambertian(0.5, 0.5, 0.5))
    n0 = d.big_box(s, s)
    d.bvh[0].left = n0                                   # a node that is its own child
    st, msg = emu_status(emu, d.finish(n0, [s]), cam, p)
    assert st == ffi.VK_ERR_BAD_ARG and "cyclic" in msg
    d = Desc()
    s = d.sphere((0, 0, 0), 1.0, d.lambertian(0.5, 0.5, 0.5))
    t = d.translate(s, (1, 0, 0))
    d.translates[0].child = t
    st, msg = emu_status(emu, d.finish(t, [s]), cam, p)
    assert st != ffi.VK_OK


def test_unsupported_shapes_reported(emu):
    cam = camera((0, 0, -5), (0, 0, 0))
    p = params(8, 8, 1)
    d = Desc()
    m = d.lambertian(0.5, 0.5, 0.5)
    s = d.sphere((0, 0, 0), 1.0, m)
    inner = d.big_box(s, s)
    lst = d.list_([inner])                                # a BVH inside a list: not linearisable
    st, msg = emu_status(emu, d.finish(lst, [s]), cam, p)
    assert st == ffi.VK_ERR_UNSUPPORTED and "list" in msg
    d = Desc()
    s = d.sphere((0, 0, 0), 1.0, d.lambertian(0.5, 0.5, 0.5))
    med = d.medium(d.translate(s, (1, 0, 0)), 0.5, d.mat(ffi.VK_MAT_ISOTROPIC, d.solid(1, 1, 1)))
    st, msg = emu_status(emu, d.finish(med, [s]), cam, p)
    assert st == ffi.VK_ERR_UNSUPPORTED and "boundary" in msg


def test_pdf_integrator_needs_lights(emu):
    cam = camera((0, 0, -5), (0, 0, 0))
    d = Desc()
    s = d.sphere((0, 0, 0), 1.0, d.lambertian(0.5, 0.5, 0.5))
    st, msg = emu_status(emu, d.finish(s, []), cam, params(8, 8, 1, integrator=ffi.VK_INTEGRATOR_PDF))
    assert st == ffi.VK_ERR_UNSUPPORTED and "lights" in msg   # Vec::random would unwrap None (hittable.rs:431)


def test_boxy_lists_become_compact_boxes(emu):
    """The lineariser stores an exact Boxy::new pattern as a 32-byte DBox (feature bit 0x100) and anything
    else as a generic list (0x4); both must render like the oracle (covered by the parity tests)."""
    cam = camera((5, 5, -12), (1, 1, 1))
    p = params(8, 8, 1)
    d = Desc()
    m = d.lambertian(0.5, 0.5, 0.5)
    box = d.boxy((0, 0, 0), (2, 3, 4), m)
    lm = d.light(5, 5, 5)
    ls = d.xz_rect(0, 1, 0, 1, 9, lm)
    desc = d.finish(d.big_box(box, Desc.flip(ls)), [ls])
    _, _, _, info = emu.render_samples(desc, cam, p)
    assert info[3] & 0x100 and not (info[3] & 0x4)
    d = Desc()
    m = d.lambertian(0.5, 0.5, 0.5)
    box = d.boxy((0, 0, 0), (2, 3, 4), m)
    d.rects[2].k = 2.5                                  # no longer the canonical pattern: stays a list
    lm = d.light(5, 5, 5)
    ls = d.xz_rect(0, 1, 0, 1, 9, lm)
    desc = d.finish(d.big_box(box, Desc.flip(ls)), [ls])
    _, _, _, info = emu.render_samples(desc, cam, p)
    assert info[3] & 0x4 and not (info[3] & 0x100)
