"""The Rust shim (vecchio_amd/rust_shim/ffi.rs) cannot be compiled here (no rustc: SURVEY §8f-4), so the one thing that CAN be
checked is checked mechanically: every `#[repr(C)]` struct of ffi.rs has the fields of the C struct of the same name in
include/vecchio_amd.h — same names, same order, same types — and the constants agree.  A header change that the shim does
not follow fails here."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "vecchio_amd.h")).read()
SHIM = open(os.path.join(ROOT, "vecchio_amd", "rust_shim", "ffi.rs")).read()

SCALAR = {"uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "float": "f32", "double": "f64", "vk_ref": "vk_ref", "int": "c_int",
          "int32_t": "i32", "char": "c_char"}


def c_structs():
    src = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*\1\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            ptr = re.match(r"const (\w+) \*(\w+)$", decl)
            if ptr:
                fields.append((ptr.group(2), "*const " + SCALAR.get(ptr.group(1), ptr.group(1))))
                continue
            ty, rest = decl.split(" ", 1)
            for name in [n.strip() for n in rest.split(",")]:
                dims = re.findall(r"\[(\d+)\]", name)
                name = name.split("[")[0]
                t = SCALAR[ty]
                for d in reversed(dims):          # float v[256][3] -> [[f32; 3]; 256]
                    t = f"[{t}; {d}]"
                fields.append((name, t))
        out[m.group(1)] = fields
    return out


def rust_structs():
    src = re.sub(r"//[^\n]*", "", SHIM)
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\][^{;]*?pub struct (\w+)\s*\{(.*?)\}", src, flags=re.S):
        body = m.group(2)
        fields, depth, cur = [], 0, ""
        for ch in body:                            # split on top-level commas ([[f32; 3]; 256] holds none, but be safe)
            if ch in "[(":
                depth += 1
            elif ch in "])":
                depth -= 1
            if ch == "," and depth == 0:
                fields.append(cur)
                cur = ""
            else:
                cur += ch
        fields.append(cur)
        parsed = []
        for f in fields:
            f = " ".join(f.split())
            if not f:
                continue
            mm = re.match(r"(?:pub )?(\w+): (.+)$", f)
            assert mm, f"unparsed Rust field in {m.group(1)}: {f!r}"
            parsed.append((mm.group(1), mm.group(2).strip()))
        out[m.group(1)] = parsed
    return out


def test_every_c_struct_has_an_identical_repr_c_twin():
    c, r = c_structs(), rust_structs()
    assert {"vk_scene_desc", "vk_camera", "vk_render_params", "vk_stats", "vk_bvh_node", "vk_perlin"} <= set(c)
    for name, fields in c.items():
        if name in ("vk_scene_info", "vk_part_info"):      # introspection for bench/tests, not bound by the shim
            continue
        assert name in r, f"{name} missing from rust_shim/ffi.rs"
        assert r[name] == fields, f"{name}:\n  header {fields}\n  ffi.rs {r[name]}"


def test_constants_agree():
    def c_const(n):
        m = re.search(rf"#define {n} (\w+)", HEADER) or re.search(rf"\b{n} = (\w+)", HEADER)
        return int(m.group(1).rstrip("u"), 0)

    for n in ("VK_ABI_VERSION", "VK_REF_FLIP", "VK_KIND_BVH", "VK_KIND_SPHERE", "VK_KIND_MOVING_SPHERE", "VK_KIND_RECT", "VK_KIND_LIST",
              "VK_KIND_MEDIUM", "VK_KIND_TRANSLATE", "VK_KIND_ROTATE"):
        m = re.search(rf"pub const {n}: u32 = ([\w_]+);", SHIM)
        assert m, n
        assert int(m.group(1).replace("_", ""), 0) == c_const(n), n


def test_every_bound_function_is_declared_in_the_header():
    block = re.search(r'extern "C" \{(.*?)\n\}', SHIM, flags=re.S).group(1)
    for fn in re.findall(r"pub fn (vk_\w+)\(", block):
        assert re.search(rf"\b{fn}\s*\(", HEADER), f"{fn} bound in ffi.rs but not declared in vecchio_amd.h"
