"""Build recipes (explicit hipcc / g++ command lines, outputs in-tree so they travel with gpurun).

-ffp-contract=off everywhere: the reference's f32 arithmetic is unfused (Rust never
contracts a*b+c) and host/device parity of every control-flow value depends on it.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vecchio_amd", "csrc")
HOST = os.path.join(ROOT, "vecchio_amd", "host")
LIB = os.path.join(ROOT, "vecchio_amd", "lib")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")
CXXFLAGS = ["-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-Wall", "-Wextra"]
# -fno-slp-vectorize: left alone, LLVM packs adjacent scalar f32 adds/multiplies of the sphere test and the shading code into
# v_pk_add_f32 / v_pk_mul_f32 behind v_mov shuffles and s_nops; on gfx950 those do not issue faster than the scalar pair and cost
# registers (measured: C2 +2.3 %, C4 +2.4 %, C3 +7.8 %, C5 +2.3 % without them; the sphere-only kernel 78 -> 74 VGPRs)
HIPFLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-fPIC",
            "-Wall", "-Wno-unused-function"]


def _fingerprint(sources, flags=()):
    """content hash of the sources (and the flags they are built with): what a built file is a function of"""
    import hashlib
    h = hashlib.sha256(" ".join(flags).encode())
    for s in sorted(sources):
        h.update(os.path.basename(s).encode())
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _newer(target, sources, flags=()):
    """Is `target` stale?  By CONTENT, not by mtime: every built file has a `<file>.src` stamp beside it holding the hash of
    the sources it was built from, so a library pushed to another machine (gpurun copies the tree; timestamps mean nothing
    there) is rebuilt exactly when it is not the build of the sources next to it."""
    stamp = target + ".src"
    if not os.path.exists(target) or not os.path.exists(stamp):
        return True
    with open(stamp) as f:
        return f.read().strip() != _fingerprint(sources, flags)


def _stamp(target, sources, flags=()):
    with open(target + ".src", "w") as f:
        f.write(_fingerprint(sources, flags) + "\n")


def _run(cmd):
    print("+", " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)


def _run_atomic(cmd_before_out, out, cmd_after_out):
    """compile to a temporary name, then rename into place: a concurrent loader never sees a half-written file"""
    tmp = f"{out}.tmp.{os.getpid()}"
    try:
        _run(cmd_before_out + [tmp] + cmd_after_out)
        os.replace(tmp, out)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def _host_deps():
    srcs = [os.path.join(HOST, f) for f in ("host.cpp", "scene.cpp", "host_api.cpp")]
    return srcs, srcs + [os.path.join(HOST, "vecchio_host.hpp"), os.path.join(HOST, "host_api.h"),
                         os.path.join(CSRC, "vk_math.h"), os.path.join(ROOT, "include", "vecchio_amd.h")]


def _device_deps():
    srcs = [os.path.join(CSRC, "vk_api.hip"), os.path.join(CSRC, "vk_linearize.cpp")]
    return srcs, srcs + [os.path.join(CSRC, f) for f in ("vk_kernels.h", "vk_trace.h", "vk_math.h", "vk_device_scene.h", "vk_linearize.h")] + \
        [os.path.join(ROOT, "include", "vecchio_amd.h"), os.path.join(ROOT, "include", "vecchio_amd_debug.h")]


def host_is_stale():
    return _newer(os.path.join(LIB, "libvecchio_host.so"), _host_deps()[1], CXXFLAGS)


def device_is_stale():
    return _newer(os.path.join(LIB, "libvecchio_amd.so"), _device_deps()[1], HIPFLAGS) or not os.path.exists(kernel_resources_path())


def build_host(force=False):
    out = os.path.join(LIB, "libvecchio_host.so")
    srcs, deps = _host_deps()
    if force or _newer(out, deps, CXXFLAGS):
        os.makedirs(LIB, exist_ok=True)
        _run_atomic([CXX] + CXXFLAGS + ["-shared", "-o"], out, srcs + ["-lz"])   # zlib: the decoded texture images are .ppm.gz
        _stamp(out, deps, CXXFLAGS)
    return out


def build_cli(force=False):
    out = os.path.join(LIB, "vecchio_cli")
    src = os.path.join(HOST, "main.cpp")
    if not os.path.exists(src):
        return None
    host = build_host()
    deps = [src, os.path.join(HOST, "host_api.h"), os.path.join(ROOT, "include", "vecchio_amd.h")]
    if force or _newer(out, deps, CXXFLAGS) or not os.path.exists(host):
        _run([CXX] + CXXFLAGS + ["-o", out, src, "-L" + LIB, "-lvecchio_host", "-ldl", "-Wl,-rpath,$ORIGIN"])
        _stamp(out, deps, CXXFLAGS)
    return out


def build_device_debug(force=False):
    """libvecchio_amd_debug.so: the product library's sources with -DVK_DEBUG_LIB — plus the instrumented (STATS) kernel builds behind
    vk_debug_phase_stats and the device arithmetic probe vk_debug_math (include/vecchio_amd_debug.h).  Tests and diagnostics only:
    the product library holds production kernels only."""
    out = os.path.join(LIB, "libvecchio_amd_debug.so")
    srcs, deps = _device_deps()
    flags = HIPFLAGS + ["-DVK_DEBUG_LIB"]
    if force or _newer(out, deps, flags):
        os.makedirs(LIB, exist_ok=True)
        _run_atomic([HIPCC] + flags + ["-shared", "-o"], out, srcs)
        _stamp(out, deps, flags)
    return out


def debug_is_stale():
    return _newer(os.path.join(LIB, "libvecchio_amd_debug.so"), _device_deps()[1], HIPFLAGS + ["-DVK_DEBUG_LIB"])


def build_device(force=False):
    """hipcc cross-compiles for gfx950 without a GPU present."""
    out = os.path.join(LIB, "libvecchio_amd.so")
    srcs, deps = _device_deps()
    if force or _newer(out, deps, HIPFLAGS) or not os.path.exists(kernel_resources_path()):
        os.makedirs(LIB, exist_ok=True)
        # -save-temps in a scratch directory: the gfx950 assembly is read back for the per-kernel spill counts
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            tmp_out = f"{out}.tmp.{os.getpid()}"
            cmd = [HIPCC] + HIPFLAGS + ["-Rpass-analysis=kernel-resource-usage", "-save-temps", "-shared", "-o", tmp_out] + srcs
            print("+", " ".join(cmd), file=sys.stderr)
            r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True, cwd=tmp)
            remarks = [l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" in l]
            other = [l for l in r.stderr.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in l]
            if other:
                print("\n".join(other), file=sys.stderr)
            if r.returncode != 0:
                if os.path.exists(tmp_out):
                    os.remove(tmp_out)
                raise subprocess.CalledProcessError(r.returncode, cmd)
            os.replace(tmp_out, out)
            spills = {}
            for f in os.listdir(tmp):
                if f.endswith("gfx950.s"):
                    spills.update(_scratch_ops(open(os.path.join(tmp, f)).read()))
        # registers / scratch / occupancy of every kernel, kept next to the library: the megakernel's
        # throughput collapses when a variant starts spilling, so tests/test_kernel_resources.py pins them.
        # ScratchOps = number of scratch_load/scratch_store instructions in the kernel's code: the remark's
        # ScratchSize does not move when the allocator spills the same slots in twice as many places.
        text = "\n".join(re.sub(r"^.*remark: [^ ]* ", "", l).replace(" [-Rpass-analysis=kernel-resource-usage]", "") for l in remarks) + "\n"
        blocks = text.split("Function Name: ")
        text = blocks[0] + "".join("Name: " + b.rstrip("\n") + f"\n   ScratchOps: {spills.get(b.split()[0], -1)}\n" for b in blocks[1:])
        with open(kernel_resources_path(), "w") as f:
            f.write(text)
        _stamp(out, deps, HIPFLAGS)
    return out


def _scratch_ops(asm):
    """kernel symbol -> number of scratch_* instructions between its label and .Lfunc_end"""
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end", asm, flags=re.S | re.M):
        out[m.group(1)] = sum(1 for l in m.group(2).split("\n") if l.lstrip().startswith("scratch_"))
    return out


def kernel_resources_path():
    return os.path.join(LIB, "kernel_resources.txt")


def build_oracle(force=False):
    """CPU oracle (test infrastructure).  oracle/_ref does not exist: the reference is Rust and
    there is no cargo/rustc in this image."""
    odir = os.path.join(ROOT, "oracle")
    out = os.path.join(odir, "_build", "liboracle.so")
    deps = [os.path.join(odir, "oracle.cpp"), os.path.join(odir, "oracle.h"), os.path.join(CSRC, "vk_math.h"),
            os.path.join(ROOT, "include", "vecchio_amd.h")]
    if force or _newer(out, deps, ("oracle",)):
        _run(["make", "-B", "-C", odir])
        _stamp(out, deps, ("oracle",))
    return out


def build_emu(force=False):
    """Test-only host build of the kernel's per-lane logic (tests/emu)."""
    edir = os.path.join(ROOT, "tests", "emu")
    out = os.path.join(edir, "_build", "libemu.so")
    srcs = [os.path.join(edir, "emu.cpp"), os.path.join(CSRC, "vk_linearize.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("vk_trace.h", "vk_math.h", "vk_device_scene.h", "vk_linearize.h")] + \
        [os.path.join(ROOT, "include", "vecchio_amd.h")]
    if force or _newer(out, deps, CXXFLAGS):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        _run([CXX] + CXXFLAGS + ["-shared", "-o", out] + srcs + ["-lpthread"])
        _stamp(out, deps, CXXFLAGS)
    return out


def build_mock_rccl(force=False):
    """Test double of librccl.so (tests/mock_rccl): the library's RCCL gather on a one-GPU box, through the DEBUG build only."""
    mdir = os.path.join(ROOT, "tests", "mock_rccl")
    out = os.path.join(mdir, "_build", "librccl_mock.so")
    src = os.path.join(mdir, "mock_rccl.cpp")
    flags = ["-O2", "-std=c++17", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
    if force or _newer(out, [src], flags):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        _run([CXX] + flags + ["-shared", "-o", out, src, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"])
        _stamp(out, [src], flags)
    return out


def build_all(force=False):
    return [build_host(force), build_device(force), build_device_debug(force), build_oracle(force), build_emu(force), build_cli(force),
            build_mock_rccl(force)]


if __name__ == "__main__":
    build_all("--force" in sys.argv)
