"""Builds the HOST-EMULATED libsdmi (test infrastructure): the library's own sources — csrc/capi.cpp, engine.cpp and the four kernel files —
compiled as plain C++ against tests/hostemu/hip/hip_runtime.h, plus tests/hostemu/emu.cpp (the block runner and thin emu_* entry points
for kernels that have no C-ABI entry of their own).  The result exports the whole C ABI of include/sdmi.h and runs on any x86-64 CPU.

    python tests/hostemu/build.py            # prints the path of the built library

Textual changes made to the COPIES that are compiled (the files themselves are the ones the GPU library is built from):
  * `extern __shared__ T x[]` (dynamic LDS) cannot also be `static`, which is what __shared__ means under the stand-in header: the arrays
    are defined in emu.cpp;
  * gemm.hip: the address-space-qualified pointer typedefs of the LDS-DMA builtin become plain pointers; an SGPR asm constraint a register;
  * `s_waitcnt` inline assembly becomes nothing (the stand-in's LDS-DMA completes at issue); attention.hip's v_permlane32_swap inline
    assembly becomes the stand-in's function; empty optimisation-barrier asm statements (AMDGPU register constraints) become nothing.
prof.cpp (HIP-event profiler) is not linked: emu.cpp records launch names instead.
"""
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "stable-diffusion-webui_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "hostemu")
KERNEL_FILES = ("elementwise.hip", "norm.hip", "gemm.hip", "attention.hip", "rowchain.hip")
HOST_FILES = ("capi.cpp", "engine.cpp")


def compiler():
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    return cxx if os.path.exists(cxx) else shutil.which("clang++")      # (_Float16 and ext_vector_type on x86: clang)


def transformed(name):
    src = open(os.path.join(CSRC, name)).read().replace("extern __shared__ ", "extern ")
    if name in ("gemm.hip", "rowchain.hip"):
        for old, new in (("typedef const __attribute__((address_space(1))) void* gptr_t;", "typedef const void* gptr_t;"),
                         ("typedef __attribute__((address_space(3))) void* lptr_t;", "typedef void* lptr_t;")) + ((('"+s"(dx)', '"+r"(dx)'),) if name == "gemm.hip" else ()):
            assert old in src, old
            src = src.replace(old, new)
    if name in ("gemm.hip", "attention.hip", "rowchain.hip"):
        src, n_wait = re.subn(r'asm volatile\("s_waitcnt[^;]*;', ";", src)
        assert n_wait > 0
    if name in ("attention.hip", "rowchain.hip"):
        src, n_swap = re.subn(r'asm volatile\("s_nop 1\\n\\tv_permlane32_swap_b32 %0, %1" : "\+v"\((\w+)\), "\+v"\((\w+)\)\);',
                              r"emu_permlane32_swap(\1, \2);", src)
        assert n_swap >= (3 if name == "attention.hip" else 1)
    src = re.sub(r'asm volatile\(""[^;]*;', ";", src)
    assert "asm volatile" not in src, name
    return src


def source_digest():
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp", ".h", ".inc"))]
    files += [os.path.join(EMU, "emu.cpp"), os.path.join(EMU, "hip", "hip_runtime.h"), os.path.abspath(__file__), os.path.join(ROOT, "include", "sdmi.h")]
    h.update((os.environ.get("SDMI_HOSTEMU_ASAN", "0") + os.environ.get("SDMI_HOSTEMU_UBSAN", "0") + os.environ.get("SDMI_HOSTEMU_TSAN", "0")).encode())
    for f in files:
        h.update(f.encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build():
    """-> path of libsdmi_hostemu.so, or None without clang++ / off x86-64.  Cached per source digest under the system temp directory."""
    cxx = compiler()
    if not cxx or os.uname().machine != "x86_64":
        return None
    out_dir = os.path.join(tempfile.gettempdir(), "sdmi_hostemu_" + source_digest())
    out = os.path.join(out_dir, "libsdmi_hostemu.so")
    if os.path.exists(out):
        return out
    work = tempfile.mkdtemp(prefix="sdmi_hostemu_build_")
    units = [os.path.join(EMU, "emu.cpp")] + [os.path.join(CSRC, f) for f in HOST_FILES]
    for name in KERNEL_FILES:
        path = os.path.join(work, name.replace(".hip", "_emu.cpp"))
        with open(path, "w") as fh:
            fh.write(transformed(name))
        units.append(path)
    f16c = ["-mf16c"] if "f16c" in open("/proc/cpuinfo").read() else []     # hardware fp16 <-> fp32 conversion: 3x faster than the soft-float calls
    flags = ["-x", "c++", "-std=c++17", "-O1", "-fPIC"] + f16c + ["-pthread", "-w", "-I" + EMU, "-I" + CSRC, "-I" + os.path.join(ROOT, "include")]
    asan = os.environ.get("SDMI_HOSTEMU_ASAN") == "1"       # an audit build: heap / stack out-of-bounds accesses of the kernels abort with a report
    if asan:                                                   #   (run with LD_PRELOAD=<clang's libclang_rt.asan-x86_64.so> ASAN_OPTIONS=detect_leaks=0)
        flags += ["-fsanitize=address", "-fno-omit-frame-pointer", "-g1", "-shared-libasan"]
    ubsan = os.environ.get("SDMI_HOSTEMU_UBSAN") == "1"     # misaligned vector accesses, signed index overflow, bad shifts, array bounds
    if ubsan:                                                  #   (run with LD_PRELOAD=<clang's libclang_rt.ubsan_standalone-x86_64.so>; reports go to stderr / UBSAN_OPTIONS=log_path)
        flags += ["-fsanitize=alignment,signed-integer-overflow,shift,bounds,integer-divide-by-zero", "-fno-sanitize-recover=all" if False else "-g1", "-shared-libsan"]
    tsan = os.environ.get("SDMI_HOSTEMU_TSAN") == "1"       # races between GPU threads of a block that no barrier orders (LDS and global memory)
    if tsan:                                                   #   (run with LD_PRELOAD=<clang's libclang_rt.tsan-x86_64.so>)
        flags += ["-fsanitize=thread", "-g1", "-shared-libsan"]
    procs = []
    for u in units:                                            # one compiler process per translation unit (gemm.hip alone is half a minute)
        obj = os.path.join(work, os.path.basename(u) + ".o")
        procs.append((u, obj, subprocess.Popen([cxx] + flags + ["-c", u, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for u, obj, pr in procs:
        log = pr.communicate(timeout=900)[0].decode()
        if pr.returncode:
            raise RuntimeError(f"host-emulation build of {u} failed:\n{log[-4000:]}")
        objs.append(obj)
    tmp_out = os.path.join(work, "libsdmi_hostemu.so")
    subprocess.run([cxx, "-shared", "-pthread"] + (["-fsanitize=address", "-shared-libasan"] if asan else []) + (["-fsanitize=undefined", "-shared-libsan"] if ubsan else []) + (["-fsanitize=thread", "-shared-libsan"] if tsan else []) + objs + ["-o", tmp_out], check=True, timeout=300)
    os.makedirs(out_dir, exist_ok=True)
    os.replace(tmp_out, out)
    shutil.rmtree(work, ignore_errors=True)
    return out


if __name__ == "__main__":
    print(build() or "unavailable (needs clang++ on x86-64)")
    sys.exit(0)
