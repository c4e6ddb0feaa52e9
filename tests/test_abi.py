"""CPU tests of the C-ABI boundary: the library builds, loads, and exports every declared symbol."""
import ctypes as C
import os
import re

import pytest

from nebulae_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            text = open(os.path.join(inc, fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names |= set(re.findall(r"\b(neb_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_builds_and_exports_every_declared_symbol():
    build.build()
    lib = C.CDLL(build.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    # the ctypes binding covers the same set
    assert set(_lib.exported_symbols()) == set(decl)


def test_binding_loads_and_reports_version():
    lib = _lib.load()
    assert b"gfx950" in lib.neb_version()
    p = _lib.SvgfParams()
    assert lib.neb_svgf_default_params(C.byref(p)) == 0
    assert abs(p.phiColor - 4.0 / 255.0) < 1e-9 and p.phiNormal == 128.0 and abs(p.alpha - 0.9) < 1e-7


def test_create_rejects_bad_arguments_without_touching_a_gpu():
    lib = _lib.load()
    ctx = C.c_void_p()
    assert lib.neb_create(None, C.byref(ctx)) == -1
    info = _lib.CreateInfo(0, 0, 16, 0, 0, 4)
    assert lib.neb_create(C.byref(info), C.byref(ctx)) == -1
    info = _lib.CreateInfo(0, 16, 16, 8, 4, 4)
    assert lib.neb_create(C.byref(info), C.byref(ctx)) == -1
    assert b"row range" in lib.neb_last_error(None)


def test_strip_exchange_entry_points_validate_arguments_without_a_gpu():
    lib = _lib.load()
    comm = C.c_void_p()
    assert lib.neb_strips_comm_create(0, 0, 0, None, C.byref(comm)) == -1 and b"bad argument" in lib.neb_strips_last_error()
    assert lib.neb_strips_unique_id(None) == -1
    assert lib.neb_strips_comm_destroy(None) == 0
    assert lib.neb_strips_exchange(None, None, None, 0, None, 0, None) == -1
    # the one-call strip frame: a null context is refused before anything else is looked at
    plan = _lib.StripPlan(2, 0, 0, 0)
    out = (C.c_uint32 * 8)()
    assert lib.neb_strip_frame(None, None, None, C.byref(plan), None) == -1
    assert lib.neb_strip_frame_begin(None, None, C.byref(plan), None, None) == -1
    assert lib.neb_strip_frame_finish(None, None, C.byref(plan), None, None) == -1
    assert lib.neb_strip_rows(None, C.byref(plan), out) == -1


def test_strip_calls_report_a_missing_rccl_instead_of_crashing():
    """A box without librccl: the strip entry points return NEB_ERR_STATE with the loader's message (the library itself still
    loads).  The not-found path is forced with NEB_RCCL_LIBRARY in a child process (the lookup happens once per process)."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys\n"
            "from nebulae_amd import _lib\n"
            "lib = _lib.load()\n"
            "buf = (C.c_char * 128)()\n"
            "rc = lib.neb_strips_unique_id(buf)\n"
            "msg = lib.neb_strips_last_error()\n"
            "rc2 = lib.neb_strips_group_begin()\n"
            "print(rc, rc2, msg.decode())\n")
    env = dict(os.environ, NEB_RCCL_LIBRARY="libneb_no_such_rccl.so", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rc, rc2, msg = out.stdout.strip().split(" ", 2)
    assert int(rc) == -4 and int(rc2) == -4
    assert "librccl not found" in msg and "libneb_no_such_rccl.so" in msg


def test_no_cpu_fallback_when_no_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nebulae_amd.svgf import SVGFDenoiser, NebError
    d = SVGFDenoiser()
    with pytest.raises(NebError):
        d.init(64, 64)


def test_cpp_mirror_header_compiles_against_the_abi(tmp_path):
    """include/nebulae_hip.hpp (Neb::SVGFDenoiser / Neb::GIPathtracer over the C ABI) compiles and links."""
    import subprocess
    src = tmp_path / "use.cpp"
    src.write_text('#include "nebulae_hip.hpp"\n'
                   'int main() { Neb::SVGFDenoiser d; Neb::GIPathtracer g(d); Neb::StripExchange x;\n'
                   '  try { x.Init(0, 0, 0, nullptr); return 3; } catch (const Neb::NebException& e) { if (e.Status != NEB_ERR_INVALID_ARG) return 4; }\n'
                   '  try { d.Init(0, 0); } catch (const Neb::NebException& e) { return e.Status == NEB_ERR_INVALID_ARG ? 0 : 2; }\n'
                   '  return 1; }\n')
    exe = tmp_path / "use"
    libdir = os.path.dirname(build.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lnebulae_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.call([str(exe)]) == 0
