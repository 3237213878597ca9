"""The C-ABI shared library: builds for gfx950, loads without a GPU and exports every
entry point include/raytracer_amd.h declares (no compute call is made here)."""
import ctypes as C
import os
import re


def declared_functions(repo_root):
    text = open(os.path.join(repo_root, "include", "raytracer_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtamd_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(repo_root):
    import __graft_entry__
    __graft_entry__.build()
    from raytracer_amd import capi
    lib = C.CDLL(capi.LIB_PATH)
    names = declared_functions(repo_root)
    assert set(names) == set(capi.EXPORTS), "capi.EXPORTS and the header disagree"
    for name in names:
        assert hasattr(lib, name), "%s is declared but not exported" % name
    capi.bind(lib)
    assert lib.rtamd_abi_version() == capi.ABI_VERSION


def test_argument_checks_need_no_device(repo_root):
    from raytracer_amd import capi
    lib = capi.library()
    handle = C.c_void_p()
    assert lib.rtamd_scene_create(None, C.byref(handle)) == -1
    assert b"null" in lib.rtamd_last_error()
    bad = capi.SceneDesc()
    bad.abi_version = 999
    assert lib.rtamd_scene_create(C.byref(bad), C.byref(handle)) == -1
    assert b"ABI" in lib.rtamd_last_error()
    assert lib.rtamd_trace_rays(None, 1, 0, None, None, None, None, 0, None, None, 0, None) == -1


def test_product_does_not_import_the_checker(repo_root):
    """Nothing under raytracer_amd/ may reach into oracle/ (the checker is test infrastructure)."""
    pkg = os.path.join(repo_root, "raytracer_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower(), "%s mentions the oracle" % os.path.join(base, f)
