"""CPU: libhx.so loads without a GPU, exports every symbol include/hx.h declares, agrees
with the ctypes structures, and fails LOUDLY (no CPU fallback) when asked to compute."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "hx.h")


@pytest.fixture(scope="module")
def lib_path():
    from rag_application_amd import build
    return build.build()


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hx_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_the_binding_binds():
    from rag_application_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_functions()


def test_library_exports_every_declared_symbol(lib_path):
    cdll = C.CDLL(lib_path)
    for name in declared_functions():
        assert getattr(cdll, name) is not None, name
    cdll.hx_abi_version.restype = C.c_int
    assert cdll.hx_abi_version() == 3


def test_struct_layouts_match_the_header(tmp_path):
    from rag_application_amd import _lib
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "hx.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",'
                    'sizeof(hx_params), sizeof(hx_stats), sizeof(hx_prof), offsetof(hx_params, rrf_k),'
                    'offsetof(hx_params, mode), offsetof(hx_prof, ms));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)],
                   check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [C.sizeof(_lib.HxParams), C.sizeof(_lib.HxStats), C.sizeof(_lib.HxProf),
                   _lib.HxParams.rrf_k.offset, _lib.HxParams.mode.offset, _lib.HxProf.ms.offset]


def test_header_is_plain_c(tmp_path):
    prog = tmp_path / "c.c"
    prog.write_text('#include "hx.h"\nint main(void){return HX_ABI_VERSION - 1;}\n')
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c",
                    str(prog), "-o", str(tmp_path / "c.o")], check=True)


def test_no_gpu_means_loud_failure_not_a_cpu_path(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from rag_application_amd import _lib, engine
    with pytest.raises(_lib.HxError) as e:
        engine.HxIndex(768)
    assert "device" in str(e.value).lower()


def test_argument_errors_are_reported_not_crashed(lib_path):
    """Bad arguments fail with rc != 0 and a message before anything touches the device (so this runs
    without a GPU): NULL pointers, limits outside [1, 2048], strides outside [1, 8192]."""
    import ctypes as C
    from rag_application_amd import _lib
    lib = _lib.lib()
    buf = (C.c_uint64 * 64)()
    cnt = (C.c_int32 * 8)()
    p, pc = C.addressof(buf), C.addressof(cnt)
    cases = [
        lambda: lib.hx_merge(0, None, 8, None, 1, 4, 0, p, pc, None),                 # NULL input
        lambda: lib.hx_merge(0, p, 0, None, 1, 4, 0, p, pc, None),                    # stride 0
        lambda: lib.hx_merge(0, p, 9000, None, 1, 4, 0, p, pc, None),                 # stride > 8192
        lambda: lib.hx_merge(0, p, 8, None, 1, 4096, 0, p, pc, None),                 # limit > 2048
        lambda: lib.hx_rrf(0, None, 4, pc, p, 4, pc, 1, 2.0, 0, 4, p, pc, None),      # NULL list
        lambda: lib.hx_h1_fuse(0, None, 2, 1, 4, 4, 4, 2.0, 0, p, pc, None),          # NULL gathered block
        lambda: lib.hx_h1_fuse(0, p, 0, 1, 4, 4, 4, 2.0, 0, p, pc, None),             # world 0
        lambda: lib.hx_h1_fuse(0, p, 100, 1, 100, 100, 10, 2.0, 0, p, pc, None),      # world x limit > 8192
        lambda: lib.hx_h1_fuse(0, p, 2, 1, 4, 4, 0, 2.0, 0, p, pc, None),             # limit 0
        lambda: lib.hx_h1_local(None, p, p, p, p, 1, 4, 4, p, None),                  # NULL index
        lambda: lib.hx_h1_local_async(None, p, p, p, p, 1, 4, 4, p, None),            # NULL index
        lambda: lib.hx_search_dense(None, p, 1, 0, 4, p, pc, None),
        lambda: lib.hx_save(None, b"/tmp/x.hx"),
        lambda: lib.hx_set_next_id(None, 5),                                          # NULL index
        lambda: lib.hx_truncate(None, 0),
        lambda: lib.hx_add_rows_dev(None, p, p, p, p, 1, None),
    ]
    for i, f in enumerate(cases):
        assert f() != 0, f"case {i} did not fail"
        assert lib.hx_last_error(), f"case {i}: no message"


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rag_application_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                for line in txt.splitlines():
                    s = line.strip()
                    if s.startswith(("import ", "from ", "#include")):
                        assert "oracle" not in s, f"{f}: {s}"
    code = "import sys; import rag_application_amd, rag_application_amd.engine, rag_application_amd.handler, " \
           "rag_application_amd.distributed, rag_application_amd.sharded, rag_application_amd.synth, rag_application_amd.embedding; " \
           "assert not [m for m in sys.modules if m.split('.')[0] == 'oracle'], 'oracle imported'"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
