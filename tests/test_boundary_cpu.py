"""The drop-in boundary stays consistent in its four statements: include/x2i.h (the contract), x2i_amd/_lib.py (the ctypes
binding the product uses), INTEGRATION.md (the binding a reference maintainer would copy) and the compiled ABI (sizeof /
offsetof from a C compiler).  A field added to one and forgotten in another shifts every later member -- this test is what
catches it (INTEGRATION.md's GemmArgs was three fields short)."""
import ctypes as C
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "x2i.h")).read()

_CTYPE = {"const void*": "c_void_p", "void*": "c_void_p", "const float*": "c_void_p", "float*": "c_void_p", "int64_t": "c_int64",
          "int32_t": "c_int32", "float": "c_float"}


def header_struct(name):
    """[(field, ctypes name)] of `typedef struct <name> {...}` in include/x2i.h, in declaration order."""
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), HDR, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"^((?:const )?\w+\s?\*?)\s*(.+)$", decl)
        ctype = m.group(1).replace(" *", "*").strip()
        for field in m.group(2).split(","):
            out.append((field.strip().lstrip("*"), _CTYPE[ctype]))
    return out


def ctypes_struct(cls):
    # ctypes aliases fixed-width names to the platform types (c_int64 is c_long on LP64): compare the canonical objects
    canon = {getattr(C, n): n for n in ("c_void_p", "c_int64", "c_int32", "c_float")}
    return [(n, canon[t]) for n, t in cls._fields_]


def markdown_struct(name):
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"class %s\(C\.Structure\):.*?_fields_ = \[(.*?)\]\n" % name, md, flags=re.S).group(1)
    return re.findall(r'\("(\w+)",\s*C\.(\w+)\)', block)


def test_gemm_args_header_binding_and_integration_doc_agree():
    from x2i_amd import _lib
    want = header_struct("x2i_gemm_args")
    assert len(want) == 29 and want[0] == ("A", "c_void_p") and want[-2] == ("w_group", "c_int32")
    assert ctypes_struct(_lib.GemmArgs) == want
    assert [(n, t) for n, t in markdown_struct("GemmArgs")] == want, "INTEGRATION.md's raw-binding struct is out of date"


def test_descriptor_structs_header_and_binding_agree():
    from x2i_amd import _lib
    assert ctypes_struct(_lib.ConvDesc) == header_struct("x2i_conv_desc")
    assert ctypes_struct(_lib.QkvDesc) == header_struct("x2i_qkv_desc")
    assert ctypes_struct(_lib.Fp8Desc) == header_struct("x2i_fp8_desc")


def test_struct_layout_matches_the_compiled_abi(tmp_path):
    """sizeof / offsetof from a C compiler over the public header == ctypes' layout (same natural-alignment rules, but checked)."""
    from x2i_amd import _lib
    structs = {"x2i_gemm_args": _lib.GemmArgs, "x2i_conv_desc": _lib.ConvDesc, "x2i_qkv_desc": _lib.QkvDesc,
               "x2i_fp8_desc": _lib.Fp8Desc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "x2i.h"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines += ["  return 0;", "}"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {}
    for ln in out.splitlines():
        s, f, v = ln.split()
        got[(s, f)] = int(v)
    for cname, cls in structs.items():
        assert got[(cname, "sizeof")] == C.sizeof(cls), cname
        for f, _ in cls._fields_:
            assert got[(cname, f)] == getattr(cls, f).offset, (cname, f)


def test_every_entry_point_is_listed_in_the_integration_table():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(x2i_\w+)\s*\(", HDR, flags=re.M))
    missing = [n for n in sorted(declared) if "`%s`" % n not in md and n not in ("x2i_groupnorm_scratch_floats", "x2i_groupnorm_moments_scratch_floats")]
    assert not missing, missing
    assert "x2i_flux_" not in HDR  # no phantom handle API in the contract
