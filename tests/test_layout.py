"""The oracle is test infrastructure: nothing under ct_icp_amd/ may import, link or execute anything under oracle/,
and the product has no CPU compute path hiding behind the HIP one."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _files(sub, exts):
    for d, _, fs in os.walk(os.path.join(ROOT, sub)):
        for f in fs:
            if f.endswith(exts):
                yield os.path.join(d, f)


def test_product_never_touches_the_oracle():
    # comments may MENTION the oracle; includes, imports, links and paths may not exist
    pat = re.compile(r"#include[^\n]*oracle|import[^\n]*oracle|from\s+oracle|libctgn_oracle|libctgn_ref|ctgn_oracle\.|oracle/|orc_[a-z_]+\(|ref_[a-z_]+\("
                     r"|numpy_check|mini_eigen|shims/")
    for path in _files("ct_icp_amd", (".py", ".hpp", ".hip", ".cpp", ".h", "Makefile")):
        txt = open(path, errors="ignore").read()
        assert not pat.search(txt), f"{path} references the oracle"


def test_only_allowed_callers_import_the_oracle():
    allowed = {os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")}
    for path in list(_files("", (".py",))):
        if path.startswith(os.path.join(ROOT, "tests")) or path.startswith(os.path.join(ROOT, "oracle")):
            continue
        if path in allowed:
            continue
        assert "oracle" not in open(path).read(), f"{path} must not use oracle/"


def test_no_compat_layers():
    for path in _files("ct_icp_amd", (".hpp", ".hip", ".cpp", ".h")):
        txt = open(path, errors="ignore").read()
        for bad in ("__HIP_PLATFORM_AMD__", "cuda_runtime.h", "__CUDACC__", "triton"):
            assert bad not in txt, f"{path}: {bad}"
