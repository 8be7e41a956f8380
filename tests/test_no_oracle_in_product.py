"""The oracle is test infrastructure: the product package and the C-ABI library must not
import, link or call anything under oracle/, and must have no CPU compute fallback."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "raytracing.jl_amd")


def _product_files():
    for d, _dirs, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                yield os.path.join(d, f)
    yield os.path.join(ROOT, "raytracing_jl_amd.py")
    yield os.path.join(ROOT, "include", "rt_segmentize.h")


def test_product_never_references_the_oracle():
    for path in _product_files():
        text = open(path).read()
        assert not re.search(r"\boracle\b|liboracle|rt_oracle|orc_", text), path


def test_library_does_not_link_the_oracle():
    lib = os.path.join(PKG, "csrc", "librt_segmentize.so")
    if os.path.exists(lib):
        blob = open(lib, "rb").read()
        assert b"liboracle" not in blob and b"orc_segmentize" not in blob
