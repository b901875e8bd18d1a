"""CPU-only checks of the boundary: the shared library loads and exports every symbol that
include/pyrodigal_amd.h declares; no compute call is made (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pyrodigal_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pga_[a-z_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    names = declared_functions()
    for must in ("pga_create", "pga_set_models", "pga_score_connections", "pga_find_genes_batch",
                 "pga_batch_create", "pga_find_genes", "pga_result_free"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from pyrodigal_amd import _cabi
    if not os.path.exists(_cabi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_cabi.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert sorted(_cabi.EXPORTS) == declared_functions()


def test_struct_sizes_match_the_header(tmp_path):
    """The ctypes mirrors against the header itself: a C program prints sizeof / offsetof of what the binding copies."""
    import subprocess
    from pyrodigal_amd import _cabi
    src = tmp_path / "sizes.c"
    src.write_text("""
#include <stdio.h>
#include <stddef.h>
#include "pyrodigal_amd.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(pga_params), sizeof(pga_gene), sizeof(pga_contig_result), sizeof(pga_training),
           sizeof(pga_result), sizeof(pga_nodes), offsetof(pga_result, mask_off), offsetof(pga_contig_result, gc));
    return 0;
}
""")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [ctypes.sizeof(_cabi.Params), ctypes.sizeof(_cabi.Gene), ctypes.sizeof(_cabi.ContigResult), _cabi.TRAINING_SIZE,
                   ctypes.sizeof(_cabi.Result), ctypes.sizeof(_cabi.Nodes), _cabi.Result.mask_off.offset, _cabi.ContigResult.gc.offset]
    assert got[:4] == [32, 88, 40, 558392]


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from pyrodigal_amd import _cabi
    with pytest.raises(RuntimeError):
        _cabi.Context(0)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pyrodigal_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".pyx", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower() or f == "benchdata.py" and "oracle" in text.lower() and "import oracle" not in text, f


def test_argument_arrays_of_a_batch_by_the_host_layer():
    """`lib._seq_pointers` (what `_cabi.Batch` hands to pga_batch_create): addresses and lengths of a list of bytes, one C loop;
    anything but bytes is the caller's to convert (TypeError -> the ctypes conversions)."""
    from pyrodigal_amd import _cabi, lib
    seqs = [b"ACGT" * 5, b"", b"N" * 33, bytes(range(65, 91))]
    ptrs, lens, total = lib._seq_pointers(seqs)
    assert total == sum(len(s) for s in seqs) and list(lens[:4]) == [len(s) for s in seqs]
    for p, s in zip(ptrs, seqs):
        assert ctypes.string_at(int(p), len(s)) == s
    with pytest.raises(TypeError):
        lib._seq_pointers([b"ACGT", "ACGT"])
    with pytest.raises(TypeError):
        lib._seq_pointers([b"ACGT", bytearray(b"ACGT")])
    p0, l0, t0 = lib._seq_pointers([])
    assert t0 == 0 and len(p0) >= 1 and len(l0) >= 1        # (never a zero-length array: the C-ABI gets a valid pointer)
    assert _cabi._seq_pointers_fn() is lib._seq_pointers
