"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports
exactly the symbols include/adaprox_hip.h declares, the ctypes mirror of the structs has the
C layout, and the product refuses to run without a device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "adaprox_hip.h")


@pytest.fixture(scope="module")
def apx():
    import __graft_entry__ as g
    lib_path = os.path.join(ROOT, "adaptive-proximal-algorithms_amd", "csrc", "libadaprox_hip.so")
    if not os.path.exists(lib_path):
        g.build()
    import adaprox_amd as apx
    return apx


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(apx_[a-z0-9_]+)\s*\(", src)) - {"apx_eval_fn", "apx_pullback_fn", "apx_allreduce_fn"})


def test_library_exports_every_declared_symbol(apx):
    lib = apx.load()
    decl = _declared()
    assert len(decl) >= 35
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in include/adaprox_hip.h but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", apx.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r"\sT\s+(apx_[a-z0-9_]+)", out)))
    assert exported == decl, "exported apx_* symbols and the header disagree"


def test_ctypes_prototypes_cover_the_header(apx):
    L = __import__("importlib").import_module("adaptive-proximal-algorithms_amd._lib")
    assert sorted(L.PROTOTYPES) == _declared()


def test_struct_layouts_match_the_header(apx, tmp_path):
    """sizeof/offsetof of every struct crossing the boundary, from a C compile of the header."""
    L = __import__("importlib").import_module("adaptive-proximal-algorithms_amd._lib")
    prog = tmp_path / "layout.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "adaprox_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(apx_prox_desc), sizeof(apx_problem_desc), sizeof(apx_rule),
         sizeof(apx_record), sizeof(apx_result), sizeof(apx_solve_opts), sizeof(apx_kernel_stat));
  printf("%zu %zu %zu %zu %zu %zu\n", offsetof(apx_problem_desc, g), offsetof(apx_problem_desc, h),
         offsetof(apx_problem_desc, A_lin), offsetof(apx_problem_desc, allreduce), offsetof(apx_problem_desc, shard_mask),
         offsetof(apx_problem_desc, comm));
  printf("%zu %zu %zu\n", offsetof(apx_result, f_evals), offsetof(apx_result, n_records), offsetof(apx_rule, xi));
  printf("%zu %zu %zu %zu\n", sizeof(apx_team_wait_slot), offsetof(apx_team_wait_slot, at_clock_ns), offsetof(apx_team_wait_slot, block),
         offsetof(apx_team_wait_slot, kind));
  return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    lines = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    sizes = list(map(int, lines[0].split()))
    assert sizes == [C.sizeof(L.ProxDesc), C.sizeof(L.ProblemDesc), C.sizeof(L.Rule), C.sizeof(L.Record),
                     C.sizeof(L.Result), C.sizeof(L.SolveOpts), C.sizeof(L.KernelStat)]
    offs = list(map(int, lines[1].split()))
    assert offs == [L.ProblemDesc.g.offset, L.ProblemDesc.h.offset, L.ProblemDesc.A_lin.offset,
                    L.ProblemDesc.allreduce.offset, L.ProblemDesc.shard_mask.offset, L.ProblemDesc.comm.offset]
    offs = list(map(int, lines[2].split()))
    assert offs == [L.Result.f_evals.offset, L.Result.n_records.offset, L.Rule.xi.offset]
    slot = list(map(int, lines[3].split()))
    assert slot == [C.sizeof(L.TeamWaitSlot), L.TeamWaitSlot.at_clock_ns.offset, L.TeamWaitSlot.block.offset, L.TeamWaitSlot.kind.offset]


def test_header_is_plain_c_and_cites_the_reference():
    src = open(HEADER).read()
    subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", HEADER], check=True)
    for cite in ("src/AdaProx.jl:312-364", "src/AdaProx.jl:418-421", "src/AdaProx.jl:34-64", "src/AdaProx.jl:367-416",
                 "src/AdaProx.jl:457-459", "src/counting.jl", "experiments/lasso/runme.jl:16-27"):
        assert cite in src, f"header does not cite {cite}"
    # no torch / C++ types in the signatures: plain pointers and sizes only
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for banned in ("torch", "at::", "c10::", "std::", "#include <hip"):
        assert banned not in code, f"{banned} leaks into the C-ABI header"


def test_no_device_means_loud_failure_not_fallback(apx):
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is visible")
    except ImportError:
        pass
    with pytest.raises(apx.ApxError) as ei:
        apx.Context(device=0)
    assert "no" in str(ei.value).lower()
    with pytest.raises(apx.ApxError):
        apx.DeviceMatrix.from_numpy(np.eye(2))


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
    leg may touch it — not the package, not tools/, not examples/, not the alias module."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", flags=re.M)
    for sub in ("adaptive-proximal-algorithms_amd", "tools", "examples", "julia", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for fn in files:
                if fn.endswith((".py", ".hip", ".h", ".cpp", ".sh", ".jl")):
                    text = open(os.path.join(dirpath, fn)).read()
                    assert not pat.search(text), f"{sub}/{fn} imports oracle/"
    assert not pat.search(open(os.path.join(ROOT, "adaprox_amd.py")).read())
    # bench.py: only inside cpu_baseline()
    bench = open(os.path.join(ROOT, "bench.py")).read()
    body = bench[bench.index("def cpu_baseline("):bench.index("def main(")]
    assert len(pat.findall(bench)) == len(pat.findall(body)) > 0


def test_rule_constructors_mirror_reference_errors(apx):
    with pytest.raises(ValueError, match="you must provide gamma > 0 if norm_A = 0"):
        apx.OurRule()                                  # src/AdaProx.jl:246
    with pytest.raises(ValueError, match="you must provide gamma > 0"):
        apx.OurRulePlus()                              # src/AdaProx.jl:288
    r = apx.OurRule(norm_A=2.0, t=0.5)                 # :244  1 / (2 * Theta * t * norm_A)
    assert r.gamma == pytest.approx(1 / (2 * 1.2 * 0.5 * 2.0))
    assert (r.t, r.delta, r.Theta) == (0.5, 0.0, 1.2)


def test_graft_entry_build_passes():
    """`__graft_entry__.build()` is the driver's "does it build" check: the in-tree make, the library loaded with every symbol of
    the header, its ABI version the header's (round 4: the check had the number 3 written into it and would have failed the round
    that made it 4), the oracle importable, the C restatement of the bench workload built."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    ge = importlib.import_module("__graft_entry__")
    ge.build()
    assert not ge.is_stale()


def test_ctypes_constants_equal_the_header(apx, tmp_path):
    """Every APX_* constant the ctypes mirror spells out — the ABI version, status codes, flags, kinds, kernel-stat slots, option
    keys — against the header's own value, from a C compile (a key renumbered in one place only would silently set another option)."""
    L = __import__("importlib").import_module("adaptive-proximal-algorithms_amd._lib")
    names = sorted(n for n in dir(L) if n.startswith("APX_") and isinstance(getattr(L, n), int))
    assert len(names) >= 50 and "APX_ABI_VERSION" in names and "APX_OPT_TEAM_PD_DEFER" in names
    prog = tmp_path / "consts.c"
    prog.write_text('#include <stdio.h>\n#include "adaprox_hip.h"\nint main(void) {\n' +
                    "".join(f'  printf("{n} %lld\\n", (long long){n});\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "consts"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    assert {n: int(v) for n, v in out.items()} == {n: getattr(L, n) for n in names}
