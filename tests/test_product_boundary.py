"""The oracle is test infrastructure: nothing the product ships may import, link or execute it (only tests/, __graft_entry__.smoke()
and bench.py's checker legs do), and the product must fail loudly -- not fall back -- when the HIP library is missing."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IMPORT = re.compile(r"^\s*(?:from\s+oracle\b|import\s+oracle\b|from\s+\.+\s*oracle\b)", re.M)


def _product_files():
    pk = os.path.join(ROOT, "hr-viton_amd")
    files = [os.path.join(pk, f) for f in sorted(os.listdir(pk)) if f.endswith(".py")]
    files += [os.path.join(pk, "csrc", f) for f in sorted(os.listdir(os.path.join(pk, "csrc"))) if f.endswith((".hip", ".h"))]
    # the reference's entry scripts, mirrored at the repo root
    files += [os.path.join(ROOT, f) for f in ("test_generator.py", "train_generator.py", "train_condition.py", "test_condition.py",
                                              "get_norm_const.py") if os.path.exists(os.path.join(ROOT, f))]
    return files


def test_no_product_file_touches_the_oracle():
    bad = []
    for path in _product_files():
        with open(path, encoding="utf-8") as fh:
            src = fh.read()
        if IMPORT.search(src) or "oracle/" in src and path.endswith((".hip", ".h")):
            bad.append(os.path.relpath(path, ROOT))
    assert not bad, f"product files referencing the oracle: {bad}"


def test_bench_uses_the_oracle_only_inside_its_checker_legs():
    """bench.py may call the oracle in ``parity()`` / ``cpu_baseline()`` closures (the checker and the reported CPU baseline), never at
    module level or inside a timed ``step``."""
    with open(os.path.join(ROOT, "bench.py"), encoding="utf-8") as fh:
        lines = fh.read().splitlines()
    for i, ln in enumerate(lines):
        if re.match(r"\s*(from oracle|import oracle)", ln):
            assert ln.startswith("        "), f"bench.py:{i + 1}: oracle import outside a nested checker function"
            # the enclosing def, walking up to the first line with smaller indentation that starts a function
            indent = len(ln) - len(ln.lstrip())
            j = i
            while j >= 0 and not (lines[j].lstrip().startswith("def ") and len(lines[j]) - len(lines[j].lstrip()) < indent):
                j -= 1
            name = lines[j].strip()
            assert any(k in name for k in ("def parity", "def cpu_baseline", "def _cpu", "def _parity", "def extra", "def _oracle")), \
                f"bench.py:{i + 1}: oracle imported inside `{name}`"


def test_missing_library_is_an_error_not_a_fallback():
    """_lib.load() raises when libhrviton_hip.so is absent (no CPU path behind the ops)."""
    with open(os.path.join(ROOT, "hr-viton_amd", "_lib.py"), encoding="utf-8") as fh:
        src = fh.read()
    assert "raise HrvError" in src and "libhrviton_hip.so" in src
    assert "torch.nn.functional.conv2d" not in src and "F.conv2d" not in src
