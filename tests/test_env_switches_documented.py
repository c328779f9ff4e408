"""Every HRV_* environment switch the product reads (hrv::env -- the cached getenv -- in csrc/, os.environ in the package, bench.py and the entry scripts) has a
row in INTEGRATION.md's switch table or is named in its text -- an undocumented switch is an A/B knob nobody can find."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = re.compile(r'(?:getenv\("|hrv::env\("|environ\.get\("|environ\[")(HRV_[A-Z0-9_]+)')


def test_every_environment_switch_is_listed_in_integration_md():
    files = []
    for d, exts in (("hr-viton_amd", (".py",)), (os.path.join("hr-viton_amd", "csrc"), (".hip", ".h"))):
        for f in sorted(os.listdir(os.path.join(ROOT, d))):
            if f.endswith(exts):
                files.append(os.path.join(ROOT, d, f))
    files += [os.path.join(ROOT, f) for f in sorted(os.listdir(ROOT)) if f.endswith(".py")]
    found = {}
    for path in files:
        with open(path, encoding="utf-8") as fh:
            for m in PAT.finditer(fh.read()):
                found.setdefault(m.group(1), os.path.relpath(path, ROOT))
    assert len(found) >= 30, sorted(found)
    with open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8") as fh:
        doc = fh.read()
    missing = {k: v for k, v in found.items() if k not in doc}
    assert not missing, f"environment switches read by the code but absent from INTEGRATION.md: {missing}"
