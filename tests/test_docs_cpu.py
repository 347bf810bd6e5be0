"""The documents cite measurement files by name; a citation of a file that is not in the tree is worth nothing to a reader."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md"]


def _cited(text):
    for m in re.finditer(r"profiles/([A-Za-z0-9_.*\-]+)", text):
        name = m.group(1).rstrip(".")
        if name and not name.endswith("_"):
            yield name


def test_cited_profile_files_exist():
    missing = []
    for doc in DOCS:
        with open(os.path.join(ROOT, doc)) as f:
            text = f.read()
        for name in set(_cited(text)):
            pattern = os.path.join(ROOT, "profiles", name)
            if "*" in name:
                if not glob.glob(pattern):
                    missing.append(f"{doc}: profiles/{name}")
            elif not os.path.exists(pattern):
                missing.append(f"{doc}: profiles/{name}")
    assert not missing, "cited but absent:\n" + "\n".join(sorted(missing))


def test_cited_tools_exist():
    missing = []
    for doc in DOCS:
        with open(os.path.join(ROOT, doc)) as f:
            text = f.read()
        for m in set(re.findall(r"tools/(?:probes/)?[A-Za-z0-9_]+\.(?:py|sh|hip)", text)):
            if not os.path.exists(os.path.join(ROOT, m)):
                missing.append(f"{doc}: {m}")
    assert not missing, "cited but absent:\n" + "\n".join(sorted(missing))
