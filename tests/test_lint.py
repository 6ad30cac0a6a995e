"""Static guard for code paths that only execute on a GPU box (bench.py main, smoke(), probes)."""
import glob
import os

import pytest

from tests.lint_names import undefined_names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(p for pat in ("*.py", "sgformer_amd/*.py", "oracle/*.py", "scripts/*.py", "tests/*.py")
               for p in glob.glob(os.path.join(ROOT, pat)))


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(p, ROOT) for p in FILES])
def test_no_undefined_names(path):
    assert undefined_names(path) == []


def test_checker_sees_function_scopes(tmp_path):
    """A name bound in ONE function is not visible in another (the bug class this guards against)."""
    f = tmp_path / "m.py"
    f.write_text("def a():\n    v = 1\n    return v\n\ndef b():\n    return v + [w for w in range(3)][0]\n")
    assert undefined_names(str(f)) == [(6, "v")]
