"""The measurement scripts under tools/ are not run by the suite; this keeps them parseable."""
import glob
import os
import py_compile
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_tools_compile(tmp_path):
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert files
    for f in files:
        py_compile.compile(f, cfile=str(tmp_path / (os.path.basename(f) + "c")), doraise=True)


def test_shell_tools_parse():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert files
    for f in files:
        subprocess.run(["bash", "-n", f], check=True)
