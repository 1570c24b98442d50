import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    # a fresh checkout has no built libraries (they are git-ignored): build what is missing once, the way
    # __graft_entry__.build() does (hipcc cross-compiles gfx950 without a GPU).  Never rebuilds an existing library.
    import subprocess
    for lib, script in (("libfsrl_hip.so", os.path.join("fsrl_amd", "csrc", "build.sh")),
                        ("libfsrl_env.so", os.path.join("fsrl_amd", "env", "csrc", "build.sh"))):
        if not os.path.exists(os.path.join(ROOT, "fsrl_amd", lib)) and not os.environ.get("FSRL_HIP_LIB"):
            try:
                subprocess.run(["bash", os.path.join(ROOT, script)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=900)
            except Exception as e:                      # noqa: BLE001 -- the tests that need the library will say so
                print(f"conftest: could not build {lib}: {e}", file=sys.stderr)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
