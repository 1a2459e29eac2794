"""The probes under tools/ include the library's kernel headers (csrc/kernels.hpp, share_mfma.hpp): a refactor there must not
break them unnoticed -- DESIGN.md's figures cite their output.  Host-side parse of every tools/*.hip (`hipcc -fsyntax-only
--cuda-host-only`: templates, declarations and launches are checked without generating device code; no GPU needed)."""
import glob
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _check(path):
    r = subprocess.run([HIPCC, "-std=c++17", "-fsyntax-only", "--cuda-host-only", "--offload-arch=gfx950", "-Wno-everything", path],
                       capture_output=True, text=True)
    return path, r.returncode, r.stderr[-1500:]


def test_every_tool_still_parses_against_the_kernel_headers():
    tools = sorted(glob.glob(os.path.join(ROOT, "tools", "*.hip")))
    assert len(tools) >= 15 and os.path.exists(HIPCC)
    with ThreadPoolExecutor(max_workers=6) as ex:
        results = list(ex.map(_check, tools))
    broken = [(os.path.basename(p), err) for p, rc, err in results if rc != 0]
    assert not broken, broken
